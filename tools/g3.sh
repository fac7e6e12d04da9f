export TMPDIR=/tmp
export ASGART_BENCH_MODE=back_to_back
rocprofv3 --output-format csv --kernel-trace -d gpurun_out/tl -o run -- python3 bench.py --workload cfg4 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/tl_bench.json 2> gpurun_out/tl_err.log
python3 tools/timeline.py gpurun_out/tl 450 > gpurun_out/timeline_b2b.txt
rm -rf gpurun_out/tl
