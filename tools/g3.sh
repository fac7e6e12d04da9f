export TMPDIR=/tmp
export ASGART_BENCH_MODE=pipelined_1_first
rocprofv3 --output-format csv --kernel-trace -d gpurun_out/tl -o run -- python3 bench.py --workload cfg4 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/tl_bench.json 2> gpurun_out/tl_err.log
python3 tools/timeline.py gpurun_out/tl 340 > gpurun_out/timeline_pipe.txt
rm -rf gpurun_out/tl
