for w in cfg4 cfg3 cfg2 cfg1; do
timeout 900 python3 bench.py --workload $w > gpurun_out/r02_bench_${w}_1gpu.json 2> gpurun_out/r02_bench_${w}_err.log
echo "$w rc=$?"
done
