export TMPDIR=/tmp
bash tools/profile_round.sh r02 cfg4 2>&1 | grep -E "rc=|summarised"
bash tools/profile_round.sh r02 cfg3 2>&1 | grep -E "rc=|summarised"
bash tools/profile_round.sh r02 cfg2 2>&1 | grep -E "rc=|summarised"
python bench.py --workload cfg1 > gpurun_out/r02_cfg1_bench.json 2> gpurun_out/r02_cfg1_bench_err.log; echo cfg1 rc=$?
python bench.py --workload cfg5 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02_cfg5_bench.json 2> gpurun_out/r02_cfg5_bench_err.log; echo cfg5 rc=$?
ASGART_BENCH_OVERLAP=0 python bench.py --workload cfg4 --no-cpu-baseline > gpurun_out/r02_cfg4_bench_sequential.json 2>/dev/null; echo seq rc=$?
./tools/bin/ubench_gather 16 > gpurun_out/r02_ubench_gather.txt 2>&1
