export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
bash tools/profile_round.sh r02 cfg4 2>&1 | grep -E "rc=|summarised"
bash tools/profile_round.sh r02 cfg3 2>&1 | grep -E "rc=|summarised"
bash tools/profile_round.sh r02 cfg2 2>&1 | grep -E "rc=|summarised"
python bench.py --workload cfg1 > gpurun_out/r02_cfg1_bench.json 2> gpurun_out/r02_cfg1_bench_err.log; echo cfg1 rc=$?
python bench.py --workload cfg5 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02_cfg5_bench.json 2> gpurun_out/r02_cfg5_bench_err.log; echo cfg5 rc=$?
python __graft_entry__.py smoke 2>&1 | tail -3
