mkdir -p gpurun_out/r3
timeout 1500 python -m pytest tests -x -q -m gpu -k "shard or multi or cfg2 or gather or passes" > gpurun_out/r3/shard_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3/shard_tests.log
ASGART_SHARD_LPT=1 timeout 1500 python -m pytest tests -x -q -m gpu -k "multi or cfg2 or gather" > gpurun_out/r3/shard_tests_lpt.log 2>&1; echo "rc=$?" >> gpurun_out/r3/shard_tests_lpt.log
timeout 1500 python tools/shard_check.py 8 cfg4 shard_lpt=1 > gpurun_out/r3/shard8_lpt.log 2>&1; echo "rc=$?" >> gpurun_out/r3/shard8_lpt.log
