mkdir -p gpurun_out/r3
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "64bit or tier6 or cfg5_shaped or wide or more_live" > gpurun_out/r3/wide_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3/wide_tests.log
timeout 2400 python tools/tune_tiers.py cfg5 "fast6w=1" "fast6w=0" > gpurun_out/r3/tune_cfg5.log 2>&1; echo "rc=$?" >> gpurun_out/r3/tune_cfg5.log
