mkdir -p gpurun_out/r3
timeout 3000 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cfg5_wide_digest or cfg5_full_properties or cfg1_ecoli" --durations=5 > gpurun_out/r3/cfg5_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3/cfg5_tests.log
