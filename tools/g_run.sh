mkdir -p gpurun_out/r3
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r3/gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3/gpu_tests.log
