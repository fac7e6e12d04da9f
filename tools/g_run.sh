mkdir -p gpurun_out/r3
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r3/gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3/gpu_tests.log
timeout 1500 python tools/tune_tiers.py cfg4 "posbits=1" "posbits=0" > gpurun_out/r3/tune_pb.log 2>&1; echo "rc=$?" >> gpurun_out/r3/tune_pb.log
