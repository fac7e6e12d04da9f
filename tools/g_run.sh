mkdir -p gpurun_out/r3
ASGART_TRACE_ALLOC=1 ASGART_DEBUG=1 python3 tools/index_build.py cfg4 3 > gpurun_out/r3/index_build.log 2>&1
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r3/gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3/gpu_tests.log
