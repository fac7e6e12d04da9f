mkdir -p gpurun_out/r3
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r3/gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3/gpu_tests.log
timeout 900 python tools/pole_synth.py --check "fast=8" "fast=0" > gpurun_out/r3/pole.log 2>&1; echo "rc=$?" >> gpurun_out/r3/pole.log
ASGART_DEBUG=1 timeout 1500 python tools/tune_tiers.py cfg4 "fast=8" "fast=0" > gpurun_out/r3/tune_cfg4.log 2>&1; echo "rc=$?" >> gpurun_out/r3/tune_cfg4.log
timeout 1500 python tools/tune_tiers.py cfg4 --pipelined "fast=8" "fast=0" > gpurun_out/r3/tune_cfg4_pipe.log 2>&1; echo "rc=$?" >> gpurun_out/r3/tune_cfg4_pipe.log
