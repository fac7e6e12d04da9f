mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "escalation or generation_wrap or random_sweep or cap or cascade or tier6" > gpurun_out/r3/t6.log 2>&1; echo "rc=$?" >> gpurun_out/r3/t6.log
ASGART_FAST=72 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "escalation or generation_wrap or random_sweep or cap or cascade or tier6 or cfg4" > gpurun_out/r3/t6b.log 2>&1; echo "rc=$?" >> gpurun_out/r3/t6b.log
TUNE_REPS=6 timeout 2400 python tools/tune_tiers.py cfg4 --pipelined "fast=8" "fast=72" "fast=72 long3=8192" "fast=72 long3=16384" > gpurun_out/r3/tune_f6.log 2>&1; echo "rc=$?" >> gpurun_out/r3/tune_f6.log
ASGART_DEBUG=1 timeout 1500 python tools/tune_tiers.py cfg4 "fast=8" "fast=72" > gpurun_out/r3/tune_f6b.log 2>&1
