mkdir -p gpurun_out/r3
timeout 2400 python -m pytest tests -x -q -m gpu -k "more_live_arms or tier6 or escalation or cascade or cap or large_max_card or 64bit" > gpurun_out/r3/cap_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3/cap_tests.log
