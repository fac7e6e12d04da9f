mkdir -p gpurun_out/r3
timeout 1500 python -m pytest tests -x -q -m gpu -k "probe_hits or families_match or battery or cfg2 or cfg3 or long_probes or trim or tail" > gpurun_out/r3/search_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3/search_tests.log
timeout 1500 python tools/tune_tiers.py cfg4 "split_search=1" "split_search=0" > gpurun_out/r3/tune_split.log 2>&1
TUNE_REPS=6 timeout 1500 python tools/tune_tiers.py cfg4 --pipelined "split_search=1" "split_search=0" >> gpurun_out/r3/tune_split.log 2>&1
