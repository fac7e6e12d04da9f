export TUNE_REPS=9
timeout 1500 python3 tools/tune_tiers.py cfg4 --pipelined '' 'progress_at=1' '' 'progress_at=1' 2>&1 | grep -v Warn
