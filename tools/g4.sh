export TUNE_REPS=7
timeout 900 python3 -m pytest tests -x -q -m gpu -k "tiers or cascade or sweep or 64bit or generation or battery or cardinality or cfg5_shaped or shards" 2>&1 | tail -3
timeout 1500 python3 tools/tune_tiers.py cfg4 '' 2>&1 | grep -v Warn
timeout 1500 python3 tools/tune_tiers.py cfg4 --pipelined '' 2>&1 | grep -v Warn
timeout 1400 python3 tools/cfg5_direct.py 2>&1 | grep -v Warn | tail -2
