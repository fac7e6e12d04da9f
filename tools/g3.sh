export TMPDIR=/tmp
python tools/pole_synth.py --copies 3800 --sub 0.03 --check '' 'force_tier=6' 'force_tier=5' 2>&1 | grep direct
python tools/tune_tiers.py cfg4 '' 2>&1 | tail -1
python -m pytest tests -m gpu -x -q -k "escalation or battery or wrap or levels or 64bit" 2>&1 | tail -3
