export TMPDIR=/tmp
python tools/pole_synth.py --copies 3800 --sub 0.03 --check '' 'test_levels=0' 'test_levels=1' 'test_levels=2' 'test_levels=3' 'force_tier=2' 'force_tier=4' 2>&1 | tail -9
echo "== diag sub 0.03"
ASGART_LIB=asgart_amd/libasgart_hip_diag.so python tools/pole_synth.py --copies 3800 --sub 0.03 2>&1 | grep -A3 "^\[extend profile 3" | head -5
python -m pytest tests -m gpu -x -q -k "not cfg4_full and not cfg3_full" 2>&1 | tail -3
