export TMPDIR=/tmp
ASGART_LIB=asgart_amd/libasgart_hip_diag.so python tools/pole_synth.py --copies 3800 --sub 0.06 '' 2>&1 | grep -A3 "^\[extend profile 3" | grep "longest slots" | head -1
