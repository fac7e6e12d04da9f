export TMPDIR=/tmp
mkdir -p gpurun_out/g2
python tools/pole_synth.py --copies 3800 --sub 0.03 --check '' 'arms2=0' 'force_tier=6' 'force_tier=5' > gpurun_out/g2/pole.log 2>&1; grep direct gpurun_out/g2/pole.log
ASGART_LIB=asgart_amd/libasgart_hip_diag.so python tools/pole_synth.py --copies 3800 --sub 0.03 '' > gpurun_out/g2/diag.log 2>&1; grep -A3 "^\[extend profile 3" gpurun_out/g2/diag.log | grep "longest slots" | head -1
timeout 900 python -m pytest tests -m gpu -x -q -k "not cfg4_full" 2>&1 | tail -8
