mkdir -p gpurun_out/r3
ASGART_LIB=asgart_amd/libasgart_hip_diag.so timeout 1200 python tools/diag_cfg.py cfg4 > gpurun_out/r3/diag_cfg4.log 2>&1
