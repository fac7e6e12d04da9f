mkdir -p gpurun_out/r3
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r3/pmc1 gpurun_out/r3/pmc2
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES -d gpurun_out/r3/pmc1 -o pole --output-format csv -- python3 tools/pole_synth.py "fast=8 cap6_pct=400 fast_nt=256 fast_s=8" "fast=8 cap6_pct=400 fast_nt=512 fast_s=4" "fast=0" > gpurun_out/r3/pmc1.log 2>&1
