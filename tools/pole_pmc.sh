#!/bin/bash
# PMC counters of the arm kernel on the isolated tandem-array segment (tools/pole_synth.py), one rocprofv3 pass per
# counter group (kernel trace only -- no other trace domains with --pmc on this pool).
# Usage (GPU box, repo root):  bash tools/pole_pmc.sh <tag> '<pole_synth option string, e.g. force_tier=4>'
# Output: gpurun_out/<tag>_pole_pmc.json (summarised by tools/pole_pmc_sum.py)
set -u
TAG=${1:-r04}; CONF=${2:-}
OUT=$PWD/gpurun_out/${TAG}_pole_pmc
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for GROUP in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
             "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES" \
             "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
             "SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_BRANCH SQ_INSTS_CBRANCH" \
             "SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  rocprofv3 --output-format csv --kernel-trace --pmc $GROUP -d "$OUT/g$i" -o run -- python3 tools/pole_synth.py "$CONF" > "$OUT/g$i.log" 2>&1
  echo "group $i rc=$?"
done
python3 tools/pole_pmc_sum.py "$OUT" "$CONF" > "$PWD/gpurun_out/${TAG}_pole_pmc.json"
find "$OUT" -type f -name '*.csv' ! -name '*counter_collection.csv' -delete 2>/dev/null
cat "$PWD/gpurun_out/${TAG}_pole_pmc.json"
