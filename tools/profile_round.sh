#!/bin/bash
# Round profile: default bench (one JSON line), rocprofv3 kernel statistics of the same command, and the two
# PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only -- see MI355X_MICROARCH.md, HBM).
# Usage (on the GPU box, from the repo root):  bash tools/profile_round.sh r01 [bench args...]
# Outputs land in gpurun_out/<tag>_*; tools/summarize_prof.py condenses them into profiles/.
set -u
TAG=${1:-r01}; shift || true
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 bench.py "$@" > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench_err.log"
echo "bench rc=$?"; tail -c 600 "$OUT/${TAG}_bench.json"
PB="--steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/${TAG}_stats" -o run -- python3 bench.py "$@" > "$OUT/${TAG}_stats_bench.json" 2> "$OUT/${TAG}_stats_err.log"
echo "stats rc=$?"
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d "$OUT/${TAG}_fetch" -o run -- python3 bench.py $PB "$@" > /dev/null 2> "$OUT/${TAG}_fetch_err.log"
echo "fetch rc=$?"
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d "$OUT/${TAG}_write" -o run -- python3 bench.py $PB "$@" > /dev/null 2> "$OUT/${TAG}_write_err.log"
echo "write rc=$?"
python3 tools/summarize_prof.py "$TAG" "$OUT" || true
# raw traces are scratch (gpurun copies back at most 64 MiB): keep the summaries only
du -sh "$OUT/${TAG}_stats" "$OUT/${TAG}_fetch" "$OUT/${TAG}_write" 2>/dev/null
find "$OUT/${TAG}_stats" "$OUT/${TAG}_fetch" "$OUT/${TAG}_write" -type f ! -name '*kernel_stats.csv' -delete 2>/dev/null
