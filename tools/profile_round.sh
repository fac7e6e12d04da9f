#!/bin/bash
# Round profile of one workload: default bench (one JSON line), rocprofv3 kernel statistics of the same command,
# and the two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only -- see MI355X_MICROARCH.md, HBM).
# Usage (on the GPU box, from the repo root):  bash tools/profile_round.sh r02 cfg4 [bench args...]
# Outputs land in gpurun_out/<tag>_<workload>_*; tools/summarize_prof.py condenses them into gpurun_out/profiles/
# (copy what should be judged into profiles/).
set -u
TAG=${1:-r02}; shift || true
WL=${1:-cfg4}; shift || true
OUT=$PWD/gpurun_out
P=${TAG}_${WL}
mkdir -p "$OUT"
export TMPDIR=/tmp
if [ -z "${SKIP_BENCH:-}" ]; then
python3 bench.py --workload "$WL" "$@" > "$OUT/${P}_bench.json" 2> "$OUT/${P}_bench_err.log"
echo "bench rc=$?"; tail -c 400 "$OUT/${P}_bench.json"
fi
# The profiled runs build the presence filters and the position-sorted lists at once (the default builds them on second
# use): every launch of the search kernels in them is then a steady-state launch, and the per-launch averages of the
# counters describe the kernels of the timed region, not a mix with the two unfiltered launches of a cold start.
export ASGART_LAZY_AUX=0
# ... and every step is issued through the passes call only (no back-to-back probe step with its single-pass launches):
# a launch of the search kernels is then always the passes of a step as ONE job
export ASGART_BENCH_MODE=library
# ... always as ONE job (option fuse_passes = 2): the library's default may time a few early calls as pipelined single-pass
# calls (fuse_passes = 1, once one segment is a call's extension), whose launches are half the size -- the timed region of a
# default bench run lies behind that and is all one-job calls
export ASGART_FUSE_PASSES=2
PB="--steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/${P}_stats" -o run -- python3 bench.py --workload "$WL" $PB > "$OUT/${P}_stats_bench.json" 2> "$OUT/${P}_stats_err.log"
echo "stats rc=$?"
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d "$OUT/${P}_fetch" -o run -- python3 bench.py --workload "$WL" $PB > /dev/null 2> "$OUT/${P}_fetch_err.log"
echo "fetch rc=$?"
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d "$OUT/${P}_write" -o run -- python3 bench.py --workload "$WL" $PB > /dev/null 2> "$OUT/${P}_write_err.log"
echo "write rc=$?"
python3 tools/summarize_prof.py "$TAG" "$OUT" "$WL" || true
# raw traces are scratch (gpurun copies back at most 64 MiB): keep the summaries only
find "$OUT/${P}_stats" "$OUT/${P}_fetch" "$OUT/${P}_write" -type f ! -name '*kernel_stats.csv' -delete 2>/dev/null
