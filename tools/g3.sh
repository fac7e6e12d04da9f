export TMPDIR=/tmp
mkdir -p gpurun_out/g3
python -m pytest tests -m gpu -x -q -k "wide or cfg5 or searcher or tail" 2>&1 | tail -3
timeout 1500 python tools/cfg5_check.py > gpurun_out/g3/cfg5.log 2>&1; echo rc=$?; tail -15 gpurun_out/g3/cfg5.log
