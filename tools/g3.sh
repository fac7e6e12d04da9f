export TMPDIR=/tmp
for i in 1 2; do python bench.py --workload cfg4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('early', d['value'], d['ms_per_step'], d['config']['passes_issued'], d['config']['mode_probe_ms'])"; done
ASGART_PROGRESS_LATE=1 python bench.py --workload cfg4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('late', d['value'], d['ms_per_step'], d['config']['passes_issued'], d['config']['mode_probe_ms'])"
python bench.py --workload cfg3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['passes_issued'], d['config']['mode_probe_ms'])"
