export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -k "not cfg4_full" 2>&1 | tail -3
ASGART_BENCH_OVERLAP=0 python bench.py --workload cfg4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('seq', d['value'], d['ms_per_step'], d['phases_ms_per_step'])"
python bench.py --workload cfg4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ovl', d['value'], d['ms_per_step'], d['phases_ms_per_step'])"
