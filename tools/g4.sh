timeout 1500 python3 -m pytest tests -x -q -m gpu -k "not cfg4 and not cfg3 and not cfg5 and not wide_suffix" 2>&1 | tail -3
ASGART_BENCH_MODE=back_to_back timeout 600 python3 bench.py --workload cfg4 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('b2b', d['ms_per_step'], d['phases_ms_per_step'], d['roofline'])"
