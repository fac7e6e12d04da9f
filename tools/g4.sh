timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 600 python3 bench.py --workload cfg4 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg4', d['value'], d['ms_per_step'], d['config']['mode_probe_ms'], d['phases_ms_per_step'])"
timeout 600 python3 bench.py --workload cfg3 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg3', d['value'], d['ms_per_step'], d['config']['mode_probe_ms'], d['phases_ms_per_step'])"
timeout 600 python3 bench.py --workload cfg2 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg2', d['value'], d['ms_per_step'], d['config']['mode_probe_ms'], d['phases_ms_per_step'])"
