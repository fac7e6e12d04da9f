timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "thousands_of_live_arms" 2>&1 | tail -5
