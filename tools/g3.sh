timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "long_probes or k22 or k31 or k42 or errors or trim_with_long or searcher or k21 or k12 or tail_corner" 2>&1 | tail -15
