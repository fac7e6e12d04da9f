"""Free device memory across create -> prepare -> search -> destroy cycles of an index (a leak of the teardown shows as a
steady decline).   python tools/leak_check.py [cycles=60] [rounds=4]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import asgart_amd  # noqa: E402
from asgart_amd import prep, synth  # noqa: E402

cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
pr = prep.prepare_records(synth.make_genome([150_000], seed=11, sd_per_mb=10, sd_len=(1000, 5000), alu_frac=0.3, l1_frac=0.0,
                                            sat_per_record=0))
st = asgart_amd.RunSettings.from_cli(reverse=True, complement=True)


def free_bytes():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]


base = None
for r in range(rounds):
    t0 = time.time()
    for _ in range(cycles):
        with asgart_amd.Index(pr.data, None) as idx:
            idx.search_duplications_raw(pr.chunks, st)
    f = free_bytes()
    base = f if base is None else base
    print(f"round {r}: free {f}  ({(f - base) / 2**20:+.1f} MiB vs round 0), {time.time() - t0:.1f} s", flush=True)
