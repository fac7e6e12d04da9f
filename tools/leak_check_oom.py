"""Which injected allocation failure (option test_fail_alloc = n) leaves device memory behind after the index is closed."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import asgart_amd  # noqa: E402
from asgart_amd import prep, synth  # noqa: E402

pr = prep.prepare_records(synth.make_genome([150_000], seed=11, sd_per_mb=10, sd_len=(1000, 5000), alu_frac=0.3, l1_frac=0.0,
                                            sat_per_record=0))
st = asgart_amd.RunSettings.from_cli(reverse=True, complement=True)
import oracle
SA = oracle.Index.build(pr.data).sa if os.environ.get("LEAK_HOST_SA") else None


def free_bytes():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]


for sweep in range(3):
    prev = free_bytes()
    for n in range(70):
        idx = asgart_amd.Index(pr.data, SA)
        what = "ok"
        try:
            idx.set_option("test_fail_alloc", n)
            try:
                idx.prepare(20)
                idx.search_duplications_raw(pr.chunks, st)
            except asgart_amd.AsgartError as e:
                what = str(e)[:100]
            idx.set_option("test_fail_alloc", -1)
        finally:
            idx.close()
        f = free_bytes()
        if f != prev:
            print(f"sweep {sweep} n={n}: free changed by {(f - prev) / 2**20:+.2f} MiB  [{what}]", flush=True)
        prev = f
    print(f"sweep {sweep} done: free {prev}", flush=True)
