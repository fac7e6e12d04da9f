"""Diagnostic: which injected allocation failure (option test_fail_alloc = n) leaves device memory behind.
Usage: python tools/oom_sweep.py   (on the GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import asgart_amd, oracle
from asgart_amd import prep
import test_gpu_parity as T

pr, cli = T._battery_case("dense_repeats")
oidx = oracle.Index.build(pr.data)
st = asgart_amd.RunSettings.from_cli(reverse=True, complement=True, **cli)

def free_bytes():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]

for sweep in range(3):
    line = []
    for n in range(0, 70):
        f0 = free_bytes()
        idx = asgart_amd.Index(pr.data, oidx.sa)
        out = "ok"
        try:
            idx.set_option("test_fail_alloc", n)
            try:
                idx.prepare(20)
                idx.search_duplications_raw(pr.chunks, st)
            except asgart_amd.AsgartError as e:
                out = "E%d" % e.code
            idx.set_option("test_fail_alloc", -1)
        finally:
            idx.close()
        d = f0 - free_bytes()
        if d or sweep == 0:
            line.append("%d:%s:%+.1f" % (n, out, d / 2**20))
    print("sweep", sweep, " ".join(line), flush=True)
