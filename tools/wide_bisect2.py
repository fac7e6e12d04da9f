import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import asgart_amd, oracle
from asgart_amd import prep, synth
pr = prep.prepare_records(synth.config_genome(5, float(sys.argv[1])))
n = len(pr.data)
idx = asgart_amd.Index(pr.data, None)
print("n", n, "gpu verifier:", idx.check_sa(), flush=True)
sa = np.empty(n, dtype=np.int64)
slab = 1 << 28
for o in range(0, n, slab):
    sa[o:o + slab] = idx.sa_read(o, min(n, o + slab))
print("sa read; min/max", int(sa.min()), int(sa.max()), flush=True)
cnt = np.bincount((sa >> 28).astype(np.int64), minlength=(n >> 28) + 1)
print("entries per 2^28 block of positions (expect 2^28 each):", cnt.tolist(), flush=True)
t0 = time.time()
r = oracle.sa_check(pr.data, sa)
print("oracle sa_check ->", r, f"({time.time()-t0:.0f}s)", flush=True)
if r > 0:
    s = r - 1
    print("around bad slot", s, sa[max(0, s - 3):s + 3].tolist(), [bytes(pr.data[int(x):int(x) + 30]) for x in sa[max(0, s - 2):s + 2]])
