import os, sys, time, hashlib, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ASGART_FORCE_WIDE"] = "1"
import asgart_amd
from asgart_amd import prep, synth
cfg, scale = int(sys.argv[1]), float(sys.argv[2])
pr = prep.prepare_records(synth.config_genome(cfg, scale))
n = len(pr.data)
t0 = time.time()
idx = asgart_amd.Index(pr.data, None)
print(f"cfg{cfg} x{scale}: n={n} wide SA {time.time()-t0:.1f}s", flush=True)
print("verifier violations:", idx.check_sa(), flush=True)
if cfg == 4 and scale == 1.0:
    d = json.load(open(os.path.join(ROOT, "tests/golden/digests.json")))["cfg4"]
    h = hashlib.sha256(); slab = 1 << 26
    for o in range(0, n, slab):
        h.update(idx.sa_read(o, min(n, o + slab)).astype("<u4").tobytes())
    print("sha matches oracle SA-IS digest:", h.hexdigest() == d["sa_sha256_u32"], flush=True)
