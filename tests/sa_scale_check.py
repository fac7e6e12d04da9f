import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, asgart_amd, oracle
from asgart_amd import prep, synth
for cfg, scale in ((2, 1.0), (3, 0.2), (3, 1.0)):
    t=time.time(); recs = synth.config_genome(cfg, scale); pr = prep.prepare_records(recs); tg=time.time()-t
    t=time.time(); sa = asgart_amd.sa_build64(pr.data); tb=time.time()-t
    t=time.time(); ok = oracle.sa_check(pr.data, sa); tc=time.time()-t
    print(f"cfg{cfg} x{scale}: n={len(pr.data)} gen {tg:.1f}s gpu-sa {tb:.2f}s check={ok} ({tc:.1f}s)", flush=True)
