"""Diagnostic build only: per-tier phase statistics of one pass (tiers run one after the other).
Usage: ASGART_LIB=asgart_amd/libasgart_hip_diag.so [DIAG_SCALE=0.1] python tools/diag_cfg.py cfg4 [rc]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import asgart_amd
from asgart_amd import prep, synth
cfg = int(sys.argv[1][3:]); rc = len(sys.argv) > 2 and sys.argv[2] == "rc"
pr = prep.prepare_records(synth.config_genome(cfg, float(os.environ.get("DIAG_SCALE", "1.0"))))
idx = asgart_amd.Index(pr.data, None); idx.prepare(20)
st = asgart_amd.RunSettings.from_cli(reverse=rc, complement=rc)
idx.search_duplications_raw(pr.chunks, st)
idx.close()
