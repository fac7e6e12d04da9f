"""Stress loop for the stall that one GPU test run of round 3 sat in for 40 minutes: the forced-tier, sweep and passes
tests, each repetition in a FRESH process (a first call on a fresh runtime was part of the one observation), with the
library's watchdog at a few seconds so that a stuck call comes back with the phase and the workgroups in flight instead
of hanging.  Any non-zero exit or time-out stops the loop and keeps the log.

    python tools/stress_tiers.py [repetitions=200] [per-run time-out in s=300]
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 300
env = dict(os.environ, ASGART_WATCHDOG_S="20")
t_all = time.time()
for r in range(reps):
    t0 = time.time()
    try:
        out = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-x", "-q", "-m", "gpu", "-k",
                              "tier or passes or sweep", "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True,
                             text=True, timeout=limit)
    except subprocess.TimeoutExpired as e:
        print(f"repetition {r}: TIMED OUT after {limit} s\n{(e.stdout or b'')[-3000:]}\n{(e.stderr or b'')[-3000:]}", flush=True)
        sys.exit(2)
    tail = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else ""
    print(f"repetition {r}: rc={out.returncode} {time.time() - t0:.1f}s  {tail}", flush=True)
    if out.returncode != 0:
        print(out.stdout[-4000:], out.stderr[-2000:], flush=True)
        sys.exit(1)
print(f"{reps} repetitions, {time.time() - t_all:.0f} s, no stall, no failure")
