"""Registers, spills, LDS and occupancy of every kernel of one HIP source (gfx950), from the compiler's own remarks.

    python tools/kernel_resources.py asgart_amd/csrc/pipeline.hip [substring of the kernel name]
"""
import re
import subprocess
import sys

src = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "--offload-device-only", "-c", src,
       "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"remark: (?:\S+: )?\s*(.*?) \[-Rpass-analysis", line)
    if not m:
        continue
    body = m.group(1).strip()
    if body.startswith("Function Name:"):
        cur = body.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in body:
        k, v = body.rsplit(":", 1)
        rows[cur][k.strip()] = v.strip()
for name, r in rows.items():
    if want not in name:
        continue
    short = re.sub(r"^_ZN6asgart\d+", "", name)
    print("%-70s VGPR %4s AGPR %3s spill V %3s S %4s scratch %4s LDS %7s occ %s" % (
        short[:70], r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("VGPRs Spill", "?"), r.get("SGPRs Spill", "?"),
        r.get("ScratchSize [bytes/lane]", "?"), r.get("LDS Size [bytes/block]", "?"), r.get("Occupancy [waves/SIMD]", "?")))
