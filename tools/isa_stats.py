"""Static instruction statistics of the kernels of one HIP source (gfx950 assembly from hipcc -S): total, VALU, SALU,
LDS, branches, lane spills of scalar registers (v_readlane / v_writelane), barriers, waits, scratch accesses.

    python tools/isa_stats.py asgart_amd/csrc/pipeline.hip extend_fast_kernel
"""
import re
import subprocess
import sys
import tempfile

src = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as d:
    out = d + "/k.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "--offload-device-only",
                    "-S", "-o", out, src], check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
name, body = None, []
rows = []


def flush():
    if name and want in name:
        ins = [l.strip() for l in body if l.startswith("\t") and not l.strip().startswith((".", ";"))]

        def cnt(p):
            return sum(1 for l in ins if re.match(p, l))

        rows.append((name, len(ins), cnt(r"v_"), cnt(r"s_(?!waitcnt|barrier|cbranch|branch|nop|endpgm)"), cnt(r"ds_"),
                     cnt(r"s_c?branch"), cnt(r"v_readlane"), cnt(r"v_writelane"), cnt(r"s_barrier"), cnt(r"s_waitcnt"),
                     cnt(r"scratch_"), cnt(r"global_|flat_|buffer_")))


for l in lines:
    m = re.match(r"(_Z\w+):\s*(;.*)?$", l)
    if m:
        flush()
        name, body = m.group(1), []
    elif l.startswith(".Lfunc_end"):
        flush()
        name, body = None, []
    elif name:
        body.append(l)
flush()
print("%-64s %6s %6s %6s %5s %6s %8s %9s %7s %7s %7s %6s" % ("kernel", "instr", "VALU", "SALU", "LDS", "branch", "readlane",
                                                               "writelane", "barrier", "waitcnt", "scratch", "vmem"))
for r in rows:
    print("%-64s %6d %6d %6d %5d %6d %8d %9d %7d %7d %7d %6d" % ((re.sub(r"^_ZN6asgart\d+", "", r[0])[:64],) + r[1:]))
