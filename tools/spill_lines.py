"""Where a kernel's register spills sit, by source line: compiles one HIP source for gfx950 with line tables, finds the
scratch_store_* / scratch_load_* instructions (vector-register spills and reloads) of every kernel whose mangled name
contains the pattern, and counts them per (file, line).

    python tools/spill_lines.py asgart_amd/csrc/pipeline.hip extend_k8_kernel [> profiles/rNN_k8_spill_lines.txt]
"""
import re
import subprocess
import sys
import tempfile
from collections import Counter

src, pat = sys.argv[1], sys.argv[2]
with tempfile.TemporaryDirectory() as tmp:
    asm = tmp + "/k.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--offload-device-only",
                    "-gline-tables-only", "-S", "-o", asm, src], check=True, stderr=subprocess.DEVNULL)
    txt = open(asm).read()
files = {}
for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', txt):
    files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
for name in re.findall(r"^(_Z\w*%s\w*):" % re.escape(pat), txt, flags=re.M):
    i = txt.index("\n" + name + ":")
    body = txt[i:txt.index("s_endpgm", i)].split("\n")
    cur, n_ins = None, 0
    per_line, kinds = Counter(), Counter()
    for line in body:
        t = line.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            continue
        if not t or t[0] in ".;" or t.endswith(":"):
            continue
        n_ins += 1
        op = t.split()[0]
        if op.startswith("scratch_"):
            per_line[(cur, op.split("_")[1])] += 1
            kinds[op] += 1
    print(f"{name}\n  {n_ins} instructions; {dict(kinds)}")
    for (loc, kind), c in per_line.most_common():
        print(f"    {loc[0]}:{loc[1]:<6} {kind:5} x {c}")
