"""Timeline of the product kernels of the LAST bench step, from a rocprofv3 --kernel-trace CSV.

Usage: python tools/timeline.py <dir-or-csv> [window_ms]
Prints start/end (ms, relative to the first kernel of the window) per dispatch of the long kernels, with queue id.
"""
import csv
import glob
import os
import sys


def short(n):
    n = n.replace("void ", "").replace("asgart::", "").replace("(anonymous namespace)::", "")
    if "rocprim" in n:
        return "rocprim"
    return n.split("(")[0][:70]


def main():
    src = sys.argv[1]
    win = float(sys.argv[2]) if len(sys.argv) > 2 else 800.0
    if os.path.isdir(src):
        src = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = []
    for r in csv.DictReader(open(src)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"),
                     r.get("Stream_Id", "?")))
    rows.sort()
    t_end = max(r[1] for r in rows)
    rows = [r for r in rows if r[0] >= t_end - win * 1e6]
    t0 = rows[0][0]
    for s, e, n, q, st in rows:
        d = (e - s) / 1e6
        if d < 0.1:
            continue
        print(f"{(s - t0) / 1e6:9.2f} {(e - t0) / 1e6:9.2f} {d:8.2f} q{q:>3} s{st:>3} {short(n)}")


if __name__ == "__main__":
    main()
