#!/usr/bin/env python3
"""Timeline of the LAST config-4 pass in a rocprofv3 kernel trace (csv): start, duration, idle gap, kernel."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
idx = [i for i, e in enumerate(ev) if "sparse_finish_kernel" in e[2]]
last = idx[-24:]
lo, hi = max(0, last[0] - int(sys.argv[2]) if len(sys.argv) > 2 else 60), min(len(ev), last[-1] + (int(sys.argv[3]) if len(sys.argv) > 3 else 60))
t0, busy_end = ev[lo][0], None
for s, e, n in ev[lo:hi]:
    short = n.replace("void ", "").replace("(anonymous namespace)::", "")[:64]
    gap = (s - busy_end) / 1e3 if busy_end else 0.0
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} gap{gap:7.1f}  {short}")
    busy_end = max(busy_end or e, e)
