#!/usr/bin/env python3
"""Timeline of the LAST screened config-4 pass in a rocprofv3 kernel trace (csv): start, duration, idle gap, kernel.
usage: c4_timeline.py <kernel_trace.csv> [marker kernel substring, default encode_group_kernel / emit_persistent]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
marks = sys.argv[2:] or ["encode_group_kernel", "emit_persistent"]
g = [i for i, e in enumerate(ev) if any(m in e[2] for m in marks)]
lo, hi = g[-2] + 1, g[-1]
while hi + 1 < len(ev) and "lens_assemble" not in ev[hi][2]:
    hi += 1
t0, busy = ev[lo][0], None
for s, e, n in ev[lo:hi + 3]:
    short = n.replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "")[:90]
    gap = (s - busy) / 1e3 if busy else 0.0
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} gap{gap:7.1f}  {short}")
    busy = max(busy or e, e)
