#!/usr/bin/env python3
"""Timeline of one bench step from a rocprofv3 kernel trace: start, duration and the idle gap before every launch
of the main stream (the stream with the most launches).   python tools/step_timeline.py <t_kernel_trace.csv> [step]"""
import csv
import sys
from collections import Counter

rows = list(csv.DictReader(open(sys.argv[1])))
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
q = Counter(r["Queue_Id"] for r in rows).most_common(1)[0][0]
rows = sorted((r for r in rows if r["Queue_Id"] == q), key=lambda r: int(r["Start_Timestamp"]))
first = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void dsic::conv_first") or "conv_first_kernel" in r["Kernel_Name"]]
lo, hi = first[which], first[which + 1] if which + 1 < 0 or which + 1 < len(first) else len(rows)
prev_end, t0, busy, gaps = None, int(rows[lo]["Start_Timestamp"]), 0, 0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0 if prev_end is None else s - prev_end
    name = r["Kernel_Name"].replace("void ", "").split("(")[0][:60]
    print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  gap {gap / 1e3:6.1f}  grid {r['Grid_Size_X']:>8s}  {name}")
    busy += e - s
    gaps += max(gap, 0)
    prev_end = max(e, prev_end or 0)
print(f"step: {(prev_end - t0) / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, gaps {gaps / 1e3:.1f} us, {hi - lo} launches")
