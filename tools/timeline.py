#!/usr/bin/env python
"""Kernel timeline of a rocprofv3 --kernel-trace CSV: the period between consecutive k_tile_scan starts (mean per group of
N calls, so the phases of a bench run -- pre-warm, timed steps, other legs -- can be told apart), the idle time on the
device inside each period, and every kernel that is not one of the voting kernels (copies show up as kernels only when the
runtime uses blit kernels).   usage: timeline.py <dir or csv> [group]"""
import csv
import glob
import os
import re
import sys
from collections import Counter

src = sys.argv[1]
grp = int(sys.argv[2]) if len(sys.argv) > 2 else 20
f = src if src.endswith(".csv") else sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
other = Counter()
scans, busy = [], []
cur_busy, last_end = 0, None
for s, e, n in rows:
    if "k_tile_scan" in n:
        if scans:
            busy.append(cur_busy)
        scans.append(s)
        cur_busy = 0
    if not re.search(r"k_(tile|compact|count|lead|select|finalize|covariance|stream)", n):
        other[re.sub(r"<.*", "", n)[:70]] += 1
    cur_busy += e - max(s, last_end or s) if last_end and s < last_end else e - s
    last_end = max(last_end or e, e)
per = [(scans[i + 1] - scans[i]) / 1e3 for i in range(len(scans) - 1)]
print("calls: %d" % len(scans))
for i in range(0, len(per), grp):
    p, b = per[i:i + grp], busy[i:i + grp]
    print("calls %4d-%4d  period mean %7.1f us  min %7.1f  max %8.1f   busy mean %6.1f us" %
          (i, i + len(p) - 1, sum(p) / len(p), min(p), max(p), sum(b) / len(b) / 1e3))
print("other kernels:", dict(other.most_common(12)))
