#!/usr/bin/env python
"""Per-kernel duration and the gap before each kernel inside one voting call, from a rocprofv3 --kernel-trace CSV of
tools/trace_calls.py.   usage: trace_gaps.py <dir or csv> [--json out.json] [--skip N first calls]"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(n):
    """this repository's kernel name without its template arguments -- except the count kernel's MODE (k_count_bf16<1>)"""
    m = re.search(r"(k_[a-z_0-9]+)(<\s*(\d+)[^>]*>)?", n)
    if not m:
        return None
    return "%s<%s>" % (m.group(1), m.group(3)) if m.group(1) == "k_count_bf16" and m.group(3) else m.group(1)


def main():
    src = sys.argv[1]
    skip = int(sys.argv[sys.argv.index("--skip") + 1]) if "--skip" in sys.argv else 10
    f = src if src.endswith(".csv") else sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(f))]
    rows = sorted(r for r in rows if r[2])
    calls, cur = [], []
    for r in rows:
        if r[2].startswith("k_tile_count") or r[2].startswith("k_tile_scan") or r[2].startswith("k_front"):
            if cur:
                calls.append(cur)
            cur = []
        cur.append(r)
    if cur:
        calls.append(cur)
    calls = calls[skip:]
    dur, gap = defaultdict(list), defaultdict(list)
    span, period = [], []
    for i, c in enumerate(calls):
        for j, (s, e, n) in enumerate(c):
            key = "%d:%s" % (j, n)
            dur[key].append((e - s) / 1e3)
            if j:
                gap[key].append((s - c[j - 1][1]) / 1e3)
            elif i:
                gap[key].append((s - calls[i - 1][-1][1]) / 1e3)
        span.append((c[-1][1] - c[0][0]) / 1e3)
        if i:
            period.append((c[0][0] - calls[i - 1][0][0]) / 1e3)
    med = lambda v: sorted(v)[len(v) // 2] if v else None  # noqa: E731
    out = {"calls": len(calls), "span_us_median": med(span), "period_us_median": med(period), "kernels": {}}
    print("%d calls, span (first start -> last end) median %.2f us, call period median %.2f us" % (
        len(calls), med(span), med(period) or 0.0))
    print("%-28s %6s %10s %10s" % ("kernel", "n", "dur_us", "gap_before_us"))
    for key in sorted(dur, key=lambda k: int(k.split(":")[0])):
        print("%-28s %6d %10.2f %10.2f" % (key, len(dur[key]), med(dur[key]), med(gap[key]) or 0.0))
        out["kernels"][key] = {"n": len(dur[key]), "dur_us_median": med(dur[key]), "gap_before_us_median": med(gap[key])}
    print("sum of durations %.2f us, sum of gaps %.2f us" % (sum(med(v) for v in dur.values()),
                                                            sum(med(v) for k, v in gap.items() if not k.startswith("0:"))))
    if "--json" in sys.argv:
        json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


main()
