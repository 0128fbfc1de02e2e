#!/usr/bin/env python
"""Summarise a tools/profile_bench.sh output directory: per-kernel time (only this repo's kernels) and PMC
counters averaged per launch.  Usage: summarize_prof.py gpurun_out/prof_<tag> [--json out.json]"""
import csv, glob, json, os, re, sys
from collections import defaultdict

def short(n):
    """this repository's kernel name without its template arguments -- except the count kernel's MODE (k_count_bf16<1>)"""
    m = re.search(r"(k_[a-z_0-9]+)(<\s*(\d+)[^>]*>)?", n)
    if not m:
        return None
    return "%s<%s>" % (m.group(1), m.group(3)) if m.group(1) == "k_count_bf16" and m.group(3) else m.group(1)

def main():
    d = sys.argv[1]
    out = {}
    for f in glob.glob(os.path.join(d, "trace", "*kernel_stats.csv")):
        rows = list(csv.DictReader(open(f)))
        tot = sum(float(r["TotalDurationNs"]) for r in rows if short(r["Name"]))
        print("%-22s %6s %12s %10s %10s %10s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "share"))
        for r in rows:
            s = short(r["Name"])
            if not s: continue
            print("%-22s %6s %12.2f %10.2f %10.2f %9.1f%%" % (s, r["Calls"], float(r["AverageNs"]) / 1e3,
                  float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
            out.setdefault("kernel_stats", {})[s] = dict(calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3,
                                                         min_us=float(r["MinNs"]) / 1e3, max_us=float(r["MaxNs"]) / 1e3)
    # steady-state duration of the count kernel at the bench batch: take launches with the modal grid/duration
    for f in glob.glob(os.path.join(d, "trace", "*kernel_trace.csv")):
        durs = defaultdict(list)
        for r in csv.DictReader(open(f)):
            s = short(r["Kernel_Name"])
            if s: durs[s].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for s, v in durs.items():
            v = sorted(v)
            big = [x for x in v if x > 0.5 * v[-1]]
            print("trace %-20s n=%4d median_us=%9.2f  top-half-mean_us=%9.2f (n=%d)" % (s, len(v), v[len(v) // 2], sum(big) / len(big), len(big)))
            out.setdefault("trace", {})[s] = dict(n=len(v), median_us=v[len(v) // 2], main_mean_us=sum(big) / len(big), main_n=len(big))
    pmc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "pmc*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            s = short(r["Kernel_Name"])
            if s: pmc[s][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for s in sorted(pmc):
        print("pmc", s)
        for c, v in sorted(pmc[s].items()):
            v = sorted(v)
            # the launches of the bench batch: within a factor two of the 90th percentile (one outlier -- a launch that
            # shared the chip with something else -- must not define the scale, as a plain "> max / 2" did)
            ref = v[min(len(v) - 1, int(0.9 * len(v)))]
            big = [x for x in v if 0.5 * ref < x < 2.0 * ref] or v
            print("    %-22s n=%4d  max=%.4g  main-mean=%.4g" % (c, len(v), v[-1], sum(big) / len(big)))
            out.setdefault("pmc", {}).setdefault(s, {})[c] = dict(n=len(v), max=v[-1], main_mean=sum(big) / len(big))
    if "--json" in sys.argv:
        json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)

main()
