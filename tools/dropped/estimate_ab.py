#!/usr/bin/env python
"""(Round 3, dropped -- DESIGN.md 4.6: this timed a build whose ESTIMATE could count in stages too; the product counts the estimate
in full, so on the shipped library the three modes are the same kernel.)
Full vs staged count pass of estimate_voting_distribution_with_mean (4096 hypotheses), one process, same batches:
whole calls by HIP events and the count pass inside the calls (pvv_problem.ev_marks); covariances compared bit for bit.

    python tools/estimate_ab.py [--batches 1,4,8,16,64] [--out f.json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import lib  # noqa: E402

lib._register_clean_pvnet_amd()
from clean_pvnet_amd import ransac_voting as ext  # noqa: E402
from clean_pvnet_amd import synth  # noqa: E402


def med(v):
    v = sorted(v)
    return v[len(v) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="1,4,8,16,64")
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--hn", type=int, default=4096)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = dict(synth.CONFIGS[a.config])
    K = cfg["K"]
    gen = {k: v for k, v in cfg.items() if k not in ("B", "hn")}
    rows = []
    for B in [int(x) for x in a.batches.split(",")]:
        batches = [synth.make_batch(B=B, **gen, first_index=1000 * r, device=dev) for r in range(2)]
        mean = [d["kpt_2d"].float().to(dev) for d in batches]
        row = {"B": B, "hn": a.hn}
        covs = {}
        for name, mode in (("full", ext.COUNT_FULL), ("staged", ext.COUNT_STAGED), ("auto", ext.COUNT_AUTO)):
            def call(i):
                d = batches[i % 2]
                return ext.estimate_voting_distribution(d["mask"], d["vertex"], mean[i % 2], a.hn, 0.99, 5, 30000, None, None, 7, False,
                                                        count_kernel=mode)
            for i in range(6):
                call(i)
            torch.cuda.synchronize()
            n = max(6, min(60, int(600 / B)))
            groups = []
            for g in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for j in range(n):
                    call(j)
                e1.record()
                groups.append((e0, e1))
            torch.cuda.synchronize()
            ms = med([x.elapsed_time(y) / n for x, y in groups])
            covs[name] = call(0)[0].cpu()
            st = ext.stage_ms_in_pipeline([d["mask"] for d in batches], [d["vertex"] for d in batches], a.hn, 0.99, 5, 30000, 7, 12,
                                          mode, True, True)[4:]
            row[name] = {"ms_per_call": round(ms, 4), "count_pass_ms": round(med([r[2] for r in st]), 4),
                         "first_launch_ms": round(med([r[5] for r in st]), 4), "k_lead_ms": round(med([r[6] for r in st]), 4)}
        row["staged_equals_full"] = bool(torch.equal(covs["full"], covs["staged"]) and torch.equal(covs["full"], covs["auto"]))
        row["speedup_call"] = round(row["full"]["ms_per_call"] / row["staged"]["ms_per_call"], 3)
        rows.append(row)
        print(json.dumps(row), flush=True)
        del batches
        torch.cuda.empty_cache()
    if a.out:
        json.dump(rows, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
