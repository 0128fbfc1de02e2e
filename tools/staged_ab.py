#!/usr/bin/env python
"""Full vs staged count pass of ransac_voting_layer_v3 (PVV_COUNT_FULL / PVV_COUNT_STAGED), one process, same batches:
whole calls (HIP events around groups of calls, cold rotating batches) and the per-stage durations inside the calls
(pvv_problem.ev_marks).  One JSON object per (config, B) on stdout; `--out file` also writes the list.

    python tools/staged_ab.py [--cases cfg3:8,cfg3:16,cfg3:32,cfg3:64,cfg4:32,cfg5:16] [--calls 60]
"""
import argparse
import json
import os
import sys
import time

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
    ap.add_argument("--cases", default="cfg3:8,cfg3:16,cfg3:32,cfg3:64,cfg4:32,cfg5:16")
    ap.add_argument("--calls", type=int, default=60)
    ap.add_argument("--rotate", type=int, default=2)
    ap.add_argument("--outlier", type=float, default=None,
                    help="override the configs' outlier fraction: pixels whose direction is random (a winner then explains "
                         "fewer pixels and the elimination bites later or not at all)")
    ap.add_argument("--sigma", type=float, default=None, help="override the configs' direction noise")
    ap.add_argument("--size", default=None, help="HxW override (T-LESS detector crops: 128x128, 256x256)")
    ap.add_argument("--fg", type=float, default=None, help="foreground fraction override")
    ap.add_argument("--hn", type=int, default=None, help="hypotheses per keypoint override")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    rows = []
    for case in args.cases.split(","):
        cfgname, B = case.split(":")
        B = int(B)
        cfg = dict(synth.CONFIGS[cfgname])
        hn, K = args.hn or cfg["hn"], cfg["K"]
        gen = {k: v for k, v in cfg.items() if k not in ("B", "hn")}
        if args.outlier is not None:
            gen["outlier"] = args.outlier
        if args.sigma is not None:
            gen["sigma"] = args.sigma
        if args.size:
            gen["H"], gen["W"] = (int(x) for x in args.size.split("x"))
        if args.fg is not None:
            gen["fg"] = args.fg
        batches = [synth.make_batch(B=B, **gen, first_index=1000 * r, device=dev) for r in range(args.rotate)]
        row = {"case": case, "hn": hn, "K": K, "outlier": gen.get("outlier", 0.0), "sigma": gen.get("sigma"), "H": gen["H"], "W": gen["W"],
               "fg": gen.get("fg")}
        outs = {}
        for name, mode in (("full", ext.COUNT_FULL), ("staged", ext.COUNT_STAGED), ("auto", ext.COUNT_AUTO)):
            def call(i):
                d = batches[i % len(batches)]
                return ext.ransac_voting_v3(d["mask"], d["vertex"], hn, 0.99, 5, 30000, None, None, 7, ext.SINGULAR_REFERENCE,
                                            count_kernel=mode)
            t0 = time.perf_counter()
            i = 0
            while time.perf_counter() - t0 < 0.05:                      # clock pre-warm
                for _ in range(8):
                    call(i)
                    i += 1
                torch.cuda.synchronize()
            groups = []
            for g in range(6):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for j in range(args.calls // 6):
                    o = call(g * 100 + j)
                b.record()
                groups.append((a, b))
            torch.cuda.synchronize()
            ms = med([a.elapsed_time(b) / (args.calls // 6) for a, b in groups])
            outs[name] = [x.cpu() for x in call(0)[:3]]
            st = ext.stage_ms_in_pipeline([d["mask"] for d in batches], [d["vertex"] for d in batches], hn, 0.99, 5, 30000, 7, 24,
                                          mode, True)[6:]
            cols = ("scan", "compact", "count_pass", "select", "finalize", "stage0", "prune")
            row[name] = {"ms_per_call": round(ms, 4), "images_per_s": round(B / ms * 1e3, 1),
                         **{c: round(med([r[j] for r in st]), 4) for j, c in enumerate(cols) if med([r[j] for r in st]) >= 0}}
        win, tn = outs["full"][1].double(), outs["full"][2].double()
        row["mean_winner_ratio"] = round(float((win / tn.clamp(min=1).view(-1, 1)).mean()), 4)
        row["staged_equals_full"] = all(torch.equal(a, b) for a, b in zip(outs["full"], outs["staged"]))
        row["speedup_call"] = round(row["full"]["ms_per_call"] / row["staged"]["ms_per_call"], 3)
        row["speedup_count_pass"] = round(row["full"]["count_pass"] / row["staged"]["count_pass"], 3)
        rows.append(row)
        print(json.dumps(row), flush=True)
        del batches
        torch.cuda.empty_cache()
    if args.out:
        json.dump(rows, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
