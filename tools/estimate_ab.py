#!/usr/bin/env python
"""estimate_voting_distribution_with_mean (4096 hypotheses, the un_pnp path of resnet18.py:72) with its count pass in full and in
stages (PVV_COUNT_FULL / PVV_COUNT_STAGED_ESTIMATE / AUTO), one process, same batches: whole calls (HIP events around groups of calls, rotating
cold batches); covariances and PnP weights compared bit for bit.  One JSON object per case on stdout.

    python tools/estimate_ab.py [--cases cfg3:1,cfg3:8,cfg3:64] [--outlier 0.095] [--hn 4096]
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
    ap.add_argument("--cases", default="cfg3:1,cfg3:4,cfg3:8,cfg3:16,cfg3:64")
    ap.add_argument("--calls", type=int, default=24)
    ap.add_argument("--rotate", type=int, default=2)
    ap.add_argument("--hn", type=int, default=4096)
    ap.add_argument("--outlier", type=float, default=None)
    ap.add_argument("--gen", default=None, help='JSON overrides of the synthetic generator, e.g. \'{"wrong_region": 0.3, "kp_outlier": [0.01, 0.4]}\'')
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    for case in args.cases.split(","):
        cfgname, B = case.split(":")
        B = int(B)
        cfg = dict(synth.CONFIGS[cfgname])
        K = cfg["K"]
        gen = {k: v for k, v in cfg.items() if k not in ("B", "hn")}
        if args.outlier is not None:
            gen["outlier"] = args.outlier
        if args.gen:
            gen.update({k: (tuple(v) if isinstance(v, list) else v) for k, v in json.loads(args.gen).items()})
        batches = [synth.make_batch(B=B, **gen, first_index=1000 * r, device=dev) for r in range(args.rotate)]
        for d in batches:
            d["mean"] = (d["kpt_2d"] + 0.3).contiguous()
        row = {"case": case, "hn": args.hn, "K": K, "outlier": gen.get("outlier", 0.0), "gen": args.gen}
        outs = {}
        for name, mode in (("full", ext.COUNT_FULL), ("staged", ext.COUNT_STAGED_ESTIMATE), ("auto", ext.COUNT_AUTO)):
            def call(i):
                d = batches[i % len(batches)]
                return ext.estimate_voting_distribution(d["mask"], d["vertex"], d["mean"], args.hn, 0.99, 5, 30000, None, None, 7, False, 0, mode)
            t0 = time.perf_counter()
            i = 0
            while time.perf_counter() - t0 < 0.05:
                for _ in range(4):
                    call(i)
                    i += 1
                torch.cuda.synchronize()
            groups = []
            per = max(1, args.calls // 6)
            for g in range(6):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for j in range(per):
                    call(g * 100 + j)
                b.record()
                groups.append((a, b))
            torch.cuda.synchronize()
            row[name] = round(med([a.elapsed_time(b) / per for a, b in groups]), 4)
            o = call(0)
            outs[name] = (o[0].cpu(), o[4].cpu())
        row["staged_equals_full"] = bool(torch.equal(outs["full"][0], outs["staged"][0]) and torch.equal(outs["full"][1], outs["staged"][1]))
        row["auto_equals_full"] = bool(torch.equal(outs["full"][0], outs["auto"][0]))
        row["speedup"] = round(row["full"] / row["staged"], 3)
        print(json.dumps(row), flush=True)
        del batches
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
