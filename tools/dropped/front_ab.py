#!/usr/bin/env python
"""Compaction workers per image (pvv_debug_option(PVV_DEBUG_COMPACT_WORKERS, W); 100000 = one block per foreground tile),
one process, same batches: whole calls (HIP events around groups of calls, cold rotating batches) and the per-stage
durations inside the calls (pvv_problem.ev_marks).  One JSON object per (config, B) on stdout; `--out file` also writes
the list.

    python tools/front_ab.py [--cases cfg2:1,cfg3:8,cfg3:64,cfg4:32,cfg5:16] [--workers 0,100000,24,32,48,64] [--calls 60]
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
    ap.add_argument("--cases", default="cfg2:1,cfg3:2,cfg3:8,cfg3:16,cfg3:32,cfg3:64,cfg4:32,cfg5:16")
    ap.add_argument("--calls", type=int, default=60)
    ap.add_argument("--workers", default="0,100000,16,24,32,48,64,0")
    ap.add_argument("--rotate", type=int, default=2)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    rows = []
    for case in args.cases.split(","):
        cfgname, B = case.split(":")
        B = int(B)
        cfg = dict(synth.CONFIGS[cfgname])
        hn, K = cfg["hn"], cfg["K"]
        gen = {k: v for k, v in cfg.items() if k not in ("B", "hn")}
        batches = [synth.make_batch(B=B, **gen, first_index=1000 * r, device=dev) for r in range(args.rotate)]
        row = {"case": case, "hn": hn, "K": K}
        outs = {}
        names = []
        for k, wk in enumerate(int(x) for x in args.workers.split(",")):
            name = "W%d%s" % (wk, "" if ("W%d" % wk) not in names else "_again")
            names.append(name)
            ext.debug_option(ext.DEBUG_COMPACT_WORKERS, wk)
            mode = ext.COUNT_AUTO
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
        ext.debug_option(ext.DEBUG_COMPACT_WORKERS, 0)
        row["all_equal"] = all(all(torch.equal(a, b) for a, b in zip(outs[names[0]], outs[n])) for n in names[1:])
        row["ms_per_call"] = {n: row[n]["ms_per_call"] for n in names}
        row["compact_us"] = {n: round(1e3 * row[n]["compact"], 2) for n in names}
        rows.append(row)
        print(json.dumps(row), flush=True)
        del batches
        torch.cuda.empty_cache()
    if args.out:
        json.dump(rows, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
