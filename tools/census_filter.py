#!/usr/bin/env python
"""Phase census of k_count_filter_runs (the staged count pass's second launch): where its time goes, per block.

Needs an instrumented build (the shipped library carries no stamps):
    tools/build_variant.sh stamps -DPVV_TUNING -DPVV_STAMPS
    gpurun -- 'PVV_LIBPATH=build/variants/stamps.so python tools/census_filter.py --out gpurun_out/filter_census.json'

Thread 0 of every block adds the shader cycles (s_memtime) between consecutive marks to one of ten phase accumulators
(count_filter_runs.hpp, PVV_FS) and writes them with its item / chunk / survivor / tile counts and its wall-clock entry
and exit (100 MHz) on exit.  Reported per case: blocks that worked, items / chunks / survivors / matrix-core tiles per
block, per phase the median and the mean of the blocks' cycles and the phase's share of all cycles spent by the working
blocks, the kernel's span (first entry -> last exit) and the blocks' lifetimes.  Thread 0 sits in wave 0: "loop" is
wave 0's own matrix-core loop, "loop_wait" the barrier behind it (= the slowest of the other three waves)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import capi  # noqa: E402
import variant_time  # noqa: E402

PHASES = ["table", "item_header", "survivor_compaction", "chunk_loads_pixels_Bstage", "A_operands", "mfma_loop",
          "loop_wait_barrier", "elimination_step", "flush", "exit"]


def census(synth, cfgname, B, calls, dev, estimate=0):
    cfg = dict(synth.CONFIGS[cfgname])
    hn = estimate if estimate else cfg["hn"]
    d = synth.make_batch(B=B, **{k: v for k, v in cfg.items() if k not in ("B", "hn")}, device=dev)
    grid = 48 * 256 + 64
    dbg = torch.zeros(16 * grid, dtype=torch.int64, device=dev)
    os.environ["PVV_DBG_PTR_FILTER"] = str(dbg.data_ptr())
    keep = []
    for _ in range(calls):
        dbg.zero_()
        if estimate:    # estimate_voting_distribution_with_mean, no count output, its pass forced into stages (PVV_COUNT_STAGED_ESTIMATE)
            capi.estimate(d["mask"], d["vertex"], d["kpt_2d"].contiguous(), hn, 0.99, max_num=cfg.get("max_num", 30000), seed=5, count_kernel=4,
                          want_counts=False)
        else:
            capi.v3(d["mask"], d["vertex"], hn, 0.99, max_num=cfg.get("max_num", 30000), seed=5, count_kernel=3)
        torch.cuda.synchronize()
        keep.append(dbg.cpu().view(-1, 16).clone())
    rows = []
    for c in keep[3:]:                                  # (the first calls ramp the clock)
        c = c[c[:, 14] != 0]
        work = c[c[:, 10] > 0].double()
        if len(work) == 0:
            continue
        cyc = work[:, :10]
        tot = cyc.sum()
        t0 = c[:, 14].min()
        span_us = float((c[:, 15].max() - t0)) / 100.0
        life = (work[:, 15] - work[:, 14]) / 100.0
        clock_ghz = float(cyc.sum(1).median() / (life.median() * 1e3))       # cycles per ns over a block's life
        ent, ext = (work[:, 14] - t0) / 100.0, (work[:, 15] - t0) / 100.0
        alive = [int(((ent <= f * span_us) & (ext > f * span_us)).sum()) for f in (0.1, 0.25, 0.5, 0.6, 0.7, 0.8, 0.9)]
        by_items = {int(k): {"blocks": int((work[:, 10] == k).sum()), "life_us_median": float(life[work[:, 10] == k].median()),
                             "exit_us_median": float(ext[work[:, 10] == k].median())} for k in work[:, 10].unique().tolist()}
        order = ext.argsort(descending=True)[:12]
        late = [{"block": int(torch.nonzero(c[:, 15].double() == work[i, 15])[0, 0]), "exit_us": round(float(ext[i]), 1), "items": int(work[i, 10]), "chunks": int(work[i, 11]),
                 "survivor_chunks": int(work[i, 12]), "tiles_wave0": int(work[i, 13]),
                 "kcycles": [round(float(cyc[i, j]) / 1e3, 1) for j in range(10)]} for i in order.tolist()]
        typical = {"chunks": float(work[:, 11].median()), "survivor_chunks": float(work[:, 12].median()), "tiles_wave0": float(work[:, 13].median())}
        rows.append({"late": late, "typical": typical, "alive": alive, "by_items": by_items, "entry_us_p90": float(ent.quantile(0.9)), "entry_us_median": float(ent.median()),
                     "blocks_launched": int(len(c)), "blocks_with_items": int(len(work)),
                     "items": int(work[:, 10].sum()), "chunks": int(work[:, 11].sum()),
                     "survivors_per_chunk_mean": float(work[:, 12].sum() / max(1.0, work[:, 11].sum())),
                     "mfma_tiles_wave0_per_chunk_mean": float(work[:, 13].sum() / max(1.0, work[:, 11].sum())),
                     "span_us": span_us, "life_us_median": float(life.median()), "life_us_max": float(life.max()),
                     "entry_us_max": float((work[:, 14].max() - t0) / 100.0), "shader_clock_ghz": clock_ghz,
                     "median_cycles": [float(cyc[:, i].median()) for i in range(10)],
                     "mean_cycles": [float(cyc[:, i].mean()) for i in range(10)],
                     "share": [float(cyc[:, i].sum() / tot) for i in range(10)]})
    med = lambda k: sorted(r[k] for r in rows)[len(rows) // 2]                       # noqa: E731
    medv = lambda k, i: sorted(r[k][i] for r in rows)[len(rows) // 2]               # noqa: E731
    out = {"case": "%s_B%d%s" % (cfgname, B, "_estimate" if estimate else ""), "hn": hn, "calls": len(rows)}
    for k in ("blocks_launched", "blocks_with_items", "items", "chunks", "survivors_per_chunk_mean", "mfma_tiles_wave0_per_chunk_mean",
              "span_us", "life_us_median", "life_us_max", "entry_us_max", "shader_clock_ghz"):
        out[k] = round(med(k), 3)
    mid = rows[len(rows) // 2]
    out["working_blocks_alive_at_fraction_of_span"] = dict(zip(("0.1", "0.25", "0.5", "0.6", "0.7", "0.8", "0.9"), mid["alive"]))
    out["by_items_per_block"] = mid["by_items"]
    out["latest_blocks"], out["typical_block"] = mid["late"], mid["typical"]
    out["entry_us_median"], out["entry_us_p90"] = round(med("entry_us_median"), 2), round(med("entry_us_p90"), 2)
    out["phases"] = {n: {"median_cycles_per_block": round(medv("median_cycles", i), 0), "mean_cycles_per_block": round(medv("mean_cycles", i), 0),
                         "share_of_working_blocks_cycles": round(medv("share", i), 4)} for i, n in enumerate(PHASES)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="cfg3:64,cfg5:16")
    ap.add_argument("--calls", type=int, default=15)
    ap.add_argument("--estimate", type=int, default=0, help="N > 0: the census of the ESTIMATE's staged pass with N hypotheses instead of v3's")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    synth = variant_time._synth()
    dev = torch.device("cuda:0")
    res = {"tool": "tools/census_filter.py on a -DPVV_TUNING -DPVV_STAMPS build (PVV_LIBPATH=%s)" % os.environ.get("PVV_LIBPATH", "<in-tree: no stamps>"),
           "what": "per-block phase cycles of k_count_filter_runs, thread 0 (wave 0), PVV_COUNT_STAGED forced; median over calls of per-call statistics",
           "cases": []}
    for c in a.cases.split(","):
        cfgname, b = c.split(":")
        r = census(synth, cfgname, int(b), a.calls, dev, a.estimate)
        res["cases"].append(r)
        print(json.dumps(r), flush=True)
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
