#!/usr/bin/env python
"""Phase census + effective shader clock of k_count_bf16 (round 6; VERDICT r5 #1a / #3).

Needs an instrumented build (the shipped library carries no stamps):
    tools/build_variant.sh stamps -DPVV_TUNING -DPVV_STAMPS
    gpurun -- 'PVV_LIBPATH=build/variants/stamps.so python tools/census_count.py --out gpurun_out/count_census.json'

Thread 0 of every block adds the SHADER cycles (s_memtime) between consecutive marks to one of eight phase accumulators and,
around the matrix-core loop, the ticks of the 100 MHz real-time counter (s_memrealtime) as well (count_bf16.hpp, PVV_CS*).  Per case:
  * share of the working blocks' cycles per phase (thread 0 sits in wave 0: "loop" is wave 0's own matrix-core loop, "loop_wait" the
    barrier behind it = the slowest of the other three waves);
  * effective shader clock INSIDE the kernel = shader cycles / real time, over the blocks' lives and over the loop phase alone;
  * the loop's cost per matrix-core tile and SIMD in shader cycles AND in ns (5 waves per SIMD share the SIMD, so a wave's loop
    time per tile / 5 is the SIMD's time per tile only where all five waves are in their loops; reported as measured, per wave);
  * how many block slots were alive over the span (occupancy of the 1280 slots).
The cases: the estimate's 4096-hypothesis full pass at B = 64 (the VALU-saturated kernel), the headline's first-stage launch
(cfg3 B = 64 staged), the full pass of the B = 8 shard (BASELINE config 3 on 8 GPUs) and of one image.
"""
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

PHASES = ["table", "item_prologue_pixels_Bstage", "A_operands", "next_group_Bstage", "mfma_loop", "loop_wait_barrier", "flush", "other"]
GRID = 48 * 256 + 64


def census(synth, cfgname, B, calls, dev, hn=0, estimate=False, count_kernel=0, prewarm_ms=60.0):
    cfg = dict(synth.CONFIGS[cfgname])
    hn = hn or cfg["hn"]
    d = synth.make_batch(B=B, **{k: v for k, v in cfg.items() if k not in ("B", "hn")}, device=dev)
    dbg = torch.zeros(64 + 16 * GRID, dtype=torch.int64, device=dev)
    os.environ["PVV_DBG_PTR"] = str(dbg.data_ptr())

    def call():
        if estimate:
            capi.estimate(d["mask"], d["vertex"], d["kpt_2d"].contiguous(), hn, 0.99, max_num=cfg.get("max_num", 30000), seed=5,
                          count_kernel=count_kernel, want_counts=False)
        else:
            capi.v3(d["mask"], d["vertex"], hn, 0.99, max_num=cfg.get("max_num", 30000), seed=5, count_kernel=count_kernel)

    # clock pre-warm: the same calls back to back for prewarm_ms, so that the stamped call runs at the clock of a busy GPU
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while True:
        for _ in range(8):
            call()
        e1.record()
        e1.synchronize()
        if e0.elapsed_time(e1) >= prewarm_ms:
            break
    keep = []
    for _ in range(calls):
        dbg.zero_()
        for _ in range(4):          # (the stamped launch is the LAST of a short burst: no idle gap in front of it)
            call()
        torch.cuda.synchronize()
        keep.append(dbg[64:].cpu().view(-1, 16).clone())
    rows = []
    for c in keep:
        c = c[c[:, 0] != 0]
        work = c[c[:, 3] > 0].double()
        if len(work) == 0:
            continue
        cyc = work[:, 4:12]
        tot = cyc.sum()
        t0 = c[:, 0].min()
        span_us = float(c[:, 1].max() - t0) / 100.0
        life_us = (work[:, 1] - work[:, 0]) / 100.0
        ent, ext = (work[:, 0] - t0) / 100.0, (work[:, 1] - t0) / 100.0
        loop_cyc, loop_us, tiles = cyc[:, 4], work[:, 12] / 100.0, work[:, 13]
        ok = (loop_us > 0) & (tiles > 0)
        hw = c[c[:, 3] > 0][:, 2]
        xcc = (hw >> 32) & 0xf
        cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5)
        per_cu = torch.bincount(torch.unique(xcc * 1024 + cu, return_counts=True)[1])
        rows.append({
            "blocks_launched": int(len(c)), "blocks_with_items": int(len(work)), "items": int(work[:, 3].sum()),
            "span_us": span_us, "life_us_median": float(life_us.median()), "life_us_max": float(life_us.max()),
            "entry_us_max": float(ent.max()),
            "alive_at_fraction_of_span": {str(f): int(((ent <= f * span_us) & (ext > f * span_us)).sum()) for f in (0.1, 0.25, 0.5, 0.75, 0.9)},
            "working_blocks_per_cu_histogram": per_cu.tolist(),
            "share": {PHASES[i]: float(cyc[:, i].sum() / tot) for i in range(8)},
            "median_cycles": {PHASES[i]: float(cyc[:, i].median()) for i in range(8)},
            "effective_clock_GHz_block_life": float(cyc.sum() / (life_us.sum() * 1e3)),
            "effective_clock_GHz_mfma_loop": float(loop_cyc[ok].sum() / (loop_us[ok].sum() * 1e3)),
            "tiles_wave0_total": float(tiles.sum()),
            "loop_cycles_per_tile_and_wave": float(loop_cyc[ok].sum() / tiles[ok].sum()),
            "loop_ns_per_tile_and_wave": float(loop_us[ok].sum() * 1e3 / tiles[ok].sum()),
        })
    if not rows:
        return {"config": cfgname, "B": B, "hn": hn, "error": "no block reported"}
    rows.sort(key=lambda r: r["span_us"])
    mid = rows[len(rows) // 2]
    mid["calls"] = len(rows)
    mid["span_us_all_calls"] = [round(r["span_us"], 2) for r in rows]
    mid["effective_clock_GHz_mfma_loop_all_calls"] = [round(r["effective_clock_GHz_mfma_loop"], 3) for r in rows]
    return {"config": cfgname, "B": B, "hn": hn, "estimate": estimate, "count_kernel": count_kernel, **mid}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--calls", type=int, default=9)
    a = ap.parse_args()
    synth = variant_time._synth()
    dev = torch.device("cuda:0")
    out = {"how": __doc__.split("\n\n")[2], "phases": PHASES, "cases": {}}
    cases = [
        ("estimate_4096_full_B64", dict(cfgname="cfg3", B=64, hn=4096, estimate=True, count_kernel=2)),
        ("cfg3_B64_staged_first_launch", dict(cfgname="cfg3", B=64, count_kernel=3)),
        ("cfg3_B64_full", dict(cfgname="cfg3", B=64, count_kernel=2)),
        ("cfg3_B8_full_shard_of_8gpu", dict(cfgname="cfg3", B=8, count_kernel=2)),
        ("cfg3_B1_full", dict(cfgname="cfg3", B=1, count_kernel=2)),
        ("cfg5_B16_full", dict(cfgname="cfg5", B=16, count_kernel=2)),
    ]
    for name, kw in cases:
        r = census(synth, calls=a.calls, dev=dev, **kw)
        out["cases"][name] = r
        print(name, json.dumps({k: r.get(k) for k in ("span_us", "blocks_with_items", "items", "effective_clock_GHz_block_life",
                                                       "effective_clock_GHz_mfma_loop", "loop_cycles_per_tile_and_wave",
                                                       "loop_ns_per_tile_and_wave", "share")}), flush=True)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
