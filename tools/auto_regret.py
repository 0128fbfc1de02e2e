#!/usr/bin/env python
"""AUTO's regret on STRUCTURED errors: time(AUTO) / min(time(FULL), time(STAGED)) per case.

AUTO picks between the full count pass and the staged one from the winners' inlier ratios the LAST calls of the shape left
in the stage hint, against a break-even table that was measured on uniform random-direction outliers (pvnet_vote.hip:
stage_hint_threshold).  Real fields fail differently: a contiguous part of the object voting for a wrong point, keypoints
of very different quality within one image, dense detector crops with both.  Per case: whole calls (HIP events around
groups of calls, rotating batches, the hint warmed by the case's own calls first), the three modes interleaved in one
process, outputs compared bit for bit.

    gpurun -- 'python tools/auto_regret.py --out gpurun_out/auto_regret.json'
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

# name, base config, B, generator overrides
CASES = [
    ("clean_cfg3_B64", "cfg3", 64, {}),
    ("uniform_outliers_0.10_cfg3_B64", "cfg3", 64, {"outlier": 0.10}),
    ("uniform_outliers_0.20_cfg3_B64", "cfg3", 64, {"outlier": 0.20}),
    ("wrong_region_0.20_cfg3_B64", "cfg3", 64, {"wrong_region": 0.20}),
    ("wrong_region_0.30_cfg3_B64", "cfg3", 64, {"wrong_region": 0.30}),
    ("wrong_region_0.40_cfg3_B64", "cfg3", 64, {"wrong_region": 0.40}),
    ("wrong_region_0.30_cfg3_B32", "cfg3", 32, {"wrong_region": 0.30}),
    ("wrong_region_0.30_cfg3_B16", "cfg3", 16, {"wrong_region": 0.30}),
    ("keypoint_spread_0.01-0.40_cfg3_B64", "cfg3", 64, {"kp_outlier": (0.01, 0.40)}),
    ("keypoint_spread_0.01-0.40_cfg3_B32", "cfg3", 32, {"kp_outlier": (0.01, 0.40)}),
    ("keypoint_spread_0.01-0.40_cfg3_B16", "cfg3", 16, {"kp_outlier": (0.01, 0.40)}),
    ("region_0.30_and_spread_cfg3_B64", "cfg3", 64, {"wrong_region": 0.30, "kp_outlier": (0.01, 0.40)}),
    ("tless_crop_256_B16_region_0.30_and_spread", "cfg3", 16, {"H": 256, "W": 256, "fg": 0.33, "wrong_region": 0.30, "kp_outlier": (0.01, 0.40)}),
    ("tless_crop_256_B16_clean", "cfg3", 16, {"H": 256, "W": 256, "fg": 0.33}),
    ("tless_crop_128_B32_region_0.30_and_spread", "cfg3", 32, {"H": 128, "W": 128, "fg": 0.33, "wrong_region": 0.30, "kp_outlier": (0.01, 0.40)}),
    ("cfg5_B16_region_0.30_and_spread", "cfg5", 16, {"wrong_region": 0.30, "kp_outlier": (0.01, 0.40)}),
    ("cfg5_B4_keypoint_spread", "cfg5", 4, {"kp_outlier": (0.01, 0.40)}),
]


def med(v):
    v = sorted(v)
    return v[len(v) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=48)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--only", default=None)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    rows = []
    for name, cfgname, B, over in CASES:
        if a.only and a.only not in name:
            continue
        cfg = dict(synth.CONFIGS[cfgname])
        hn = cfg["hn"]
        gen = {k: v for k, v in cfg.items() if k not in ("B", "hn")}
        gen.update(over)
        batches = [synth.make_batch(B=B, **gen, first_index=1000 * r, device=dev) for r in range(2)]
        modes = (("full", ext.COUNT_FULL), ("staged", ext.COUNT_STAGED), ("auto", ext.COUNT_AUTO))

        def call(mode, i):
            d = batches[i % 2]
            return ext.ransac_voting_v3(d["mask"], d["vertex"], hn, 0.99, 5, 30000, None, None, 7, ext.SINGULAR_REFERENCE, count_kernel=mode)
        ext.shutdown()                                             # every case starts without a hint, like a new workload
        for i in range(12):                                        # AUTO's own calls leave the hint this case's ratios
            call(ext.COUNT_AUTO, i)
        torch.cuda.synchronize()
        hint = ext.stage_hint(batches[0]["mask"], batches[0]["vertex"], hn)
        t0 = time.perf_counter()
        i = 0
        while time.perf_counter() - t0 < 0.05:                     # clock pre-warm
            for _ in range(8):
                call(ext.COUNT_AUTO, i)
                i += 1
            torch.cuda.synchronize()
        ms = {m: [] for m, _ in modes}
        per = max(4, a.calls // a.rounds)
        for r in range(a.rounds):                                  # interleaved: every round times all three modes
            for m, code in modes:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                call(code, 0)
                e0.record()
                for j in range(per):
                    call(code, r * per + j)
                e1.record()
                torch.cuda.synchronize()
                ms[m].append(e0.elapsed_time(e1) / per)
        outs = {m: [x.cpu() for x in call(code, 0)[:3]] for m, code in modes}
        win, tn = outs["full"][1].double(), outs["full"][2].double()
        ratio = win / tn.clamp(min=1).view(-1, 1)
        t = {m: med(v) for m, v in ms.items()}
        st = ext.stage_ms_in_pipeline([d["mask"] for d in batches], [d["vertex"] for d in batches], hn, 0.99, 5, 30000, 7, 12, ext.COUNT_AUTO, True)[6:]
        row = {"case": name, "B": B, "hn": hn, "K": cfg["K"], "H": gen["H"], "W": gen["W"], "overrides": {k: v for k, v in over.items()},
               "ms_full": round(t["full"], 4), "ms_staged": round(t["staged"], 4), "ms_auto": round(t["auto"], 4),
               "auto_staged_its_calls": bool(med([r[5] for r in st]) >= 0),
               "regret": round(t["auto"] / min(t["full"], t["staged"]), 4),
               "winner_ratio_mean": round(float(ratio.mean()), 4), "winner_ratio_min_keypoint": round(float(ratio.min()), 4),
               "winner_ratio_max_keypoint": round(float(ratio.max()), 4), "tn_mean": round(float(tn.mean()), 1),
               "hint": {"has_data": bool(hint[0]), "mean_ratio": round(hint[1], 4), "auto_threshold": round(hint[2], 4)},
               "outputs_equal_full_staged_auto": all(torch.equal(x, y) and torch.equal(x, z) for x, y, z in zip(outs["full"], outs["staged"], outs["auto"]))}
        rows.append(row)
        print(json.dumps(row), flush=True)
        del batches
        torch.cuda.empty_cache()
    res = {"tool": "tools/auto_regret.py", "what": "time(AUTO) / min(time(FULL), time(STAGED)), whole pvv_ransac_voting_v3 calls, medians of %d interleaved rounds" % a.rounds,
           "worst_regret": max(r["regret"] for r in rows) if rows else None, "cases": rows}
    print(json.dumps({"worst_regret": res["worst_regret"]}))
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
