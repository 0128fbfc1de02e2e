#!/usr/bin/env python
"""Stage times of the ESTIMATE's calls (4096 hypotheses, config 3 at B images): HIP events at the stage boundaries inside calls
(ext.stage_ms_in_pipeline) for the full count pass and the pass in stages -> per mode the medians
[scan, compact + hypotheses, count pass, covariance, 0, first count launch, k_lead] in ms; with SURV=1 also the share of the
hypotheses within 0.1 / 0.2 of the best ratio (what the staged pass cannot drop) and the quantiles of the ratios.
On a tuning build (LD_PRELOAD=build/variants/tuning.so) PVV_STAGE_EIGHTH / PVV_RUN_R select the first stage and the run length.

    B=64 SURV=1 python tools/estimate_stages.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import lib  # noqa: E402

lib._register_clean_pvnet_amd()
from clean_pvnet_amd import ransac_voting as ext  # noqa: E402
from clean_pvnet_amd import synth  # noqa: E402

dev = torch.device("cuda", 0)
B = int(os.environ.get("B", "64"))
cfg = dict(synth.CONFIGS["cfg3"])
gen = {k: v for k, v in cfg.items() if k not in ("B", "hn")}
d = synth.make_batch(B=B, **gen, device=dev)
m, v = d["mask"], d["vertex"]
med = lambda x: sorted(x)[len(x) // 2]  # noqa: E731
for name, mode in (("full", ext.COUNT_FULL), ("staged", ext.COUNT_STAGED_ESTIMATE)):
    ms = ext.stage_ms_in_pipeline([m], [v], 4096, 0.99, 5, 30000, 3, 12, mode, True, True, [], [d["kpt_2d"].contiguous()])[4:]
    print(name, "B=%d" % B, "eighth=" + os.environ.get("PVV_STAGE_EIGHTH", "default"), "run_r=" + os.environ.get("PVV_RUN_R", "default"),
          [round(med(c), 4) for c in zip(*ms)], flush=True)
if os.environ.get("SURV"):
    mean = d["kpt_2d"].contiguous()
    cov, hyp, counts, tn, w = ext.estimate_voting_distribution(m, v, mean, 4096, 0.99, 5, 30000, None, None, 7, True, 0, ext.COUNT_FULL)
    r = counts.float() / tn.float()[:, None, None]
    mx = r.max(dim=2, keepdim=True).values
    print("share of the hypotheses within 0.1 of the best ratio: %.4f   within 0.2: %.4f   mean best ratio %.4f" %
          (float((r >= mx - 0.1).float().mean()), float((r >= mx - 0.2).float().mean()), float(mx.mean())))
    q = torch.quantile(r.flatten()[:4000000], torch.tensor([0.1, 0.25, 0.5, 0.75, 0.9], device=dev))
    print("ratio quantiles 10/25/50/75/90 %:", [round(x, 4) for x in q.tolist()])
