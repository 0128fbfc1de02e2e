#!/usr/bin/env python
"""The un_pnp path three ways in one process, whole calls on two rotating batches of config 3 (ms per step, two rounds each):
the one fused call as the library runs it (pvv_decode_keypoint_un_pnp; rows counted as two passes where the estimate stages),
the same call with its single full count pass forced (PVV_COUNT_FULL), and the reference's two calls on the int64 mask.

    BS=64,32,24 python tools/un_pnp_ab.py
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import lib
lib._register_clean_pvnet_amd()
from clean_pvnet_amd import ransac_voting as ext
from clean_pvnet_amd import synth
from clean_pvnet_amd.ransac_voting_gpu import ransac_voting_layer_v3, estimate_voting_distribution_with_mean
dev = torch.device("cuda", 0)
for B in [int(x) for x in os.environ.get("BS", "64").split(",")]:
    cfg = dict(synth.CONFIGS["cfg3"]); gen = {k: v for k, v in cfg.items() if k not in ("B", "hn")}
    bs = [synth.make_batch(B=B, **gen, first_index=1000 * r, device=dev) for r in range(2)]
    xs = []
    for d in bs:
        m, v = d["mask"], d["vertex"]
        x = torch.empty(B, 2 + 18, 480, 640, device=dev)
        x[:, 0] = 3.0 * (m == 0); x[:, 1] = 3.0 * (m != 0); x[:, 2:] = v.permute(0, 3, 4, 1, 2).reshape(B, 18, 480, 640)
        xs.append((x[:, :2], x[:, 2:].permute(0, 2, 3, 1).view(B, 480, 640, 9, 2)))
    def fused(i):
        s, v = xs[i % 2]
        return ext.decode_keypoint_un_pnp(s, v, 512, 4096, 0.99, 5, 30000, None, None, None, 7 + i, ext.SINGULAR_REFERENCE, 0)
    def fused_full(i):
        s, v = xs[i % 2]
        return ext.decode_keypoint_un_pnp(s, v, 512, 4096, 0.99, 5, 30000, None, None, None, 7 + i, ext.SINGULAR_REFERENCE, 0, ext.COUNT_FULL)
    def two(i):
        d = bs[i % 2]
        mean = ransac_voting_layer_v3(d["mask"], d["vertex"], 512, inlier_thresh=0.99)
        return estimate_voting_distribution_with_mean(d["mask"], d["vertex"], mean)
    def timeit(f, n=40):
        for i in range(8): f(i)
        torch.cuda.synchronize(); t = time.perf_counter()
        for i in range(n): f(i)
        torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t) / n
    r = {}
    for rep in range(2):
        for name, f in (("fused", fused), ("fused_full", fused_full), ("two_calls", two)):
            r.setdefault(name, []).append(round(timeit(f), 4))
    print("B", B, r, flush=True)
