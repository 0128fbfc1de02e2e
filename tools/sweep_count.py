#!/usr/bin/env python
"""Sweep the count kernel's scheduling knobs (tuning build: tools/build_variant.sh <name> -DPVV_TUNING, then
PVV_LIBPATH=build/variants/<name>.so) over batch sizes: kernel time by HIP events around re-launches.
usage: sweep_count.py [--batches 1,8] [--grid 5,10,15] [--items 2,4,8] [--config cfg3]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import capi  # noqa: E402
import variant_time  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="1,2,4,8,16,32,64")
    ap.add_argument("--grid", default="5,10,15")
    ap.add_argument("--items", default="128,512,1280", help="target work items (absolute)")
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--hn", type=int, default=0)
    a = ap.parse_args()
    synth = variant_time._synth()
    cfg = dict(synth.CONFIGS[a.config])
    hn = a.hn or cfg["hn"]
    dev = torch.device("cuda:0")
    L = capi.load()
    for B in [int(x) for x in a.batches.split(",")]:
        d = synth.make_batch(B, cfg["H"], cfg["W"], cfg["K"], device=dev,
                             **{k: v for k, v in cfg.items() if k not in ("B", "H", "W", "K", "hn")})
        mask, vertex = d["mask"], d["vertex"]
        p = capi.problem(mask, vertex, hn, 0.99, seed=12345)
        n = L.pvv_workspace_bytes(ctypes.byref(p))
        ws = torch.empty(n, dtype=torch.uint8, device=dev)
        out = torch.empty(p.B, p.K, 2, device=dev)
        win = torch.empty(p.B, p.K, dtype=torch.int32, device=dev)
        tn = torch.empty(p.B, dtype=torch.int32, device=dev)
        ref = None
        line = []
        for g in [int(x) for x in a.grid.split(",")]:
            for it in [int(x) for x in a.items.split(",")]:
                os.environ["PVV_GRID_PER_CU"] = str(g)
                os.environ["PVV_TARGET_ITEMS"] = str(it)
                capi.check(L.pvv_ransac_voting_v3(ctypes.byref(p), capi.ptr(mask), capi.ptr(vertex), None, None,
                                                  capi.ptr(ws), n, capi.ptr(out), capi.ptr(win), capi.ptr(tn), capi.stream()))
                torch.cuda.synchronize()
                w = win.cpu().clone()
                if ref is None:
                    ref = w
                assert torch.equal(w, ref), "winner counts changed with the schedule"
                for _ in range(3):
                    L.pvv_rerun_count_kernel(ctypes.byref(p), capi.ptr(ws), n, 0, capi.stream())
                ms = []
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10):
                        L.pvv_rerun_count_kernel(ctypes.byref(p), capi.ptr(ws), n, 0, capi.stream())
                    e1.record()
                    torch.cuda.synchronize()
                    ms.append(e0.elapsed_time(e1) / 10)
                line.append("g%d/i%d %.1f" % (g, it, 1e3 * sorted(ms)[len(ms) // 2]))
        print("B=%-3d us: %s" % (B, "  ".join(line)), flush=True)


if __name__ == "__main__":
    main()
