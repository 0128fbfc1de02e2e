"""Time the inlier-count kernel of ONE build of libpvnet_vote.so (PVV_LIBPATH, default: the in-tree library) on a
BASELINE config, through the C ABI: one full pvv_ransac_voting_v3 for the state, then groups of back-to-back
pvv_rerun_count_kernel between HIP events.  Prints one JSON line with the kernel time and a checksum of the winning
inlier counts and keypoints, so experimental builds can be compared for speed AND equality in one gpurun call:

    gpurun -- 'for v in build/variants/*.so; do PVV_LIBPATH=$v python tools/variant_time.py --tag $v; done'
"""
import argparse
import ctypes
import hashlib
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import capi  # noqa: E402


def _synth():
    spec = importlib.util.spec_from_file_location("pvv_synth", os.path.join(ROOT, "clean-pvnet_amd", "synth.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--hn", type=int, default=0)
    ap.add_argument("--groups", type=int, default=5)
    ap.add_argument("--per-group", type=int, default=10)
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    synth = _synth()
    cfg = dict(synth.CONFIGS[a.config])
    B = a.batch or cfg["B"]
    hn = a.hn or cfg["hn"]
    dev = torch.device("cuda:0")
    d = synth.make_batch(B, cfg["H"], cfg["W"], cfg["K"], device=dev,
                         **{k: v for k, v in cfg.items() if k not in ("B", "H", "W", "K", "hn")})
    mask, vertex = d["mask"], d["vertex"]
    thresh = cfg.get("thresh", 0.99)
    max_num = cfg.get("max_num", 30000)
    L = capi.load()
    p = capi.problem(mask, vertex, hn, thresh, max_num=max_num, seed=12345)
    n = L.pvv_workspace_bytes(ctypes.byref(p))
    ws = torch.empty(n, dtype=torch.uint8, device=dev)
    out = torch.empty(p.B, p.K, 2, device=dev)
    win = torch.empty(p.B, p.K, dtype=torch.int32, device=dev)
    tn = torch.empty(p.B, dtype=torch.int32, device=dev)
    capi.check(L.pvv_ransac_voting_v3(ctypes.byref(p), capi.ptr(mask), capi.ptr(vertex), None, None, capi.ptr(ws), n,
                                      capi.ptr(out), capi.ptr(win), capi.ptr(tn), capi.stream()))
    torch.cuda.synchronize()
    for _ in range(3):
        capi.check(L.pvv_rerun_count_kernel(ctypes.byref(p), capi.ptr(ws), n, 0, capi.stream()))
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.groups)]
    for e0, e1 in evs:
        e0.record()
        for _ in range(a.per_group):
            L.pvv_rerun_count_kernel(ctypes.byref(p), capi.ptr(ws), n, 0, capi.stream())
        e1.record()
    torch.cuda.synchronize()
    ms = sorted(e0.elapsed_time(e1) / a.per_group for e0, e1 in evs)
    evals = int(tn.sum().item()) * p.K * hn
    h = hashlib.sha1(win.cpu().numpy().tobytes() + out.cpu().numpy().tobytes()).hexdigest()[:12]
    print(json.dumps({"tag": a.tag, "lib": os.path.basename(capi.LIBPATH), "config": a.config, "B": B, "hn": hn,
                      "kernel_ms_min": round(ms[0], 4), "kernel_ms_avg": round(sum(ms) / len(ms), 4),
                      "teval_per_s": round(evals / (sum(ms) / len(ms) * 1e-3) / 1e12, 2),
                      "win_sum": int(win.sum().item()), "sha": h,
                      "env": {k: v for k, v in os.environ.items() if k.startswith("PVV_") and k != "PVV_LIBPATH"}}),
          flush=True)


if __name__ == "__main__":
    main()
