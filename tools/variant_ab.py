"""Interleaved A/B timing of the inlier-count kernel of several builds of libpvnet_vote.so in ONE process (same
device buffers, alternating groups of launches), because run-to-run and box-to-box spread (+-2 %) hides the small
steps.  Prints mean, standard error and the ratio to the first library per build.

    gpurun -- 'python tools/variant_ab.py build/variants/a.so build/variants/b.so --rounds 40'
"""
import argparse
import ctypes
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


def _load(path):
    L = ctypes.CDLL(os.path.abspath(path))
    L.pvv_last_error.restype = ctypes.c_char_p
    L.pvv_workspace_bytes.restype = ctypes.c_size_t
    L.pvv_workspace_bytes.argtypes = [ctypes.POINTER(capi.Problem)]
    L.pvv_default_cap.restype = ctypes.c_int32
    L.pvv_default_cap.argtypes = [ctypes.c_int32] * 3
    vp = ctypes.c_void_p
    L.pvv_ransac_voting_v3.argtypes = [ctypes.POINTER(capi.Problem), vp, vp, vp, vp, vp, ctypes.c_size_t, vp, vp, vp, vp]
    L.pvv_rerun_count_kernel.argtypes = [ctypes.POINTER(capi.Problem), vp, ctypes.c_size_t, ctypes.c_int, vp]
    L.pvv_decode_keypoint_v3.argtypes = [ctypes.POINTER(capi.Problem), vp, vp, vp, vp, vp, ctypes.c_size_t, vp, vp, vp, vp, vp]
    return L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--hn", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=30)
    ap.add_argument("--per-group", type=int, default=10)
    ap.add_argument("--mode", default="count", choices=["count", "v3", "decode"],
                    help="count: re-launches of the inlier-count kernel; v3: whole pvv_ransac_voting_v3 calls; decode: whole "
                         "pvv_decode_keypoint_v3 calls on seg logits + the planar vertex view (one [B,2+2K,H,W] tensor, resnet18.py:93)")
    ap.add_argument("--outlier", type=float, default=None, help="fraction of foreground pixels with a random direction")
    ap.add_argument("--fg", type=float, default=None, help="foreground fraction (0.0985 at 480x640: the dense stress, tn ~ 30000)")
    ap.add_argument("--size", default=None, help="HxW override (T-LESS detector crops: 128x128, 256x256)")
    ap.add_argument("--count-kernel", type=int, default=0, help="pvv_problem.count_kernel (0 AUTO, 2 FULL, 3 STAGED)")
    ap.add_argument("--rotate", type=int, default=1, help="v3 mode: cycle over this many distinct device-resident batches (cold caches)")
    a = ap.parse_args()
    synth = _synth()
    cfg = dict(synth.CONFIGS[a.config])
    if a.outlier is not None:
        cfg["outlier"] = a.outlier
    if a.fg is not None:
        cfg["fg"] = a.fg
    if a.size:
        cfg["H"], cfg["W"] = (int(x) for x in a.size.split("x"))
    B = a.batch or cfg["B"]
    hn = a.hn or cfg["hn"]
    dev = torch.device("cuda:0")
    d = synth.make_batch(B, cfg["H"], cfg["W"], cfg["K"], device=dev,
                         **{k: v for k, v in cfg.items() if k not in ("B", "H", "W", "K", "hn")})
    mask, vertex = d["mask"], d["vertex"]
    others = [synth.make_batch(B, cfg["H"], cfg["W"], cfg["K"], device=dev, seed=50 + i,
                               **{k: v for k, v in cfg.items() if k not in ("B", "H", "W", "K", "hn")}) for i in range(a.rotate - 1)]
    st = capi.stream()
    net = []
    if a.mode == "decode":           # the network's output tensor per rotating batch: seg = x[:, :2], vertex = the planar view of x[:, 2:]
        for o in [d] + others:
            K = cfg["K"]
            x = torch.empty(B, 2 + 2 * K, cfg["H"], cfg["W"], device=dev)
            x[:, 0] = 3.0 * (o["mask"] == 0)
            x[:, 1] = 3.0 * (o["mask"] != 0)
            x[:, 2:] = o["vertex"].permute(0, 3, 4, 1, 2).reshape(B, 2 * K, cfg["H"], cfg["W"])
            net.append((x, x[:, :2], x[:, 2:].permute(0, 2, 3, 1).view(B, cfg["H"], cfg["W"], K, 2),
                        torch.empty(B, cfg["H"], cfg["W"], dtype=torch.int64, device=dev)))
        vertex = net[0][2]
    runs = []
    for spec in a.libs:
        # "lib.so@PVV_GRID_PER_CU=30,PVV_X=1": environment for THIS build only (the library reads its knobs once, at
        # its first launch, so the same file can be compared with itself under different settings via a copy)
        path, _, envs = spec.partition("@")
        saved = {}
        for kv in filter(None, envs.split(",")):
            k, _, v = kv.partition("=")
            saved[k] = os.environ.get(k)
            os.environ[k] = v
        if envs:
            import shutil, tempfile
            tmp = os.path.join(tempfile.mkdtemp(), os.path.basename(path))
            shutil.copy(path, tmp)
            path = tmp
        L = _load(path)
        p = capi.Problem()
        p.B, p.H, p.W, p.K, _ = vertex.shape
        p.hn = hn
        p.mask_elem_size = mask.element_size()
        p.min_num, p.max_num = 5, cfg.get("max_num", 30000)
        p.cap = L.pvv_default_cap(p.H, p.W, p.max_num)
        p.inlier_thresh = cfg.get("thresh", 0.99)
        p.mask_stride[:] = mask.stride()
        p.vertex_stride[:] = vertex.stride()
        p.seed = 12345
        p.count_kernel = a.count_kernel
        if a.mode == "decode":
            p.seg_classes = 2
            p.seg_stride[:] = net[0][1].stride()
        n = L.pvv_workspace_bytes(ctypes.byref(p))
        ws = torch.empty(n, dtype=torch.uint8, device=dev)
        out = torch.empty(p.B, p.K, 2, device=dev)
        win = torch.empty(p.B, p.K, dtype=torch.int32, device=dev)
        tn = torch.empty(p.B, dtype=torch.int32, device=dev)
        rc = L.pvv_ransac_voting_v3(ctypes.byref(p), capi.ptr(mask), capi.ptr(vertex), None, None, capi.ptr(ws), n,
                                    capi.ptr(out), capi.ptr(win), capi.ptr(tn), st)
        assert rc == 0, L.pvv_last_error()
        torch.cuda.synchronize()
        args = (ctypes.byref(p), capi.ptr(mask), capi.ptr(vertex), None, None, capi.ptr(ws), n, capi.ptr(out), capi.ptr(win),
                capi.ptr(tn), st)
        for _ in range(2):       # first launches happen here, under this build's environment
            L.pvv_rerun_count_kernel(ctypes.byref(p), capi.ptr(ws), n, 0, st)
        torch.cuda.synchronize()
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        rot = [args] + [(ctypes.byref(p), capi.ptr(o["mask"]), capi.ptr(o["vertex"]), None, None, capi.ptr(ws), n, capi.ptr(out),
                         capi.ptr(win), capi.ptr(tn), st) for o in others]
        if a.mode == "decode":
            rot = [(ctypes.byref(p), capi.ptr(sg), capi.ptr(vx), None, None, capi.ptr(ws), n, capi.ptr(mo), capi.ptr(out), capi.ptr(win),
                    capi.ptr(tn), st) for (_x, sg, vx, mo) in net]
        runs.append(dict(path=spec, L=L, p=p, ws=ws, n=n, win=int(win.sum().item()), out=out.double().sum().item(), ms=[],
                         args=args, rot=rot, k=0, keep=(out, win, tn)))
    def launch(r):
        if a.mode == "decode":
            r["k"] += 1
            rc = r["L"].pvv_decode_keypoint_v3(*r["rot"][r["k"] % len(r["rot"])])
            assert rc == 0, r["L"].pvv_last_error()
        elif a.mode == "v3":
            r["k"] += 1
            r["L"].pvv_ransac_voting_v3(*r["rot"][r["k"] % len(r["rot"])])
        else:
            r["L"].pvv_rerun_count_kernel(ctypes.byref(r["p"]), capi.ptr(r["ws"]), r["n"], 0, st)

    for r in runs:
        for _ in range(5):
            launch(r)
    torch.cuda.synchronize()
    for _ in range(a.rounds):
        for r in runs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.per_group):
                launch(r)
            e1.record()
            torch.cuda.synchronize()
            r["ms"].append(e0.elapsed_time(e1) / a.per_group)
    if a.mode == "decode":
        for r in runs:
            r["mask_sum"] = int(net[r["k"] % len(net)][3].sum().item())
            r["win"], r["out"] = int(r["keep"][1].sum().item()), r["keep"][0].double().sum().item()
    base = None
    for r in runs:
        t = torch.tensor(r["ms"], dtype=torch.float64)
        mean, sem = t.mean().item(), (t.std().item() / len(t) ** 0.5)
        base = base or mean
        print(json.dumps({"lib": os.path.basename(r["path"]), "mode": a.mode, "config": a.config, "B": B, "hn": hn,
                          "ms_mean": round(mean, 4), "ms_sem": round(sem, 5), "ms_min": round(t.min().item(), 4),
                          "ratio": round(mean / base, 4), "win_sum": r["win"], "out_sum": round(r["out"], 3), "mask_sum": r.get("mask_sum")}), flush=True)


if __name__ == "__main__":
    main()
