#!/usr/bin/env python
"""Every BASELINE config (and the shards / default path the verdicts ask about) on ONE MI355X, one JSON file:

    gpurun -- 'python tools/config_bench.py --out gpurun_out/<dir>/configs.json'
    cp gpurun_out/<dir>/configs.json profiles/rNN_configs.json

Per row: images/s and ms per call of ``ransac_voting_layer_v3`` through the drop-in Python API measured three ways
(host wall clock over back-to-back calls; per-call HIP events on the launch stream: median / p10 / p90; one captured
HIP graph replayed back to back = the GPU-side floor without host launch cost), the host-only cost of a call (calls
enqueued without waiting, on an idle stream), the inlier-count kernel's duration (re-launches between HIP events),
evaluations/s and the dense-field roofline fraction of that kernel.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

ROWS = [
    # name, config, B, overrides
    ("cfg2_B1", "cfg2", 1, {}),
    ("cfg2_dense_tn30000_B1", "cfg2", 1, {"fg": 0.0985}),      # the dense stress of config 2 (tn ~ 30000)
    ("cfg3_B2", "cfg3", 2, {}),
    ("cfg3_B4", "cfg3", 4, {}),
    ("cfg3_B8_shard_of_8gpu", "cfg3", 8, {}),
    ("cfg3_B16", "cfg3", 16, {}),
    ("cfg3_B32", "cfg3", 32, {}),
    ("cfg3_B64", "cfg3", 64, {}),
    ("cfg4_B32", "cfg4", 32, {}),
    ("cfg4_B4_shard_of_8gpu", "cfg4", 4, {}),
    ("cfg5_B16", "cfg5", 16, {}),
    ("cfg5_B2_shard_of_8gpu", "cfg5", 2, {}),
    ("default_path_hn128_maxnum100_B64", "cfg3", 64, {"hn": 128, "max_num": 100}),
    ("default_path_hn128_maxnum100_B1", "cfg3", 1, {"hn": 128, "max_num": 100}),
    # the layout the REAL caller passes (resnet18.py:66-69,93-94): seg logits and the vertex field are channel
    # slices of ONE [B, 2+2K, H, W] network output -- the vertex a strided planar view, the mask an argmax away
    ("cfg3_B64_planar_vertex", "cfg3", 64, {"planar": True}),                 # int64 mask given, planar vertex
    ("cfg3_B64_decode_fused", "cfg3", 64, {"seg": True}),                     # decode_keypoint_v3: argmax inside the scan
    ("cfg3_B64_decode_unfused", "cfg3", 64, {"seg": True, "unfused": True}),  # torch.argmax + v3, as resnet18.py:69-71 runs it
    ("cfg2_B1_decode_fused", "cfg2", 1, {"seg": True}),
    ("cfg2_B1_decode_unfused", "cfg2", 1, {"seg": True, "unfused": True}),
    # T-LESS runs the voting on DETECTOR CROPS, one per detection (lib/datasets/tless_test/pvnet.py:75-89, 128x128 and
    # 256x256, B = #detections, K = 9): small images whose tiles are dense in foreground
    ("tless_crop128_B8", "cfg3", 8, {"H": 128, "W": 128, "fg": 0.35}),
    ("tless_crop256_B8", "cfg3", 8, {"H": 256, "W": 256, "fg": 0.35}),
    ("tless_crop256_B16_decode_fused", "cfg3", 16, {"H": 256, "W": 256, "fg": 0.35, "seg": True}),
]


def make_case(name, dev):
    """-> dict(call, data, stage(inner) -> rows of stage_ms_in_pipeline, B, H, W, K, hn, max_num) of one row."""
    import lib
    lib._register_clean_pvnet_amd()
    from clean_pvnet_amd import ransac_voting as ext
    from clean_pvnet_amd import synth
    from lib.csrc.ransac_voting.ransac_voting_gpu import ransac_voting_layer_v3
    _n, cfgname, B, over = [r for r in ROWS if r[0] == name][0]
    cfg = dict(synth.CONFIGS[cfgname])
    cfg.update({k: over[k] for k in ("H", "W") if k in over})
    H, W, K = cfg["H"], cfg["W"], cfg["K"]
    hn = over.get("hn", cfg["hn"])
    max_num = over.get("max_num", 30000)
    gen = {k: v for k, v in cfg.items() if k not in ("B", "hn")}
    if "fg" in over:
        gen["fg"] = over["fg"]
    d = synth.make_batch(B=B, **gen, device=dev, planar=bool(over.get("planar") or over.get("seg")))
    mask, vertex = d["mask"], d["vertex"]
    case = dict(B=B, H=H, W=W, K=K, hn=hn, max_num=max_num, data=d, layout="contiguous [B,H,W,K,2] vertex, int64 mask")
    if over.get("seg"):
        x = torch.empty(B, 2 + 2 * K, H, W, device=dev)
        x[:, :2] = torch.randn(B, 2, H, W, device=dev) * 0.1
        x[:, 0] += 3.0 * (mask == 0)
        x[:, 1] += 3.0 * (mask != 0)
        x[:, 2:] = vertex.permute(0, 3, 4, 1, 2).reshape(B, 2 * K, H, W)
        seg, ver = x[:, :2], x[:, 2:]
        vtx = ver.permute(0, 2, 3, 1).view(B, H, W, K, 2)
        case["layout"] = "seg logits + planar vertex: channel slices of one [B,2+2K,H,W] tensor (resnet18.py:93-94)"
        if over.get("unfused"):
            case["call"] = lambda: ransac_voting_layer_v3(torch.argmax(seg, 1), vtx, hn, inlier_thresh=0.99, max_num=max_num)
            case["stage"] = lambda inner, reps=36: ext.stage_ms_in_pipeline([torch.argmax(seg, 1)], [vtx], hn, 0.99, 5, max_num, 1, reps, ext.COUNT_AUTO, inner)
        else:
            case["call"] = lambda: ext.decode_keypoint_v3(seg, vtx, hn, 0.99, 5, max_num, None, None, 7, ext.SINGULAR_REFERENCE)[0]
            case["stage"] = lambda inner, reps=36: ext.stage_ms_in_pipeline([], [vtx], hn, 0.99, 5, max_num, 1, reps, ext.COUNT_AUTO, inner, False, [seg])
        case["tn"] = lambda: ext.decode_keypoint_v3(seg, vtx, hn, 0.99, 5, max_num, None, None, 7, ext.SINGULAR_REFERENCE)[3]
        case["keep"] = x
    else:
        if over.get("planar"):
            case["layout"] = "planar vertex (storage [B,2K,H,W], the view resnet18.py:66-68 makes), int64 mask"
        case["call"] = lambda: ransac_voting_layer_v3(mask, vertex, hn, inlier_thresh=0.99, max_num=max_num)
        case["stage"] = lambda inner, reps=36: ext.stage_ms_in_pipeline([mask], [vertex], hn, 0.99, 5, max_num, 1, reps, ext.COUNT_AUTO, inner)
        case["tn"] = lambda: ext.ransac_voting_v3(mask, vertex, hn, 0.99, 5, max_num, None, None, 1, ext.SINGULAR_REFERENCE)[2]
    return case


def pct(v, q):
    v = sorted(v)
    return v[min(len(v) - 1, int(q * len(v)))]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--rows", default="", help="comma-separated row names (default: all)")
    ap.add_argument("--calls", type=int, default=200)
    args = ap.parse_args()
    import lib
    lib._register_clean_pvnet_amd()
    from clean_pvnet_amd import ransac_voting as ext
    from clean_pvnet_amd import synth
    from lib.csrc.ransac_voting.ransac_voting_gpu import ransac_voting_layer_v3
    dev = torch.device("cuda:0")
    want = set(args.rows.split(",")) if args.rows else None
    res = {"device": torch.cuda.get_device_name(0), "rows": {}}
    for name, cfgname, B, over in ROWS:
        if want and name not in want:
            continue
        case = make_case(name, dev)
        H, W, K, hn, max_num, d = case["H"], case["W"], case["K"], case["hn"], case["max_num"], case["data"]
        call = case["call"]
        n = args.calls if B <= 16 else max(20, args.calls // 4)

        for _ in range(10):
            out = call()
        torch.cuda.synchronize()
        # (a) host wall clock over back-to-back calls
        t0 = time.perf_counter()
        for _ in range(n):
            out = call()
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        # (b) per-call HIP events
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        evs[0].record()
        for i in range(n):
            call()
            evs[i + 1].record()
        torch.cuda.synchronize()
        per = [evs[i].elapsed_time(evs[i + 1]) for i in range(n)]
        # (c) host-only cost of a call: enqueue on an idle stream, no waiting (small bursts so the queue never fills)
        host = []
        for _ in range(10):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(8):
                call()
            host.append((time.perf_counter() - t1) / 8)
        torch.cuda.synchronize()
        # (d) one captured graph replayed back to back
        graph_ms = None
        try:
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(3):
                    call()
                with torch.cuda.graph(g, stream=s):
                    gout = call()
            for _ in range(5):
                g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            graph_ms = e0.elapsed_time(e1) / n
            del g, gout
        except Exception as ex:  # noqa: BLE001
            graph_ms = "failed: %s" % (str(ex)[:100],)
        # (e) the kernels as they run inside the calls: HIP events at the stage boundaries (pvv_problem.ev_marks).  The count
        #     pass = one k_count_bf16 launch, or -- staged (AUTO on large batches) -- k_count_bf16<first> + k_lead +
        #     k_count_bf16<filter>
        tn = case["tn"]()
        stage_names = ("k_tile_scan", "k_compact_hyp", "count_pass", "k_select_refit", "k_finalize_v3", "count_first_launch", "k_lead")
        stages = {}
        for inner in (False, True):        # the pass without records inside it; then its split
            st = case["stage"](inner)[6:]
            for j, nm in enumerate(stage_names):
                if (j >= 5) == inner and pct([r[j] for r in st], 0.5) >= 0:
                    stages[nm] = round(pct([r[j] for r in st], 0.5), 4)
        k_ms = stages["count_pass"]
        tn_sum = int(tn.sum().item())
        evals = tn_sum * K * hn
        alg = synth.dense_field_bytes(B, H, W, K, hn)
        err = float((out - d["kpt_2d"]).abs().max())
        row = {"B": B, "H": H, "W": W, "K": K, "hn": hn, "max_num": max_num, "layout": case["layout"], "tn_mean": round(tn_sum / B, 1),
               "calls": n,
               "wall_ms_per_call": round(1e3 * wall / n, 4), "images_per_s_wall": round(B * n / wall, 1),
               "host_enqueue_ms_per_call_backlogged": round(1e3 * t_enq / n, 4),
               "host_ms_per_call_idle_stream": round(1e3 * sorted(host)[len(host) // 2], 4),
               "event_ms_per_call_median": round(pct(per, 0.5), 4), "event_ms_p10": round(pct(per, 0.1), 4),
               "event_ms_p90": round(pct(per, 0.9), 4),
               "graph_replay_ms_per_call": round(graph_ms, 4) if isinstance(graph_ms, float) else graph_ms,
               "count_kernel_ms": round(k_ms, 4), "count_pass_staged": "k_lead" in stages, "kernels_inside_calls_ms": stages,
               "evaluations": evals,
               "tevals_per_s": round(evals / (k_ms * 1e-3) / 1e12, 2),
               "dense_field_bytes": alg, "roofline_frac_hbm_8TBs": round(alg / (k_ms * 1e-3) / 8e12, 4),
               "known_answer_max_err_px": round(err, 3)}
        res["rows"][name] = row
        print(name, json.dumps(row), flush=True)
        del d, case, call
        torch.cuda.empty_cache()
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
