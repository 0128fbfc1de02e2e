#!/usr/bin/env python
"""bench.py -- images/s of the RANSAC-voting hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL)

A *step* is one ``ransac_voting_layer_v3(mask, vertex, 512, inlier_thresh=0.99)`` call through the
drop-in Python API on a device-resident synthetic batch (free-running device RNG, exactly the call of
resnet18.py:71), followed -- for N > 1 -- by the RCCL all_gather of the ``[B,K,2]`` keypoints.
Workload: BASELINE config 3's image shape (480x640, K=9, 512 hypotheses, ~2 % foreground) at the batch
the roofline target is quoted on, B = 64 PER GPU (weak scaling: N GPUs vote on 64*N images).

One JSON line on rank 0.  Besides the contract fields it carries
  roofline      the inlier-count kernel (dominant): dense-field algorithmic bytes / its duration,
                measured here with HIP events around re-launches of that kernel alone
  cpu_baseline  the oracle (oracle/vote_oracle.c, OpenMP) on the host cores for a bounded sample of
                the same images, rank 0, N = 1 only
  extra         per-phase numbers: kernel duration, evaluations/s, B=1 latency (config 2),
                v3 + estimate_voting_distribution_with_mean throughput
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is what a copy reaches


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU")
    ap.add_argument("--config", default="cfg3", help="image shape / K / hn / foreground of this BASELINE config")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extras", action="store_true",
                    help="also time config 2 (B=1 latency) and v3+estimate; off by default so that a rocprofv3 "
                         "--stats run of the default command sees the count kernel at ONE problem size")
    ap.add_argument("--cpu-sample", type=int, default=8, help="distinct images the CPU oracle cycles over")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU path exists)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ     # a 1-rank torchrun exercises the RCCL path too
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import lib
    lib._register_clean_pvnet_amd()
    from clean_pvnet_amd import dist as pdist
    from clean_pvnet_amd import ransac_voting as ext
    from clean_pvnet_amd import synth
    from lib.csrc.ransac_voting.ransac_voting_gpu import (estimate_voting_distribution_with_mean,
                                                          ransac_voting_layer_v3)

    cfg = dict(synth.CONFIGS[args.config])
    B, H, W, K, hn = args.batch, cfg["H"], cfg["W"], cfg["K"], cfg["hn"]
    thresh = 0.99
    gen_cfg = {k: v for k, v in cfg.items() if k not in ("B", "hn")}
    data = synth.make_batch(B=B, **gen_cfg, first_index=rank * B, device=dev)     # excluded from timing
    mask, vertex = data["mask"], data["vertex"]
    global_batch = B * world
    torch.manual_seed(1234 + rank)

    pending = []

    def step():
        """vote on this rank's shard, then enqueue the RCCL all_gather of the [B,K,2] keypoints; the collective of
        step i overlaps with the voting of step i+1 and is waited for before the next one is enqueued (at most one
        in flight) and at the end of the timed region -- every step's exchange completes inside it."""
        local = ransac_voting_layer_v3(mask, vertex, hn, inlier_thresh=thresh)
        if not use_dist:
            return local
        while pending:
            pending.pop()[1].wait()
        out_w = pdist.gather_results(local, global_batch, async_op=True)
        pending.append(out_w)
        return out_w[0]

    def sync():
        while pending:
            pending.pop()[1].wait()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = global_batch * args.steps / elapsed

    # known-answer sanity of what was timed: voting recovers the keypoints the field was built from
    err = float((out[rank * B:(rank + 1) * B] - data["kpt_2d"]).abs().max())

    result = None
    if rank == 0:
        # ---- roofline of the dominant kernel: HIP events on the launch stream around re-launches ----
        _o, win, tn, ws = ext.ransac_voting_v3(mask, vertex, hn, thresh, 5, 30000, None, None, 1, ext.SINGULAR_REFERENCE)
        # groups of back-to-back re-launches between one event pair each: the host's launch latency is hidden behind
        # the previous launch, so the figure is the kernel's duration (plus the ~1.5 us kernel boundary), which is
        # what rocprofv3 --kernel-trace reports for it
        groups, per_group = 5, 10
        for _ in range(3):
            ext.rerun_count_kernel(mask, vertex, hn, thresh, 5, 30000, ws, False)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(groups)]
        for a, b in evs:
            a.record()
            for _ in range(per_group):
                ext.rerun_count_kernel(mask, vertex, hn, thresh, 5, 30000, ws, False)
            b.record()
        torch.cuda.synchronize()
        k_ms = sorted(a.elapsed_time(b) / per_group for a, b in evs)
        k_avg_ms = sum(k_ms) / len(k_ms)
        alg_bytes = synth.dense_field_bytes(B, H, W, K, hn)
        achieved = alg_bytes / (k_avg_ms * 1e-3) / 1e9
        tn_cpu = tn.cpu()
        evals = int(tn_cpu.sum().item()) * K * hn
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "count_kernel_pmc.json")
        if os.path.exists(pmc_path):
            try:
                pmc = json.load(open(pmc_path))
                if pmc.get("workload") == "%s_B%d" % (args.config, B):
                    traffic = pmc.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "kernel": {"exact": "k_count_inliers", "fast": "k_count_fast"}.get(os.environ.get("PVV_COUNT_KERNEL", ""), "k_count_bf16"), "kernel_ms_avg": round(k_avg_ms, 4),
                    "kernel_ms_median": round(k_ms[len(k_ms) // 2], 4), "algorithmic_bytes": alg_bytes,
                    "evaluations": evals, "gevals_per_s": round(evals / (k_avg_ms * 1e-3) / 1e9, 1)}

        extra = {"tn_mean": round(float(tn_cpu.float().mean()), 1), "known_answer_max_err_px": round(err, 3)}
        if world == 1 and args.extras:
            # config 2: latency of one 480x640 image (B = 1), same call
            m1, v1 = mask[:1], vertex[:1]
            for _ in range(10):
                ransac_voting_layer_v3(m1, v1, hn, inlier_thresh=thresh)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n1 = 100
            for _ in range(n1):
                ransac_voting_layer_v3(m1, v1, hn, inlier_thresh=thresh)
            torch.cuda.synchronize()
            extra["cfg2_B1_ms_per_image"] = round(1e3 * (time.perf_counter() - t1) / n1, 4)
            # the un_pnp path of resnet18.py:71-72: v3 + estimate (4096 hypotheses)
            for _ in range(2):
                estimate_voting_distribution_with_mean(mask, vertex, out)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            n2 = 5
            for _ in range(n2):
                mean = ransac_voting_layer_v3(mask, vertex, hn, inlier_thresh=thresh)
                estimate_voting_distribution_with_mean(mask, vertex, mean)
            torch.cuda.synchronize()
            extra["v3_plus_estimate_images_per_s"] = round(B * n2 / (time.perf_counter() - t2), 1)
            # the reference's default (non-un_pnp) call, resnet18.py:75: 128 hypotheses on ~100 subsampled pixels
            for _ in range(3):
                ransac_voting_layer_v3(mask, vertex, 128, inlier_thresh=thresh, max_num=100)
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            for _ in range(20):
                kp = ransac_voting_layer_v3(mask, vertex, 128, inlier_thresh=thresh, max_num=100)
            torch.cuda.synchronize()
            extra["default_path_hn128_maxnum100_images_per_s"] = round(B * 20 / (time.perf_counter() - t4), 1)
            extra["default_path_known_answer_max_err_px"] = round(float((kp - data["kpt_2d"]).abs().max()), 2)
            # decode_keypoint with torch.argmax + v3 vs the argmax fused into the mask scan
            from clean_pvnet_amd.decode import decode_keypoint
            x = torch.randn(B, 2 + 2 * K, H, W, device=dev) * 0.1
            x[:, 1] += 3.0 * (mask != 0)
            x[:, 0] += 3.0 * (mask == 0)
            x[:, 2:] = vertex.permute(0, 3, 4, 1, 2).reshape(B, 2 * K, H, W)
            seg, ver = x[:, :2], x[:, 2:]

            def unfused():
                vtx = ver.permute(0, 2, 3, 1).view(B, H, W, K, 2)
                m = torch.argmax(seg, 1)
                return ransac_voting_layer_v3(m, vtx, hn, inlier_thresh=thresh, max_num=30000)

            def fused():
                return ext.decode_keypoint_v3(seg, ver.permute(0, 2, 3, 1).view(B, H, W, K, 2), hn, thresh, 5, 30000,
                                              None, None, 7, ext.SINGULAR_REFERENCE)[0]
            def un_pnp_two_calls():      # resnet18.py:69-72 as the reference runs it: argmax, v3, estimate
                vtx = ver.permute(0, 2, 3, 1).view(B, H, W, K, 2)
                m = torch.argmax(seg, 1)
                mean = ransac_voting_layer_v3(m, vtx, 512, inlier_thresh=0.99)
                return estimate_voting_distribution_with_mean(m, vtx, mean)

            def un_pnp_one_pass():       # the same, one mask scan / compaction / hypothesis + count launch
                return decode_keypoint({"seg": seg, "vertex": ver}, un_pnp=True)["var"]
            for name, fn in (("decode_unfused_images_per_s", unfused), ("decode_fused_images_per_s", fused),
                             ("decode_un_pnp_two_calls_images_per_s", un_pnp_two_calls),
                             ("decode_un_pnp_one_pass_images_per_s", un_pnp_one_pass)):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                t3 = time.perf_counter()
                for _ in range(20):
                    fn()
                torch.cuda.synchronize()
                extra[name] = round(B * 20 / (time.perf_counter() - t3), 1)
            del decode_keypoint
            # the ADD-S nearest-neighbour search at a LINEMOD-sized model (5841 points, both clouds)
            import numpy as np
            from clean_pvnet_amd.nn_utils import find_nearest_point_idx
            from oracle import vote_oracle as _vo
            rng = np.random.RandomState(0)
            ref = (rng.randn(5841, 3) * 0.05).astype(np.float32)
            que = (ref + rng.randn(5841, 3) * 0.002).astype(np.float32)
            for _ in range(3):
                idx = find_nearest_point_idx(ref, que)
            t5 = time.perf_counter()
            for _ in range(20):
                idx = find_nearest_point_idx(ref, que)
            extra["adds_nn_5841pts_ms_host_call"] = round(1e3 * (time.perf_counter() - t5) / 20, 3)
            t6 = time.perf_counter()
            want = _vo.find_nearest_point_idx(ref, que)
            extra["adds_nn_5841pts_ms_cpu_oracle"] = round(1e3 * (time.perf_counter() - t6), 2)
            extra["adds_nn_indices_equal_oracle"] = bool((idx == want).all())

        cpu_baseline = None
        if world == 1 and not args.no_cpu_baseline:
            cpu_baseline = cpu_leg(mask, vertex, tn_cpu, hn, K, thresh, args.cpu_sample, synth, out)

        result = {
            "metric": "images/sec RANSAC-vote (480x640, K=9, 512 hyp)", "value": round(value, 1), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s image shape, B=%d per GPU: %dx%d, K=%d, %d hypotheses, ~%.0f%% foreground, "
                                   "int64 mask, contiguous [B,H,W,K,2] f32 vertex; ransac_voting_layer_v3 (+ RCCL "
                                   "all_gather of [B,K,2] for N>1)" % (args.config, B, H, W, K, hn,
                                                                       100 * (cfg["fg"] if not isinstance(cfg["fg"], tuple) else cfg["fg"][1])),
                       "batch_per_gpu": B, "global_batch": global_batch, "H": H, "W": W, "K": K, "hn": hn,
                       "inlier_thresh": thresh, "parallelism": "batch-sharded x%d" % world},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "extra": extra,
        }
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return result


def cpu_leg(mask, vertex, tn, hn, K, thresh, n_sample, synth, gpu_out, budget_s=12.0):
    """The CPU oracle (a port: the reference has no CPU path) on images of the same batch, all host cores via
    OpenMP over hypotheses, repeated until ~budget_s seconds of CPU work have been timed; the keypoints it finds
    are cross-checked against the GPU's (different RNG draws, same field => same keypoints within a pixel or so)."""
    import numpy as np
    from oracle import vote_oracle
    vote_oracle.lib()
    n = min(n_sample, mask.shape[0])
    m = mask[:n].cpu().numpy()
    v = vertex[:n].cpu().numpy()
    idxs = synth.make_idxs([int(t) for t in tn[:n]], hn, K).numpy()
    vote_oracle.ransac_voting_layer_v3(m[:1], v[:1], hn, thresh, idxs=idxs[:1])       # warm-up
    # host CPUs visible != CPUs usable (cgroup quotas): pick the OpenMP thread count that is actually fastest
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cand, best = sorted({max(1, avail >> s) for s in range(0, 6)} | {min(avail, 8)}, reverse=True), None
    for nthr in cand:
        vote_oracle.set_num_threads(nthr)
        ts = time.perf_counter()
        for _ in range(2):
            vote_oracle.ransac_voting_layer_v3(m[:1], v[:1], hn, thresh, idxs=idxs[:1])
        dt1 = (time.perf_counter() - ts) / 2
        if best is None or dt1 < best[0]:
            best = (dt1, nthr)
    vote_oracle.set_num_threads(best[1])
    t0 = time.perf_counter()
    done = 0
    outs = {}
    while time.perf_counter() - t0 < budget_s:
        i = done % n
        outs[i] = vote_oracle.ransac_voting_layer_v3(m[i:i + 1], v[i:i + 1], hn, thresh, idxs=idxs[i:i + 1])
        done += 1
    dt = time.perf_counter() - t0
    diff = max(float(np.abs(outs[i] - gpu_out[i:i + 1].cpu().numpy()).max()) for i in outs)
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(done / dt, 3), "unit": "images/s", "cores": vote_oracle.num_threads(), "kind": "port",
            "sample": "%d single-image ransac_voting_layer_v3 calls cycling over %d of the timed 480x640 images "
                      "(compaction in numpy, hypotheses + counting + refit in C/OpenMP), %.1f s" % (done, n, dt),
            "cpu_model": model, "host_cpus": os.cpu_count(), "max_abs_diff_vs_gpu_px": round(diff, 3)}


if __name__ == "__main__":
    main()
