#!/usr/bin/env python
"""bench.py -- images/s of the RANSAC-voting hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL)

A *step* is one ``ransac_voting_layer_v3(mask, vertex, 512, inlier_thresh=0.99)`` call through the drop-in Python API
on a device-resident synthetic batch (free-running device RNG, exactly the call of resnet18.py:71), followed -- whenever
a process group exists -- by the RCCL all_gather of the ``[B,K,2]`` keypoints, waited for INSIDE the step it belongs to.

Workload = BASELINE config 3: 480x640, K=9, 512 hypotheses, ~2 % foreground, GLOBAL batch 64.
  N = 1   the whole batch on one GPU (the configuration the roofline target is quoted on);
  N > 1   STRONG scaling, as config 3 says ("batch=64 sharded over 8xMI355X"): the same 64 images, 64/N per GPU,
          contiguous shards (clean_pvnet_amd.dist.shard_bounds), every image generated from its global index.
          The weak-scaling figure (64 images PER GPU) and the variant that overlaps the exchange of step i with the
          voting of step i+1 are reported in ``extra`` (--no-weak skips them).
Steps cycle over --rotate (default 3) distinct device-resident batches, so that neither the 256 MiB Infinity Cache nor
the L2 holds a step's inputs from the step before.

One JSON line on rank 0.  Besides the contract fields it carries
  step_ms       per-step HIP events on the launch stream: median / p10 / p90 (the contract's ``ms_per_step`` is the wall
                clock over the K steps, barrier + synchronize on both sides, max over ranks)
  roofline      the inlier-count kernel (dominant): dense-field algorithmic bytes / its duration, measured here with HIP
                events around re-launches of that kernel alone; ``traffic`` from the committed PMC file (static)
  roofline_valu the same kernel against what actually bounds it: fp32 VALU issue
  cpu_baseline  the oracle (oracle/vote_oracle.c, OpenMP) on the host cores for a bounded sample of the same images,
                rank 0, N = 1 only
  extra         per-phase numbers; with --extras also B=1 latency (config 2), v3 + estimate, the default path, decode
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
# VALU-issue model of k_count_bf16 (tools/microbench): one 16-pixel x 32-hypothesis matrix-core tile
# (512 evaluations) costs 21 VALU instructions in the steady-state loop, ~4.2 cycles each per SIMD; 1024 SIMDs
VALU_PER_TILE, CYCLES_PER_VALU, N_SIMD, EVALS_PER_TILE = 21, 4.2, 1024, 512


def pct(v, q):
    v = sorted(v)
    return v[min(len(v) - 1, int(q * len(v)))]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64, help="GLOBAL batch (images over all GPUs)")
    ap.add_argument("--rotate", type=int, default=3, help="distinct device-resident batches the steps cycle over")
    ap.add_argument("--prewarm-ms", type=float, default=60.0,
                    help="untimed steps run for this long BEFORE the --warmup steps: after the (GPU-idle) data generation the "
                         "chip needs ~60-100 steps (20-30 ms) to reach steady clocks -- tools/clock_ramp.py: 0.335 ms/step at "
                         "step 8, 0.300 at step 30, 0.279 from step 60 on -- so a 25-step run would time the ramp, not the path")
    ap.add_argument("--config", default="cfg3", help="image shape / K / hn / foreground of this BASELINE config")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-two-stream", action="store_true",
                    help="N = 1: skip the two-stream extra (profiling runs: overlapped launches would blur per-kernel durations)")
    ap.add_argument("--no-weak", action="store_true", help="N > 1: skip the weak-scaling and overlapped variants")
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
    H, W, K, hn = cfg["H"], cfg["W"], cfg["K"], cfg["hn"]
    thresh = 0.99
    gen_cfg = {k: v for k, v in cfg.items() if k not in ("B", "hn")}
    global_batch = args.batch
    lo, hi = pdist.shard_bounds(global_batch, world, rank)            # this rank's contiguous shard
    B = hi - lo
    # --rotate distinct batches (excluded from timing); batch r holds the global images r*global_batch + [lo, hi)
    batches = [synth.make_batch(B=B, **gen_cfg, first_index=r * global_batch + lo, device=dev) if B > 0 else None
               for r in range(max(1, args.rotate))]
    torch.manual_seed(1234 + rank)
    rccl_ranks = None
    if use_dist:
        probe = pdist.gather_results(torch.full((B, 1), float(rank), device=dev), global_batch)     # a real collective
        rccl_ranks = int(probe.unique().numel()) if global_batch >= world else dist.get_world_size()
        assert dist.get_world_size() == world

    def vote(d):
        if d is None:                                                 # a rank without images still enters the collective
            return torch.zeros((0, K, 2), device=dev)
        return ransac_voting_layer_v3(d["mask"], d["vertex"], hn, inlier_thresh=thresh)

    prewarm_done = [0]

    def run(step_fn, warmup, steps):
        """(clock pre-warm, untimed) + W untimed + exactly K timed steps, barrier + synchronize on both sides, max over
        ranks; per-step HIP events."""
        if args.prewarm_ms > 0:
            t_pre = time.perf_counter()
            n_pre = 0
            while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms or n_pre % 8:
                step_fn(n_pre)                 # a multiple of 8 steps on every rank (collectives stay matched) ...
                n_pre += 1
                if n_pre % 8 == 0:
                    torch.cuda.synchronize()   # ... and the clock is read with the queue drained
                    if use_dist:               # same count everywhere: rank 0 decides
                        flag = torch.tensor([1.0 if (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms else 0.0], device=dev)
                        dist.broadcast(flag, 0)
                        if flag.item() == 0.0:
                            break
            prewarm_done[0] = max(prewarm_done[0], n_pre)
        for i in range(warmup):
            step_fn(i)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        t0 = time.perf_counter()
        evs[0].record()
        out = None
        for i in range(steps):
            out = step_fn(warmup + i)
            evs[i + 1].record()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        per = [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]
        return elapsed, per, out

    def step(i):
        """vote on this rank's shard of batch i, then the RCCL all_gather of the [B,K,2] keypoints -- enqueued behind the
        voting kernels and completed (stream-ordered) inside this step: nothing of step i overlaps step i+1."""
        local = vote(batches[i % len(batches)])
        return pdist.gather_results(local, global_batch) if use_dist else local

    elapsed, per_step, out = run(step, args.warmup, args.steps)
    ms_per_step = 1e3 * elapsed / args.steps
    value = global_batch * args.steps / elapsed
    last = batches[(args.warmup + args.steps - 1) % len(batches)]
    # known-answer sanity of what was timed: voting recovers the keypoints the field was built from
    err = float((out[lo:hi] - last["kpt_2d"]).abs().max()) if B > 0 else 0.0

    # N = 1 extra (never `value`): the same steps alternating over two streams, as a caller that decodes a sequence of
    # batches would issue them -- the scan / compaction of step i+1 run under the (VALU-bound) count kernel of step i.
    # Splitting ONE call over streams loses instead (tools/two_stream.py), so the library does not do that by itself.
    two_stream = None
    if not use_dist and B > 0 and not args.no_two_stream:
        from clean_pvnet_amd.pipeline import StreamRing
        ring = StreamRing(2, dev)
        def alternating_step(i):
            return ring.run(vote, batches[i % len(batches)])
        n2 = max(10, args.steps // 2)
        ts_el, _per, _o = run(alternating_step, 6, n2)
        ring.join()
        two_stream = {"two_stream_images_per_s": round(global_batch * n2 / ts_el, 1), "two_stream_ms_per_step": round(1e3 * ts_el / n2, 4)}

    # N > 1 extras: weak scaling (global_batch images PER GPU) and the exchange overlapped with the next step's voting
    weak = None
    if use_dist and world > 1 and not args.no_weak:
        wb = [synth.make_batch(B=global_batch, **gen_cfg, first_index=(100 + r) * global_batch * world + rank * global_batch,
                               device=dev) for r in range(2)]
        def weak_step(i):
            d = wb[i % 2]
            local = ransac_voting_layer_v3(d["mask"], d["vertex"], hn, inlier_thresh=thresh)
            return pdist.gather_results(local, global_batch * world)
        w_el, w_per, _ = run(weak_step, 5, max(10, args.steps // 4))
        pending = []
        def overlapped_step(i):
            local = vote(batches[i % len(batches)])
            while pending:
                pending.pop()[1].wait()
            ow = pdist.gather_results(local, global_batch, async_op=True)
            pending.append(ow)
            return ow[0]
        o_el, o_per, _ = run(overlapped_step, 5, max(10, args.steps // 4))
        while pending:
            pending.pop()[1].wait()
        n2 = max(10, args.steps // 4)
        weak = {"weak_scaling_images_per_s": round(global_batch * world * n2 / w_el, 1),
                "weak_scaling_ms_per_step": round(1e3 * w_el / n2, 4), "weak_batch_per_gpu": global_batch,
                "strong_overlapped_exchange_images_per_s": round(global_batch * n2 / o_el, 1)}

    # per-rank duration of the dominant kernel, HIP events on the launch stream around groups of back-to-back re-launches
    # of that kernel alone (pvv_rerun_count_kernel; the host's launch latency hides behind the previous launch, so the
    # figure is the kernel's duration plus the ~1.5 us kernel boundary -- what rocprofv3 --kernel-trace reports for it).
    # tools/count_kernel_timing.py shows this, a differential measurement (steps with one extra count launch), HIP events
    # recorded around the launch inside full calls (pvv_problem.ev_count_begin/end) and rocprofv3's own timestamps of
    # the in-pipeline launches agreeing within 2 % in one process; the events-inside-calls figure is reported in `extra`
    # (it reads up to ~10 % long on some boxes and under the profiler: the event packets themselves).
    k_ms, k_avg_ms, k_incall_ms, tn_cpu = [0.0], 0.0, 0.0, torch.zeros(0)
    if B > 0:
        d0 = batches[0]
        t_pre = time.perf_counter()                                   # the clocks dropped during the idle moments since the timed
        while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:  # region: same pre-warm as there, then measure
            for i in range(8):
                vote(batches[i % len(batches)])
            torch.cuda.synchronize()
        _o, win, tn, ws = ext.ransac_voting_v3(d0["mask"], d0["vertex"], hn, thresh, 5, 30000, None, None, 1, ext.SINGULAR_REFERENCE)
        groups, per_group = 5, 10
        for _ in range(3):
            ext.rerun_count_kernel(d0["mask"], d0["vertex"], hn, thresh, 5, 30000, ws, False)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(groups)]
        for a, b in evs:
            a.record()
            for _ in range(per_group):
                ext.rerun_count_kernel(d0["mask"], d0["vertex"], hn, thresh, 5, 30000, ws, False)
            b.record()
        torch.cuda.synchronize()
        k_ms = sorted(a.elapsed_time(b) / per_group for a, b in evs)
        k_avg_ms = sum(k_ms) / len(k_ms)
        ic = ext.count_kernel_ms_in_pipeline([d["mask"] for d in batches], [d["vertex"] for d in batches], hn, thresh, 5, 30000, 12, 24)
        k_incall_ms = sum(ic[6:]) / len(ic[6:])
        tn_cpu = torch.cat([ext.ransac_voting_v3(d["mask"], d["vertex"], hn, thresh, 5, 30000, None, None, 1,
                                                 ext.SINGULAR_REFERENCE)[2].cpu() for d in batches]).view(len(batches), -1)
        tn_cpu = tn_cpu.float().mean(0)                               # foreground pixels per image slot, mean over the batches
    per_rank_kernel_ms = [round(k_avg_ms, 4)]
    if use_dist:
        g = torch.zeros(world, dtype=torch.float64, device=dev)
        g[rank] = k_avg_ms
        dist.all_reduce(g)
        per_rank_kernel_ms = [round(float(x), 4) for x in g.cpu()]

    result = None
    if rank == 0:
        alg_bytes = synth.dense_field_bytes(B, H, W, K, hn)
        achieved = alg_bytes / (k_avg_ms * 1e-3) / 1e9 if k_avg_ms > 0 else 0.0
        evals = int(round(float(tn_cpu.sum().item()))) * K * hn     # per launch (mean over the rotating batches)
        traffic, traffic_source = None, None
        pmc_path = os.path.join(ROOT, "profiles", "count_kernel_pmc.json")
        if os.path.exists(pmc_path):
            try:
                pmc = json.load(open(pmc_path))
                if pmc.get("workload") == "%s_B%d" % (args.config, B):
                    traffic = pmc.get("hbm_bytes_per_launch")
                    traffic_source = "profiles/count_kernel_pmc.json (static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, not measured in this run)"
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                    "kernel": "k_count_bf16", "kernel_ms_avg": round(k_avg_ms, 4),
                    "kernel_ms_median": round(k_ms[len(k_ms) // 2], 4),
                    "kernel_ms_how": "HIP events on the launch stream around %d groups of 10 back-to-back re-launches of the kernel alone" % len(k_ms),
                    "algorithmic_bytes": alg_bytes,
                    "evaluations": evals, "gevals_per_s": round(evals / (k_avg_ms * 1e-3) / 1e9, 1) if k_avg_ms else 0.0,
                    "note": "contract figure (SURVEY 8d dense-field bytes / kernel time); the kernel reads the compacted "
                            "foreground only and is bound by VALU issue -- see roofline_valu"}
        clock_ghz = torch.cuda.get_device_properties(dev).clock_rate / 1e6 if hasattr(torch.cuda.get_device_properties(dev), "clock_rate") else 2.4
        peak_evals = N_SIMD * clock_ghz * 1e9 * EVALS_PER_TILE / (VALU_PER_TILE * CYCLES_PER_VALU)
        ach_evals = evals / (k_avg_ms * 1e-3) if k_avg_ms else 0.0
        roofline_valu = {"bound": "valu_issue", "achieved": round(ach_evals / 1e12, 3), "peak": round(peak_evals / 1e12, 3),
                         "unit": "T evaluations/s", "frac": round(ach_evals / peak_evals, 4),
                         "model": "%d SIMDs x %.2f GHz (device max clock) x %d evaluations per matrix-core tile / (%d VALU x %.1f "
                                  "cycles): the steady-state loop with no prologue, no flagged tiles and no idle SIMD"
                                  % (N_SIMD, clock_ghz, EVALS_PER_TILE, VALU_PER_TILE, CYCLES_PER_VALU)}

        extra = {"tn_mean": round(float(tn_cpu.float().mean()), 1) if tn_cpu.numel() else 0.0,
                 "known_answer_max_err_px": round(err, 3),
                 "rotating_batches": len(batches), "bytes_per_batch_per_gpu": int(B * H * W * (8 + K * 8)),
                 "prewarm": {"ms": args.prewarm_ms, "untimed_steps": prewarm_done[0],
                             "why": "clock ramp after the GPU-idle data generation (tools/clock_ramp.py); the timed region is exactly --steps steps"},
                 "images_per_gpu": B, "rccl_ranks": rccl_ranks, "per_rank_count_kernel_ms": per_rank_kernel_ms,
                 "count_kernel_ms_events_inside_calls": round(k_incall_ms, 4),
                 "exchange": ("all_gather_into_tensor of [%d,%d,2] f32 inside every step" % (global_batch, K)) if use_dist else None}
        if weak:
            extra.update(weak)
        if two_stream:
            extra.update(two_stream)
        if world == 1 and args.extras:
            extras_leg(extra, batches[0], out, ext, synth, ransac_voting_layer_v3, estimate_voting_distribution_with_mean,
                       B, H, W, K, hn, thresh, dev)

        cpu_baseline = None
        if world == 1 and not args.no_cpu_baseline:
            d0 = batches[0]
            gpu0 = ransac_voting_layer_v3(d0["mask"], d0["vertex"], hn, inlier_thresh=thresh)
            tn0 = ext.ransac_voting_v3(d0["mask"], d0["vertex"], hn, thresh, 5, 30000, None, None, 1, ext.SINGULAR_REFERENCE)[2].cpu()
            cpu_baseline = cpu_leg(d0["mask"], d0["vertex"], tn0, hn, K, thresh, args.cpu_sample, synth, gpu0)

        result = {
            "metric": "images/sec RANSAC-vote (480x640, K=9, 512 hyp)", "value": round(value, 1), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE config 3 (%s): %dx%d, K=%d, %d hypotheses, ~%.0f%% foreground, int64 mask, contiguous "
                                   "[B,H,W,K,2] f32 vertex; GLOBAL batch %d in contiguous shards of %d images per GPU; "
                                   "ransac_voting_layer_v3 (+ RCCL all_gather of [B,K,2] inside the step for N>1); steps "
                                   "cycle over %d distinct device-resident batches"
                                   % (args.config, H, W, K, hn, 100 * (cfg["fg"] if not isinstance(cfg["fg"], tuple) else cfg["fg"][1]),
                                      global_batch, B, len(batches)),
                       "batch_per_gpu": B, "global_batch": global_batch, "H": H, "W": W, "K": K, "hn": hn,
                       "inlier_thresh": thresh, "parallelism": "batch-sharded x%d (strong scaling)" % world},
            "step_ms": {"median": round(pct(per_step, 0.5), 4), "p10": round(pct(per_step, 0.1), 4),
                        "p90": round(pct(per_step, 0.9), 4), "wall": round(ms_per_step, 4),
                        "how": "torch.cuda.Event pairs around every step on the launch stream (rank 0); wall = perf_counter over the timed region / steps, max over ranks"},
            "roofline": roofline, "roofline_valu": roofline_valu, "cpu_baseline": cpu_baseline, "extra": extra,
        }
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return result


def extras_leg(extra, data, out, ext, synth, ransac_voting_layer_v3, estimate_voting_distribution_with_mean, B, H, W, K, hn,
               thresh, dev):
    """--extras (N = 1): config 2 latency, v3 + estimate, the reference's default path, decode_keypoint, ADD-S search."""
    mask, vertex = data["mask"], data["vertex"]
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
    # the whole post-network step of one frame with cfg.test.un_pnp (resnet18.py:65-72 + evaluators/linemod/pvnet.py:118-132):
    # fused decode_keypoint (mask, keypoints, covariances, PnP weights) + the batched uncertainty-PnP refinement, B = 1 and B
    from clean_pvnet_amd.un_pnp_utils import uncertainty_pnp_batched
    import numpy as _np
    rng0 = _np.random.RandomState(1)
    kpt3d = torch.tensor(rng0.uniform(-0.06, 0.06, (K, 3)), device=dev)
    Kcam = torch.tensor([[572.4114, 0, 325.2611], [0, 573.57043, 242.04899], [0, 0, 1.0]], device=dev, dtype=torch.float64)
    for nb, tag in ((1, "B1"), (B, "B%d" % B)):
        segb, verb = seg[:nb], ver[:nb]
        init = torch.tensor(_np.tile(_np.array([0.1, -0.2, 0.3, 0.0, 0.0, 0.8]), (nb, 1)), device=dev)

        def frame():
            o = decode_keypoint({"seg": segb, "vertex": verb}, un_pnp=True, weights=True)
            return uncertainty_pnp_batched(o["kpt_2d"], o["var_weights"], kpt3d, Kcam, init)
        for _ in range(3):
            frame()
        torch.cuda.synchronize()
        t7 = time.perf_counter()
        nrep = 50 if nb == 1 else 10
        for _ in range(nrep):
            frame()
        torch.cuda.synchronize()
        extra["un_pnp_decode_plus_pnp_%s_ms_per_call" % tag] = round(1e3 * (time.perf_counter() - t7) / nrep, 4)
        o = decode_keypoint({"seg": segb, "vertex": verb}, un_pnp=True, weights=True)
        for _ in range(3):
            uncertainty_pnp_batched(o["kpt_2d"], o["var_weights"], kpt3d, Kcam, init)
        torch.cuda.synchronize()
        t8 = time.perf_counter()
        for _ in range(50):
            uncertainty_pnp_batched(o["kpt_2d"], o["var_weights"], kpt3d, Kcam, init)
        torch.cuda.synchronize()
        extra["uncertainty_pnp_batched_%s_ms_per_call" % tag] = round(1e3 * (time.perf_counter() - t8) / 50, 4)
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
