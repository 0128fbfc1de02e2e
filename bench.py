#!/usr/bin/env python
"""bench.py -- images/s of the RANSAC-voting hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: the driver launches it under torch.distributed.run, one rank per GPU, RCCL; launched BARE with --gpus N > 1 it
     starts its N ranks itself the same way)

A *step* is one ``ransac_voting_layer_v3(mask, vertex, 512, inlier_thresh=0.99)`` call through the drop-in Python API
on a device-resident synthetic batch (free-running device RNG, exactly the call of resnet18.py:71), followed -- whenever
a process group exists -- by the RCCL all_gather of the ``[B,K,2]`` keypoints: enqueued asynchronously behind the step's
voting kernels (RCCL's own stream) and waited for once the NEXT step's kernels have been launched, so the 72 B/image
exchange runs beside them -- or completed inside its step (the launch stream waits for it), whichever a 30-step
calibration before the timed region finds faster on this node (``--exchange auto``, the default; ``extra.exchange`` and
``extra.exchange_calibration`` say which and why; the other variant is reported in ``extra`` for N > 1).  Either way all
K exchanges complete inside the timed region: the last one is ordered before the closing barrier and device synchronize.

Workload = BASELINE config 3: 480x640, K=9, 512 hypotheses, ~2 % foreground, 64 images.
  N = 1   the 64 images on one GPU (the configuration the roofline target is quoted on);
  N > 1   STRONG scaling (default since round 5, ``"scaling": "strong"``: VERDICT r4 #1a): BASELINE config 3 read literally,
          "batch=64 sharded over 8xMI355X" -- the same 64 images in contiguous shards of 64/N
          (clean_pvnet_amd.dist.shard_bounds), one in-place RCCL all_gather of the [64,K,2] keypoints per step.  ``value`` is
          that; the same run also measures WEAK scaling (64 images PER GPU, global batch 64 N: every rank decodes the batch its
          own network produced -- how the path is deployed) and reports it as the top-level ``value_weak`` (``--scaling weak``
          swaps the two; ``value_strong`` / ``value_weak`` always name both).  Every image is generated from its global index.
Steps cycle over --rotate (default 3) distinct device-resident batches, so that neither the 256 MiB Infinity Cache nor
the L2 holds a step's inputs from the step before.

One JSON line on rank 0.  Besides the contract fields it carries
  value_at_rho_0.90  the same call on fields with 9.5 % outlier pixels (the staged count's gain depends on clean fields)
  step_ms          per-step HIP events on the launch stream: median / p10 / p90 (the contract's ``ms_per_step`` is the wall
                   clock over the K steps, barrier + synchronize on both sides, max over ranks)
  roofline         the WHOLE CALL against HBM: SURVEY 8d's dense-field bytes / ms_per_step (bounded: the call consumes the
                   field once); ``traffic`` = the bytes its kernels really move (profiles/call_pmc.json, static), ``traffic_frac``
  roofline_contract_count_pass   rounds 1-3's contract figure (dense-field bytes / the count pass): not a bound, kept for continuity
  roofline_scan, roofline_compact   the two HBM-facing kernels against what they move, and against the box's own
                   streaming-read rate (pvv_stream_read_probe, measured in this run)
  roofline_valu    the count pass on ISSUED VALU instructions (SQ_INSTS_VALU, static) / its duration inside calls (live) against
                   1024 SIMDs x max clock / 2 cycles; ``busy_frac`` = SQ_ACTIVE_INST_VALU x 4 / SIMD cycles from the counters
  cpu_baseline     the oracle (oracle/vote_oracle.c) on the host: one thread and OpenMP over the cores (the thread probe's table
                   included), a bounded sample of the timed images -- and the SAME images with the SAME index pairs through the
                   GPU path, cross-checked (winner counts equal, means within the contract); rank 0, N = 1 only
  extra            per-kernel durations inside calls, rank / shard bookkeeping, the un_pnp path (v3 + estimate, two calls and one
                   fused pass, the estimate's count pass with its VALU block), the fused decode on the real caller's layout,
                   predicted_8gpu (from the tracked one-GPU profile; unmeasured); with --extras also B=1 latency (config 2),
                   the default path, uncertainty PnP, ADD-S
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is what a copy reaches
# VALU issue: 1024 SIMDs; a wave64 VALU instruction occupies its SIMD for 2 cycles at full rate (MI355X_MICROARCH.md's
# table: reached by v_fma / v_mul / v_add / v_sub at 8 waves per SIMD of nothing else, tools/microbench/valu_rate3.hip ->
# profiles/r04_microbench.txt).  The ceiling below is that full-rate one at the device's max clock -- the contract's roof.  What the
# count loop's OWN instruction mix can issue at the kernel's occupancy, and the clock the chip really runs at inside the kernels, are
# measured (round 6: tools/microbench/count_pipe3.hip, tools/census_count.py -> profiles/call_pmc.json) and reported beside it as
# roofline.frac_at_effective_clock and roofline.frac_of_measured_mix_roof.  (The "GHz implied" of the older microbenchmarks -- wave
# cycles / kernel time -- is a lower bound, not the clock: waves of one SIMD finish far apart.)
# VALU_PER_TILE: one 16-pixel x 32-hypothesis matrix-core tile (512 evaluations) costs 21 VALU in the steady-state loop.
VALU_PER_TILE, CYCLES_PER_VALU, N_SIMD, EVALS_PER_TILE = 21, 2.0, 1024, 512


def smi_snapshot(dev_index):
    """Clocks and power of one GPU as rocm-smi reports them right now (None when the tool is missing or says nothing useful)."""
    try:
        out = subprocess.run(["rocm-smi", "-d", str(dev_index), "--showclocks", "--showpower", "--showtemp", "--json"],
                             capture_output=True, text=True, timeout=20).stdout
        j = json.loads(out[out.index("{"):])
        card = next(iter(j.values()))
        keep = {k: v for k, v in card.items() if any(t in k.lower() for t in ("sclk", "mclk", "fclk", "power", "temperature (sensor junction)"))}
        return keep or None
    except Exception as e:                                                          # never lose the bench line to this
        return {"unavailable": str(e)[:80]}


def load_profile(name):
    """A tracked, static profile file (profiles/<name>) or None."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None


def latest_configs_profile():
    """(name, rows) of the newest profiles/rNN_configs.json."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_configs.json")))
    for f in reversed(files):
        try:
            return os.path.basename(f), json.load(open(f))["rows"]
        except Exception:
            continue
    return None, {}


def pct(v, q):
    v = sorted(v)
    return v[min(len(v) - 1, int(q * len(v)))]


def relaunch_under_torchrun(n):
    """``python bench.py --gpus N`` with N > 1 and no WORLD_SIZE in the environment: start the N ranks ourselves, exactly
    as the driver does (torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1), and pass on rank 0's line."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    print("bench.py: --gpus %d without WORLD_SIZE: launching %s" % (n, " ".join(cmd[1:8])), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


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
    ap.add_argument("--no-side-legs", action="store_true",
                    help="N = 1: skip the noisy-field leg and the un_pnp leg (profiling runs: one problem size per kernel)")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="N > 1: strong (default) = the SAME --batch images cut into N contiguous shards (BASELINE config 3 read "
                         "literally: batch=64 sharded over the GPUs); weak = --batch images PER GPU, global batch N x --batch (each "
                         "rank decodes the batch its own network produced: the deployment); the other mode is measured too and "
                         "reported as the top-level value_weak / value_strong")
    ap.add_argument("--no-weak", "--no-other-scaling", dest="no_other", action="store_true",
                    help="N > 1: skip the other scaling mode and the overlapped-exchange variant")
    ap.add_argument("--exchange", default="auto", choices=["auto", "overlapped", "in-step"],
                    help="N > 1: 'overlapped' enqueues the all_gather of step i asynchronously on RCCL's stream and waits for "
                         "it after step i+1's voting has been launched -- the collective runs beside the next step's kernels, "
                         "every one of the K exchanges still completes inside the timed region; 'in-step' makes the launch "
                         "stream wait for the exchange before the next step starts; 'auto' (default) times 30 steps of each "
                         "before the timed region (max over ranks, so every rank decides alike) and takes the faster one.  The "
                         "other one is reported in extra")
    ap.add_argument("--no-sustained", action="store_true",
                    help="skip the sustained leg (>= 2 s / >= 10 000 steps of the timed call; extra.sustained)")
    ap.add_argument("--exchange-impl", default="auto", choices=["auto", "rccl", "torch"],
                    help="how the all_gather is issued: 'rccl' = ncclAllGather through a ctypes binding, directly on the launch stream "
                         "(clean_pvnet_amd/rccl.py: no cross-stream events); 'torch' = torch.distributed.all_gather_into_tensor (its own "
                         "stream + two events per step); 'auto' (default) = rccl when every rank could create its communicator, else torch")
    ap.add_argument("--extras", action="store_true",
                    help="also time config 2 (B=1 latency) and v3+estimate; off by default so that a rocprofv3 "
                         "--stats run of the default command sees the count kernel at ONE problem size")
    ap.add_argument("--cpu-sample", type=int, default=8, help="distinct images the CPU oracle cycles over")
    ap.add_argument("--backend", default=None, choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (default: nccl = RCCL; gloo when the ranks outnumber the node's GPUs)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # a bare `python bench.py --gpus N`: start the ranks ourselves (the driver's own torchrun command keeps working)
        sys.exit(relaunch_under_torchrun(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:                                            # the launcher's world is the truth; never die on plumbing
        print("bench.py: --gpus %d but WORLD_SIZE=%d: using WORLD_SIZE" % (args.gpus, world), file=sys.stderr, flush=True)
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU path exists)"
    n_dev = torch.cuda.device_count()
    # More ranks than GPUs on this node (a 1-GPU box asked for --gpus 2): ranks share devices.  RCCL refuses two ranks on
    # one device, so the exchange then runs on gloo (results staged through the host) -- every other line of the N > 1
    # path is the same, which is what tests/test_gpu_dist.py uses this for.  The JSON line says so (`extra.backend`).
    oversubscribed = world > n_dev
    dev = torch.device("cuda", local_rank % n_dev)
    torch.cuda.set_device(dev)
    backend = args.backend or ("gloo" if oversubscribed else "nccl")
    use_dist = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ     # a 1-rank torchrun exercises the RCCL path too
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    cdev = dev if backend == "nccl" else torch.device("cpu")         # where the small control tensors of the bench live

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
    weak = args.scaling == "weak"
    global_batch = args.batch * world if weak else args.batch          # weak: --batch images per GPU; strong: in total
    lo, hi = pdist.shard_bounds(global_batch, world, rank)            # this rank's contiguous shard
    B = hi - lo
    # --rotate distinct batches (excluded from timing); batch r holds the global images r*global_batch + [lo, hi)
    batches = [synth.make_batch(B=B, **gen_cfg, first_index=r * global_batch + lo, device=dev) if B > 0 else None
               for r in range(max(1, args.rotate))]
    torch.manual_seed(1234 + rank)
    coll_ranks, shard_sizes = None, [B]
    if use_dist:
        assert dist.get_world_size() == world
        sizes = torch.zeros(world, dtype=torch.int64, device=cdev)   # a real collective: every rank reports its shard
        sizes[rank] = B
        dist.all_reduce(sizes)
        shard_sizes = [int(x) for x in sizes.cpu()]
        probe = pdist.gather_results(torch.full((B, 1), float(rank), device=dev), global_batch)
        coll_ranks = max(int(probe.unique().numel()), sum(1 for x in shard_sizes if x > 0)) if global_batch >= world else world
        assert sum(shard_sizes) == global_batch, "shards %r do not add up to the global batch %d" % (shard_sizes, global_batch)

    def vote(d, out=None):
        if d is None:                                                 # a rank without images still enters the collective
            return torch.zeros((0, K, 2), device=dev) if out is None else out
        return ransac_voting_layer_v3(d["mask"], d["vertex"], hn, inlier_thresh=thresh, out=out)

    # The exchange (round 5, VERDICT r4 #1b): two persistent gather buffers, rotated; the voting call writes this rank's keypoints
    # straight into its rows of the buffer (out=) and ONE in-place all_gather_into_tensor fills in the other ranks' -- no output
    # allocation, no pad, no copy per step (clean_pvnet_amd.dist.GatherBuffer).  Two, so that the result of step i stays intact
    # while step i + 1 is being enqueued (the overlapped variant waits for the collective of step i only then).
    comm, comm_note = None, None
    if use_dist and backend == "nccl" and args.exchange_impl != "torch":
        from clean_pvnet_amd import rccl
        comm = rccl.Comm.create(dev)                                   # collective: every rank, same decision on every rank
        comm_note = None if comm is not None else ("direct RCCL communicator unavailable (%s): torch.distributed's collective is used" % (rccl.Comm.last_error or "?"))
        if comm is None and args.exchange_impl == "rccl":
            raise RuntimeError(comm_note)
    gbufs = [pdist.GatherBuffer(global_batch, (K, 2), dev, comm=comm) for _ in range(2)] if use_dist else None

    prewarm_done = [0]

    def run(step_fn, warmup, steps):
        """(clock pre-warm, untimed) + W untimed + exactly K timed steps, barrier + synchronize on both sides, max over
        ranks; per-step HIP events."""
        if args.prewarm_ms > 0:
            t_pre = time.perf_counter()
            n_pre = 0
            while True:
                for _ in range(8):             # groups of 8 steps; whether another group follows is decided by rank 0's
                    step_fn(n_pre)             # clock ALONE and broadcast, so every rank runs the same number of steps
                    n_pre += 1                 # and the collectives inside them stay matched (each rank used to
                torch.cuda.synchronize()       # re-test its own clock after the broadcast)
                go = (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms
                if use_dist:
                    flag = torch.tensor([1.0 if go else 0.0], device=cdev)
                    dist.broadcast(flag, 0)
                    go = flag.item() != 0.0
                if not go:
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
            t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        per = [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]
        return elapsed, per, out

    def in_step(i):
        """vote on this rank's shard of batch i, then the RCCL all_gather of the [B,K,2] keypoints -- enqueued behind the
        voting kernels and completed (stream-ordered) inside this step: nothing of step i overlaps step i+1."""
        if not use_dist:
            return vote(batches[i % len(batches)])
        buf = gbufs[i % 2]
        vote(batches[i % len(batches)], out=buf.mine)
        return buf.gather()

    pending = []

    def overlapped_step(i):
        """the same, with the exchange enqueued asynchronously (RCCL's own stream, ordered behind this step's voting
        kernels) and waited for only after the NEXT step's voting has been launched: the 72 B/image all_gather runs beside
        the next step's kernels.  The last exchange is ordered before the closing barrier (same communicator) and the
        closing device synchronize, so all K exchanges complete inside the timed region."""
        buf = gbufs[i % 2]
        vote(batches[i % len(batches)], out=buf.mine)
        while pending:
            pending.pop()[1].wait()
        ow = buf.gather(async_op=True)
        pending.append(ow)
        return ow[0]

    def drain():
        while pending:
            pending.pop()[1].wait()

    calibration = None
    if use_dist and comm is not None:
        overlapped = False                                            # stream-ordered on the launch stream: nothing to overlap or wait for
    elif use_dist and args.exchange == "auto":
        # which way of exchanging is faster depends on the node (collective latency vs what a concurrent RCCL kernel costs
        # the voting kernels): decide by measurement, before the timed region; `run` returns the max over ranks, which is
        # the same number on every rank
        c_in, _p, _o = run(in_step, 3, 30)
        c_ov, _p, _o = run(overlapped_step, 3, 30)
        drain()
        calibration = {"in_step_ms_per_step": round(1e3 * c_in / 30, 4), "overlapped_ms_per_step": round(1e3 * c_ov / 30, 4)}
        # within noise (3 %) the simpler in-step exchange is taken: overlapping costs two more cross-stream events per step
        # and measured SLOWER on the one-rank group; it has to earn its place
        overlapped = c_ov < 0.97 * c_in
        calibration["rule"] = "overlapped only if >= 3 % faster than in-step"
    else:
        overlapped = use_dist and args.exchange == "overlapped"
    step = overlapped_step if overlapped else in_step

    elapsed, per_step, out = run(step, args.warmup, args.steps)
    drain()
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

    # N = 1 extras that belong in the DEFAULT line:
    #  * the same call on fields with ~10 % outlier pixels (winner ratio rho ~ 0.90 instead of the clean 0.995): the staged
    #    count's gain depends on clean fields (profiles/DESIGN_rounds_1-4.md 4.6), so the headline is quoted beside this one;
    #  * the path cfg.test.un_pnp runs (resnet18.py:70-72): v3 (512 hypotheses) + the 4096-hypothesis estimate, as the
    #    reference's two calls and as this library's one fused pass over seg logits + the planar vertex tensor.
    noisy, un_pnp = None, None
    if not use_dist and B > 0 and not args.no_side_legs:
        nb = [synth.make_batch(B=B, **{**gen_cfg, "outlier": 0.095}, first_index=(50 + r) * global_batch, device=dev) for r in range(2)]
        def noisy_step(i):
            d = nb[i % 2]
            return ransac_voting_layer_v3(d["mask"], d["vertex"], hn, inlier_thresh=thresh)
        n3 = max(20, args.steps // 4)
        nz_el, nz_per, _o = run(noisy_step, 10, n3)       # (its warm-up also lets the stage hint see the new fields)
        hint = ext.stage_hint(nb[0]["mask"], nb[0]["vertex"], hn)
        noisy = {"images_per_s": round(B * n3 / nz_el, 1), "ms_per_step": round(1e3 * nz_el / n3, 4), "steps": n3,
                 "outlier_pixel_fraction": 0.095, "mean_winner_ratio_rho": round(hint[1], 4) if hint[0] else None,
                 "auto_stage_threshold": round(hint[2], 4), "count_pass_staged_by_auto": bool(hint[0] and hint[1] >= hint[2]),
                 "vs_clean_headline": round((B * n3 / nz_el) / value, 4),
                 "what": "the timed call on 2 rotating batches whose foreground has 9.5 % random-direction (outlier) pixels; "
                         "AUTO picks the count mode from the stage hint (profiles/DESIGN_rounds_1-4.md 4.6)"}
        del nb
        un_pnp = un_pnp_leg(batches[0], out, ext, ransac_voting_layer_v3, estimate_voting_distribution_with_mean, B, H, W, K, hn, thresh, dev,
                            run, max(4, args.steps // 25))
        un_pnp.update(decode_leg(batches, ext, ransac_voting_layer_v3, B, H, W, K, hn, thresh, dev, run, max(20, args.steps // 4), value))
        for i in range(8):                                  # leave the stage hint as the clean batches set it
            vote(batches[i % len(batches)])
        torch.cuda.synchronize()

    # N > 1 extras: the OTHER scaling mode (strong when the headline is weak and vice versa) and the exchange overlapped with
    # the next step's voting
    other = None
    if use_dist and world > 1 and not args.no_other:
        o_global = args.batch if weak else args.batch * world          # the other mode's global batch
        olo, ohi = pdist.shard_bounds(o_global, world, rank)
        ob = [synth.make_batch(B=ohi - olo, **gen_cfg, first_index=(100 + r) * o_global + olo, device=dev) if ohi > olo else None
              for r in range(2)]
        obufs = [pdist.GatherBuffer(o_global, (K, 2), dev, comm=comm) for _ in range(2)]
        def other_step(i):
            buf = obufs[i % 2]
            vote(ob[i % 2], out=buf.mine)
            return buf.gather()
        n2 = max(10, args.steps // 2)
        w_el, w_per, _ = run(other_step, 5, n2)
        x_el, x_per, _ = run(in_step if overlapped else overlapped_step, 5, n2)
        drain()
        oname = "strong" if weak else "weak"
        other = {"%s_scaling_images_per_s" % oname: round(o_global * n2 / w_el, 1),
                 "%s_scaling_ms_per_step" % oname: round(1e3 * w_el / n2, 4),
                 "%s_scaling_global_batch" % oname: o_global, "%s_scaling_images_per_gpu" % oname: ohi - olo,
                 ("in_step_exchange_images_per_s" if overlapped else "overlapped_exchange_images_per_s"): round(global_batch * n2 / x_el, 1)}
        other["_value"] = round(o_global * n2 / w_el, 1)
        del ob, obufs

    # The sustained leg (VERDICT r4 #4b): the timed call again for >= 2 s and >= 10 000 steps on the same rotating batches --
    # long enough for the clocks to settle under this load and for an outside observer (the driver's gpu_busy sampler) to
    # see the device busy -- with the clocks and the power rocm-smi reports before and after.  Expectation: within 3 % of `value`.
    sustained = None
    if not args.no_sustained and not args.no_side_legs and B > 0:
        n_s = max(10000, int(2.0 / max(elapsed / args.steps, 1e-6)) + 1)
        smi0 = smi_snapshot(local_rank % n_dev) if rank == 0 else None
        s_el, s_per, _ = run(step, 0, n_s)
        drain()
        smi1 = smi_snapshot(local_rank % n_dev) if rank == 0 else None
        sustained = {"images_per_s": round(global_batch * n_s / s_el, 1), "steps": n_s, "seconds": round(s_el, 3),
                     "ms_per_step": round(1e3 * s_el / n_s, 4), "vs_value": round((global_batch * n_s / s_el) / value, 4),
                     "step_ms_p10_p50_p90": [round(pct(s_per, q), 4) for q in (0.1, 0.5, 0.9)],
                     "rocm_smi_before": smi0, "rocm_smi_after": smi1,
                     "what": "the timed step (same call, same rotating batches, same exchange) for max(10 000 steps, 2 s), wall clock with "
                             "barrier + synchronize on both sides, max over ranks; rocm-smi read right before and right after"}

    # Durations of the kernels AS THEY RUN INSIDE FULL CALLS: HIP events recorded by the library at the stage boundaries of
    # `reps` calls on the launch stream (pvv_problem.ev_marks), cycling over the rotating batches exactly as the timed steps
    # do -- the same sample rocprofv3 --kernel-trace sees (VERDICT r2 #3c).  The count pass = everything between the
    # compaction and the arg-max: one k_count_bf16 launch, or -- staged (count_prune.hpp) -- its two launches and k_lead.
    # A second figure re-runs the count pass alone (pvv_rerun_count_kernel; a staged pass has to clear the counters first,
    # so a memset node is inside that figure).
    stage, k_avg_ms, k_med_ms, k_rerun_ms, tn_cpu, probe = None, 0.0, 0.0, 0.0, torch.zeros(0), None
    staged_path = False
    if B > 0:
        d0 = batches[0]

        def rewarm():
            t_pre = time.perf_counter()                                   # the clocks drop during idle moments: same pre-warm
            while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:  # as the timed region, then measure
                for i in range(8):
                    vote(batches[i % len(batches)])
                torch.cuda.synchronize()
        rewarm()
        reps = 36
        # first without records INSIDE the count pass (they cost ~2 us each): the pass as rocprofv3 sees it; then with them,
        # for the split of a staged pass into its launches
        names = ("k_tile_scan", "k_compact_hyp", "count_pass", "k_select_refit", "k_finalize_v3", "count_first_launch", "k_lead")
        stage = {}
        for inner in (False, True):
            ms = ext.stage_ms_in_pipeline([d["mask"] for d in batches], [d["vertex"] for d in batches], hn, thresh, 5, 30000, 12, reps,
                                          ext.COUNT_AUTO, inner)[6:]
            for j, nm in enumerate(names):
                if (j >= 5) != inner:
                    continue
                col = sorted(r[j] for r in ms if r[j] >= 0)
                if col:
                    stage[nm] = {"avg_ms": round(sum(col) / len(col), 4), "median_ms": round(col[len(col) // 2], 4)}
            rewarm()
        staged_path = "k_lead" in stage
        k_avg_ms, k_med_ms = stage["count_pass"]["avg_ms"], stage["count_pass"]["median_ms"]
        stage["sum_avg_ms"] = round(sum(stage[nm]["avg_ms"] for nm in names[:5]), 4)
        rewarm()
        _o, win, tn, ws = ext.ransac_voting_v3(d0["mask"], d0["vertex"], hn, thresh, 5, 30000, None, None, 1, ext.SINGULAR_REFERENCE)
        groups, per_group = 5, 10
        # (a staged pass is re-run only on an explicit PVV_COUNT_STAGED: the workspace is a v3 call's)
        ck = ext.COUNT_STAGED if staged_path else ext.COUNT_FULL
        for _ in range(3):
            ext.rerun_count_kernel(d0["mask"], d0["vertex"], hn, thresh, 5, 30000, ws, True, ck)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(groups)]
        for a, b in evs:
            a.record()
            for _ in range(per_group):
                ext.rerun_count_kernel(d0["mask"], d0["vertex"], hn, thresh, 5, 30000, ws, True, ck)
            b.record()
        torch.cuda.synchronize()
        k_rerun_ms = sum(a.elapsed_time(b) / per_group for a, b in evs) / groups
        tn_cpu = torch.cat([ext.ransac_voting_v3(d["mask"], d["vertex"], hn, thresh, 5, 30000, None, None, 1,
                                                 ext.SINGULAR_REFERENCE)[2].cpu() for d in batches]).view(len(batches), -1)
        tn_cpu = tn_cpu.float().mean(0)                               # foreground pixels per image slot, mean over the batches
        # "the achievable number from a streaming-read microbenchmark on the box" -- one read-once pass over
        # each rotating batch's vertex field (1.4 GB at B = 64: nothing of it is cache-resident), 16-byte loads per lane
        if rank == 0:
            sink = torch.zeros(1, dtype=torch.int32, device=dev)
            bufs = [d["vertex"] for d in batches]
            for i in range(3):
                ext.stream_read_probe(bufs[i % len(bufs)], sink)
            pa, pb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            pa.record()
            nprobe = 6
            for i in range(nprobe):
                ext.stream_read_probe(bufs[i % len(bufs)], sink)
            pb.record()
            torch.cuda.synchronize()
            nbytes = sum(bufs[i % len(bufs)].numel() * 4 for i in range(nprobe))
            probe = {"GBs": round(nbytes / (pa.elapsed_time(pb) * 1e-3) / 1e9, 1), "bytes_per_pass": int(bufs[0].numel() * 4),
                     "how": "pvv_stream_read_probe: read-once pass over a rotating batch's vertex field, 16 B per lane, 4 loads in flight, persistent grid; HIP events around %d passes" % nprobe}
    per_rank_kernel_ms = [round(k_avg_ms, 4)]
    if use_dist:
        g = torch.zeros(world, dtype=torch.float64, device=cdev)
        g[rank] = k_avg_ms
        dist.all_reduce(g)
        per_rank_kernel_ms = [round(float(x), 4) for x in g.cpu()]

    result = None
    if rank == 0:
        alg_bytes = synth.dense_field_bytes(B, H, W, K, hn)
        achieved = alg_bytes / (k_avg_ms * 1e-3) / 1e9 if k_avg_ms > 0 else 0.0
        evals = int(round(float(tn_cpu.sum().item()))) * K * hn     # per launch (mean over the rotating batches)
        stream_gbs = probe["GBs"] if probe else None
        wl = "%s_B%d" % (args.config, B)
        # Static counter figures of THIS workload (profiles/call_pmc.json: per kernel of one call, separate rocprofv3 --pmc
        # passes of the bench command, FETCH_SIZE corrected as MI355X_MICROARCH.md prescribes; tools/refresh_profiles.py)
        call_pmc = load_profile("call_pmc.json")
        if not call_pmc or call_pmc.get("workload") != wl:
            call_pmc = None
        # the staged pass's second launch: k_count_filter_runs (round 4: runs of chunks) -- or round 3's k_count_bf16<2> where the
        # library predicts one-chunk runs; the profile says which one this workload ran
        second = "k_count_bf16<2>" if call_pmc and "k_count_filter_runs" not in call_pmc["kernels"] else "k_count_filter_runs"
        pass_names = ("k_count_bf16<1>", "k_lead", second) if staged_path else ("k_count_bf16<0>",)
        call_names = ("k_tile_scan", "k_compact_hyp") + pass_names + ("k_select_refit", "k_finalize_v3")

        def pmc_sum(names, field):
            if not call_pmc or any(n not in call_pmc["kernels"] for n in names):
                return None
            return sum(call_pmc["kernels"][n].get(field, 0) for n in names)
        pmc_src = ("profiles/call_pmc.json (static: separate rocprofv3 --pmc passes of `%s`, %s; not measured in this run)"
                   % (call_pmc.get("command", "bench.py"), call_pmc.get("round", "?"))) if call_pmc else None
        call_traffic = pmc_sum(call_names, "hbm_bytes")
        pass_traffic = pmc_sum(pass_names, "hbm_bytes")
        # ---- THE roofline block: the WHOLE CALL against HBM.  `achieved` = SURVEY 8d's dense-field bytes of
        # the batch / the step's wall time -- the call consumes the dense field exactly once, so this is bounded by the peak
        # as long as the call is slower than one streaming read of its input; `traffic` = the bytes the call's kernels really
        # move (it compacts the field to ~2 % of it after the first two kernels), `traffic_frac` = that / time / peak.
        call_gbs = alg_bytes / (ms_per_step * 1e-3) / 1e9
        roofline = {"bound": "hbm (dense-read EQUIVALENT: not the binding resource, see roofline)", "achieved": round(call_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(call_gbs / HBM_PEAK_GBS, 4),
                    "traffic": call_traffic,
                    "traffic_frac": round(call_traffic / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if call_traffic else None,
                    "traffic_source": pmc_src,
                    "kernel": "whole ransac_voting_layer_v3 call: " + " + ".join(call_names),
                    "algorithmic_bytes": alg_bytes, "ms": round(ms_per_step, 4),
                    "ms_how": "ms_per_step: wall clock of the timed region / steps (the contract's own figure); cross-check: "
                              "extra.kernels_inside_calls_ms.sum_avg_ms = HIP events at the stage boundaries inside calls",
                    "frac_of_stream_read_this_box": round(call_gbs / stream_gbs, 4) if stream_gbs else None,
                    "note": "dense-read-equivalent: [B,H,W,K,2] f32 + u8 mask + hypotheses + counts, once, per step.  The call is "
                            "NOT HBM-bound: ~60 % of it is the VALU-bound inlier-count pass on the compacted foreground "
                            "(roofline_valu); the HBM-facing kernels are roofline_scan (streams the mask) and roofline_compact "
                            "(gathers); the contract's count-pass figure, which exceeds 1, is roofline_contract_count_pass"}
        # the contract figure of rounds 1-3, kept for continuity: dense-field bytes / duration of the count pass alone
        roofline_contract = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac_not_a_bound": round(achieved / HBM_PEAK_GBS, 4), "traffic": pass_traffic, "traffic_source": pmc_src,
                    "kernel": ("inlier-count pass: " + " + ".join(pass_names)) if staged_path else "k_count_bf16",
                    "kernel_ms_avg": round(k_avg_ms, 4), "kernel_ms_median": round(k_med_ms, 4),
                    "kernel_ms_how": "HIP events recorded at the stage boundaries INSIDE full calls on the launch stream (pvv_problem.ev_marks), "
                                     "30 calls cycling over the rotating batches -- the sample rocprofv3 --kernel-trace sees",
                    "kernel_ms_rerun_alone_incl_counter_memset": round(k_rerun_ms, 4),
                    "algorithmic_bytes": alg_bytes,
                    "evaluations_full_pass": evals,
                    "note": "SURVEY 8d dense-field bytes / duration of the count pass.  The pass reads the COMPACTED foreground "
                            "(see traffic), not the dense field, and eliminates hypotheses exactly (count_prune.hpp): this "
                            "'bandwidth' is not bounded by the HBM peak and measures nothing as a fraction (VERDICT r3 weak #2)"}
        mask_bytes = B * H * W * batches[0]["mask"].element_size() if B > 0 else 0
        roofline_scan, roofline_compact = None, None
        if stage:
            sc_gbs = mask_bytes / (stage["k_tile_scan"]["avg_ms"] * 1e-3) / 1e9
            roofline_scan = {"kernel": "k_tile_scan", "bound": "hbm", "bytes": mask_bytes, "ms_avg": stage["k_tile_scan"]["avg_ms"],
                             "achieved": round(sc_gbs, 1), "unit": "GB/s", "peak": HBM_PEAK_GBS, "frac": round(sc_gbs / HBM_PEAK_GBS, 4),
                             "stream_read_GBs_this_box": stream_gbs,
                             "frac_of_stream_read": round(sc_gbs / stream_gbs, 4) if stream_gbs else None,
                             "what": "the only pass over the [B,H,W] mask (%d-byte elements), read once, cold (rotating batches)" % batches[0]["mask"].element_size()}
            tn_sum = float(tn_cpu.sum().item())
            # what the compaction must move: 2 B list entry + K gathers of 8 B in, 8 B coords + K x 8 B planar rows out per
            # foreground pixel; the tile table; 16 B of vertex field + 8 B out per hypothesis.  Gathers fetch 32-byte sectors
            # (64 B lines), so the measured FETCH_SIZE is higher -- profiles/ holds it (static)
            cmp_bytes = int(tn_sum * (2 + K * 8 + 8 + K * 8) + B * K * hn * (2 * 8 + 8 + 4))
            cm_gbs = cmp_bytes / (stage["k_compact_hyp"]["avg_ms"] * 1e-3) / 1e9
            roofline_compact = {"kernel": "k_compact_hyp", "bound": "latency (a queue of short-lived gather blocks)", "algorithmic_bytes": cmp_bytes,
                                "ms_avg": stage["k_compact_hyp"]["avg_ms"], "achieved": round(cm_gbs, 1), "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                "frac": round(cm_gbs / HBM_PEAK_GBS, 4)}
            for blk, nm in ((roofline_scan, "k_tile_scan"), (roofline_compact, "k_compact_hyp")):
                t = pmc_sum((nm,), "hbm_bytes")
                if t:
                    blk["traffic"] = t
                    blk["traffic_GBs"] = round(t / (blk["ms_avg"] * 1e-3) / 1e9, 1)
                    blk["traffic_source"] = pmc_src
        # ---- the count pass against fp32 VALU issue, on ISSUED instructions: SQ_INSTS_VALU of the pass's
        # kernels (static, per launch) / the pass's duration inside calls (live) against 1024 SIMDs x max clock / 2 cycles
        # per wave64 VALU instruction (full rate).  `busy_frac` is the counter figure proper -- the share of the pass's SIMD
        # cycles in which a VALU instruction was executing: sum_k SQ_ACTIVE_INST_VALU_k x 4 / 1024 SIMDs over
        # sum_k GRBM_GUI_ACTIVE_k / 8 XCDs -- which also sees the half-rate instructions of the loop (v_alignbit, v_min3,
        # v_cmp: ~4 cycles) and the clock the chip really ran at.
        clock_ghz = torch.cuda.get_device_properties(dev).clock_rate / 1e6 if hasattr(torch.cuda.get_device_properties(dev), "clock_rate") else 2.4
        issued = pmc_sum(pass_names, "SQ_INSTS_VALU")
        act, gui = pmc_sum(pass_names, "SQ_ACTIVE_INST_VALU"), pmc_sum(pass_names, "GRBM_GUI_ACTIVE")
        peak_issue = N_SIMD * clock_ghz * 1e9 / CYCLES_PER_VALU
        roofline_valu = {"bound": "valu_issue", "kernel": " + ".join(pass_names), "ms_avg": round(k_avg_ms, 4), "unit": "T wave-instructions/s",
                         "issued_valu_wave_instructions": issued,
                         "achieved": round(issued / (k_avg_ms * 1e-3) / 1e12, 4) if issued and k_avg_ms else None,
                         "peak": round(peak_issue / 1e12, 4),
                         "frac": round(issued / (k_avg_ms * 1e-3) / peak_issue, 4) if issued and k_avg_ms else None,
                         "busy_frac": round(act * 4 / N_SIMD / (gui / 8), 4) if act and gui else None,
                         "busy_frac_per_kernel": ({n: call_pmc["kernels"][n].get("valu_busy") for n in pass_names} if call_pmc and issued else None),
                         "evaluations_of_a_full_pass": evals, "source": pmc_src,
                         "model": "%d SIMDs x %.2f GHz (device max clock) / %.0f cycles per wave64 VALU instruction: ISSUED instructions, "
                                  "whatever they compute -- prologues, spill traffic and flagged tiles included" % (N_SIMD, clock_ghz, CYCLES_PER_VALU)}

        # ---- THE roofline block of the contract (round 5, VERDICT r4 #4a): the DOMINANT pass -- the inlier count, ~60 % of the call --
        # against the resource that binds it.  That is VALU issue, not HBM (the pass reads the compacted 2 % of the field) and not
        # the matrix pipe (mfma_busy_frac: the bf16 MFMA pipe idles three quarters of the time): issued VALU wave-instructions of
        # the pass's kernels (SQ_INSTS_VALU, static, profiles/call_pmc.json) / its duration inside calls (HIP events, live) against
        # 1024 SIMDs x max clock / 2 cycles per wave64 instruction.  `traffic` = the HBM bytes of the pass (PMC, static).
        mfma_cyc = pmc_sum(pass_names, "SQ_VALU_MFMA_BUSY_CYCLES")
        dense_equivalent = roofline
        # Round 6: the same fraction against the clock the chip really ran at INSIDE these kernels -- shader cycles /
        # real time stamped around the matrix-core loops of an instrumented build (static, profiles/call_pmc.json `effective_clock_GHz`;
        # weighted by the kernels' live durations) -- and against what the loop's own instruction mix can issue: the per-tile loop
        # alone, no MFMA, no memory, at the kernel's occupancy sustains one VALU instruction per `simd_cycles_per_valu_instruction_
        # valu_alone` SIMD cycles (tools/microbench/count_pipe3.hip; its v_alignbit / v_min3 / v_cmp do not issue at the 2 cycles of
        # the guide's full-rate figure), 4.0 beside the MFMA that feeds it.
        eff_clock, mix = None, None
        if call_pmc and issued and k_avg_ms:
            wsum, csum = 0.0, 0.0
            for nm in pass_names:
                kb_ = call_pmc["kernels"].get(nm, {})
                w_ = (stage.get({"k_count_bf16<1>": "count_first_launch", "k_lead": "k_lead"}.get(nm, ""), {}) or {}).get("avg_ms") or kb_.get("avg_us", 0.0) * 1e-3
                if kb_.get("effective_clock_GHz") and w_:
                    wsum += w_
                    csum += w_ * kb_["effective_clock_GHz"]
            eff_clock = round(csum / wsum, 3) if wsum else None
            mix = call_pmc.get("count_loop_microbench")
        ach = issued / (k_avg_ms * 1e-3) if issued and k_avg_ms else None

        def _frac(clock, cycles_per_inst):
            return round(ach / (N_SIMD * clock * 1e9 / cycles_per_inst), 4) if ach and clock and cycles_per_inst else None
        roofline = {"bound": "valu_issue", "kernel": "inlier-count pass: " + " + ".join(pass_names), "unit": "T wave-instructions/s",
                    "achieved": roofline_valu["achieved"], "peak": roofline_valu["peak"], "frac": roofline_valu["frac"],
                    "traffic": pass_traffic, "ms": round(k_avg_ms, 4), "share_of_call": round(k_avg_ms / ms_per_step, 3) if ms_per_step else None,
                    "issued_valu_wave_instructions": issued,
                    "effective_clock_GHz": eff_clock,
                    "frac_at_effective_clock": _frac(eff_clock, CYCLES_PER_VALU),
                    "peak_at_effective_clock": round(N_SIMD * eff_clock * 1e9 / CYCLES_PER_VALU / 1e12, 4) if eff_clock else None,
                    "frac_of_measured_mix_roof": ({"valu_alone": _frac(eff_clock, mix["simd_cycles_per_valu_instruction_valu_alone"]),
                                                   "beside_the_mfma": _frac(eff_clock, mix["simd_cycles_per_valu_instruction_with_mfma"]),
                                                   "simd_cycles_per_valu_instruction": {"valu_alone": mix["simd_cycles_per_valu_instruction_valu_alone"],
                                                                                        "beside_the_mfma": mix["simd_cycles_per_valu_instruction_with_mfma"]},
                                                   "what": "the pass's issued VALU instructions / time against 1024 SIMDs x the effective clock / the SIMD cycles ONE VALU "
                                                           "instruction of the count loop's own mix costs in a microbenchmark of that loop (count_pipe3.hip: alone, and "
                                                           "beside the bf16 MFMA it consumes) -- how much of the pass is its loop running at the loop's own ceiling"}
                                                  if mix and eff_clock else None),
                    "wave_wait_inst_frac_per_kernel": ({n: call_pmc["kernels"][n].get("wave_wait_inst_frac") for n in pass_names} if call_pmc and issued else None),
                    "busy_frac_counter": roofline_valu["busy_frac"],
                    "busy_frac_counter_note": "SQ_ACTIVE_INST_VALU x 4 / SIMD cycles: that counter ticks one quad-cycle per issued VALU instruction, so this is the "
                                              "issue rate at an assumed 4 cycles per instruction and the counter's own clock -- NOT independent evidence of a saturated pipe "
                                              "(VERDICT r5 weak #4); the independent figures are effective_clock_GHz (stamps) and frac_of_measured_mix_roof (microbenchmark)",
                    "mfma_busy_frac": round(mfma_cyc / N_SIMD / (gui / 8), 4) if mfma_cyc and gui else None,
                    "source": pmc_src, "model": roofline_valu["model"],
                    "ms_how": "HIP events at the stage boundaries inside full calls on the launch stream (pvv_problem.ev_marks), live; the "
                              "instruction counts, the effective clock and the microbenchmark figures are static (separate rocprofv3 --pmc passes / "
                              "instrumented builds of this tree, tracked under profiles/)",
                    "hbm_view": {"dense_equivalent_frac": dense_equivalent["frac"], "call_traffic_frac": dense_equivalent["traffic_frac"],
                                 "see": "roofline_dense_equivalent (the whole call against SURVEY 8d's dense-field bytes) and roofline_scan"}}

        extra = {"tn_mean": round(float(tn_cpu.float().mean()), 1) if tn_cpu.numel() else 0.0,
                 "known_answer_max_err_px": round(err, 3),
                 "rotating_batches": len(batches), "bytes_per_batch_per_gpu": int(B * H * W * (8 + K * 8)),
                 "prewarm": {"ms": args.prewarm_ms, "untimed_steps": prewarm_done[0],
                             "why": "clock ramp after the GPU-idle data generation (tools/clock_ramp.py); the timed region is exactly --steps steps"},
                 "images_per_gpu": B, "shard_sizes": shard_sizes, "backend": (backend if use_dist else None),
                 "oversubscribed_ranks_per_gpu": (-(-world // n_dev) if oversubscribed else None),
                 "collective_ranks": coll_ranks, "rccl_ranks": (coll_ranks if backend == "nccl" else None),
                 "per_rank_count_kernel_ms": per_rank_kernel_ms,
                 "kernels_inside_calls_ms": stage, "stream_read_probe": probe, "count_pass_staged": staged_path,
                 "exchange_impl": ("rccl: ncclAllGather issued directly on the launch stream (clean_pvnet_amd/rccl.py)" if comm is not None
                                   else ("torch.distributed.all_gather_into_tensor" + (": " + comm_note if comm_note else ""))) if use_dist else None,
                 "exchange": (("in-place all_gather of [%d,%d,2] f32 per step (the voting call writes this rank's rows of a "
                               "persistent buffer, clean_pvnet_amd.dist.GatherBuffer), " % (global_batch, K)) +
                              ("stream-ordered behind the step's voting kernels on the launch stream" if comm is not None else
                               ("enqueued asynchronously behind the step's voting kernels and waited for after the next step's "
                                "launches (it runs beside them); all K complete before the closing barrier + synchronize"
                                if overlapped else "completed inside the step (the launch stream waits for it)"))) if use_dist else None,
                 "exchange_calibration": calibration}
        if world > 1:
            extra["scaling_vs_n1_profile"] = predict_from_profile(args.config, global_batch, world, max(shard_sizes), value, weak)
        if other:
            extra.update({k: v for k, v in other.items() if not k.startswith("_")})
        if sustained:
            extra["sustained"] = sustained
            extra["sustained_images_per_s"] = sustained["images_per_s"]
        if noisy:
            extra["noisy_field"] = noisy
        if un_pnp:
            extra.update(un_pnp)
        extra["predicted_8gpu"] = predict_8gpu(args.config, value if world == 1 else None)
        if two_stream:
            extra.update(two_stream)
        if world == 1 and args.extras:
            extras_leg(extra, batches[0], out, ext, synth, ransac_voting_layer_v3, estimate_voting_distribution_with_mean,
                       B, H, W, K, hn, thresh, dev)

        cpu_baseline = None
        if world == 1 and not args.no_cpu_baseline:
            d0 = batches[0]
            tn0 = ext.ransac_voting_v3(d0["mask"], d0["vertex"], hn, thresh, 5, 30000, None, None, 1, ext.SINGULAR_REFERENCE)[2].cpu()
            cpu_baseline = cpu_leg(d0["mask"], d0["vertex"], tn0, hn, K, thresh, args.cpu_sample, synth, ext)

        result = {
            "metric": metric_name(H, W, K, hn, world, args.batch, weak), "value": round(value, 1), "unit": "images/s",
            "value_strong": round(value, 1) if (not weak or world == 1) else (other or {}).get("_value"),
            "value_weak": round(value, 1) if (weak or world == 1) else (other or {}).get("_value"),
            "value_at_rho_0.90": noisy["images_per_s"] if noisy else None,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE config 3 (%s): %dx%d, K=%d, %d hypotheses, ~%.0f%% foreground, int64 mask, contiguous "
                                   "[B,H,W,K,2] f32 vertex; global batch %d in contiguous shards of %d images per GPU; "
                                   "ransac_voting_layer_v3 (+ one RCCL all_gather of [B,K,2] per step for N>1, see extra.exchange); steps "
                                   "cycle over %d distinct device-resident batches"
                                   % (args.config, H, W, K, hn, 100 * (cfg["fg"] if not isinstance(cfg["fg"], tuple) else cfg["fg"][1]),
                                      global_batch, B, len(batches)),
                       "batch_per_gpu": B, "global_batch": global_batch, "H": H, "W": W, "K": K, "hn": hn,
                       "inlier_thresh": thresh,
                       "arithmetic": "IEEE binary32, one rounding per source-level operation, NO fused multiply-add: inlier counts are "
                                     "bit-exact against the reference's kernel compiled without contraction (oracle/_ref, tests/test_ref_pin.py); "
                                     "nvcc's default contraction moves <= 1e-5 of the reference's own inlier decisions (libref_ransac_voting_fma.so) -- "
                                     "unverifiable without nvcc; normal equations in binary64 (DESIGN.md 3)",
                       "parallelism": ("x%d GPUs, %d images per GPU (weak scaling: global batch %d)" % (world, B, global_batch)) if weak
                                      else ("batch-sharded x%d (strong scaling: %d images in total)" % (world, global_batch))},
            "step_ms": {"median": round(pct(per_step, 0.5), 4), "p10": round(pct(per_step, 0.1), 4),
                        "p90": round(pct(per_step, 0.9), 4), "wall": round(ms_per_step, 4),
                        "how": "torch.cuda.Event pairs around every step on the launch stream (rank 0); wall = perf_counter over the timed region / steps, max over ranks"},
            # N > 1: what a SCALE record needs to be judged without reading `extra` (round 6, VERDICT r5 #2c)
            "multi_gpu": ({"ranks": world, "backend": backend, "rccl_ranks": (coll_ranks if backend == "nccl" else None),
                           "collective_ranks": coll_ranks, "shard_sizes": shard_sizes,
                           "exchange_impl": ("rccl_direct" if comm is not None else "torch_distributed"),
                           "exchange_fallback_reason": comm_note,
                           "oversubscribed_ranks_per_gpu": (-(-world // n_dev) if oversubscribed else None),
                           "per_rank_count_pass_ms": per_rank_kernel_ms} if use_dist else None),
            "roofline": roofline, "roofline_dense_equivalent": dense_equivalent, "roofline_contract_count_pass": roofline_contract, "roofline_scan": roofline_scan,
            "roofline_compact": roofline_compact, "roofline_valu": roofline_valu, "cpu_baseline": cpu_baseline, "extra": extra,
        }
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.barrier()
        torch.cuda.synchronize()
        if comm is not None:
            comm.destroy()
        dist.destroy_process_group()
    return result


def metric_name(H, W, K, hn, world, batch, weak):
    """BASELINE.json's metric; for N > 1 the scaling mode is part of the NAME, so that a weak-scaling value (N x 64 images)
    cannot be mistaken for BASELINE config 3 read literally (64 images sharded over the GPUs) -- ADVICE r3."""
    base = "images/sec RANSAC-vote (%dx%d, K=%d, %d hyp)" % (H, W, K, hn)
    if world == 1:
        return base
    return base + (", weak scaling: %d images per GPU" % batch if weak else ", strong scaling: batch=%d sharded over %d GPUs" % (batch, world))


def predict_8gpu(config, n1_value):
    """What the tracked one-GPU profile predicts for the driver's first 8-GPU run (none has been measured: every box this
    was developed on has one GPU).  Weak: 8 x the one-GPU rate minus one exchange per step; strong (BASELINE config 3 read
    literally): bounded by the time ONE shard of 8 images takes on one GPU -- a latency chain -- plus the exchange."""
    name, rows = latest_configs_profile()
    # the exchange: what the one-rank RCCL group adds to a step on the test box (tracked lines of the same round), else 15 us
    exch_ms, exch_src = 0.015, "assumed (12-15 us measured on the one-rank RCCL group in rounds 2-4)"
    try:
        rnd = name.split("_")[0]
        a, b = load_profile(rnd + "_bench_torchrun_1rank.json"), load_profile(rnd + "_bench_default.json")
        if a and b:
            exch_ms = max(0.0, a["ms_per_step"] - b["ms_per_step"])
            exch_src = "profiles/%s_bench_torchrun_1rank.json - profiles/%s_bench_default.json (ms_per_step; a ONE-rank group: the in-place all_gather is no device work there -- eight ranks pay RCCL's latency, unmeasured)" % (rnd, rnd)
    except Exception:
        pass
    try:
        full = next(v for k, v in rows.items() if k == "%s_B64" % config or k.startswith("%s_B64_" % config))
        part = next(v for k, v in rows.items() if k.startswith("%s_B8" % config))
        ms64, ms8 = full["event_ms_per_call_median"], part["event_ms_per_call_median"]
        return {"strong_images_per_s": round(64 / ((ms8 + exch_ms) * 1e-3), 1), "strong_speedup_vs_1gpu": round(ms64 / (ms8 + exch_ms), 2),
                "weak_images_per_s": round(8 * 64 / ((ms64 + exch_ms) * 1e-3), 1), "weak_efficiency": round(ms64 / (ms64 + exch_ms), 3),
                "inputs": {"shard_of_8": next(k for k in rows if k.startswith("%s_B8" % config)),
                           "batch_of_64": next(k for k in rows if k == "%s_B64" % config or k.startswith("%s_B64_" % config)),
                           "field": "event_ms_per_call_median", "file": "profiles/%s" % name},
                "shard_of_8_ms_per_call": ms8, "batch_of_64_ms_per_call": ms64, "exchange_ms": round(exch_ms, 4), "exchange_ms_source": exch_src,
                "source": "profiles/%s (one MI355X, warm caches); UNMEASURED on 8 GPUs" % name}
    except Exception as e:                                                          # never lose the bench line to this
        return {"note": "prediction unavailable: %s" % (e,)}


def decode_leg(batches, ext, ransac_voting_layer_v3, B, H, W, K, hn, thresh, dev, run, n, headline):
    """The layout the REAL caller passes (resnet18.py:65-71,93-94): seg logits and the vertex field are channel slices of ONE
    [B, 2+2K, H, W] network output per batch -- the vertex a strided planar view, the mask an argmax away.  Timed like the
    headline (rotating batches, pre-warm, barriered region): (a) pvv_decode_keypoint_v3 -- the class argmax inside the mask
    scan (k_tile_scan_seg2), the int64 `mask` output written from the tile lists on the side stream beside the count
    pass; (b) what the reference's code does with the same tensors: torch.argmax, then ransac_voting_layer_v3."""
    nets = []
    for d in batches:
        x = torch.empty(B, 2 + 2 * K, H, W, device=dev)
        x[:, :2] = torch.randn(B, 2, H, W, device=dev) * 0.1
        x[:, 0] += 3.0 * (d["mask"] == 0)
        x[:, 1] += 3.0 * (d["mask"] != 0)
        x[:, 2:] = d["vertex"].permute(0, 3, 4, 1, 2).reshape(B, 2 * K, H, W)
        nets.append((x[:, :2], x[:, 2:].permute(0, 2, 3, 1).view(B, H, W, K, 2)))
    keep = [None]

    def fused(i):
        seg, vtx = nets[i % len(nets)]
        o = ext.decode_keypoint_v3(seg, vtx, hn, thresh, 5, 30000, None, None, 7 + i, ext.SINGULAR_REFERENCE)
        keep[0] = o[1]
        return o[0]

    def unfused(i):
        seg, vtx = nets[i % len(nets)]
        return ransac_voting_layer_v3(torch.argmax(seg, 1), vtx, hn, inlier_thresh=thresh)
    el, _p, kp = run(fused, 5, n)
    seg0 = nets[(5 + n - 1) % len(nets)][0]
    mask_ok = bool(torch.equal(keep[0], torch.argmax(seg0, 1)))
    err = float((kp - batches[(5 + n - 1) % len(batches)]["kpt_2d"]).abs().max())
    res = {"decode_fused_images_per_s": round(B * n / el, 1), "decode_fused_ms_per_step": round(1e3 * el / n, 4),
           "decode_fused_vs_headline": round(B * n / el / headline, 4), "decode_fused_mask_equals_torch_argmax": mask_ok,
           "decode_fused_known_answer_max_err_px": round(err, 3)}
    el, _p, _o = run(unfused, 3, max(10, n // 2))
    res["decode_unfused_argmax_plus_v3_images_per_s"] = round(B * max(10, n // 2) / el, 1)
    st = ext.stage_ms_in_pipeline([], [v for _s, v in nets], hn, thresh, 5, 30000, 12, 24, ext.COUNT_AUTO, False, False, [s_ for s_, _v in nets])[6:]
    med = lambda j: sorted(r[j] for r in st)[len(st) // 2]                         # noqa: E731
    seg_bytes, mask_bytes = B * 2 * H * W * 4, B * H * W * 8
    res["decode_fused_scan"] = {"kernel": "k_tile_scan_seg2<false> (two f32 seg planes, 16-byte loads; the int64 mask is written by "
                                          "k_mask_from_lists on the side stream)", "bound": "hbm", "bytes_read": seg_bytes,
                                "ms_median": round(med(0), 4), "achieved": round(seg_bytes / (med(0) * 1e-3) / 1e9, 1), "unit": "GB/s",
                                "peak": HBM_PEAK_GBS, "frac": round(seg_bytes / (med(0) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                "mask_bytes_written_beside_the_count_pass": mask_bytes,
                                "stages_ms_median": {"scan": round(med(0), 4), "compact_hyp": round(med(1), 4), "count_pass": round(med(2), 4),
                                                     "select_refit": round(med(3), 4), "finalize": round(med(4), 4)}}
    del nets
    return res


def un_pnp_leg(data, out, ext, ransac_voting_layer_v3, estimate_voting_distribution_with_mean, B, H, W, K, hn, thresh, dev, run, n):
    """The path cfg.test.un_pnp runs (resnet18.py:70-72), timed like the headline (pre-warm + barriered region): (a) the
    reference's two calls on the argmax mask (round 5: the library counts the estimate of a batch this large IN STAGES), (b)
    decode_keypoint(un_pnp=True) on seg logits + planar vertex -- ONE fused call (pvv_decode_keypoint_un_pnp) that, where the
    estimate stages, counts its rows as two passes over the one compaction -- and (c) the same call with its single full count
    pass forced; and the estimate's count pass inside calls, in full (k_count_bf16<0>, 4096 hypotheses: its VALU block) and as
    AUTO runs it."""
    from clean_pvnet_amd.decode import decode_keypoint
    mask, vertex = data["mask"], data["vertex"]

    def two_calls(_i):
        mean = ransac_voting_layer_v3(mask, vertex, hn, inlier_thresh=thresh)
        return estimate_voting_distribution_with_mean(mask, vertex, mean)[1]
    el, _p, _o = run(two_calls, 2, n)
    res = {"v3_plus_estimate_images_per_s": round(B * n / el, 1), "v3_plus_estimate_ms_per_step": round(1e3 * el / n, 4)}
    x = torch.empty(B, 2 + 2 * K, H, W, device=dev)
    x[:, 0] = 3.0 * (mask == 0)
    x[:, 1] = 3.0 * (mask != 0)
    x[:, 2:] = vertex.permute(0, 3, 4, 1, 2).reshape(B, 2 * K, H, W)
    seg, ver = x[:, :2], x[:, 2:]

    def auto_path(_i):
        return decode_keypoint({"seg": seg, "vertex": ver}, un_pnp=True)["var"]
    el, _p, _o = run(auto_path, 2, n)
    two = bool(ext.estimate_counts_in_stages(B, H, W, K, 4096, 30000))
    res.update({"un_pnp_decode_keypoint_images_per_s": round(B * n / el, 1), "un_pnp_decode_keypoint_ms_per_step": round(1e3 * el / n, 4),
                "un_pnp_decode_keypoint_path": "one fused call (pvv_decode_keypoint_un_pnp): one scan + compaction + hypothesis launch; " +
                                               ("the rows counted as two passes -- v3's 512 columns, then the estimate's 4096 in stages" if two
                                                else "one full count pass over all 4608 columns")})
    vtx = ver.permute(0, 2, 3, 1).view(B, H, W, K, 2)

    def one_pass(_i):
        return ext.decode_keypoint_un_pnp(seg, vtx, hn, 4096, thresh, 5, 30000, None, None, None, 7 + _i, ext.SINGULAR_REFERENCE, 0, ext.COUNT_FULL)[2]
    el, _p, _o = run(one_pass, 2, n)
    res.update({"un_pnp_fused_one_pass_images_per_s": round(B * n / el, 1), "un_pnp_fused_one_pass_ms_per_step": round(1e3 * el / n, 4)})
    del x
    # (about the keypoints v3 finds: the staged pass walks the chunks nearest to them first)
    kp = ransac_voting_layer_v3(mask, vertex, hn, inlier_thresh=thresh).contiguous()
    sta = ext.stage_ms_in_pipeline([mask], [vertex], 4096, thresh, 5, 30000, 3, 10, ext.COUNT_AUTO, False, True, [], [kp])[4:]
    res["estimate_4096_count_pass_as_auto_runs_it_ms"] = round(sorted(r[2] for r in sta)[len(sta) // 2], 4)
    res["estimate_4096_counted_in_stages_by_auto"] = two
    st = ext.stage_ms_in_pipeline([mask], [vertex], 4096, thresh, 5, 30000, 3, 10, ext.COUNT_FULL, False, True)[4:]
    med = lambda j: sorted(r[j] for r in st)[len(st) // 2]                         # noqa: E731
    est_ms = med(2)
    tn_e = ext.ransac_voting_v3(mask, vertex, hn, thresh, 5, 30000, None, None, 1, ext.SINGULAR_REFERENCE)[2].sum().item()
    evals_e = int(tn_e) * K * 4096
    clock_ghz = torch.cuda.get_device_properties(dev).clock_rate / 1e6 if hasattr(torch.cuda.get_device_properties(dev), "clock_rate") else 2.4
    # the full pass executes every evaluation: issued VALU ~ (21 in the loop + prologue share) per 512-evaluation tile; the
    # counter-backed figure for this kernel at 4096 hypotheses is in profiles/ when the estimate was profiled
    tiles = evals_e / EVALS_PER_TILE
    peak_issue = N_SIMD * clock_ghz * 1e9 / CYCLES_PER_VALU
    est_pmc = (load_profile("call_pmc.json") or {}).get("estimate_4096", {})
    issued = est_pmc.get("SQ_INSTS_VALU")
    res["estimate_4096_count_pass"] = {
        "kernel": "k_count_bf16<0> (4096 hypotheses, full pass)", "ms_inside_calls_median": round(est_ms, 4), "evaluations": evals_e,
        "T_evaluations_per_s": round(evals_e / (est_ms * 1e-3) / 1e12, 3),
        "loop_only_valu_frac": round(tiles * VALU_PER_TILE / (est_ms * 1e-3) / peak_issue, 4),
        "issued_valu_wave_instructions": issued, "issued_valu_frac": round(issued / (est_ms * 1e-3) / peak_issue, 4) if issued else None,
        "effective_clock_GHz": est_pmc.get("effective_clock_GHz"),
        "issued_valu_frac_at_effective_clock": (round(issued / (est_ms * 1e-3) / (N_SIMD * est_pmc["effective_clock_GHz"] * 1e9 / CYCLES_PER_VALU), 4)
                                                if issued and est_pmc.get("effective_clock_GHz") else None),
        "simd_cycles_per_valu_instruction_at_effective_clock": (round(est_ms * 1e-3 * est_pmc["effective_clock_GHz"] * 1e9 * N_SIMD / issued, 3)
                                                                if issued and est_pmc.get("effective_clock_GHz") else None),
        "loop_microbench_simd_cycles_per_valu_instruction": ((load_profile("call_pmc.json") or {}).get("count_loop_microbench") or {}).get("simd_cycles_per_valu_instruction_with_mfma"),
        "busy_frac_from_counters": est_pmc.get("valu_busy"),
        "busy_frac_note": "SQ_ACTIVE_INST_VALU x 4 / SIMD cycles = issue rate at an assumed 4 cycles per instruction (a counter identity, not a saturation proof)",
        "model": "loop_only: %d VALU per 512-evaluation matrix-core tile x tiles / time against 1024 SIMDs x %.2f GHz / %.0f cycles "
                 "(full rate; the loop's v_alignbit / v_min3 / v_cmp are half rate); issued: SQ_INSTS_VALU (static, "
                 "profiles/call_pmc.json) / time against the same ceiling" % (VALU_PER_TILE, clock_ghz, CYCLES_PER_VALU),
        "scan_ms": round(med(0), 4), "compact_hyp_ms": round(med(1), 4), "covariance_ms": round(med(3), 4)}
    return res


def predict_from_profile(config, global_batch, world, shard, value, weak):
    """What the tracked single-GPU profile (the newest profiles/rNN_configs.json: ms per call of this config at every shard size)
    predicts for this run, so that a surprising scaling curve can be read against it.  Weak scaling (the same batch on
    every GPU) adds only the exchange to a step; strong scaling of one batch is bounded by the time ONE SHARD takes on one
    GPU, and a shard of 8 images is latency-bound."""
    try:
        src_name, rows = latest_configs_profile()
        def row(b):                        # "cfg3_B8_shard_of_8gpu", "cfg2_B1" (= cfg3 at B = 1), ...
            for c in (config, "cfg2" if config == "cfg3" else config):
                for k, v in rows.items():
                    if k == "%s_B%d" % (c, b) or k.startswith("%s_B%d_" % (c, b)):
                        return v
            return None
        n1_batch = shard if weak else global_batch            # what ONE GPU runs at N = 1 under this scaling mode
        full, part = row(n1_batch), row(shard)
        if not full or not part:
            return {"note": "no profile row for %s at B=%d / B=%d" % (config, n1_batch, shard)}
        n1 = n1_batch / (full["event_ms_per_call_median"] * 1e-3)
        pred = global_batch / (part["event_ms_per_call_median"] * 1e-3)           # without the exchange (10-30 us)
        return {"n1_profile_images_per_s": round(n1, 1), "shard_ms_per_call_profile": part["event_ms_per_call_median"],
                "predicted_images_per_s_without_exchange": round(pred, 1),
                "predicted_speedup": round(pred / n1, 2), "predicted_efficiency": round(pred / n1 / world, 3),
                "efficiency_vs_n1_profile": round(value / n1 / world, 3),
                "measured_over_predicted": round(value / pred, 3),
                "source": "profiles/%s (one MI355X, event_ms_per_call_median of the shard size)" % src_name}
    except Exception as e:                                                          # never lose the bench line to this
        return {"note": "prediction unavailable: %s" % (e,)}


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
    # the same 64-image call with the mask as uint8 (1 B/pixel -- SURVEY 8d's algorithmic figure -- instead of the int64
    # torch.argmax emits, which is what the headline feeds): what a caller who can choose the mask's dtype gets
    m8 = mask.to(torch.uint8)
    for _ in range(10):
        ransac_voting_layer_v3(m8, vertex, hn, inlier_thresh=thresh)
    torch.cuda.synchronize()
    t8 = time.perf_counter()
    for _ in range(50):
        ransac_voting_layer_v3(m8, vertex, hn, inlier_thresh=thresh)
    torch.cuda.synchronize()
    extra["uint8_mask_images_per_s_one_batch_replayed"] = round(B * 50 / (time.perf_counter() - t8), 1)
    del m8
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
    # the estimate's own count kernel (4096 hypotheses, always the full pass: the estimate weighs every hypothesis) as it
    # runs inside the calls, with its VALU roofline (the call the network makes with test.un_pnp)
    st = ext.stage_ms_in_pipeline([mask], [vertex], 4096, thresh, 5, 30000, 3, 12, ext.COUNT_FULL, False, True)[4:]
    est_ms = sorted(r[2] for r in st)[len(st) // 2]
    tn_e = ext.ransac_voting_v3(mask, vertex, hn, thresh, 5, 30000, None, None, 1, ext.SINGULAR_REFERENCE)[2].sum().item()
    evals_e = int(tn_e) * K * 4096
    clock_ghz = torch.cuda.get_device_properties(dev).clock_rate / 1e6 if hasattr(torch.cuda.get_device_properties(dev), "clock_rate") else 2.4
    peak_e = N_SIMD * clock_ghz * 1e9 * EVALS_PER_TILE / (VALU_PER_TILE * CYCLES_PER_VALU)
    extra["estimate_4096_count_kernel"] = {"ms_inside_calls_median": round(est_ms, 4), "evaluations": evals_e,
                                           "T_evaluations_per_s": round(evals_e / (est_ms * 1e-3) / 1e12, 3),
                                           "roofline_valu_frac": round(evals_e / (est_ms * 1e-3) / peak_e, 4),
                                           "scan_ms": round(sorted(r[0] for r in st)[len(st) // 2], 4),
                                           "compact_hyp_ms": round(sorted(r[1] for r in st)[len(st) // 2], 4),
                                           "covariance_ms": round(sorted(r[3] for r in st)[len(st) // 2], 4)}
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


def cpu_leg(mask, vertex, tn, hn, K, thresh, n_sample, synth, ext, rep_s=1.0, reps=3):
    """SURVEY 8(d)'s CPU baseline: the oracle (a port: the reference has no CPU path) on images of the timed batch -- the whole
    per-image layer in C since round 5 (compaction: orc_compact_v3; hypotheses, counting over OpenMP threads, winner, refit:
    orc_v3_image; no numpy in the timed loop, VERDICT r4 #4c).  One thread, and the host cores: for each candidate thread count
    `reps` repetitions of >= `rep_s` seconds, each cycling over the sampled images; the count with the best MEDIAN is the
    baseline, its median is `value`, min / max its spread -- the repetitions that choose the thread count ARE the measurement
    (round 4 probed on one cache-warm image and then measured something else: 299.7 vs 94.8 images/s at the same setting).
    The SAME images with the SAME injected index pairs go once through the GPU path and are cross-checked in this run: winner
    inlier counts equal, keypoint means within the contract (1e-4 px + 2e-6 relative)."""
    import numpy as np
    from oracle import vote_oracle
    vote_oracle.lib()
    n = min(n_sample, mask.shape[0])
    m = mask[:n].cpu().numpy()
    v = vertex[:n].cpu().numpy()
    idxs_t = synth.make_idxs([int(t) for t in tn[:n]], hn, K)
    idxs = idxs_t.numpy()
    g_out, g_win, _g_tn, _ws = ext.ransac_voting_v3(mask[:n], vertex[:n], hn, thresh, 5, 30000, idxs_t.to(mask.device), None, 0,
                                                   ext.SINGULAR_REFERENCE)
    g_out, g_win = g_out.cpu().numpy(), g_win.cpu().numpy()

    def one(i, det=None):
        return vote_oracle.ransac_voting_layer_v3(m[i:i + 1], v[i:i + 1], hn, thresh, idxs=idxs[i:i + 1], details=det, compact_in_c=True)

    def rate(nthr):
        """-> images/s of `reps` repetitions of >= rep_s seconds at nthr OpenMP threads, HYPOTHESIS-parallel (one image at a time)"""
        vote_oracle.set_num_threads(nthr)
        one(0)
        out = []
        for _ in range(reps):
            t0, done = time.perf_counter(), 0
            while done < 1 or time.perf_counter() - t0 < rep_s:
                one(done % n)
                done += 1
            out.append(done / (time.perf_counter() - t0))
        return sorted(out)

    def rate_images(nthr):
        """-> images/s at nthr OpenMP threads, IMAGE-parallel (orc_v3_batch: one thread per image, serial inside; round 6, VERDICT
        r5 #5): `reps` repetitions, each one C call over enough images for >= rep_s seconds (sized from a first call of one image per
        thread)"""
        vote_oracle.set_num_threads(nthr)
        t0 = time.perf_counter()
        vote_oracle.v3_batch(m, v, hn, thresh, idxs, total=nthr)
        per_round = max(1e-3, time.perf_counter() - t0)                  # seconds for one image per thread
        total = max(nthr, int(nthr * min(64.0, max(1.0, rep_s / per_round))))
        out = []
        for _ in range(reps):
            t0 = time.perf_counter()
            vote_oracle.v3_batch(m, v, hn, thresh, idxs, total=total)
            out.append(total / (time.perf_counter() - t0))
        return sorted(out)
    single = rate(1)
    # host CPUs visible != CPUs usable (cgroup quotas): the thread count is chosen by measurement, and the quota is printed
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            quota = {"file": path, "content": open(path).read().strip()}
            break
        except OSError:
            pass
    cand = {avail, max(1, avail // 2), max(1, avail // 4), min(avail, 8)}
    quota_cpus = None
    try:                                                                 # cgroup v2 "quota period": the CPUs' worth of time this job may use
        q, per = quota["content"].split()[:2]
        if q != "max":
            quota_cpus = max(1, -(-int(q) // int(per)))
            cand |= {min(avail, quota_cpus), min(avail, 2 * quota_cpus)}
    except Exception:                                                    # noqa: BLE001  (no quota file, or cgroup v1's single number)
        pass
    cand = sorted(cand, reverse=True)
    table, best = [], None
    for form, fn in (("hypothesis_parallel", rate), ("image_parallel", rate_images)):
        for nthr in cand:
            if form == "image_parallel" and nthr * (2 * 4 * (1 + K) * int(m.shape[1]) * int(m.shape[2])) > 48 * 2 ** 30:
                continue                                                 # (per-thread scratch: coords + direct of a whole image)
            r = fn(nthr)
            table.append({"form": form, "threads": nthr, "images_per_s_median": round(r[len(r) // 2], 2), "min": round(r[0], 2), "max": round(r[-1], 2)})
            if best is None or r[len(r) // 2] > best[1][len(best[1]) // 2]:
                best = (nthr, r, form)
    vote_oracle.set_num_threads(best[0])
    outs, wins = {}, {}
    for i in range(n):                                               # the cross-check: every sampled image once, untimed
        det = []
        outs[i] = one(i, det)
        wins[i] = det[0]["win_counts"] if not det[0].get("skipped") else np.zeros(K, np.int32)
    diff = max(float(np.abs(outs[i] - g_out[i:i + 1]).max()) for i in outs)
    within = all(bool((np.abs(outs[i] - g_out[i:i + 1]) <= 1e-4 + 2e-6 * np.abs(outs[i])).all()) for i in outs)
    win_eq = all(bool(np.array_equal(wins[i], g_win[i])) for i in wins)
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    r = best[1]
    return {"value": round(r[len(r) // 2], 3), "unit": "images/s", "cores": best[0], "kind": "port",
            "spread": {"min": round(r[0], 3), "max": round(r[-1], 3), "repetitions": reps, "seconds_each": rep_s},
            "form": best[2],
            "sample": "ransac_voting_layer_v3 of the oracle cycling over %d of the timed 480x640 images, everything in C (orc_compact_v3 + "
                      "orc_v3_image), timed two ways -- hypothesis-parallel (OpenMP inside one image, image after image) and image-parallel "
                      "(orc_v3_batch: one thread per image, serial inside) -- at %d thread counts each, %d repetitions of >= %.1f s; `value` = "
                      "the best median, `cores` = its thread count, `form` = which of the two" % (n, len(cand), reps, rep_s),
            "single_thread": {"value": round(single[len(single) // 2], 3), "unit": "images/s", "cores": 1,
                              "spread": {"min": round(single[0], 3), "max": round(single[-1], 3)}},
            "cpu_model": model, "host_cpus": os.cpu_count(), "usable_cpus": avail, "sched_getaffinity_count": avail, "cgroup_cpu_quota": quota, "cgroup_quota_cpus": quota_cpus,
            "thread_probe": {"table": table, "picked": best[0],
                             "how": "%d repetitions of >= %.1f s per candidate thread count, the same loop as the figure itself (visible CPUs != "
                                    "usable CPUs under cgroup quotas: the figure varies from box to box -- quote it with this table)" % (reps, rep_s)},
            "same_idxs_gpu_check": {"images": n, "means_max_abs_diff": float("%.3g" % diff), "means_within_1e-4_contract": within,
                                    "win_counts_equal": win_eq,
                                    "how": "the sampled images with the same injected index pairs through ext.ransac_voting_v3 in this run"}}


if __name__ == "__main__":
    main()
