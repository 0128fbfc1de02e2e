#!/usr/bin/env python
"""Host-side anatomy of the exchange in a one-rank RCCL group (what the test box offers): (1) host time per
all_gather_into_tensor, sync / async + wait / async; (2) the bench's overlapped step -- vote, wait for the previous
exchange, enqueue this one -- with the host time of each part, to see whether any of them blocks on the GPU.

    python tools/coll_host.py            (sets up its own 1-rank group on 127.0.0.1)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
x = torch.zeros(64, 9, 2, device=dev)
out = torch.empty(64, 9, 2, device=dev)
for name, fn in (("sync", lambda: dist.all_gather_into_tensor(out, x)),
                 ("async+wait", lambda: dist.all_gather_into_tensor(out, x, async_op=True).wait()),
                 ("async", lambda: dist.all_gather_into_tensor(out, x, async_op=True))):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(1000):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(name, "host us/call %.1f" % ((t1 - t0) * 1e3), " incl. drain %.1f" % ((t2 - t0) * 1e3))

import lib  # noqa: E402

lib._register_clean_pvnet_amd()
from clean_pvnet_amd import dist as pdist, synth  # noqa: E402
from clean_pvnet_amd.ransac_voting_gpu import ransac_voting_layer_v3  # noqa: E402

cfg = dict(synth.CONFIGS["cfg3"])
gen = {k: v for k, v in cfg.items() if k not in ("B", "hn")}
batches = [synth.make_batch(B=64, **gen, first_index=1000 * r, device=dev) for r in range(3)]


def vote(i):
    d = batches[i % 3]
    return ransac_voting_layer_v3(d["mask"], d["vertex"], 512, inlier_thresh=0.99)


def loop(kind, n=300):
    pending = []
    acc = [0.0, 0.0, 0.0]
    for i in range(60):
        vote(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        a = time.perf_counter()
        local = vote(i)
        b = time.perf_counter()
        if kind == "overlapped":
            while pending:
                pending.pop()[1].wait()
        c = time.perf_counter()
        if kind == "overlapped":
            pending.append(pdist.gather_results(local, 64, async_op=True))
        elif kind == "overlapped-keep":           # the Work objects and outputs are kept alive (no destructor in the loop)
            pending.append(pdist.gather_results(local, 64, async_op=True))
        elif kind == "in-step":
            pdist.gather_results(local, 64)
        d = time.perf_counter()
        acc[0] += b - a; acc[1] += c - b; acc[2] += d - c
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for w in pending:
        w[1].wait()
    print("%-16s host us/step: vote %.1f  wait %.1f  gather %.1f | loop %.1f us/step, incl. drain %.1f" %
          (kind, acc[0] / n * 1e6, acc[1] / n * 1e6, acc[2] / n * 1e6, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))


for kind in ("none", "in-step", "overlapped", "overlapped-keep", "none"):
    loop(kind)
dist.destroy_process_group()
