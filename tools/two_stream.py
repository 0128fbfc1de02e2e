#!/usr/bin/env python
"""Does overlapping consecutive calls help?  cfg3 (480x640, K=9, 512 hyp), three rotating device-resident batches:

  one    : B=64 calls back to back on one stream (what bench.py times)
  two    : the same calls alternating over two streams (call n+1's scan/compaction may run under call n's count kernel)
  halves : every batch as two B=32 calls, one per stream

    gpurun -- 'python tools/two_stream.py > gpurun_out/two_stream.json'
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    import lib
    lib._register_clean_pvnet_amd()
    from clean_pvnet_amd import synth
    from lib.csrc.ransac_voting.ransac_voting_gpu import ransac_voting_layer_v3
    dev = torch.device("cuda:0")
    cfg = dict(synth.CONFIGS["cfg3"])
    gen = {k: v for k, v in cfg.items() if k not in ("B", "hn")}
    batches = [synth.make_batch(B=64, **gen, seed=100 + i, device=dev) for i in range(3)]
    halves = [[(d["mask"][:32].contiguous(), d["vertex"][:32].contiguous()), (d["mask"][32:].contiguous(), d["vertex"][32:].contiguous())]
              for d in batches]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    steps = 300

    def vote(m, v):
        return ransac_voting_layer_v3(m, v, 512, inlier_thresh=0.99)

    def run_one():
        for i in range(steps):
            d = batches[i % 3]
            vote(d["mask"], d["vertex"])

    def run_two():
        for i in range(steps):
            d = batches[i % 3]
            with torch.cuda.stream(streams[i & 1]):
                vote(d["mask"], d["vertex"])

    def run_halves():
        for i in range(steps):
            for s in range(2):
                with torch.cuda.stream(streams[s]):
                    vote(*halves[i % 3][s])

    main_s = torch.cuda.current_stream()

    def run_joined(parts):
        def fn():
            for i in range(steps):
                d = batches[i % 3]
                n = 64 // parts
                fork = torch.cuda.Event()
                fork.record(main_s)
                for s in range(parts):
                    st = streams[s % 2] if parts > 1 else main_s
                    st.wait_event(fork)
                    with torch.cuda.stream(st):
                        vote(d["mask"][s * n:(s + 1) * n], d["vertex"][s * n:(s + 1) * n])
                for st in streams:
                    e = torch.cuda.Event()
                    e.record(st)
                    main_s.wait_event(e)
        return fn

    res = {}
    for name, fn in (("one", run_one), ("two", run_two), ("halves", run_halves), ("joined2", run_joined(2)), ("joined4", run_joined(4)), ("one_again", run_one)):
        fn()                                        # warm (clocks, allocator pools of both streams)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[name] = {"ms_per_batch": round(dt / steps * 1e3, 4), "images_per_s": round(64 * steps / dt, 1)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
