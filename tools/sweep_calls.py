#!/usr/bin/env python
"""Whole ransac_voting_layer_v3 calls (AUTO, after two warm-up calls that fill the stage hint) over a grid of sizes the benchmark
does not run -- to find mis-sized grids / work items: a dip in evaluations per second between neighbouring sizes is one.

    python tools/sweep_calls.py [--cases "480x640:0.02:512:1,2,4,8;256x256:0.35:512:4,8,16"] [--mode 0|2|3]
Case = HxW:foreground fraction:hypotheses:batch sizes.  One line per (case, B): ms per call, us per image, Tera-evaluations/s.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import lib  # noqa: E402

lib._register_clean_pvnet_amd()
from clean_pvnet_amd import ransac_voting as ext  # noqa: E402
from clean_pvnet_amd import synth  # noqa: E402

DEFAULT = ("480x640:0.02:512:1,2,3,4,6,8,10,12,14,16,20,24,28,32,40,48,56,64,80,96,128;"
           "480x640:0.02:1024:4,8,12,16,24,32;480x640:0.02:2048:4,8,12,16,24,32;"
           "480x640:0.05:512:2,4,8,16,32;480x640:0.0985:512:2,4,8,16;"
           "256x256:0.35:512:2,4,6,8,12,16,24,32;128x128:0.35:512:8,16,32,64,128;"
           "720x1280:0.02:512:1,4,16,32;1080x1920:0.02:512:1,4,16")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default=DEFAULT)
    ap.add_argument("--mode", type=int, default=0, help="pvv count_kernel: 0 AUTO, 2 FULL, 3 STAGED")
    ap.add_argument("--K", type=int, default=9)
    ap.add_argument("--calls", type=int, default=40)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    for case in a.cases.split(";"):
        size, fg, hn, bs = case.split(":")
        H, W = (int(x) for x in size.split("x"))
        fg, hn = float(fg), int(hn)
        for B in (int(x) for x in bs.split(",")):
            ds = [synth.make_batch(B=B, H=H, W=W, K=a.K, fg=fg, sigma=0.05, seed=7 + 1000 * r, device=dev) for r in range(2)]

            def call(i):
                d = ds[i % 2]
                return ext.ransac_voting_v3(d["mask"], d["vertex"], hn, 0.99, 5, 30000, None, None, 7, ext.SINGULAR_REFERENCE, count_kernel=a.mode)
            for i in range(6):
                o = call(i)
            torch.cuda.synchronize()
            tn = float(o[2].double().sum())
            ts = []
            for g in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(a.calls // 5):
                    call(i)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / (a.calls // 5))
            ms = sorted(ts)[2]
            ev = a.K * hn * tn
            print("%-10s fg %.4f hn %4d B %3d  tn/img %6.0f  %.4f ms/call  %6.2f us/img  %5.2f Tevals/s" % (size, fg, hn, B, tn / B, ms, 1e3 * ms / B, ev / ms / 1e9), flush=True)
            del ds
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
