#!/usr/bin/env python
"""Back-to-back ``ransac_voting_layer_v3`` calls of one tools/config_bench.py row, to be run under
``rocprofv3 --kernel-trace`` (then ``tools/trace_gaps.py <dir>``): which kernels a call launches, how long each runs
and how large the gaps between them are.   usage: trace_calls.py <row name> [calls]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import config_bench  # noqa: E402


def main():
    rowname = sys.argv[1]
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    import lib
    lib._register_clean_pvnet_amd()
    from clean_pvnet_amd import synth
    from lib.csrc.ransac_voting.ransac_voting_gpu import ransac_voting_layer_v3
    name, cfgname, B, over = [r for r in config_bench.ROWS if r[0] == rowname][0]
    cfg = dict(synth.CONFIGS[cfgname])
    hn, max_num = over.get("hn", cfg["hn"]), over.get("max_num", 30000)
    dev = torch.device("cuda:0")
    d = synth.make_batch(B=B, **{k: v for k, v in cfg.items() if k not in ("B", "hn")}, device=dev)
    for _ in range(calls):
        ransac_voting_layer_v3(d["mask"], d["vertex"], hn, inlier_thresh=0.99, max_num=max_num)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
