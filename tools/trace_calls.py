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
    dev = torch.device("cuda:0")
    case = config_bench.make_case(rowname, dev)
    for _ in range(calls):
        case["call"]()
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
