#!/usr/bin/env python
"""Like trace_calls.py but through the C ABI (tests/capi.py), so that PVV_LIBPATH can point at an experimental build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, capi, config_bench, variant_time
synth = variant_time._synth()
rowname = sys.argv[1]; calls = int(sys.argv[2]) if len(sys.argv) > 2 else 60
name, cfgname, B, over = [r for r in config_bench.ROWS if r[0] == rowname][0]
cfg = dict(synth.CONFIGS[cfgname])
hn, max_num = over.get("hn", cfg["hn"]), over.get("max_num", 30000)
dev = torch.device("cuda:0")
d = synth.make_batch(B=B, **{k: v for k, v in cfg.items() if k not in ("B", "hn")}, device=dev)
for _ in range(calls):
    capi.v3(d["mask"], d["vertex"], hn, 0.99, max_num=max_num, seed=5)
torch.cuda.synchronize()
