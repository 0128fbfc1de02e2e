"""Survivors per (work item, group) of the staged count's second launch, read from a debug build
(tools/build_variant.sh dbgns -DPVV_TUNING --py <patch adding atomics on dbg[0..4]>): PVV_LIBPATH=build/variants/dbgns.so"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
dev = torch.device("cuda:0")
dbg = torch.zeros(512, dtype=torch.int64, device=dev)
os.environ["PVV_DBG_PTR"] = hex(dbg.data_ptr())
import capi
import importlib.util
spec = importlib.util.spec_from_file_location("pvv_synth", os.path.join(ROOT, "clean-pvnet_amd", "synth.py"))
synth = importlib.util.module_from_spec(spec); spec.loader.exec_module(synth)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = dict(synth.CONFIGS["cfg3"])
d = synth.make_batch(B, cfg["H"], cfg["W"], cfg["K"], device=dev, **{k: v for k, v in cfg.items() if k not in ("B", "H", "W", "K", "hn")})
out, win, tn = capi.v3(d["mask"], d["vertex"], 512, 0.99, seed=5, count_kernel=3)
torch.cuda.synchronize()
v = dbg[:5].tolist()
print("items*groups", v[1], "mean survivors %.1f" % (v[0] / max(1, v[1])), "mean tiles %.2f" % (v[2] / max(1, v[1])), "mean L* %.1f" % (v[3] / max(1, v[1])), "mean R %.1f" % (v[4] / max(1, v[1])), "tn", tn[:4].tolist())
