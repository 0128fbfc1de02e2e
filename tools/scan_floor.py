import os, sys
sys.path.insert(0, os.getcwd())
import torch, lib
lib._register_clean_pvnet_amd()
from clean_pvnet_amd import ransac_voting as ext, synth
dev = torch.device("cuda", 0)
cfg = dict(synth.CONFIGS["cfg3"]); gen = {k: v for k, v in cfg.items() if k not in ("B", "hn")}
batches = [synth.make_batch(B=64, **gen, first_index=1000 * r, device=dev) for r in range(3)]
def med(v): v = sorted(v); return v[len(v)//2]
for name, masks in (("real masks", [d["mask"] for d in batches]), ("all background", [torch.zeros_like(d["mask"]) for d in batches]),
                    ("uint8 real", [d["mask"].to(torch.uint8) for d in batches])):
    st = ext.stage_ms_in_pipeline(masks, [d["vertex"] for d in batches], 512, 0.99, 5, 30000, 7, 30, ext.COUNT_AUTO, False)[6:]
    print("%-16s scan %.2f us  compact %.2f us" % (name, 1e3 * med([r[0] for r in st]), 1e3 * med([r[1] for r in st])))
