import os, sys
ROOT="/root/repo" if os.path.isdir("/root/repo/tools") else os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests")); sys.path.insert(0, os.path.join(ROOT,"tools"))
import torch, capi, variant_time
synth = variant_time._synth()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = dict(synth.CONFIGS["cfg3"]); hn = cfg["hn"]
dev = torch.device("cuda:0")
d = synth.make_batch(B=B, **{k: v for k, v in cfg.items() if k not in ("B", "hn")}, device=dev)
GRID = 48*256+64
dbg = torch.zeros(64 + 4*GRID, dtype=torch.int64, device=dev)
os.environ["PVV_DBG_PTR"] = str(dbg.data_ptr())
for _ in range(20):
    dbg.zero_(); capi.v3(d["mask"], d["vertex"], hn, 0.99, max_num=30000, seed=5, count_kernel=3); torch.cuda.synchronize()
c = dbg.cpu()[64:].view(-1,4); c = c[c[:,0] != 0]
t0 = int(c[:,0].min()); ent=(c[:,0]-t0).double()/100; ext=(c[:,1]-t0).double()/100; items=c[:,3]
print("blocks", len(c), "items", int(items.sum()), "exit max %.1f" % ext.max())
for k in range(0, int(items.max())+1):
    m = items==k
    if m.any(): print(" blocks with %d items: %d; entry median %.1f max %.1f; exit median %.1f p90 %.1f max %.1f; life median %.1f" % (k, int(m.sum()), ent[m].median(), ent[m].max(), ext[m].median(), ext[m].quantile(0.9), ext[m].max(), (ext-ent)[m].median()))
for t in (5,10,20,30,35,40,45,50,55):
    print("  t=%2d us: %4d blocks alive" % (t, int(((ent<=t)&(ext>t)&(items>0)).sum())))
