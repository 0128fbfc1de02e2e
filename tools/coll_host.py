import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29531")
dist.init_process_group("nccl", rank=0, world_size=1)
dev=torch.device("cuda",0); torch.cuda.set_device(dev)
x=torch.zeros(64,9,2,device=dev); out=torch.empty(64,9,2,device=dev)
for name,fn in (("sync", lambda: dist.all_gather_into_tensor(out,x)),
                ("async+wait", lambda: dist.all_gather_into_tensor(out,x,async_op=True).wait()),
                ("async", lambda: dist.all_gather_into_tensor(out,x,async_op=True))):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(1000): fn()
    t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print(name, "host us/call %.1f"%((t1-t0)*1e3), " incl. drain %.1f"%((t2-t0)*1e3))
