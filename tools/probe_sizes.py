import sys, torch
sys.path.insert(0, "/root/repo")
import lib; lib._register_clean_pvnet_amd()
from clean_pvnet_amd import ransac_voting as ext
dev = torch.device("cuda:0")
big = torch.empty(3 << 30, dtype=torch.uint8, device=dev).random_(0, 255)
sink = torch.zeros(1, dtype=torch.int32, device=dev)
for mb in (16, 64, 157, 512, 1400):
    n = mb << 20
    views = [big[i * n:(i + 1) * n] for i in range(min(8, (3 << 30) // n))]
    for i in range(6): ext.stream_read_probe(views[i % len(views)], sink)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(20): ext.stream_read_probe(views[i % len(views)], sink)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    print("probe %5d MB: %.1f us per pass, %.0f GB/s" % (mb, us, n / us / 1e3))
