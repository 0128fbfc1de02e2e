import sys, time, torch
sys.path.insert(0, '.')
import lib; lib._register_clean_pvnet_amd()
from clean_pvnet_amd import synth, ransac_voting as ext
dev = torch.device('cuda')
d = synth.make_batch(**{**synth.CONFIGS['cfg2'], 'B': 1}, device=dev)
m, v = d['mask'], d['vertex']
def call(): return ext.ransac_voting_v3(m, v, 512, 0.99, 5, 30000, None, None, 1, ext.SINGULAR_REFERENCE)
for _ in range(5): call()
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(200): call()
torch.cuda.synchronize(); print('eager  B=1 us/image', (time.perf_counter()-t)/200*1e6)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g): out = call()
for _ in range(5): g.replay()
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(200): g.replay()
torch.cuda.synchronize(); print('graph  B=1 us/image', (time.perf_counter()-t)/200*1e6)
