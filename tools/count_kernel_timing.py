import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lib; lib._register_clean_pvnet_amd()
from clean_pvnet_amd import synth, ransac_voting as ext
dev = torch.device('cuda')
cfg = dict(synth.CONFIGS['cfg3']); gen = {k: v for k, v in cfg.items() if k not in ('B', 'hn')}
bs = [synth.make_batch(B=64, **gen, first_index=r * 64, device=dev) for r in range(3)]
def call(i, extra):
    d = bs[i % 3]
    o, w, t, ws = ext.ransac_voting_v3(d['mask'], d['vertex'], 512, 0.99, 5, 30000, None, None, i, 0)
    for _ in range(extra):
        ext.rerun_count_kernel(d['mask'], d['vertex'], 512, 0.99, 5, 30000, ws, False)
for i in range(300): call(i, 0)
torch.cuda.synchronize()
def group(extra, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): call(i, extra)
    e1.record()
    return e0, e1, n
res = {0: [], 1: [], 2: []}
evs = []
for r in range(12):
    for x in (0, 1, 2):
        evs.append((x,) + group(x))
torch.cuda.synchronize()
for x, e0, e1, n in evs: res[x].append(e0.elapsed_time(e1) / n)
m = {x: sum(v) / len(v) for x, v in res.items()}
print('step ms: plain %.4f, +1 count %.4f, +2 counts %.4f  => count kernel in pipeline %.4f / %.4f ms' % (m[0], m[1], m[2], m[1] - m[0], (m[2] - m[0]) / 2))
ms = ext.count_kernel_ms_in_pipeline([d['mask'] for d in bs], [d['vertex'] for d in bs], 512, 0.99, 5, 30000, 1, 30)
print('events inside calls: mean %.4f' % (sum(ms) / len(ms)))
d = bs[0]; o, w, t, ws = ext.ransac_voting_v3(d['mask'], d['vertex'], 512, 0.99, 5, 30000, None, None, 1, 0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(5): ext.rerun_count_kernel(d['mask'], d['vertex'], 512, 0.99, 5, 30000, ws, False)
e0.record()
for _ in range(30): ext.rerun_count_kernel(d['mask'], d['vertex'], 512, 0.99, 5, 30000, ws, False)
e1.record(); torch.cuda.synchronize()
print('relaunched alone: %.4f' % (e0.elapsed_time(e1) / 30))
