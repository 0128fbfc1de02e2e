import sys, time, torch
sys.path.insert(0, '.')
import lib; lib._register_clean_pvnet_amd()
from clean_pvnet_amd import synth
from lib.csrc.ransac_voting.ransac_voting_gpu import ransac_voting_layer_v3
dev = torch.device('cuda')
cfg = dict(synth.CONFIGS['cfg3']); gen = {k: v for k, v in cfg.items() if k not in ('B', 'hn')}
bs = [synth.make_batch(B=64, **gen, first_index=r * 64, device=dev) for r in range(3)]
torch.cuda.synchronize(); time.sleep(0.5)
n = 120
evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
evs[0].record()
for i in range(n):
    d = bs[i % 3]; ransac_voting_layer_v3(d['mask'], d['vertex'], 512, inlier_thresh=0.99); evs[i + 1].record()
torch.cuda.synchronize()
per = [evs[i].elapsed_time(evs[i + 1]) for i in range(n)]
print('steps 0-9   ', ' '.join('%.3f' % x for x in per[:10]))
print('steps 10-29 ', ' '.join('%.3f' % x for x in per[10:30]))
print('steps 30-59 mean %.4f  60-119 mean %.4f' % (sum(per[30:60]) / 30, sum(per[60:]) / 60))
