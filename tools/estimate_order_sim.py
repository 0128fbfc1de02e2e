#!/usr/bin/env python
"""CPU simulation (numpy, no GPU): how many (hypothesis, pixel) pairs the ESTIMATE's staged pass must evaluate under different
pixel orders, on synthetic config-3 images -- 4096 random-pair hypotheses per keypoint, the reference's inlier test, the pass's own
rule (first stage = chunks 1, 5 mod 8; a hypothesis is dropped once  partial + remaining - misses seen < best - ceil(tn / 10) - 2,
checked after every 512-pixel chunk).  Printed per (image, keypoint) as a share of the full pass:
  row     the second launch in row-major chunk order (rounds 4-5 until experiment (15))
  band    chunks ranked by |y - y_keypoint| (what count_filter_runs.hpp does now)
  pixel   the remaining pixels sorted by their distance to the keypoint (would need per-keypoint row copies)
  nearest_first_stage   additionally the first stage on the NEAREST quarter of the pixels
  keep    share of the hypotheses within the 0.1 window: evaluated in full under any order

    python tools/estimate_order_sim.py
"""
import sys, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lib; lib._register_clean_pvnet_amd()
from clean_pvnet_amd import synth
cfg = dict(synth.CONFIGS["cfg3"]); gen = {k: v for k, v in cfg.items() if k not in ("B", "hn")}
d = synth.make_batch(B=2, **gen, device="cpu")
rng = np.random.default_rng(0)
PC = 512
tot = {}
for b in range(2):
    mask = d["mask"][b].numpy() != 0
    ys, xs = np.nonzero(mask); tn = len(ys)
    coords = np.stack([xs, ys], 1).astype(np.float32)
    for k in range(0, 9, 2):
        dirs = d["vertex"][b].numpy()[ys, xs, k, :]          # [tn,2]
        kp = d["kpt_2d"][b, k].numpy()
        hn = 4096
        i0 = rng.integers(0, tn, hn); i1 = rng.integers(0, tn, hn)
        # intersection of two lines p + t d
        p0, d0, p1, d1 = coords[i0], dirs[i0], coords[i1], dirs[i1]
        n0 = np.stack([d0[:,1], -d0[:,0]],1); n1 = np.stack([d1[:,1], -d1[:,0]],1)
        c0 = (n0*p0).sum(1); c1 = (n1*p1).sum(1)
        det = n0[:,0]*n1[:,1]-n0[:,1]*n1[:,0]; det[det==0] = 1e-9
        hx = (c0*n1[:,1]-c1*n0[:,1])/det; hy = (n0[:,0]*c1-n1[:,0]*c0)/det
        H = np.stack([hx,hy],1)                              # [hn,2]
        diff = H[:,None,:]-coords[None,:,:]                   # [hn,tn,2]
        nd = np.linalg.norm(diff,axis=2)+1e-9; ndir = np.linalg.norm(dirs,axis=1)+1e-9
        cos = (diff*dirs[None]).sum(2)/nd/ndir[None]
        inl = cos > 0.99                                      # [hn,tn]
        full = inl.sum(1); best = full.max()
        nch = (tn+PC-1)//PC
        first = [c for c in range(nch) if c % 8 in (1,5)]
        rest = [c for c in range(nch) if c % 8 not in (1,5)]
        bound = best - ((tn+9)//10 + 2)                       # L* ~ best (leaders' sure inliers ~ exact)
        def chunk_px(c): return np.arange(c*PC, min(tn,(c+1)*PC))
        fpx = np.concatenate([chunk_px(c) for c in first])
        partial = inl[:,fpx].sum(1); R_rem = tn-len(fpx)
        def simulate(order_chunks=None, order_px=None):
            # returns processed (hyp,pixel) pairs in the second launch
            alive = partial + R_rem - bound >= 0
            miss = np.zeros(hn, int); work = 0
            if order_px is None: seq = [chunk_px(c) for c in order_chunks]
            else: seq = [order_px[i:i+PC] for i in range(0, len(order_px), PC)]
            for px in seq:
                work += alive.sum()*len(px)
                miss[alive] += (~inl[alive][:,px]).sum(1)
                alive &= (partial + R_rem - bound - miss >= 0)
            return work
        yc = {c: coords[min(c*PC+PC//2, tn-1),1] for c in rest}
        w_row = simulate(order_chunks=rest)
        w_band = simulate(order_chunks=sorted(rest, key=lambda c: abs(yc[c]-kp[1])))
        rpx = np.concatenate([chunk_px(c) for c in rest])
        dist = np.linalg.norm(coords[rpx]-kp[None],axis=1)
        w_px = simulate(order_px=rpx[np.argsort(dist)])
        # nearest FIRST STAGE too: first = nearest quarter of pixels, rest sorted by distance
        alld = np.linalg.norm(coords-kp[None],axis=1); o = np.argsort(alld)
        nf = len(fpx); f2 = o[:nf]; r2 = o[nf:]
        partial2 = inl[:,f2].sum(1)
        alive = partial2 + (tn-nf) - bound >= 0; miss = np.zeros(hn,int); work2 = 0
        for i in range(0,len(r2),PC):
            px = r2[i:i+PC]; work2 += alive.sum()*len(px); miss[alive] += (~inl[alive][:,px]).sum(1); alive &= (partial2 + (tn-nf) - bound - miss >= 0)
        fs = hn*len(fpx); fullw = hn*tn
        r = dict(row=float((fs+w_row)/fullw), band=float((fs+w_band)/fullw), pixel=float((fs+w_px)/fullw), nearest_first_stage=float((fs+work2)/fullw), keep=float((full >= bound).mean()))
        print(b, k, tn, {a: round(v,3) for a,v in r.items()}, flush=True)
        for a,v in r.items(): tot.setdefault(a,[]).append(v)
print({a: round(float(np.mean(v)),3) for a,v in tot.items()})
