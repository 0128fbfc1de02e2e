"""Seeded synthetic {mask, vertex} batches for tests, smoke and bench.

The vertex field follows the semantics of the reference's ground-truth generator
``compute_vertex`` (/root/reference/lib/utils/pvnet/pvnet_data_utils.py:30-44): on foreground
pixels the unit vector from the pixel to each keypoint (norm clamp ``norm[norm<1e-3] += 1e-3``),
plus N(0, sigma^2) noise per component; on background U(-1,1) garbage, because a network's
output is arbitrary there.  Every image is generated from ``seed + global image index`` only, so
a batch sharded over N ranks is the same data for every N.
"""
import math

import torch

# BASELINE.json configs (restated in BASELINE.md section 2)
CONFIGS = {
    "cfg1": dict(B=1, H=128, W=128, K=4, hn=64, fg=0.10, sigma=0.02),
    "cfg2": dict(B=1, H=480, W=640, K=9, hn=512, fg=0.02, sigma=0.05),
    "cfg3": dict(B=64, H=480, W=640, K=9, hn=512, fg=0.02, sigma=0.05),
    "cfg4": dict(B=32, H=480, W=640, K=9, hn=1024, fg=(0.002, 0.01), sigma=0.05, outlier=(0.3, 0.5), occlude=True),
    "cfg5": dict(B=16, H=540, W=720, K=17, hn=2048, fg=0.08, sigma=0.05),
}


def _image(i, H, W, K, fg, sigma, outlier, occlude, seed, device, mask_dtype, wrong_region=0.0, kp_outlier=None):
    g = torch.Generator(device="cpu").manual_seed(seed + i)
    u = lambda lo, hi: lo + (hi - lo) * torch.rand((), generator=g).item()  # noqa: E731
    frac = u(*fg) if isinstance(fg, (tuple, list)) else fg
    out_frac = u(*outlier) if isinstance(outlier, (tuple, list)) else outlier
    aspect = u(0.6, 1.6)
    area = frac * H * W
    a = math.sqrt(area * aspect / math.pi)
    b = area / (math.pi * a)
    cx, cy = u(0.3 * W, 0.7 * W), u(0.3 * H, 0.7 * H)
    th = u(0.0, math.pi)
    ys = torch.arange(H, device=device, dtype=torch.float32).view(H, 1)
    xs = torch.arange(W, device=device, dtype=torch.float32).view(1, W)
    xr = (xs - cx) * math.cos(th) + (ys - cy) * math.sin(th)
    yr = -(xs - cx) * math.sin(th) + (ys - cy) * math.cos(th)
    m = ((xr / a) ** 2 + (yr / b) ** 2) <= 1.0
    if occlude:
        for _ in range(3):
            ox, oy = u(cx - a, cx + a), u(cy - b, cy + b)
            ow, oh = u(0.2 * a, 0.8 * a), u(0.2 * b, 0.8 * b)
            m &= ~((xs >= ox) & (xs < ox + ow) & (ys >= oy) & (ys < oy + oh))
    r = 1.5 * max(a, b)
    kx = torch.tensor([u(cx - r, cx + r) for _ in range(K)], device=device)
    ky = torch.tensor([u(cy - r, cy + r) for _ in range(K)], device=device)

    dg = torch.Generator(device=device).manual_seed(seed + i)
    vx = kx.view(1, 1, K) - xs.view(1, W, 1)
    vy = ky.view(1, 1, K) - ys.view(H, 1, 1)
    norm = torch.sqrt(vx * vx + vy * vy)
    norm = torch.where(norm < 1e-3, norm + 1e-3, norm)
    v = torch.stack([vx / norm, vy / norm], -1)                       # [H,W,K,2]
    v = v + sigma * torch.randn(v.shape, generator=dg, device=device)
    if out_frac > 0:
        ang = 2 * math.pi * torch.rand((H, W, K), generator=dg, device=device)
        rnd = torch.stack([torch.cos(ang), torch.sin(ang)], -1)
        sel = torch.rand((H, W, 1, 1), generator=dg, device=device) < out_frac
        v = torch.where(sel, rnd, v)
    # STRUCTURED errors (round 5: what AUTO's stage decision is tested against besides uniform random outliers):
    #  wrong_region  a contiguous slab of the object (one end of the ellipse, ~this fraction of its extent along the major axis)
    #                whose field votes, consistently, for a WRONG point per keypoint -- a decoy 0.5-1.5 object sizes away;
    #  kp_outlier    (lo, hi): every keypoint gets its own fraction of random-direction pixels, uniform in [lo, hi] -- the winners'
    #                inlier ratios then spread over keypoints within one image instead of sitting at one value.
    if wrong_region and wrong_region > 0:
        slab = xr > a * (1.0 - 2.0 * float(wrong_region))
        ang_d = torch.tensor([u(0.0, 2 * math.pi) for _ in range(K)], device=device)
        rad_d = torch.tensor([u(0.5, 1.5) * max(a, b) for _ in range(K)], device=device)
        dkx, dky = kx + rad_d * torch.cos(ang_d), ky + rad_d * torch.sin(ang_d)
        wx = dkx.view(1, 1, K) - xs.view(1, W, 1)
        wy = dky.view(1, 1, K) - ys.view(H, 1, 1)
        wn = torch.sqrt(wx * wx + wy * wy).clamp(min=1e-3)
        vw = torch.stack([wx / wn, wy / wn], -1) + sigma * torch.randn(v.shape, generator=dg, device=device)
        v = torch.where(slab.view(H, W, 1, 1), vw, v)
    if kp_outlier is not None:
        fk = torch.tensor([u(*kp_outlier) for _ in range(K)], device=device)
        ang = 2 * math.pi * torch.rand((H, W, K), generator=dg, device=device)
        rnd = torch.stack([torch.cos(ang), torch.sin(ang)], -1)
        sel = torch.rand((H, W, K), generator=dg, device=device) < fk.view(1, 1, K)
        v = torch.where(sel.unsqueeze(-1), rnd, v)
    bg = 2 * torch.rand(v.shape, generator=dg, device=device) - 1
    v = torch.where(m.view(H, W, 1, 1), v, bg)
    return m.to(mask_dtype), v.float(), torch.stack([kx, ky], -1)


def make_batch(B, H, W, K, fg=0.02, sigma=0.05, outlier=0.0, occlude=False, seed=1234, first_index=0,
               device="cpu", mask_dtype=torch.int64, planar=False, wrong_region=0.0, kp_outlier=None, **_unused):
    """-> dict(mask [B,H,W], vertex [B,H,W,K,2] float32, kpt_2d [B,K,2]).

    ``planar=True`` returns the vertex as the strided view ``decode_keypoint`` produces
    (resnet18.py:66-68): storage ``[B,2K,H,W]``, ``permute(0,2,3,1).view(B,H,W,K,2)``."""
    masks, verts, kpts = [], [], []
    for i in range(B):
        m, v, k = _image(first_index + i, H, W, K, fg, sigma, outlier, occlude, seed, device, mask_dtype, wrong_region, kp_outlier)
        masks.append(m); verts.append(v); kpts.append(k)
    mask = torch.stack(masks)
    vertex = torch.stack(verts)
    if planar:
        store = vertex.view(B, H, W, 2 * K).permute(0, 3, 1, 2).contiguous()    # [B,2K,H,W]
        vertex = store.permute(0, 2, 3, 1).view(B, H, W, K, 2)
    return dict(mask=mask, vertex=vertex, kpt_2d=torch.stack(kpts))


def make_idxs(tn, hn, K, seed=1234, first_index=0):
    """Injected hypothesis index pairs: per image i, ``randint(0, tn[i])`` from a CPU generator seeded
    with ``seed + global index`` -> ``[B,hn,K,2]`` int32 (zeros where tn[i] == 0)."""
    out = torch.zeros((len(tn), hn, K, 2), dtype=torch.int32)
    for i, t in enumerate(tn):
        if int(t) > 0:
            g = torch.Generator(device="cpu").manual_seed(10_000_019 * (seed + first_index + i) + 7)
            out[i] = torch.randint(0, int(t), (hn, K, 2), generator=g, dtype=torch.int32)
    return out


def dense_field_bytes(B, H, W, K, hn):
    """ALGORITHMIC bytes of one inlier-count pass (SURVEY.md section 8d / BASELINE.md): the dense
    [B,H,W,K,2] f32 field + a u8 mask + hypotheses in + int32 counts out."""
    return B * (H * W * (K * 8 + 1) + hn * K * 8 + hn * K * 4)
