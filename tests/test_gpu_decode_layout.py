"""GPU: the layout the real caller passes to decode_keypoint (resnet18.py:66-69,93-94) -- a two-class seg and the planar
vertex field as channel slices of ONE [B, 2+2K, H, W] tensor -- through every instantiation of the fused mask scan:

  k_tile_scan_seg2<false> + k_mask_from_lists on the side stream   two contiguous seg planes, a batch of >= 2^21 pixels, lists
                                                                   that stay complete (un_pnp path: max_num = 30000)
  k_tile_scan_seg2<true>                                           the same planes on a small batch, or when subsampling has its
                                                                   own pass (the default call: max_num = 100)
  k_tile_scan<.., false, ..> with the argmax inside                strided planes, H*W not a multiple of 4, unaligned slices

Every one of them must give torch.argmax's mask (first maximum; a NaN beats everything, the first NaN wins), and the
keypoints of the unfused call on the same draws, bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _network_output(synth, gpu, B, H, W, K, seed, seg_offset=0, seg_stride=1):
    d = synth.make_batch(B=B, H=H, W=W, K=K, fg=0.06, sigma=0.05, seed=seed, planar=True)
    g = torch.Generator().manual_seed(seed)
    C = 2
    x = torch.empty(B, seg_offset + C * seg_stride + 2 * K, H, W)
    seg = x[:, seg_offset:seg_offset + C * seg_stride:seg_stride]
    seg.copy_(torch.randn(B, C, H, W, generator=g) * 0.1)
    seg[:, 0] += 1.0
    seg[:, 1][d["mask"] != 0] += 4.0
    # torch.argmax's corner cases, at tile starts / ends / the last pixels of the image
    pts = [0, 1, 2047, 2048, 4095, H * W - 1, H * W - 2, H * W // 2 + 3]
    for i, p in enumerate(pts):
        if not 0 <= p < H * W:
            continue
        y, xx = divmod(p, W)
        b = i % B
        if i % 4 == 0:
            seg[b, 0, y, xx] = float("nan")                     # NaN in class 0 -> index 0
        elif i % 4 == 1:
            seg[b, 1, y, xx] = float("nan")                     # NaN in class 1 -> index 1
        elif i % 4 == 2:
            seg[b, :, y, xx] = float("nan")                     # both NaN -> the first
        else:
            seg[b, :, y, xx] = 0.75                             # tie -> the first
    if B > 2:
        seg[B - 1, 1] = -9.0                                    # an image without foreground
    ver = x[:, seg_offset + C * seg_stride:]
    ver.copy_(d["vertex"].permute(0, 3, 4, 1, 2).reshape(B, 2 * K, H, W))
    x = x.to(gpu)
    return x[:, seg_offset:seg_offset + C * seg_stride:seg_stride], x[:, seg_offset + C * seg_stride:], d


CASES = [
    # B, H, W, K, seg_offset, seg_stride, what
    (24, 300, 404, 3, 0, 1, "seg2 + deferred mask, partial last tile"),     # 2.9 M pixels, H*W = 59.2 tiles
    (3, 300, 404, 3, 0, 1, "seg2 writes the mask itself (small batch)"),
    (3, 96, 128, 3, 1, 1, "seg planes start 1 plane into the tensor (still aligned)"),
    (3, 96, 128, 3, 0, 2, "every second channel: gc = 2 planes"),
    (3, 33, 35, 3, 0, 1, "H*W % 4 != 0: generic argmax scan"),
    (2, 97, 131, 3, 1, 1, "odd plane size, offset slice: unaligned -> generic scan"),
]


@pytest.mark.parametrize("B,H,W,K,off,stride,what", CASES, ids=[c[-1] for c in CASES])
def test_fused_decode_equals_torch_argmax_plus_v3(synth, pkg, gpu, B, H, W, K, off, stride, what):
    from clean_pvnet_amd import ransac_voting as ext
    seg, ver, d = _network_output(synth, gpu, B, H, W, K, seed=300 + B + H, seg_offset=off, seg_stride=stride)
    vertex = ver.permute(0, 2, 3, 1).view(B, H, W, K, 2)
    mask_ref = torch.argmax(seg, 1)
    tn = [int(v) for v in (mask_ref != 0).sum((1, 2)).cpu()]
    hn = 64
    idxs = synth.make_idxs(tn, hn, K, seed=17).to(gpu)
    for max_num in (30000, 150):                                # 150: k_tile_subsample rewrites the lists -> never deferred
        sel = torch.rand(B, H, W, generator=torch.Generator().manual_seed(5)).to(gpu) if max_num == 150 else None
        ii = None if max_num == 150 else idxs                   # (injected pairs address the subsampled list: device RNG there)
        out_ref, win_ref, tn_ref, _ws = ext.ransac_voting_v3(mask_ref, vertex, hn, 0.99, 5, max_num, ii, sel, 11, ext.SINGULAR_REFERENCE)
        out, mask, win, tnn = ext.decode_keypoint_v3(seg, vertex, hn, 0.99, 5, max_num, ii, sel, 11, ext.SINGULAR_REFERENCE)
        assert mask.dtype == torch.int64 and torch.equal(mask, mask_ref), what
        assert torch.equal(tnn, tn_ref) and torch.equal(win, win_ref) and torch.equal(out, out_ref), what
    if B > 2:
        assert tn[B - 1] == 0 and bool((out[B - 1] == 0).all())


def test_fused_un_pnp_pass_with_a_deferred_mask(synth, pkg, gpu):
    """pvv_decode_keypoint_un_pnp on a batch large enough to defer the mask: equal to argmax + v3 + estimate under one seed."""
    from clean_pvnet_amd import ransac_voting as ext
    B, H, W, K = 10, 480, 640, 4
    seg, ver, d = _network_output(synth, gpu, B, H, W, K, seed=901)
    vertex = ver.permute(0, 2, 3, 1).view(B, H, W, K, 2)
    mask_ref = torch.argmax(seg, 1)
    kpt, mask, cov, w, win, tnn = ext.decode_keypoint_un_pnp(seg, vertex, 128, 512, 0.99, 5, 30000, None, None, None, 77, ext.SINGULAR_REFERENCE)
    assert torch.equal(mask, mask_ref)
    kpt2, win2, tn2, _ws = ext.ransac_voting_v3(mask_ref, vertex, 128, 0.99, 5, 30000, None, None, 77, ext.SINGULAR_REFERENCE)
    cov2, _h, _c, _t, w2 = ext.estimate_voting_distribution(mask_ref, vertex, kpt2, 512, 0.99, 5, 30000, None, None, 77, False)
    assert torch.equal(kpt, kpt2) and torch.equal(win, win2) and torch.equal(tnn, tn2) and torch.equal(cov, cov2)
    assert np.abs(kpt[:B - 1].cpu().numpy() - d["kpt_2d"][:B - 1].numpy()).max() < 20.0     # (a sanity bound: keypoints lie up to 1.5 object radii away, sigma = 0.05)


def test_deferred_mask_under_concurrent_streams_and_back_to_back_calls(synth, pkg, gpu):
    """The side stream is one per device and its events are reused: calls issued back to back on two caller streams must
    each get their own mask."""
    from clean_pvnet_amd import ransac_voting as ext
    B, H, W, K = 8, 480, 640, 2
    cases = [_network_output(synth, gpu, B, H, W, K, seed=40 + i) for i in range(3)]
    refs = [torch.argmax(c[0], 1) for c in cases]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = []
    for rep in range(12):
        i = rep % 3
        seg, ver, _d = cases[i]
        with torch.cuda.stream(streams[rep % 2]):
            o = ext.decode_keypoint_v3(seg, ver.permute(0, 2, 3, 1).view(B, H, W, K, 2), 64, 0.99, 5, 30000, None, None, 3, ext.SINGULAR_REFERENCE)
        outs.append((i, o[1]))
    torch.cuda.synchronize()
    for i, m in outs:
        assert torch.equal(m, refs[i])


def test_fused_decode_of_a_large_batch_inside_a_captured_graph(synth, pkg, gpu):
    """While the caller's stream is being captured no side stream is used (nothing may be created or forked under capture): the
    scan writes the mask itself.  The captured call replays with the same result as the eager one (which defers the mask)."""
    from clean_pvnet_amd import ransac_voting as ext
    B, H, W, K = 8, 480, 640, 3
    seg, ver, _d = _network_output(synth, gpu, B, H, W, K, seed=77)
    vertex = ver.permute(0, 2, 3, 1).view(B, H, W, K, 2)
    eager = ext.decode_keypoint_v3(seg, vertex, 64, 0.99, 5, 30000, None, None, 5, ext.SINGULAR_REFERENCE)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2):
            ext.decode_keypoint_v3(seg, vertex, 64, 0.99, 5, 30000, None, None, 5, ext.SINGULAR_REFERENCE)
        with torch.cuda.graph(g, stream=s):
            out = ext.decode_keypoint_v3(seg, vertex, 64, 0.99, 5, 30000, None, None, 5, ext.SINGULAR_REFERENCE)
    for _ in range(3):
        out[1].zero_()
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out[1], torch.argmax(seg, 1)) and torch.equal(out[1], eager[1])
    assert torch.equal(out[0], eager[0]) and torch.equal(out[2], eager[2])


def test_deferred_mask_from_several_host_threads(synth, pkg, gpu):
    """Four host threads, each with its own stream, decode different batches at the same time: one side stream per device and
    a ring of events behind a mutex -- every call gets its own mask."""
    import threading
    from clean_pvnet_amd import ransac_voting as ext
    B, H, W, K = 8, 480, 640, 2
    cases = [_network_output(synth, gpu, B, H, W, K, seed=600 + i) for i in range(4)]
    refs = [torch.argmax(c[0], 1) for c in cases]
    torch.cuda.synchronize()
    errs = []

    def work(i):
        try:
            st = torch.cuda.Stream()
            seg, ver, _d = cases[i]
            with torch.cuda.stream(st):
                for _ in range(10):
                    o = ext.decode_keypoint_v3(seg, ver.permute(0, 2, 3, 1).view(B, H, W, K, 2), 64, 0.99, 5, 30000, None, None, 3,
                                               ext.SINGULAR_REFERENCE)
                    if not torch.equal(o[1], refs[i]):
                        errs.append(i)
            st.synchronize()
        except Exception as e:                                                  # noqa: BLE001
            errs.append(repr(e))
    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
