"""GPU parity of the compaction workers (round 3, compaction.hpp): k_compact_hyp gives every image W blocks, each of which
takes every W-th FOREGROUND tile of the image (found in one pass over the tile table), instead of one block per tile.

Every case runs with the library's W, with one worker per image (a worker walks all foreground tiles), with 3 (several
rounds) and with more workers than tiles (one block per foreground tile, the rest leave) -- and must agree bit for bit in
EVERYTHING that depends on the compacted rows: tn, status, the pixel every device-drawn index pair resolved to, all K*hn
hypotheses and ALL their inlier counts (estimate entry), keypoints and winner counts (v3 entry).  The oracle comparisons
of the other GPU test files run with the library's W."""
import pytest
import torch

from tests import capi

pytestmark = pytest.mark.gpu

FRONTS = (("auto", 0), ("one", 1), ("three", 3), ("classic", 100000))


def _each_front(fn):
    L = capi.load()
    res = {}
    try:
        for name, mode in FRONTS:
            capi.check(L.pvv_debug_option(1, mode))
            res[name] = fn()
            torch.cuda.synchronize()
    finally:
        capi.check(L.pvv_debug_option(1, 0))
    return res


def _assert_all_equal(res):
    ref = res["classic"]
    for name in ("auto", "one", "three"):
        assert len(res[name]) == len(ref)
        for i, (a, b) in enumerate(zip(res[name], ref)):
            if a.is_floating_point():                          # bit patterns: a skipped image's NaN must be the same NaN
                a, b = a.contiguous().view(torch.int32), b.contiguous().view(torch.int32)
            assert torch.equal(a, b), "W = %s differs from one block per tile in output %d" % (name, i)


def _both_layers(mask, vertex, hn, thresh, gpu, idxs=None, selection=None, **kw):
    """-> per front: (v3 out, win, tn, status, draws, estimate cov, hyps, counts, tn)"""
    B, H, W, K, _ = vertex.shape
    m, v = mask.to(gpu), vertex.to(gpu)
    i = None if idxs is None else idxs.to(gpu)
    s = None if selection is None else selection.to(gpu)
    mean = torch.zeros(B, K, 2, device=gpu) + torch.tensor([W / 2.0, H / 2.0], device=gpu)

    def run():
        status = torch.full((B,), -1, dtype=torch.int32, device=gpu)
        draws = torch.full((B, K, hn, 2), -7, dtype=torch.int32, device=gpu)
        out, win, tn = capi.v3(m, v, hn, thresh, idxs=i, selection=s, status=status, draws_out=draws, seed=11, **kw)
        cov, hyp, counts, tn2 = capi.estimate(m, v, mean, hn, thresh, idxs=i, selection=s, seed=11, **kw)
        return [t.cpu() for t in (out, win, tn, status, draws, cov, hyp, counts, tn2)]

    return _each_front(run)


@pytest.mark.parametrize("B,H,W,K,hn,dtype", [
    (5, 480, 640, 9, 64, torch.int64),        # the benchmark's image size, 150 tiles, int64 mask as argmax emits it
    (3, 480, 640, 9, 64, torch.uint8),
    (1, 480, 640, 4, 32, torch.bool),
    (7, 120, 160, 3, 32, torch.int32),        # 10 tiles
    (4, 32, 40, 2, 16, torch.int16),          # one tile per image
    (2, 540, 720, 17, 32, torch.int64),       # 190 tiles (config 5's size; nothing subsampled at fg = 2 %)
    (70, 96, 128, 2, 16, torch.uint8),        # more images than one generation of scan blocks sees side by side
])
def test_workers_agree_device_rng(synth, pkg, gpu, B, H, W, K, hn, dtype):
    d = synth.make_batch(B=B, H=H, W=W, K=K, fg=(0.01, 0.06), sigma=0.05, seed=4100 + B, mask_dtype=dtype)
    res = _both_layers(d["mask"], d["vertex"], hn, 0.99, gpu)
    _assert_all_equal(res)
    assert int(res["auto"][2].min()) > 0


def test_workers_agree_injected_index_pairs_and_oracle_rows(oracle, synth, pkg, gpu):
    """Injected index pairs (the subsampling is then NOT fused: scan, k_tile_subsample, compaction) -- and the rows
    themselves against numpy: tn, and the estimate's hypotheses from known pairs pin coords and dirs of the drawn rows."""
    B, H, W, K, hn = 4, 240, 320, 5, 48
    d = synth.make_batch(B=B, H=H, W=W, K=K, fg=(0.02, 0.08), sigma=0.03, seed=77)
    tn = [int(x) for x in (d["mask"] != 0).sum((1, 2))]
    idxs = synth.make_idxs(tn, hn, K, seed=78)
    res = _both_layers(d["mask"], d["vertex"], hn, 0.99, gpu, idxs=idxs)
    _assert_all_equal(res)
    assert res["auto"][2].tolist() == tn
    # draws_out = pixel (y*W+x) of the injected ROW index: row r of the compacted list must be the r-th foreground pixel
    draws = res["auto"][4]                                   # [B,K,hn,2]
    for b in range(B):
        nz = torch.nonzero(d["mask"][b].reshape(-1)).reshape(-1).to(torch.int32)
        want = nz[idxs[b].long()]                            # [hn,K,2]
        assert torch.equal(draws[b], want.permute(1, 0, 2).contiguous())


@pytest.mark.parametrize("inject", [False, True])
def test_workers_agree_when_some_images_are_subsampled(synth, pkg, gpu, inject):
    """max_num between the images' foreground_num: fused subsampling with the device RNG (a worker recounts the survivors
    before each of its tiles), k_tile_subsample with injected draws."""
    B, H, W, K, hn = 9, 120, 160, 3, 32
    d = synth.make_batch(B=B, H=H, W=W, K=K, fg=(0.03, 0.20), sigma=0.05, seed=5150, mask_dtype=torch.uint8)
    max_num = 1500                                            # 16 * 1500 >= 19200: the subsampling is fused
    fg = d["mask"].long().sum((1, 2))
    assert int((fg > max_num).sum()) >= 2 and int((fg <= max_num).sum()) >= 2, fg.tolist()
    sel = idxs = None
    if inject:
        sel = torch.rand(d["mask"].shape, generator=torch.Generator().manual_seed(9))
        keep = (d["mask"] != 0) & ((fg <= max_num).view(-1, 1, 1) | (sel < (torch.tensor(float(max_num)) / fg.float()).view(-1, 1, 1)))
        idxs = synth.make_idxs([int(x) for x in keep.sum((1, 2))], hn, K, seed=10)
    res = _both_layers(d["mask"], d["vertex"], hn, 0.99, gpu, idxs=idxs, selection=sel, max_num=max_num)
    _assert_all_equal(res)
    status = res["auto"][3].tolist()
    assert [bool(s & 2) for s in status] == (fg > max_num).tolist()     # PVV_STATUS_SUBSAMPLED


def test_workers_agree_on_skipped_empty_full_and_truncated_images(synth, pkg, gpu):
    B, H, W, K, hn = 6, 96, 128, 2, 16
    d = synth.make_batch(B=B, H=H, W=W, K=K, fg=0.05, sigma=0.05, seed=31, mask_dtype=torch.uint8)
    mask = d["mask"].clone()
    mask[1] = 0                                               # empty: skipped
    mask[2] = 0
    mask[2, 95, 125:128] = 1                                  # 3 pixels in the LAST tile only: below min_num
    mask[3] = 1                                               # every pixel: foreground_num = H*W > max_num -> subsampled
    mask[4] = 0
    mask[4, 0, 0:40] = 1                                      # first tile only
    res = _both_layers(mask, d["vertex"], hn, 0.99, gpu, max_num=4000)
    _assert_all_equal(res)
    tn, status = res["auto"][2].tolist(), res["auto"][3].tolist()
    assert tn[1] == 0 and tn[2] == 0 and status[1] == 1 and status[2] == 1     # PVV_STATUS_SKIPPED
    assert status[3] & 2 and tn[4] == 40
    # list truncated at cap (a caller-chosen cap below the foreground count): rows beyond cap are never written
    res = _both_layers(d["mask"], d["vertex"], hn, 0.99, gpu, max_num=4000, cap=300)
    _assert_all_equal(res)
    assert all(t == 300 for t in res["auto"][2].tolist()) and all(s & 4 for s in res["auto"][3].tolist())   # TRUNCATED


def test_workers_agree_on_a_strided_mask_and_a_planar_vertex(synth, pkg, gpu):
    """non-contiguous mask (the scan instantiation without read-ahead) and the strided vertex view of decode_keypoint"""
    B, H, W, K, hn = 3, 200, 256, 4, 32
    d = synth.make_batch(B=B, H=H, W=W, K=K, fg=0.05, sigma=0.05, seed=61, planar=True)
    wide = torch.zeros(B, H, 2 * W, dtype=torch.int64)
    wide[:, :, ::2] = d["mask"]
    m = wide.to(gpu)[:, :, ::2]
    assert not m.is_contiguous()
    v = d["vertex"].to(gpu)
    mean = torch.zeros(B, K, 2, device=gpu)

    def run():
        out, win, tn = capi.v3(m, v, hn, 0.99, seed=3)
        cov, hyp, counts, tn2 = capi.estimate(m, v, mean, hn, 0.99, seed=3)
        return [t.cpu() for t in (out, win, tn, cov, hyp, counts, tn2)]

    _assert_all_equal(_each_front(run))


def test_workers_agree_through_the_fused_argmax(synth, pkg, gpu):
    from clean_pvnet_amd import ransac_voting as ext
    B, H, W, K, hn = 4, 240, 320, 9, 64
    d = synth.make_batch(B=B, H=H, W=W, K=K, fg=(0.02, 0.06), sigma=0.05, seed=91)
    seg = torch.randn(B, 2, H, W, generator=torch.Generator().manual_seed(5)) * 0.1
    seg[:, 1] += (d["mask"] != 0).float() * 2 - 1
    s, v = seg.to(gpu), d["vertex"].to(gpu)

    def run():
        return [t.cpu() for t in ext.decode_keypoint_v3(s, v, hn, 0.99, 5, 30000, None, None, 17, ext.SINGULAR_ZERO)]

    res = _each_front(run)
    _assert_all_equal(res)
    assert torch.equal(res["auto"][1], seg.argmax(1))


def test_workers_agree_on_images_of_more_than_256_tiles(synth, pkg, gpu):
    B, H, W, K, hn = 2, 720, 1280, 2, 16                      # 450 tiles: two rounds of the table pass
    d = synth.make_batch(B=B, H=H, W=W, K=K, fg=0.01, sigma=0.05, seed=8, mask_dtype=torch.uint8)
    _assert_all_equal(_both_layers(d["mask"], d["vertex"], hn, 0.99, gpu, max_num=100000))


def test_debug_option_rejects_unknown_values(pkg):
    L = capi.load()
    assert L.pvv_debug_option(1, -1) != 0 and L.pvv_debug_option(2, 0) != 0
    assert b"debug option" in L.pvv_last_error()
    capi.check(L.pvv_debug_option(1, 0))
