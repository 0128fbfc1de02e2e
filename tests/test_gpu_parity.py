"""GPU parity tests: the HIP path against the CPU oracle on identical seeded inputs.

Contract (BASELINE.json north_star): inlier counts / masks / indices bit-exact; keypoint means and
covariances within 1e-4 (absolute for means -- pixel units; absolute + relative for covariances).
Every layer is exercised both through the pybind11 module (what clean-pvnet imports) and through the
raw C ABI of include/pvnet_vote.h (tests/capi.py).
"""
import os

import numpy as np
import pytest
import torch

from tests import capi
from tests import tolerances as tol

pytestmark = pytest.mark.gpu

ATOL = 1e-4


def _np(t):
    return t.detach().cpu().numpy()


def _compacted(oracle, synth, cfg, seed=1234, **over):
    d = synth.make_batch(**{**synth.CONFIGS[cfg], **over, "B": 1}, seed=seed)
    fg, coords, direct = oracle.compact_v3(_np(d["mask"][0]), _np(d["vertex"][0]))
    return coords, direct


def _rand_idxs(tn, hn, vn, seed):
    rng = np.random.RandomState(seed)
    return rng.randint(0, tn, size=(hn, vn, 2)).astype(np.int32)


# --------------------------------------------------------------------------------------------------
# extension-module surface (reference layouts)
# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg,hn", [("cfg1", 64), ("cfg2", 512), ("cfg1", 37)])
def test_generate_hypothesis_bit_exact(oracle, synth, pkg, gpu, cfg, hn):
    from clean_pvnet_amd import ransac_voting as ext
    coords, direct = _compacted(oracle, synth, cfg)
    tn, vn, _ = direct.shape
    idxs = _rand_idxs(tn, hn, vn, 1)
    idxs[0, :, 1] = idxs[0, :, 0]                       # t0 == t1 -> degenerate -> (0,0)
    want = oracle.generate_hypothesis(direct, coords, idxs)
    got = ext.generate_hypothesis(torch.from_numpy(direct).to(gpu), torch.from_numpy(coords).to(gpu),
                                  torch.from_numpy(idxs).to(gpu))
    assert got.shape == (hn, vn, 2) and got.dtype == torch.float32
    np.testing.assert_array_equal(_np(got).view(np.uint32), want.view(np.uint32))
    assert (want[0] == 0).all()


@pytest.mark.parametrize("cfg,hn,thresh", [("cfg1", 64, 0.99), ("cfg2", 96, 0.99), ("cfg1", 5, 0.999)])
def test_voting_for_hypothesis_bit_exact(oracle, synth, pkg, gpu, cfg, hn, thresh):
    from clean_pvnet_amd import ransac_voting as ext
    coords, direct = _compacted(oracle, synth, cfg)
    tn, vn, _ = direct.shape
    hyp = oracle.generate_hypothesis(direct, coords, _rand_idxs(tn, hn, vn, 2))
    want = oracle.voting_for_hypothesis(direct, coords, hyp, np.zeros((hn, vn, tn), np.uint8), thresh)
    inl = torch.zeros(hn, vn, tn, dtype=torch.uint8, device=gpu)
    ret = ext.voting_for_hypothesis(torch.from_numpy(direct).to(gpu), torch.from_numpy(coords).to(gpu),
                                    torch.from_numpy(hyp).to(gpu), inl, thresh)
    assert ret is None                                   # in place, like the reference
    np.testing.assert_array_equal(_np(inl), want)
    assert want.sum() > 0
    # "only writes 1": pre-set bytes survive (ransac_voting_kernel.cu:124-125 never stores 0)
    inl2 = torch.full((hn, vn, tn), 7, dtype=torch.uint8, device=gpu)
    ext.voting_for_hypothesis(torch.from_numpy(direct).to(gpu), torch.from_numpy(coords).to(gpu),
                              torch.from_numpy(hyp).to(gpu), inl2, thresh)
    np.testing.assert_array_equal(_np(inl2), np.where(want == 1, 1, 7))


def test_vanishing_point_pair_bit_exact(oracle, synth, pkg, gpu):
    from clean_pvnet_amd import ransac_voting as ext
    coords, direct = _compacted(oracle, synth, "cfg1")
    tn, vn, _ = direct.shape
    hn = 48
    idxs = _rand_idxs(tn, hn, vn, 3)
    want_h = oracle.generate_hypothesis_vanishing_point(direct, coords, idxs)
    d, c = torch.from_numpy(direct).to(gpu), torch.from_numpy(coords).to(gpu)
    got_h = ext.generate_hypothesis_vanishing_point(d, c, torch.from_numpy(idxs).to(gpu))
    assert got_h.shape == (hn, vn, 3)
    np.testing.assert_array_equal(_np(got_h).view(np.uint32), want_h.view(np.uint32))
    want = oracle.voting_for_hypothesis_vanishing_point(direct, coords, want_h, np.zeros((hn, vn, tn), np.uint8), 0.99)
    inl = torch.zeros(hn, vn, tn, dtype=torch.uint8, device=gpu)
    ext.voting_for_hypothesis_vanishing_point(d, c, got_h, inl, 0.99)
    np.testing.assert_array_equal(_np(inl), want)
    assert want.sum() > 0


@pytest.mark.parametrize("cfg,hn,thresh", [("cfg1", 64, 0.99), ("cfg2", 512, 0.99), ("cfg2", 100, 0.999),
                                           ("cfg1", 1, 0.99), ("cfg1", 700, 0.99)])
def test_count_inliers_bit_exact(oracle, synth, pkg, gpu, cfg, hn, thresh):
    """The hot kernel on reference layouts: counts equal the oracle's, and equal
    voting_for_hypothesis + sum (ransac_voting_gpu.py:156-159) computed on the GPU."""
    from clean_pvnet_amd import ransac_voting as ext
    coords, direct = _compacted(oracle, synth, cfg)
    tn, vn, _ = direct.shape
    hyp = oracle.generate_hypothesis(direct, coords, _rand_idxs(tn, hn, vn, 4))
    want = oracle.count_inliers(direct, coords, hyp, thresh)
    d, c, h = (torch.from_numpy(x).to(gpu) for x in (direct, coords, hyp))
    got = ext.count_inliers(d, c, h, thresh)
    assert got.dtype == torch.int32 and got.shape == (hn, vn)
    np.testing.assert_array_equal(_np(got), want)
    inl = torch.zeros(hn, vn, tn, dtype=torch.uint8, device=gpu)
    ext.voting_for_hypothesis(d, c, h, inl, thresh)
    np.testing.assert_array_equal(_np(inl.sum(2, dtype=torch.int32)), want)
    assert want.max() > 0


def test_count_inliers_adversarial_near_threshold(oracle, pkg, gpu):
    """Pixels whose cosine to the hypothesis sits within a few ulps of the threshold, plus the
    norm < 1e-6 guards, zero directions, NaN/Inf -- decisions must still be bit-exact."""
    from clean_pvnet_amd import ransac_voting as ext
    rng = np.random.RandomState(7)
    tn, vn, hn = 4096, 3, 128
    thresh = 0.99
    coords = np.stack([rng.randint(0, 640, tn), rng.randint(0, 480, tn)], 1).astype(np.float32)
    hyp = (rng.rand(hn, vn, 2) * [640, 480]).astype(np.float32)
    hyp[0] = coords[0]                                   # hypothesis exactly on a pixel: norm2 = 0
    hyp[1] = coords[1] + np.float32(5e-7)                # norm2 below the 1e-6 guard
    direct = np.empty((tn, vn, 2), np.float32)
    ang0 = np.arccos(thresh)
    for ti in range(tn):
        h = hyp[rng.randint(hn), :, :]                   # aim each pixel at some hypothesis ...
        d = h - coords[ti]
        base = np.arctan2(d[:, 1], d[:, 0])
        off = ang0 * (1 + rng.uniform(-3e-6, 3e-6, vn)) * rng.choice([-1, 1], vn)   # ... at ~acos(thresh)
        scale = rng.choice([1.0, 1e-3, 37.5, 1e-7], vn, p=[0.7, 0.1, 0.15, 0.05])
        direct[ti, :, 0] = np.cos(base + off) * scale
        direct[ti, :, 1] = np.sin(base + off) * scale
    direct[5] = 0.0
    direct[6, 0] = [np.nan, 1.0]
    direct[7, 1] = [np.inf, 1.0]
    direct[8] = 1e-7
    want = oracle.count_inliers(direct, coords, hyp, thresh)
    got = ext.count_inliers(torch.from_numpy(direct).to(gpu), torch.from_numpy(coords).to(gpu),
                            torch.from_numpy(hyp).to(gpu), thresh)
    np.testing.assert_array_equal(_np(got), want)
    assert 0 < want.sum() < tn * vn * hn
    # and through the raw C ABI
    L = capi.load()
    d, c, h = (torch.from_numpy(x).to(gpu) for x in (direct, coords, hyp))
    cnt = torch.empty(hn, vn, dtype=torch.int32, device=gpu)
    capi.check(L.pvv_count_inliers(capi.ptr(d), capi.ptr(c), capi.ptr(h), capi.ptr(cnt), tn, vn, hn, thresh,
                                   capi.stream()))
    np.testing.assert_array_equal(_np(cnt), want)


def test_extension_rejects_bad_inputs(pkg, gpu):
    from clean_pvnet_amd import ransac_voting as ext
    d = torch.zeros(10, 3, 2, device=gpu)
    c = torch.zeros(10, 2, device=gpu)
    i = torch.zeros(4, 3, 2, dtype=torch.int32, device=gpu)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        ext.generate_hypothesis(d.cpu(), c, i)
    with pytest.raises(RuntimeError, match="contiguous"):
        ext.generate_hypothesis(d.transpose(0, 1).contiguous().transpose(0, 1), c, i)
    with pytest.raises(RuntimeError, match="dtype"):
        ext.generate_hypothesis(d, c, i.long())
    with pytest.raises(RuntimeError, match="idxs must be"):
        ext.generate_hypothesis(d, c, i[:, :2].contiguous())
    with pytest.raises(RuntimeError, match="inliers must be"):
        ext.voting_for_hypothesis(d, c, torch.zeros(4, 3, 2, device=gpu),
                                  torch.zeros(4, 3, 9, dtype=torch.uint8, device=gpu), 0.99)


# --------------------------------------------------------------------------------------------------
# ransac_voting_layer_v3
# --------------------------------------------------------------------------------------------------
def _v3_case(oracle, synth, gpu, cfg, B, planar=False, mask_dtype=torch.int64, hn=None, thresh=0.99,
             max_num=30000, seed=1234, **over):
    c = {**synth.CONFIGS[cfg], **over, "B": B}
    hn = hn or c["hn"]
    d = synth.make_batch(**c, seed=seed, mask_dtype=mask_dtype, planar=planar)
    mask, vertex = d["mask"], d["vertex"]
    tn = [int(x) for x in (mask != 0).sum((1, 2))]
    idxs = synth.make_idxs(tn, hn, c["K"], seed=seed)
    return d, mask, vertex, idxs, hn, tn


def _check_v3(oracle, got_out, got_win, got_tn, mask, vertex, idxs, hn, thresh, singular="reference", selection=None,
              max_num=30000):
    details = []
    want = oracle.ransac_voting_layer_v3(_np(mask), _np(vertex), hn, thresh, idxs=_np(idxs), details=details,
                                         singular=singular, selection=None if selection is None else _np(selection),
                                         max_num=max_num)
    want_tn = np.array([r["tn"] for r in details], np.int32)
    want_win = np.stack([r["win_counts"] if not r["skipped"] else np.zeros(vertex.shape[3], np.int32)
                         for r in details])
    np.testing.assert_array_equal(_np(got_tn), want_tn)
    np.testing.assert_array_equal(_np(got_win), want_win)          # inlier counts: bit-exact
    tol.assert_means_close(_np(got_out), want)
    return want, details


@pytest.mark.parametrize("cfg,B,planar,mask_dtype", [
    ("cfg1", 1, False, torch.int64),
    ("cfg1", 3, True, torch.uint8),
    ("cfg2", 1, False, torch.int64),
    ("cfg2", 2, True, torch.int64),
    ("cfg2", 2, False, torch.bool),
    ("cfg1", 2, False, torch.int32),
    ("cfg1", 2, False, torch.int16),
])
def test_v3_parity_pybind_and_cabi(oracle, synth, pkg, gpu, cfg, B, planar, mask_dtype):
    from clean_pvnet_amd import ransac_voting as ext
    d, mask, vertex, idxs, hn, tn = _v3_case(oracle, synth, gpu, cfg, B, planar, mask_dtype)
    m, v, i = mask.to(gpu), vertex.to(gpu), idxs.to(gpu)
    if planar:
        store = vertex.permute(0, 3, 4, 1, 2).reshape(B, -1, *vertex.shape[1:3]).contiguous().to(gpu)   # [B,2K,H,W]
        v = store.permute(0, 2, 3, 1).view(*vertex.shape)
        assert not v.is_contiguous()
    mm = m.view(torch.uint8) if m.dtype == torch.bool else m
    out, win, tnn, _ws = ext.ransac_voting_v3(mm, v, hn, 0.99, 5, 30000, i, None, 0, ext.SINGULAR_REFERENCE)
    want, _ = _check_v3(oracle, out, win, tnn, mask, vertex, idxs, hn, 0.99)
    # voting recovers the keypoints the field was built from (compute_vertex known answer)
    assert np.abs(want - _np(d["kpt_2d"])).max() < 6.0   # sigma=0.05 rad noise, keypoints up to ~100 px away
    out2, win2, tn2 = capi.v3(mm, v, hn, 0.99, idxs=i)
    np.testing.assert_array_equal(_np(out2), _np(out))
    np.testing.assert_array_equal(_np(win2), _np(win))


def test_v3_python_layer_matches_reference_signature_use(oracle, synth, pkg, gpu):
    """The three call forms of resnet18.py:71-75 through the drop-in import path."""
    from lib.csrc.ransac_voting.ransac_voting_gpu import (estimate_voting_distribution_with_mean,
                                                          ransac_voting_layer_v3)
    d, mask, vertex, idxs, hn, tn = _v3_case(oracle, synth, gpu, "cfg2", 2)
    m, v = mask.to(gpu), vertex.to(gpu)
    mean = ransac_voting_layer_v3(m, v, 512, inlier_thresh=0.99, idxs=idxs.to(gpu))
    assert mean.shape == (2, 9, 2) and mean.dtype == torch.float32 and mean.is_cuda
    _check_v3(oracle, mean, *_aux(oracle, mask, vertex, idxs, hn, gpu), mask, vertex, idxs, hn, 0.99)
    # free-running RNG forms: statistical parity = known-answer recovery
    torch.manual_seed(0)
    mean_r = ransac_voting_layer_v3(m, v, 512, inlier_thresh=0.99)
    assert np.abs(_np(mean_r) - _np(d["kpt_2d"])).max() < 6.0
    torch.manual_seed(0)
    np.testing.assert_array_equal(_np(ransac_voting_layer_v3(m, v, 512, inlier_thresh=0.99)), _np(mean_r))
    mean2, var = estimate_voting_distribution_with_mean(m, v, mean_r)
    assert mean2 is mean_r and var.shape == (2, 9, 2, 2)
    assert torch.isfinite(var).all() and (var[:, :, 0, 0] > 0).all() and (var[:, :, 1, 1] > 0).all()
    kp = ransac_voting_layer_v3(m, v, 128, inlier_thresh=0.99, max_num=100)       # resnet18.py:75
    assert np.abs(_np(kp) - _np(d["kpt_2d"])).max() < 6.0


def _aux(oracle, mask, vertex, idxs, hn, gpu):
    """win_counts / tn of the product for a v3 problem (through the pybind module)."""
    from clean_pvnet_amd import ransac_voting as ext
    _o, win, tn, _ws = ext.ransac_voting_v3(mask.to(gpu), vertex.to(gpu), hn, 0.99, 5, 30000, idxs.to(gpu), None, 0,
                                            ext.SINGULAR_REFERENCE)
    return win, tn


def test_v3_empty_tiny_and_multiclass_images(oracle, synth, pkg, gpu):
    """B>1 mixing: empty mask, < min_num pixels, a multi-class mask (byte values 2 count double in
    foreground_num, ransac_voting_gpu.py:125-129) and a normal image."""
    from clean_pvnet_amd import ransac_voting as ext
    d, mask, vertex, idxs, hn, tn = _v3_case(oracle, synth, gpu, "cfg1", 5)
    mask[0] = 0
    mask[1] = 0
    mask[1, 10, 10:14] = 1                               # 4 px < min_num=5 -> skipped
    mask[2] = 0
    mask[2, 20, 20:23] = 2                               # 3 px, byte-sum 6 >= 5 -> NOT skipped
    mask[3][mask[3] != 0] = 3
    tn = [int(x) for x in (mask != 0).sum((1, 2))]
    idxs = synth.make_idxs(tn, hn, 4, seed=99)
    out, win, tnn, _ws = ext.ransac_voting_v3(mask.to(gpu), vertex.to(gpu), hn, 0.99, 5, 30000, idxs.to(gpu), None, 0,
                                              ext.SINGULAR_ZERO)
    want, det = _check_v3(oracle, out, win, tnn, mask, vertex, idxs, hn, 0.99, singular="zero")
    assert (want[0] == 0).all() and (want[1] == 0).all()
    assert det[0]["skipped"] and det[1]["skipped"] and not det[2]["skipped"]
    assert _np(tnn).tolist() == [0, 0, 3, tn[3], tn[4]]


@pytest.mark.parametrize("singular", ["reference", "zero"])
def test_v3_singular_keypoint_policies(oracle, synth, pkg, gpu, singular):
    """A keypoint nobody votes for (count 0 -> winner (0,0) -> ATA = 0): reference policy turns the
    whole image into ATb (b_inv's identity fallback), 'zero' only that keypoint."""
    from clean_pvnet_amd.ransac_voting_gpu import ransac_voting_layer, ransac_voting_layer_v3
    d, mask, vertex, idxs, hn, tn = _v3_case(oracle, synth, gpu, "cfg1", 2)
    vertex[0, :, :, 1, :] = 0.0                          # zero directions: norm1 < 1e-6 everywhere
    m, v, i = mask.to(gpu), vertex.to(gpu), idxs.to(gpu)
    got = ransac_voting_layer_v3(m, v, hn, inlier_thresh=0.99, idxs=i, singular=singular)
    det = []
    want = oracle.ransac_voting_layer_v3(_np(mask), _np(vertex), hn, 0.99, idxs=_np(idxs), singular=singular, details=det)
    assert det[0]["singular"][1] == 1 and det[0]["win_counts"][1] == 0
    if singular == "reference":
        tol.assert_means_close(_np(got), want)       # ATb is huge: relative
        assert np.abs(want[0]).max() > 1e3
    else:
        tol.assert_means_close(_np(got), want)
        assert (want[0, 1] == 0).all() and np.abs(want[0, 0] - _np(d["kpt_2d"])[0, 0]).max() < 2
    v1 = ransac_voting_layer(m, v, hn, inlier_thresh=0.99, idxs=i)             # v1: whole image zeros
    assert (_np(v1)[0] == 0).all()
    tol.assert_means_close(_np(v1)[1], want[1])


def test_v3_subsample_with_injected_selection(oracle, synth, pkg, gpu):
    """foreground_num > max_num (ransac_voting_gpu.py:135-138) with the U(0,1) draws injected."""
    from clean_pvnet_amd import ransac_voting as ext
    c = {**synth.CONFIGS["cfg1"], "B": 2, "fg": 0.3}
    d = synth.make_batch(**c, seed=5)
    mask, vertex = d["mask"], d["vertex"]
    max_num = 1500
    g = torch.Generator().manual_seed(3)
    selection = torch.rand(mask.shape, generator=g)
    fg = mask.sum((1, 2)).float()
    assert (fg > max_num).all()
    kept = (mask != 0) & (selection < (torch.tensor(float(max_num)) / fg).view(-1, 1, 1))
    tn = [int(x) for x in kept.sum((1, 2))]
    hn = 64
    idxs = synth.make_idxs(tn, hn, c["K"], seed=5)
    out, win, tnn, _ws = ext.ransac_voting_v3(mask.to(gpu), vertex.to(gpu), hn, 0.99, 5, max_num, idxs.to(gpu),
                                              selection.to(gpu), 0, ext.SINGULAR_REFERENCE)
    assert _np(tnn).tolist() == tn
    _check_v3(oracle, out, win, tnn, mask, vertex, idxs, hn, 0.99, selection=selection, max_num=max_num)
    # device RNG instead of the injected draws: statistically the same subsample size
    out_r, _w, tn_r, _ws = ext.ransac_voting_v3(mask.to(gpu), vertex.to(gpu), hn, 0.99, 5, max_num, None, None, 12345,
                                                ext.SINGULAR_REFERENCE)
    assert np.abs(_np(tn_r) - max_num).max() < 6 * np.sqrt(max_num)
    assert np.abs(_np(out_r) - _np(d["kpt_2d"])).max() < 6.0


def test_v3_subsample_fused_into_compaction_at_480x640(oracle, synth, pkg, gpu):
    """480x640 = 150 tiles: the subsampling of P:135-138 happens inside k_compact_hyp (no k_tile_subsample launch), every
    block redoing the draws of the tiles before it.  ~12 % foreground (37 k pixels > max_num = 30000), injected draws:
    tn, every count and the means must equal the oracle's; the second image (2 % foreground) is not subsampled."""
    from clean_pvnet_amd import ransac_voting as ext
    d0 = synth.make_batch(B=1, H=480, W=640, K=2, fg=0.12, sigma=0.05, seed=77)
    d1 = synth.make_batch(B=1, H=480, W=640, K=2, fg=0.02, sigma=0.05, seed=78)
    mask, vertex = torch.cat([d0["mask"], d1["mask"]]), torch.cat([d0["vertex"], d1["vertex"]])
    selection = torch.rand(mask.shape, generator=torch.Generator().manual_seed(8))
    fg = mask.sum((1, 2)).float()
    assert float(fg[0]) > 30000 > float(fg[1])
    prob = (torch.tensor(30000.0) / fg).view(-1, 1, 1)
    kept = (mask != 0) & ((fg <= 30000).view(-1, 1, 1) | (selection < prob))
    tn = [int(x) for x in kept.sum((1, 2))]
    hn = 64
    idxs = synth.make_idxs(tn, hn, 2, seed=77)
    out, win, tnn, _ws = ext.ransac_voting_v3(mask.to(gpu), vertex.to(gpu), hn, 0.99, 5, 30000, idxs.to(gpu), selection.to(gpu), 0,
                                              ext.SINGULAR_REFERENCE)
    assert _np(tnn).tolist() == tn
    _check_v3(oracle, out, win, tnn, mask, vertex, idxs, hn, 0.99, selection=selection, max_num=30000)


def _counter_rng_draws(seed, image, n_pixels):
    """numpy restatement of the library's counter RNG for the subsample draws (vote_common.hpp rng_u32, stream 0, key
    (image, pixel)): splitmix64 finaliser, top 24 bits -> U[0,1)."""
    def mix64(z):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))
    with np.errstate(over="ignore"):
        k = mix64(np.array([seed], np.uint64) + np.uint64(0x9E3779B97F4A7C15) * np.uint64(1))
        x = (np.uint64(image) << np.uint64(32)) | np.arange(n_pixels, dtype=np.uint64)
        u = (mix64(k ^ x) >> np.uint64(32)).astype(np.uint32)
    return ((u >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24))


def test_fused_subsample_with_the_device_rng_equals_injected_draws_at_480x640(oracle, synth, pkg, gpu):
    """The path a dense image takes in production: 480x640 (150 tiles), ~12 % foreground (37 k pixels > max_num = 30000), no
    injected tensors -- k_compact_hyp subsamples itself, every tile block re-evaluating the counter RNG for the listed pixels of
    the tiles before it (round 4: as one sequence of 16-byte list groups).  The same call with the SAME draws injected as a
    selection tensor (numpy restatement of the generator) must give identical tn, winners and means; tn equals the count of
    draws below max_num / foreground_num; the second image (2 % foreground) is not subsampled; and a call that takes the other
    route (injected index pairs -> k_tile_subsample) keeps the same pixels."""
    from clean_pvnet_amd import ransac_voting as ext
    d0 = synth.make_batch(B=1, H=480, W=640, K=3, fg=0.12, sigma=0.05, seed=91)
    d1 = synth.make_batch(B=1, H=480, W=640, K=3, fg=0.02, sigma=0.05, seed=92)
    mask, vertex = torch.cat([d0["mask"], d1["mask"]]), torch.cat([d0["vertex"], d1["vertex"]])
    seed, hn = 424242, 128
    sel = torch.from_numpy(np.stack([_counter_rng_draws(seed, b, 480 * 640).reshape(480, 640) for b in range(2)]))
    fg = mask.sum((1, 2)).float()
    assert float(fg[0]) > 30000 > float(fg[1])
    prob = (torch.tensor(30000.0) / fg).view(-1, 1, 1)
    kept = (mask != 0) & ((fg <= 30000).view(-1, 1, 1) | (sel < prob))
    tn_want = [int(x) for x in kept.sum((1, 2))]
    m, v = mask.to(gpu), vertex.to(gpu)
    out, win, tn, _ws = ext.ransac_voting_v3(m, v, hn, 0.99, 5, 30000, None, None, seed, ext.SINGULAR_REFERENCE)
    out_i, win_i, tn_i, _ws = ext.ransac_voting_v3(m, v, hn, 0.99, 5, 30000, None, sel.to(gpu), seed, ext.SINGULAR_REFERENCE)
    assert _np(tn).tolist() == tn_want == _np(tn_i).tolist()
    assert torch.equal(out, out_i) and torch.equal(win, win_i)
    assert np.abs(_np(out) - np.concatenate([_np(d0["kpt_2d"]), _np(d1["kpt_2d"])])).max() < 10.0     # (128 hypotheses, sigma = 0.05: a sanity bound)
    idxs = synth.make_idxs(tn_want, hn, 3, seed=9)
    out_t, win_t, tn_t, _ws = ext.ransac_voting_v3(m, v, hn, 0.99, 5, 30000, idxs.to(gpu), None, seed, ext.SINGULAR_REFERENCE)
    assert _np(tn_t).tolist() == tn_want
    _check_v3(oracle, out_t, win_t, tn_t, mask, vertex, idxs, hn, 0.99, selection=sel, max_num=30000)


# --------------------------------------------------------------------------------------------------
# estimate_voting_distribution_with_mean
# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg,B,round_hyp,min_hyp", [("cfg1", 2, 64, 256), ("cfg2", 1, 256, 4096)])
def test_estimate_parity(oracle, synth, pkg, gpu, cfg, B, round_hyp, min_hyp):
    from clean_pvnet_amd.ransac_voting_gpu import estimate_voting_distribution_with_mean
    c = {**synth.CONFIGS[cfg], "B": B}
    d = synth.make_batch(**c, seed=77)
    mask, vertex = d["mask"], d["vertex"]
    if B > 1:
        mask[1][mask[1] != 0] = 2                        # class 2: `mask == 1` is empty -> skipped image
    tn = [int(x) for x in (mask == 1).sum((1, 2))]
    idxs = synth.make_idxs(tn, min_hyp, c["K"], seed=77)
    mean = d["kpt_2d"] + 0.25
    det = []
    _m, want = oracle.estimate_voting_distribution_with_mean(_np(mask), _np(vertex), _np(mean), round_hyp, min_hyp,
                                                             idxs=_np(idxs), details=det)
    m, v, mu, i = mask.to(gpu), vertex.to(gpu), mean.to(gpu), idxs.to(gpu)
    ret_mean, cov, hyp, ratio = estimate_voting_distribution_with_mean(m, v, mu, round_hyp, min_hyp, idxs=i,
                                                                      output_hyp=True)
    assert ret_mean is mu and cov.shape == (B, c["K"], 2, 2)
    tol.assert_cov_close(_np(cov), want)
    r0 = det[0]
    np.testing.assert_array_equal(_np(hyp[0]).view(np.uint32), r0["hypo_pts"].transpose(1, 0, 2).view(np.uint32))
    want_ratio = (r0["counts"].astype(np.float32) / np.float32(r0["tn"])).T
    np.testing.assert_array_equal(_np(ratio[0]), want_ratio)
    # raw C ABI: same covariance, bit-exact counts
    cov2, hyp2, counts2, tn2 = capi.estimate(m, v, mu, min_hyp, 0.99, idxs=i)
    np.testing.assert_array_equal(_np(cov2), _np(cov))
    np.testing.assert_array_equal(_np(counts2[0]), r0["counts"].T)
    assert _np(tn2).tolist() == tn


# --------------------------------------------------------------------------------------------------
# full-size, size-independent properties (BASELINE configs 3-5)
# --------------------------------------------------------------------------------------------------
def test_full_size_cfg3_properties_and_sampled_oracle(oracle, synth, pkg, gpu):
    """B=64, 480x640, K=9, 512 hypotheses on one GPU: (i) EVERY image against the oracle (winner counts, means),
    (ii) permuting the batch permutes the result, (iii) run-to-run determinism, (iv) fused counts ==
    legacy vote + sum for one image."""
    from clean_pvnet_amd import ransac_voting as ext
    c = synth.CONFIGS["cfg3"]
    d = synth.make_batch(**c, device=gpu)
    mask, vertex = d["mask"], d["vertex"]
    tn = [int(x) for x in (mask != 0).sum((1, 2)).cpu()]
    idxs = synth.make_idxs(tn, c["hn"], c["K"]).to(gpu)
    out, win, tnn, _ws = ext.ransac_voting_v3(mask, vertex, c["hn"], 0.99, 5, 30000, idxs, None, 0, ext.SINGULAR_REFERENCE)
    assert _np(tnn).tolist() == tn
    assert np.abs(_np(out) - _np(d["kpt_2d"])).max() < 6.0
    for bi in range(c["B"]):
        det = []
        want = oracle.ransac_voting_layer_v3(_np(mask[bi:bi + 1]), _np(vertex[bi:bi + 1]), c["hn"], 0.99,
                                             idxs=_np(idxs[bi:bi + 1]), details=det)
        np.testing.assert_array_equal(_np(win[bi]), det[0]["win_counts"])
        tol.assert_means_close(_np(out[bi:bi + 1]), want)
    perm = torch.randperm(c["B"], generator=torch.Generator().manual_seed(1)).to(gpu)
    out_p, win_p, _t, _w = ext.ransac_voting_v3(mask[perm], vertex[perm], c["hn"], 0.99, 5, 30000, idxs[perm], None, 0,
                                                ext.SINGULAR_REFERENCE)
    np.testing.assert_array_equal(_np(out_p), _np(out[perm]))
    np.testing.assert_array_equal(_np(win_p), _np(win[perm]))
    out_2, _w2, _t2, _ws2 = ext.ransac_voting_v3(mask, vertex, c["hn"], 0.99, 5, 30000, idxs, None, 0, ext.SINGULAR_REFERENCE)
    np.testing.assert_array_equal(_np(out_2), _np(out))


@pytest.mark.parametrize("cfg,B", [("cfg4", 4), ("cfg5", 2)])
def test_stress_configs_every_image_all_counts(oracle, synth, pkg, gpu, cfg, B):
    """cfg4 (sparse/occluded, 1024 hyps, outlier pixels) and cfg5 (540x720, K=17, 2048 hyps, tn capped by
    max_num=30000 -> subsampling with injected draws) at reduced batch, full image size: EVERY image's tn, winner
    counts and means against the oracle, and all K*hn inlier counts (through the estimate entry of the C ABI, whose
    foreground `mask == 1` coincides with v3's on these 0/1 masks)."""
    from clean_pvnet_amd import ransac_voting as ext
    c = {**synth.CONFIGS[cfg], "B": B}
    d = synth.make_batch(**c, seed=4321)
    mask, vertex = d["mask"], d["vertex"]
    selection = torch.rand(mask.shape, generator=torch.Generator().manual_seed(8))
    fg = mask.sum((1, 2)).float()
    keep = (mask != 0) & ((fg <= 30000).view(-1, 1, 1) | (selection < (torch.tensor(30000.0) / fg).view(-1, 1, 1)))
    tn = [int(x) for x in keep.sum((1, 2))]
    if cfg == "cfg5":
        assert (fg > 30000).all()
    idxs = synth.make_idxs(tn, c["hn"], c["K"], seed=4321)
    out, win, tnn, _ws = ext.ransac_voting_v3(mask.to(gpu), vertex.to(gpu), c["hn"], 0.99, 5, 30000, idxs.to(gpu),
                                              selection.to(gpu), 0, ext.SINGULAR_REFERENCE)
    assert _np(tnn).tolist() == tn
    _want, det = _check_v3(oracle, out, win, tnn, mask, vertex, idxs, c["hn"], 0.99, selection=selection)
    cov, hyp, counts, tn2 = capi.estimate(mask.to(gpu), vertex.to(gpu), out, c["hn"], 0.99, idxs=idxs.to(gpu),
                                          selection=selection.to(gpu))
    assert _np(tn2).tolist() == tn
    for bi in range(B):
        np.testing.assert_array_equal(_np(counts[bi]), det[bi]["counts"].T)          # all K*hn counts, bit-exact
        np.testing.assert_array_equal(_np(hyp[bi]), det[bi]["hypo_pts"].transpose(1, 0, 2))


@pytest.mark.parametrize("planar", [False, True])
def test_estimate_and_un_pnp_at_480x640_with_4096_hypotheses(oracle, synth, pkg, gpu, planar):
    """The un_pnp path at the size the network runs it (resnet18.py:71-72): 480x640, K=9, B=4, v3 with 512 hypotheses
    and the estimate with 16 x 256 = 4096, injected index pairs -- every count, the covariances, and the fused
    decode_keypoint(un_pnp=True) pass against the oracle."""
    from clean_pvnet_amd.decode import decode_keypoint
    from clean_pvnet_amd.ransac_voting_gpu import estimate_voting_distribution_with_mean, ransac_voting_layer_v3
    B, K, hn, hn_est = 4, 9, 512, 4096
    c = {**synth.CONFIGS["cfg2"], "B": B}
    d = synth.make_batch(**c, seed=555, planar=planar)
    mask, vertex = d["mask"], d["vertex"]
    tn = [int(x) for x in (mask == 1).sum((1, 2))]
    idxs = synth.make_idxs(tn, hn, K, seed=555)
    idxs_est = synth.make_idxs(tn, hn_est, K, seed=556)
    m, v = mask.to(gpu), vertex.to(gpu)
    mean = ransac_voting_layer_v3(m, v, hn, inlier_thresh=0.99, idxs=idxs.to(gpu))
    want_mean = oracle.ransac_voting_layer_v3(_np(mask), _np(vertex), hn, 0.99, idxs=_np(idxs))
    tol.assert_means_close(_np(mean), want_mean)
    _mm, cov, hyp, ratio = estimate_voting_distribution_with_mean(m, v, mean, idxs=idxs_est.to(gpu), output_hyp=True)
    det = []
    _m2, want_cov = oracle.estimate_voting_distribution_with_mean(_np(mask), _np(vertex), _np(mean), idxs=_np(idxs_est), details=det)
    tol.assert_cov_close(_np(cov), want_cov)                         # same mean on both sides
    for bi in range(B):
        want_ratio = (det[bi]["counts"].astype(np.float32) / np.float32(det[bi]["tn"])).T
        np.testing.assert_array_equal(_np(ratio[bi]), want_ratio)    # all 4096 x 9 counts of every image
        np.testing.assert_array_equal(_np(hyp[bi]), det[bi]["hypo_pts"].transpose(1, 0, 2))
    # the same through Resnet18.decode_keypoint's mirror, one fused pass (two-class seg built around the mask)
    H, W = c["H"], c["W"]
    x = torch.empty(B, 2 + 2 * K, H, W)
    x[:, 0] = 1.0
    x[:, 1] = torch.where(mask != 0, torch.tensor(4.0), torch.tensor(-4.0))
    x[:, 2:] = vertex.permute(0, 3, 4, 1, 2).reshape(B, 2 * K, H, W)
    x = x.to(gpu)
    o = decode_keypoint({"seg": x[:, :2], "vertex": x[:, 2:]}, un_pnp=True, idxs=idxs.to(gpu), idxs_est=idxs_est.to(gpu))
    assert torch.equal(o["mask"], m)
    np.testing.assert_array_equal(_np(o["kpt_2d"]), _np(mean))       # same kernels, same draws: bit-identical
    np.testing.assert_array_equal(_np(o["var"]), _np(cov))


def test_decode_keypoint_beyond_1024_images_and_sharding_invariance(synth, pkg, gpu):
    """decode_keypoint cuts batches beyond 1024 images into several launches (it used to raise), and -- given
    seed= / first_image= -- a batch decoded in shards equals the batch decoded at once (device RNG)."""
    from clean_pvnet_amd.decode import decode_keypoint
    B, H, W, K = 1030, 24, 32, 3
    d = synth.make_batch(B=8, H=H, W=W, K=K, fg=0.3, sigma=0.03, seed=77, planar=True)
    rep = (B + 7) // 8
    mask = d["mask"].repeat(rep, 1, 1)[:B]
    vertex_store = d["vertex"].permute(0, 3, 4, 1, 2).reshape(8, 2 * K, H, W).repeat(rep, 1, 1, 1)[:B]
    x = torch.empty(B, 2 + 2 * K, H, W)
    x[:, 0] = 1.0
    x[:, 1] = torch.where(mask != 0, torch.tensor(4.0), torch.tensor(-4.0))
    x[:, 2:] = vertex_store
    x = x.to(gpu)
    for un_pnp in (False, True):
        whole = decode_keypoint({"seg": x[:, :2], "vertex": x[:, 2:]}, un_pnp=un_pnp, seed=99)
        assert whole["kpt_2d"].shape == (B, K, 2) and whole["mask"].shape == (B, H, W)
        a = decode_keypoint({"seg": x[:500, :2], "vertex": x[:500, 2:]}, un_pnp=un_pnp, seed=99, first_image=0)
        b = decode_keypoint({"seg": x[500:, :2], "vertex": x[500:, 2:]}, un_pnp=un_pnp, seed=99, first_image=500)
        assert torch.equal(torch.cat([a["kpt_2d"], b["kpt_2d"]]), whole["kpt_2d"])
        assert torch.equal(torch.cat([a["mask"], b["mask"]]), whole["mask"])
        if un_pnp:
            assert whole["var"].shape == (B, K, 2, 2)
            assert torch.equal(torch.cat([a["var"], b["var"]]), whole["var"])
        assert float((whole["kpt_2d"][:8].cpu() - d["kpt_2d"]).abs().max()) < 4.0


@pytest.mark.parametrize("max_num,byte_mask", [(30000, False), (5000, False), (2000, False), (30000, True), (48960, True),
                                               (30001, True)])
def test_device_rng_draws_replayed_through_the_oracle(oracle, synth, pkg, gpu, max_num, byte_mask):
    """The hypothesis blocks of k_compact_hyp with the DEVICE RNG (no injected index pairs -- the path production
    runs): pvv_problem.d_draws_out reports the pixel every draw resolved to; mapping those pixels to rows of the
    oracle's compacted list and injecting them reproduces hypotheses, all counts and means exactly.  max_num = 5000
    subsamples inside k_compact_hyp (max_num >= 1/16 of the image: the index pairs then come from rejection sampling over
    the survivors), max_num = 2000 through k_tile_subsample (the lists are rewritten first).  byte_mask: a 0/255 uint8
    mask -- foreground_num sums the byte VALUES (P:126), so 9216 pixels are subsampled to ~118 with probability 0.013:
    too small for rejection sampling, the hypothesis blocks list the survivors instead.  max_num = 48960 with the byte mask:
    survival probability 1/48, just above the 1/64 where the survivor list takes over -- every index needs ~48 tries and one
    in 200 more than 256 (the try counter used to wrap at 256, those hypotheses silently became (0,0))."""
    import ctypes
    c = {**synth.CONFIGS["cfg2"], "B": 2, "H": 240, "W": 320, "fg": 0.12}
    d = synth.make_batch(**c, seed=31)
    if byte_mask:
        d["mask"] = (d["mask"] * 255).to(torch.uint8)
    mask, vertex = d["mask"].to(gpu), d["vertex"].to(gpu)
    B, H, W, K, hn = 2, 240, 320, c["K"], 256
    sel = torch.rand(B, H, W, generator=torch.Generator().manual_seed(5))          # injected: the oracle needs the same draws
    if max_num == 30001:
        # an injected selection may keep ANY number of pixels.  Draws of zero keep every one of the 9216 foreground
        # pixels although the survival probability is 0.013 (< 1/64: the survivor-list path) -- more than the list's 8192
        # entries: the hypothesis blocks must fall back to rejection sampling over ALL survivors, not sample a truncated list
        sel = torch.zeros(B, H, W)
    L = capi.load()
    draws = torch.full((B, K, hn, 2), -7, dtype=torch.int32, device=gpu)
    p = capi.problem(mask, vertex, hn, 0.99, max_num=max_num, seed=2024, draws_out=draws, cap=H * W)
    n = L.pvv_workspace_bytes(ctypes.byref(p))
    ws = torch.empty(n, dtype=torch.uint8, device=gpu)
    out = torch.empty(B, K, 2, device=gpu)
    win = torch.empty(B, K, dtype=torch.int32, device=gpu)
    tn = torch.empty(B, dtype=torch.int32, device=gpu)
    capi.check(L.pvv_ransac_voting_v3(ctypes.byref(p), capi.ptr(mask), capi.ptr(vertex), None, capi.ptr(sel.to(gpu)), capi.ptr(ws), n,
                                      capi.ptr(out), capi.ptr(win), capi.ptr(tn), capi.stream()))
    torch.cuda.synchronize()
    dr = _np(draws)
    assert (dr >= 0).all()
    # pixel -> row of the oracle's compacted (and subsampled) list
    idxs = np.zeros((B, hn, K, 2), np.int32)
    for bi in range(B):
        fg, coords, direct = oracle.compact_v3(_np(mask[bi]), _np(vertex[bi]), max_num, _np(sel[bi]))
        assert coords.shape[0] == int(tn[bi]) and (fg > max_num) == (max_num < 30000 or byte_mask)
        assert not byte_mask or (60 < coords.shape[0] < 200 if max_num == 30000 else
                                 (coords.shape[0] > 8192 if max_num == 30001 else 120 < coords.shape[0] < 280))
        row = -np.ones(H * W, np.int64)
        row[(coords[:, 1] * W + coords[:, 0]).astype(np.int64)] = np.arange(coords.shape[0])
        r = row[dr[bi]]                                            # [K,hn,2]
        assert (r >= 0).all(), "a draw resolved to a pixel that did not survive the subsample"
        idxs[bi] = r.transpose(1, 0, 2)
        # the draws are spread over the whole list (uniform index pairs, not stuck on a tile)
        assert len(np.unique(r)) > 0.5 * min(coords.shape[0], hn * K) or (byte_mask and max_num != 30001)
        if max_num == 30001:
            assert r.max() > 8192                                  # rows beyond the survivor list's capacity are drawn too
    _check_v3(oracle, out, win, tn, mask, vertex, torch.from_numpy(idxs), hn, 0.99, selection=sel, max_num=max_num)


def test_fused_un_pnp_equals_the_two_calls_under_one_seed(synth, pkg, gpu):
    """Device RNG: the v3 layer and the estimate draw from different streams of one seed (they used to share
    their first draws), and the fused un_pnp pass draws exactly what the two calls draw."""
    from clean_pvnet_amd import ransac_voting as ext
    B, H, W, K, hn, hn_est = 3, 96, 128, 5, 64, 160
    d = synth.make_batch(B=B, H=H, W=W, K=K, fg=0.08, sigma=0.05, seed=91, planar=True)
    x = torch.empty(B, 2 + 2 * K, H, W)
    x[:, 0] = 1.0
    x[:, 1] = torch.where(d["mask"] != 0, torch.tensor(4.0), torch.tensor(-4.0))
    x[:, 2:] = d["vertex"].permute(0, 3, 4, 1, 2).reshape(B, 2 * K, H, W)
    x = x.to(gpu)
    seg, vertex = x[:, :2], x[:, 2:].permute(0, 2, 3, 1).view(B, H, W, K, 2)
    kpt2, mask2, win2, tn2 = ext.decode_keypoint_v3(seg, vertex, hn, 0.99, 5, 30000, None, None, 4711, ext.SINGULAR_REFERENCE)
    cov2, hyp2, _c, _t, w2 = ext.estimate_voting_distribution(mask2, vertex, kpt2, hn_est, 0.99, 5, 30000, None, None, 4711, True)
    kpt, mask, cov, w, win, tnn = ext.decode_keypoint_un_pnp(seg, vertex, hn, hn_est, 0.99, 5, 30000, None, None, None, 4711,
                                                             ext.SINGULAR_REFERENCE)
    for a, b in ((mask, mask2), (kpt, kpt2), (cov, cov2), (w, w2), (win, win2), (tnn, tn2)):
        assert torch.equal(a, b)
    # counter-based generator: a longer run of the estimate extends a shorter one
    _cv, hyp_s, _cc, _tt, _ww = ext.estimate_voting_distribution(mask2, vertex, kpt2, hn, 0.99, 5, 30000, None, None, 4711, True)
    assert torch.equal(hyp_s, hyp2[:, :, :hn])


# --------------------------------------------------------------------------------------------------
# shapes / thresholds / non-finite inputs
# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("K,hn,thresh", [
    (1, 64, 0.99),        # single keypoint
    (17, 100, 0.99),      # hn not a multiple of 64 (R=2 tile with padded lanes)
    (4, 130, 0.999),      # reference default threshold (kappa = 22.3), R=4
    (4, 700, 0.99),       # two hypothesis tiles of 512, second one ragged
    (3, 64, 0.3),         # threshold outside the fast path's range -> exact kernel
    (3, 64, 0.99999),     # ditto, above
])
def test_v3_parity_shapes_and_thresholds(oracle, synth, pkg, gpu, K, hn, thresh):
    from clean_pvnet_amd import ransac_voting as ext
    c = {**synth.CONFIGS["cfg1"], "B": 2, "K": K}
    d = synth.make_batch(**c, seed=31 + K)
    mask, vertex = d["mask"], d["vertex"]
    tn = [int(x) for x in (mask != 0).sum((1, 2))]
    idxs = synth.make_idxs(tn, hn, K, seed=31 + K)
    out, win, tnn, _ws = ext.ransac_voting_v3(mask.to(gpu), vertex.to(gpu), hn, thresh, 5, 30000, idxs.to(gpu), None, 0,
                                              ext.SINGULAR_ZERO)
    det = []
    want = oracle.ransac_voting_layer_v3(_np(mask), _np(vertex), hn, thresh, idxs=_np(idxs), details=det, singular="zero")
    np.testing.assert_array_equal(_np(win), np.stack([r["win_counts"] for r in det]))
    tol.assert_means_close(_np(out), want)
    # every one of the hn*K counts, not just the winners
    m, v = mask.to(gpu), vertex.to(gpu)
    mean = torch.zeros(2, K, 2, device=gpu)
    if hn % 1 == 0:
        cov, hyp, counts, tn2 = capi.estimate(m, v, mean, hn, thresh, idxs=idxs.to(gpu))
        det2 = []
        oracle.estimate_voting_distribution_with_mean(_np(mask), _np(vertex), _np(mean), hn, hn, inlier_thresh=thresh,
                                                      idxs=_np(idxs), details=det2)
        for bi in range(2):
            np.testing.assert_array_equal(_np(counts[bi]), det2[bi]["counts"].T)


def test_v3_nonfinite_and_degenerate_vertex_values(oracle, synth, pkg, gpu):
    """NaN / Inf / zero / huge direction vectors on foreground pixels: hypotheses built from them are NaN, Inf
    or astronomically far (exact-loop fallback of the fast kernel); counts must still be bit-exact."""
    from clean_pvnet_amd import ransac_voting as ext
    c = {**synth.CONFIGS["cfg1"], "B": 2}
    d = synth.make_batch(**c, seed=41)
    mask, vertex = d["mask"], d["vertex"]
    ys, xs = np.nonzero(_np(mask[0]))
    rng = np.random.RandomState(0)
    pick = rng.choice(len(ys), 60, replace=False)
    vals = [np.nan, np.inf, -np.inf, 0.0, 1e30, 1e-30, 3e19, 1e-7]
    for j, pi in enumerate(pick):
        vertex[0, ys[pi], xs[pi], j % 4, j % 2] = vals[j % len(vals)]
    # two nearly parallel directions -> a hypothesis ~1e9 px away; exactly parallel -> degenerate (0,0)
    ys1, xs1 = np.nonzero(_np(mask[1]))
    vertex[1, ys1[0], xs1[0], 0] = torch.tensor([1.0, 0.0])
    j = int(np.argmax(ys1 != ys1[0]))                                     # first pixel of the next row
    vertex[1, ys1[j], xs1[j], 0] = torch.tensor([1.0, 2e-6])
    tn = [int(x) for x in (mask != 0).sum((1, 2))]
    hn = 256
    idxs = synth.make_idxs(tn, hn, 4, seed=41)
    idxs[0, :60, :, 0] = torch.from_numpy(np.searchsorted(np.flatnonzero(_np(mask[0]).ravel()),
                                                          ys[pick] * mask.shape[2] + xs[pick]).astype(np.int32))[:, None]
    idxs[1, 0, 0] = torch.tensor([0, j], dtype=torch.int32)               # the near-parallel pair
    mean = torch.zeros(2, 4, 2)
    det = []
    with np.errstate(all="ignore"):
        oracle.estimate_voting_distribution_with_mean(_np(mask), _np(vertex), _np(mean), hn, hn, idxs=_np(idxs), details=det)
    cov, hyp, counts, tnn = capi.estimate(mask.to(gpu), vertex.to(gpu), mean.to(gpu), hn, 0.99, idxs=idxs.to(gpu))
    for bi in range(2):
        got_h, want_h = _np(hyp[bi]), det[bi]["hypo_pts"].transpose(1, 0, 2)
        np.testing.assert_array_equal(got_h, want_h)                       # NaN == NaN here; x86 and gfx950 differ in the
        fin = np.isfinite(want_h)                                          # default NaN's sign bit only
        np.testing.assert_array_equal(got_h[fin].view(np.uint32), want_h[fin].view(np.uint32))
        np.testing.assert_array_equal(_np(counts[bi]), det[bi]["counts"].T)
    assert not np.isfinite(det[0]["hypo_pts"]).all()                        # the case really contains NaN/Inf hypotheses
    assert np.abs(det[1]["hypo_pts"][np.isfinite(det[1]["hypo_pts"])]).max() > 1e5


def test_estimate_subsample_and_multiclass(oracle, synth, pkg, gpu):
    """estimate with foreground > max_num (fresh count after the subsample, P:219-223) and a mask holding classes
    {0,1,2}: only `== 1` pixels vote."""
    from clean_pvnet_amd.ransac_voting_gpu import estimate_voting_distribution_with_mean
    c = {**synth.CONFIGS["cfg1"], "B": 2, "fg": 0.3}
    d = synth.make_batch(**c, seed=51)
    mask, vertex = d["mask"], d["vertex"]
    mask[1, :, :64][mask[1, :, :64] != 0] = 2
    max_num = 1000
    selection = torch.rand(mask.shape, generator=torch.Generator().manual_seed(4))
    fg = (mask == 1).sum((1, 2)).float()
    kept = (mask == 1) & ((fg <= max_num).view(-1, 1, 1) | (selection < (torch.tensor(float(max_num)) / fg).view(-1, 1, 1)))
    tn = [int(x) for x in kept.sum((1, 2))]
    idxs = synth.make_idxs(tn, 128, 4, seed=51)
    mean = d["kpt_2d"].clone()
    _m, want = oracle.estimate_voting_distribution_with_mean(_np(mask), _np(vertex), _np(mean), 64, 128, max_num=max_num,
                                                             idxs=_np(idxs), selection=_np(selection))
    _m2, cov = estimate_voting_distribution_with_mean(mask.to(gpu), vertex.to(gpu), mean.to(gpu), 64, 128, max_num=max_num,
                                                      idxs=idxs.to(gpu), selection=selection.to(gpu))
    tol.assert_cov_close(_np(cov), want)


# --------------------------------------------------------------------------------------------------
# decode_keypoint (resnet18.py:65-76): argmax fused into the mask scan
# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("C", [2, 3])
def test_decode_keypoint_fused_argmax_equals_unfused(oracle, synth, pkg, gpu, C):
    from clean_pvnet_amd import ransac_voting as ext
    from clean_pvnet_amd.decode import decode_keypoint
    c = {**synth.CONFIGS["cfg1"], "B": 3, "K": 9}
    d = synth.make_batch(**c, seed=61, planar=True)
    B, H, W, K = 3, c["H"], c["W"], 9
    # one network output tensor [B, C + 2K, H, W]; seg / vertex are channel slices of it (resnet18.py:93-94)
    x = torch.empty(B, C + 2 * K, H, W)
    g = torch.Generator().manual_seed(1)
    x[:, :C] = torch.randn(B, C, H, W, generator=g) * 0.1
    fg = d["mask"] != 0
    x[:, 1][fg] += 3.0                                              # class 1 wins on the object
    if C == 3:
        x[:, 2, :, : W // 4][fg[:, :, : W // 4]] += 6.0             # class 2 wins on its left quarter
    x[0, 0, 5, 7] = float("nan")                                    # NaN logit: torch.argmax returns its index
    x[1, 1, 6, 8] = float("nan")
    x[2, :C, 9, 9] = 1.25                                           # tie -> first index
    x[:, C:] = d["vertex"].permute(0, 3, 4, 1, 2).reshape(B, 2 * K, H, W)
    x = x.to(gpu)
    seg, ver = x[:, :C], x[:, C:]
    mask_ref = torch.argmax(seg, 1)
    vertex = ver.permute(0, 2, 3, 1).view(B, H, W, K, 2)
    tn = [int(v) for v in ((mask_ref & 0xFF) != 0).sum((1, 2)).cpu()]
    hn = 128
    idxs = synth.make_idxs(tn, hn, K, seed=61).to(gpu)
    out_ref, win_ref, tn_ref, _ws = ext.ransac_voting_v3(mask_ref, vertex, hn, 0.99, 5, 30000, idxs, None, 0, ext.SINGULAR_REFERENCE)
    out, mask, win, tnn = ext.decode_keypoint_v3(seg, vertex, hn, 0.99, 5, 30000, idxs, None, 0, ext.SINGULAR_REFERENCE)
    assert mask.dtype == torch.int64
    np.testing.assert_array_equal(_np(mask), _np(mask_ref))
    assert int(mask_ref[0, 5, 7]) == 0 and int(mask_ref[1, 6, 8]) == 1 and int(mask_ref[2, 9, 9]) == 0
    np.testing.assert_array_equal(_np(tnn), _np(tn_ref))
    np.testing.assert_array_equal(_np(win), _np(win_ref))
    np.testing.assert_array_equal(_np(out), _np(out_ref))           # same kernels downstream: bit-identical
    want = oracle.ransac_voting_layer_v3(_np(mask_ref), _np(vertex), hn, 0.99, idxs=_np(idxs))
    tol.assert_means_close(_np(out), want)
    # the dict-updating mirror of the reference method, both branches
    o = decode_keypoint({"seg": seg, "vertex": ver}, un_pnp=False)
    assert set(o) == {"seg", "vertex", "mask", "kpt_2d"} and o["kpt_2d"].shape == (B, K, 2)
    np.testing.assert_array_equal(_np(o["mask"]), _np(mask_ref))
    o = decode_keypoint({"seg": seg, "vertex": ver}, un_pnp=True)
    assert set(o) == {"seg", "vertex", "mask", "kpt_2d", "var"} and o["var"].shape == (B, K, 2, 2)
    if C == 2:
        assert np.abs(_np(o["kpt_2d"]) - _np(d["kpt_2d"])).max() < 6.0


@pytest.mark.parametrize("policy", ["reference", "zero"])
def test_decode_keypoint_un_pnp_one_pass_equals_the_two_calls(oracle, synth, pkg, gpu, policy):
    """resnet18.py:71-72 (v3 with 512 hypotheses, then the estimate with 4096) as ONE pass over a two-class seg: mask,
    keypoints, covariances and PnP weights bit-identical to decode_keypoint_v3 + estimate_voting_distribution on the
    same injected draws; keypoints and covariances within tolerance of the oracle; more than two classes are refused."""
    from clean_pvnet_amd import ransac_voting as ext
    from clean_pvnet_amd.decode import decode_keypoint
    B, H, W, K, hn, hn_est = 3, 96, 128, 5, 64, 160
    d = synth.make_batch(B=B, H=H, W=W, K=K, fg=0.08, sigma=0.05, seed=91, planar=True)
    x = torch.empty(B, 2 + 2 * K, H, W)
    g = torch.Generator().manual_seed(2)
    x[:, :2] = torch.randn(B, 2, H, W, generator=g) * 0.1
    x[:, 0] += 1.0
    x[:, 1][d["mask"] != 0] += 4.0
    x[2, 1] = -9.0                                                  # an image without foreground: skipped by both layers
    x[:, 2:] = d["vertex"].permute(0, 3, 4, 1, 2).reshape(B, 2 * K, H, W)
    x = x.to(gpu)
    seg, ver = x[:, :2], x[:, 2:]
    vertex = ver.permute(0, 2, 3, 1).view(B, H, W, K, 2)
    mask_ref = torch.argmax(seg, 1)
    tn = [int(v) for v in (mask_ref != 0).sum((1, 2)).cpu()]
    assert tn[2] == 0 and min(tn[:2]) > 100
    idxs = synth.make_idxs(tn, hn, K, seed=91).to(gpu)
    idxs_est = synth.make_idxs(tn, hn_est, K, seed=92).to(gpu)
    pol = ext.SINGULAR_REFERENCE if policy == "reference" else ext.SINGULAR_ZERO
    kpt2, mask2, win2, tn2 = ext.decode_keypoint_v3(seg, vertex, hn, 0.99, 5, 30000, idxs, None, 0, pol)
    cov2, _h, _c, _t, w2 = ext.estimate_voting_distribution(mask2, vertex, kpt2, hn_est, 0.99, 5, 30000, idxs_est, None, 0, False)
    kpt, mask, cov, w, win, tnn = ext.decode_keypoint_un_pnp(seg, vertex, hn, hn_est, 0.99, 5, 30000, idxs, idxs_est, None, 0, pol)
    for a, b in ((mask, mask2), (kpt, kpt2), (cov, cov2), (w, w2), (win, win2), (tnn, tn2)):
        assert torch.equal(a, b)
    want = oracle.ransac_voting_layer_v3(_np(mask_ref), _np(vertex), hn, 0.99, idxs=_np(idxs), singular=policy)
    tol.assert_means_close(_np(kpt), want)
    _m, want_cov = oracle.estimate_voting_distribution_with_mean(_np(mask_ref), _np(vertex), want, hn_est, hn_est,
                                                                 idxs=_np(idxs_est))
    np.testing.assert_allclose(_np(cov), want_cov, rtol=1e-3, atol=1e-3)    # mean differs by <= 1e-4 px from the oracle's
    # the dict mirror takes the one-pass route for a two-class seg (512 + 4096 hypotheses, device RNG)
    o = decode_keypoint({"seg": seg, "vertex": ver}, un_pnp=True, weights=True)
    assert set(o) == {"seg", "vertex", "mask", "kpt_2d", "var", "var_weights"}
    assert torch.equal(o["mask"], mask_ref) and o["var"].shape == (B, K, 2, 2) and o["var_weights"].shape == (B, K, 3)
    assert np.abs(_np(o["kpt_2d"][:2]) - _np(d["kpt_2d"][:2])).max() < 6.0
    seg3 = torch.cat([seg, seg[:, :1] - 5.0], 1)
    with pytest.raises(RuntimeError, match="seg_classes == 2"):
        ext.decode_keypoint_un_pnp(seg3, vertex, hn, hn_est, 0.99, 5, 30000, idxs, idxs_est, None, 0, pol)


# --------------------------------------------------------------------------------------------------
# the guard band of the default (bf16 matrix-core prefilter) kernel, through the batched path
# --------------------------------------------------------------------------------------------------
def _adversarial_image(oracle, synth, seed, thresh, hn, K=4):
    """A cfg1-like image whose foreground directions are re-aimed so that, for every pixel not used to generate a
    hypothesis, the cosine to some hypothesis sits within ~1e-6 (relative, in angle) of the threshold."""
    c = {**synth.CONFIGS["cfg1"], "B": 1, "K": K, "fg": 0.12}
    d = synth.make_batch(**c, seed=seed)
    mask, vertex = d["mask"], d["vertex"].clone()
    fg, coords, direct = oracle.compact_v3(_np(mask[0]), _np(vertex[0]))
    tn = coords.shape[0]
    rng = np.random.RandomState(seed)
    used = rng.choice(tn, 40, replace=False)                       # hypotheses come from these pixels only
    idxs = used[rng.randint(0, 40, size=(hn, K, 2))].astype(np.int32)
    hyp = oracle.generate_hypothesis(direct, coords, idxs)
    free = np.setdiff1d(np.arange(tn), used)
    ang0 = np.arccos(np.float64(np.float32(thresh)))
    ys, xs = coords[:, 1].astype(int), coords[:, 0].astype(int)
    for ti in free:
        for vi in range(K):
            h = hyp[rng.randint(hn), vi].astype(np.float64)
            dd = h - coords[ti]
            if not np.isfinite(dd).all() or np.abs(dd).max() < 1e-3:
                continue
            base = np.arctan2(dd[1], dd[0])
            off = ang0 * (1 + rng.uniform(-2e-6, 2e-6)) * rng.choice([-1, 1])
            s = rng.choice([1.0, 0.37, 12.5])
            vertex[0, ys[ti], xs[ti], vi, 0] = float(np.cos(base + off) * s)
            vertex[0, ys[ti], xs[ti], vi, 1] = float(np.sin(base + off) * s)
    return mask, vertex, torch.from_numpy(idxs)[None], tn


@pytest.mark.parametrize("thresh,seed", [(0.99, 3), (0.999, 4), (0.9, 5)])
def test_prefilter_guard_band_adversarial(oracle, synth, pkg, gpu, thresh, seed):
    hn = 256
    mask, vertex, idxs, tn = _adversarial_image(oracle, synth, seed, thresh, hn)
    mean = torch.zeros(1, 4, 2)
    det = []
    oracle.estimate_voting_distribution_with_mean(_np(mask), _np(vertex), _np(mean), hn, hn, inlier_thresh=thresh,
                                                  idxs=_np(idxs), details=det)
    cov, hyp, counts, tnn = capi.estimate(mask.to(gpu), vertex.to(gpu), mean.to(gpu), hn, thresh, idxs=idxs.to(gpu))
    want = det[0]["counts"].T
    np.testing.assert_array_equal(_np(counts[0]), want)              # all hn*K counts bit-exact
    assert int(_np(tnn)[0]) == tn and want.max() > 0
    # sanity of the construction: a sizeable share of evaluations really is within 1e-5 of the threshold
    fg, coords, direct = oracle.compact_v3(_np(mask[0]), _np(vertex[0]))
    hy = det[0]["hypo_pts"]
    dd = hy[:, None, :, :] - coords[None, :, None, :]                # [hn,tn,K,2]
    cosv = (dd * direct[None]).sum(-1) / (np.linalg.norm(dd, axis=-1) * np.linalg.norm(direct, axis=-1)[None] + 1e-30)
    near = np.abs(cosv - thresh) < 1e-5
    assert near.sum() > 0.5 * (tn - 40) * 4


def test_count_kernel_variants_agree(synth, pkg, gpu):
    """pvv_problem.count_kernel: the matrix-core prefilter (AUTO) and the reference's own arithmetic for every
    evaluation (EXACT) give identical counts, winners and means on the same seeded problem (v3 and the estimate)."""
    from clean_pvnet_amd import ransac_voting as ext
    c = {**synth.CONFIGS["cfg2"], "B": 2}
    d = synth.make_batch(**c, seed=77)
    tn = [int(x) for x in (d["mask"] != 0).sum((1, 2))]
    idxs = synth.make_idxs(tn, 512, 9, seed=77).to(gpu)
    m, v = d["mask"].to(gpu), d["vertex"].to(gpu)
    res = {}
    for name, k in (("auto", ext.COUNT_AUTO), ("exact", ext.COUNT_EXACT)):
        out, win, tnn, ws = ext.ransac_voting_v3(m, v, 512, 0.99, 5, 30000, idxs, None, 0, ext.SINGULAR_REFERENCE,
                                                 count_kernel=k)
        cov, hyp, counts, t2, wts = ext.estimate_voting_distribution(m, v, out, 512, 0.99, 5, 30000, idxs, None, 0, True,
                                                                     count_kernel=k)
        res[name] = (out.cpu(), win.cpu(), counts.cpu(), cov.cpu())
    for a_, b_ in zip(res["auto"], res["exact"]):
        assert torch.equal(a_, b_)
    assert int(res["exact"][2].sum()) > 0


def test_uncertainty_pnp_weights_match_evaluator_host_loop(oracle, synth, pkg, gpu):
    """SURVEY 8(f) rank 3, first half: inv(sqrtm(var)) -> (wxx,wxy,wyy), the loop of evaluators/linemod/pvnet.py:118-128,
    fused into the covariance kernel and as a standalone tensor function; oracle = numpy + scipy.linalg.sqrtm."""
    import scipy.linalg
    from clean_pvnet_amd.ransac_voting_gpu import estimate_voting_distribution_with_mean, uncertainty_pnp_weights
    c = {**synth.CONFIGS["cfg1"], "B": 3, "K": 5}
    d = synth.make_batch(**c, seed=91)
    mask, vertex = d["mask"], d["vertex"]
    mask[2] = 0                                                          # skipped image: cov = mean mean^T (rank 1)
    tn = [int(x) for x in (mask == 1).sum((1, 2))]
    idxs = synth.make_idxs(tn, 256, 5, seed=91)
    mean = d["kpt_2d"] + 0.1
    _m, cov, w = estimate_voting_distribution_with_mean(mask.to(gpu), vertex.to(gpu), mean.to(gpu), 64, 256,
                                                        idxs=idxs.to(gpu), return_weights=True)
    var = _np(cov)

    def host_loop(var_b):                                                # the reference's loop, verbatim semantics
        out = []
        for vi in range(var_b.shape[0]):
            if var_b[vi, 0, 0] < 1e-6 or np.sum(np.isnan(var_b)[vi]) > 0:
                out.append(np.zeros([2, 2], np.float32))
            else:
                out.append(np.linalg.inv(scipy.linalg.sqrtm(var_b[vi].astype(np.float64))))
        return np.asarray(out).reshape(-1, 4)[:, (0, 1, 3)]
    for bi in range(2):                                                  # well-conditioned images
        want = host_loop(var[bi])
        np.testing.assert_allclose(_np(w[bi]), want, rtol=2e-4, atol=1e-5)
        np.testing.assert_allclose(_np(uncertainty_pnp_weights(cov[bi])), want, rtol=2e-4, atol=1e-5)
    assert (_np(w[2]) == 0).all() or np.isfinite(_np(w[2])).all()        # rank-1 covariance: zeros (not positive definite) or finite
    bad = torch.tensor([[[1e-7, 0.0], [0.0, 1.0]], [[float("nan"), 0.0], [0.0, 1.0]], [[4.0, 0.0], [0.0, 9.0]]], device=gpu)
    np.testing.assert_allclose(_np(uncertainty_pnp_weights(bad)), [[0, 0, 0], [0, 0, 0], [0.5, 0, 1 / 3]], atol=1e-6)


def test_whole_call_can_be_captured_in_a_hip_graph(oracle, synth, pkg, gpu):
    """The library only enqueues kernels on the given stream (no allocation, no sync), so one voting call can be
    captured in a HIP graph and replayed; the replay must reproduce the eager result bit for bit."""
    from clean_pvnet_amd import ransac_voting as ext
    c = {**synth.CONFIGS["cfg2"], "B": 1}
    d = synth.make_batch(**c, seed=13)
    tn = [int(x) for x in (d["mask"] != 0).sum((1, 2))]
    idxs = synth.make_idxs(tn, 512, 9, seed=13).to(gpu)
    m, v = d["mask"].to(gpu), d["vertex"].to(gpu)
    eager, win_e, _tn, _ws = ext.ransac_voting_v3(m, v, 512, 0.99, 5, 30000, idxs, None, 0, ext.SINGULAR_REFERENCE)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            ext.ransac_voting_v3(m, v, 512, 0.99, 5, 30000, idxs, None, 0, ext.SINGULAR_REFERENCE)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out, win, tn_g, ws = ext.ransac_voting_v3(m, v, 512, 0.99, 5, 30000, idxs, None, 0, ext.SINGULAR_REFERENCE)
    for _ in range(3):
        out.zero_()
        g.replay()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(_np(out), _np(eager))
    np.testing.assert_array_equal(_np(win), _np(win_e))
    want = oracle.ransac_voting_layer_v3(_np(d["mask"]), _np(d["vertex"]), 512, 0.99, idxs=_np(idxs))
    tol.assert_means_close(_np(out), want)


def _soak_cases():
    """14 cases by default; PVV_SOAK_CASES=N draws the first N of the same sequence (a long run after kernel changes:
    `PVV_SOAK_CASES=300 pytest tests/test_gpu_parity.py -k soak`)."""
    rng = np.random.RandomState(2026)
    cases = []
    for i in range(int(os.environ.get("PVV_SOAK_CASES", "14"))):
        cases.append(dict(H=int(rng.choice([48, 96, 130, 200])), W=int(rng.choice([64, 100, 160, 257])),
                          K=int(rng.choice([1, 2, 5, 9, 17])), hn=int(rng.choice([32, 33, 64, 200, 512, 1000])),
                          thresh=float(rng.choice([0.6, 0.9, 0.99, 0.995, 0.999, 0.9999])),
                          fg=float(rng.choice([0.01, 0.05, 0.2, 0.6])), sigma=float(rng.choice([0.0, 0.02, 0.1, 0.5])),
                          B=int(rng.choice([1, 2, 3])), seed=1000 + i))
    kinds = ["int64", "uint8", "uint8x2", "int32", "bool", "int16"]          # the mask types the scan is instantiated for
    for i, c in enumerate(cases):
        c["mask_kind"] = kinds[i % len(kinds)]
    return cases


def _mask_as(mask, kind):
    if kind == "uint8x2":                       # two classes: v3 votes with every non-zero pixel, estimate with `== 1` only
        odd = (torch.arange(mask[0].numel()).view(mask[0].shape) % 3 == 0).to(torch.uint8)
        return (mask != 0).to(torch.uint8) * (1 + odd)
    return mask.to(getattr(torch, kind))


@pytest.mark.parametrize("case", _soak_cases(), ids=lambda c: "H%dW%dK%dhn%dT%g" % (c["H"], c["W"], c["K"], c["hn"], c["thresh"]))
def test_randomized_soak_all_counts_bit_exact(oracle, synth, pkg, gpu, case):
    """Random shapes / keypoint counts / hypothesis counts / thresholds / noise levels: every one of the B*K*hn inlier counts
    and the hypotheses themselves must equal the oracle's, and the v3 means must be within 1e-4."""
    from clean_pvnet_amd import ransac_voting as ext
    c = dict(case)
    thresh, hn, seed, kind = c.pop("thresh"), c.pop("hn"), c.pop("seed"), c.pop("mask_kind")
    d = synth.make_batch(**c, seed=seed)
    mask, vertex = _mask_as(d["mask"], kind), d["vertex"]
    tn = [int(x) for x in (mask != 0).sum((1, 2))]                # v3's foreground: non-zero
    tn1 = [int(x) for x in (mask == 1).sum((1, 2))]               # estimate's: == 1
    if min(tn1) < 5:
        pytest.skip("degenerate synthetic mask")
    if max(int(m.to(torch.uint8).long().sum()) for m in mask) > 30000:
        pytest.skip("subsampled case: covered by test_full_hd_frame_with_subsampling (needs injected selection draws)")
    idxs = synth.make_idxs(tn1, hn, c["K"], seed=seed)
    mean = torch.zeros(c["B"], c["K"], 2)
    det = []
    oracle.estimate_voting_distribution_with_mean(_np(mask), _np(vertex), _np(mean), hn, hn, inlier_thresh=thresh,
                                                  idxs=_np(idxs), details=det)
    cov, hyp, counts, tnn = capi.estimate(mask.to(gpu), vertex.to(gpu), mean.to(gpu), hn, thresh, idxs=idxs.to(gpu))
    assert tnn.tolist() == tn1
    for bi in range(c["B"]):
        np.testing.assert_array_equal(_np(counts[bi]), det[bi]["counts"].T)
        np.testing.assert_array_equal(_np(hyp[bi]), det[bi]["hypo_pts"].transpose(1, 0, 2))
    idxs = synth.make_idxs(tn, hn, c["K"], seed=seed + 1)
    out, win, t2, _ws = ext.ransac_voting_v3(mask.to(gpu), vertex.to(gpu), hn, thresh, 5, 30000, idxs.to(gpu), None, 0,
                                             ext.SINGULAR_ZERO)
    assert t2.tolist() == tn
    want = oracle.ransac_voting_layer_v3(_np(mask), _np(vertex), hn, thresh, idxs=_np(idxs), singular="zero")
    tol.assert_means_close(_np(out), want)


@pytest.mark.parametrize("dtype", [torch.int64, torch.uint8])
def test_strided_mask_views_equal_the_contiguous_mask(synth, pkg, gpu, dtype):
    """The mask is taken by strides (pvv_problem.mask_stride), not copied: a view with a padded row pitch, a channel slice of
    a [B,2,H,W] tensor and an every-other-column view give the results of their contiguous copies (same seed; the
    read-ahead scan is for contiguous masks only, these take the one-block-per-tile instantiation)."""
    from clean_pvnet_amd.ransac_voting_gpu import estimate_voting_distribution_with_mean, ransac_voting_layer_v3
    d = synth.make_batch(B=3, H=96, W=160, K=5, fg=0.08, sigma=0.03, seed=21, device=gpu, mask_dtype=dtype)
    mask, vertex = d["mask"], d["vertex"]
    B, H, W = mask.shape
    padded = torch.zeros(B, H, W + 7, dtype=dtype, device=gpu)
    padded[:, :, 3:3 + W] = mask
    two = torch.stack([1 - mask.clamp(max=1), mask], 1)                   # [B,2,H,W]
    wide = torch.zeros(B, H, 2 * W, dtype=dtype, device=gpu)
    wide[:, :, ::2] = mask
    views = [padded[:, :, 3:3 + W], two[:, 1], wide[:, :, ::2]]
    want = ransac_voting_layer_v3(mask, vertex, 128, inlier_thresh=0.99, seed=9)
    want_cov = estimate_voting_distribution_with_mean(mask, vertex, want, 64, 256, seed=9)[1]
    for v in views:
        assert not v.is_contiguous() and torch.equal(v, mask)
        got = ransac_voting_layer_v3(v, vertex, 128, inlier_thresh=0.99, seed=9)
        assert torch.equal(got, want)
        assert torch.equal(estimate_voting_distribution_with_mean(v, vertex, got, 64, 256, seed=9)[1], want_cov)


def test_calls_on_two_streams_are_independent(synth, pkg, gpu):
    """The library launches on the caller's stream and owns no global state: calls enqueued on two streams at once (each
    with its own workspace, as the shim allocates them) give the results of the same calls made one after the other."""
    from clean_pvnet_amd.ransac_voting_gpu import estimate_voting_distribution_with_mean, ransac_voting_layer_v3
    da = synth.make_batch(**{**synth.CONFIGS["cfg2"], "B": 3}, seed=5, device=gpu)
    db = synth.make_batch(**{**synth.CONFIGS["cfg1"], "B": 7}, seed=6, device=gpu)
    ref_a = ransac_voting_layer_v3(da["mask"], da["vertex"], 512, inlier_thresh=0.99, seed=1)
    ref_b = ransac_voting_layer_v3(db["mask"], db["vertex"], 64, inlier_thresh=0.99, seed=2)
    ref_c = estimate_voting_distribution_with_mean(db["mask"], db["vertex"], ref_b, seed=3)[1]
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    outs_a, outs_b, outs_c = [], [], []
    for _ in range(6):
        with torch.cuda.stream(sa):
            outs_a.append(ransac_voting_layer_v3(da["mask"], da["vertex"], 512, inlier_thresh=0.99, seed=1))
        with torch.cuda.stream(sb):
            ob = ransac_voting_layer_v3(db["mask"], db["vertex"], 64, inlier_thresh=0.99, seed=2)
            outs_b.append(ob)
            outs_c.append(estimate_voting_distribution_with_mean(db["mask"], db["vertex"], ob, seed=3)[1])
    torch.cuda.synchronize()
    for o in outs_a:
        assert torch.equal(o, ref_a)
    for o in outs_b:
        assert torch.equal(o, ref_b)
    for o in outs_c:
        assert torch.equal(o, ref_c)


def test_stream_ring_pipelines_a_sequence_of_batches(synth, pkg, gpu):
    """clean_pvnet_amd.pipeline.StreamRing: consecutive batches on alternating streams, inputs produced on the current
    stream right before each call, results consumed on the current stream after join() -- equal to the serial calls."""
    from clean_pvnet_amd.pipeline import StreamRing
    from clean_pvnet_amd.ransac_voting_gpu import estimate_voting_distribution_with_mean, ransac_voting_layer_v3
    cfg = {**synth.CONFIGS["cfg2"], "B": 2}
    data = [synth.make_batch(**cfg, seed=300 + i, device=gpu) for i in range(5)]
    want = [ransac_voting_layer_v3(d["mask"], d["vertex"], 256, inlier_thresh=0.99, seed=40 + i) for i, d in enumerate(data)]
    want_cov = [estimate_voting_distribution_with_mean(d["mask"], d["vertex"], w, 64, 512, seed=40 + i)[1]
                for i, (d, w) in enumerate(zip(data, want))]
    torch.cuda.synchronize()
    ring = StreamRing(2)
    got, got_cov = [], []
    for i, d in enumerate(data):
        mask = d["mask"].clone()                     # an input produced on the current stream just before the call
        vertex = d["vertex"] * 1.0

        def both(m, v, seed):
            k = ransac_voting_layer_v3(m, v, 256, inlier_thresh=0.99, seed=seed)
            return k, estimate_voting_distribution_with_mean(m, v, k, 64, 512, seed=seed)[1]
        k, c = ring.run(both, mask, vertex, 40 + i)
        got.append(k)
        got_cov.append(c)
        del mask, vertex                             # the allocator must not hand the memory out before the side stream is done
    ring.join()
    total = sum(g.sum() for g in got)                # consumed on the current stream
    torch.cuda.synchronize()
    assert torch.isfinite(total)
    for g, w in zip(got, want):
        assert torch.equal(g, w)
    for g, w in zip(got_cov, want_cov):
        assert torch.equal(g, w)
    with pytest.raises(ValueError):
        StreamRing(0)


def test_foreground_sizes_around_tile_and_chunk_edges(oracle, pkg, gpu):
    """The count kernel cuts an image's tn compacted pixels into 512-pixel chunks and 16-pixel tiles that go round-robin
    to 4 waves, skipping empty tiles: tn just below / at / above every boundary (and one pixel over a chunk), all
    B*K*hn counts bit-exact, means within 1e-4."""
    from clean_pvnet_amd import ransac_voting as ext
    sizes = [5, 15, 16, 17, 31, 33, 63, 64, 65, 127, 128, 129, 255, 257, 511, 512, 513, 527, 529, 1023, 1024, 1025, 1537]
    B, H, W, K, hn = len(sizes), 40, 48, 3, 96
    g = torch.Generator().manual_seed(31)
    mask = torch.zeros(B, H, W, dtype=torch.int64)
    for i, n in enumerate(sizes):
        mask[i].view(-1)[torch.randperm(H * W, generator=g)[:n]] = 1
    vertex = torch.randn(B, H, W, K, 2, generator=g)
    # aim most directions at a common point so that counts are far from 0 and from tn
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    for k in range(K):
        d = torch.stack([10.0 + 9 * k - xs, 20.0 - ys], -1)
        d = d / d.norm(dim=-1, keepdim=True).clamp(min=1e-3)
        keep = torch.rand(B, H, W, generator=g) < 0.7
        vertex[:, :, :, k] = torch.where(keep[..., None], d + 0.03 * torch.randn(B, H, W, 2, generator=g), vertex[:, :, :, k])
    idxs = torch.zeros(B, hn, K, 2, dtype=torch.int32)
    for i, n in enumerate(sizes):
        idxs[i] = torch.randint(0, n, (hn, K, 2), generator=g, dtype=torch.int32)
    mean = torch.zeros(B, K, 2)
    det = []
    oracle.estimate_voting_distribution_with_mean(_np(mask), _np(vertex), _np(mean), hn, hn, inlier_thresh=0.99, idxs=_np(idxs),
                                                  details=det)
    cov, hyp, counts, tnn = capi.estimate(mask.to(gpu), vertex.to(gpu), mean.to(gpu), hn, 0.99, idxs=idxs.to(gpu))
    assert _np(tnn).tolist() == sizes
    for bi in range(B):
        np.testing.assert_array_equal(_np(counts[bi]), det[bi]["counts"].T, err_msg="tn = %d" % sizes[bi])
    out, win, _t, _ws = ext.ransac_voting_v3(mask.to(gpu), vertex.to(gpu), hn, 0.99, 5, 30000, idxs.to(gpu), None, 0, ext.SINGULAR_ZERO)
    want = oracle.ransac_voting_layer_v3(_np(mask), _np(vertex), hn, 0.99, idxs=_np(idxs), singular="zero")
    tol.assert_means_close(_np(out), want)


def test_batches_beyond_1024_images_are_split_by_the_python_layer(oracle, pkg, gpu):
    """The count kernels keep their work-item table in LDS (<= 1024 images per launch, PVV_E_ARG beyond); the Python
    layers split larger batches.  1030 tiny images, three of them checked against the oracle."""
    from clean_pvnet_amd.ransac_voting_gpu import estimate_voting_distribution_with_mean, ransac_voting_layer_v3
    B, H, W, K, hn = 1030, 16, 24, 2, 32
    g = torch.Generator().manual_seed(9)
    mask = (torch.rand(B, H, W, generator=g) < 0.4).to(torch.int64)
    mask[5] = 0                                                           # an empty image in the first chunk
    vertex = torch.randn(B, H, W, K, 2, generator=g)
    tn = [int(x) for x in (mask != 0).sum((1, 2))]
    idxs = torch.zeros(B, hn, K, 2, dtype=torch.int32)
    for i, t in enumerate(tn):
        if t > 0:
            idxs[i] = torch.randint(0, t, (hn, K, 2), generator=g, dtype=torch.int32)
    out = ransac_voting_layer_v3(mask.to(gpu), vertex.to(gpu), hn, inlier_thresh=0.9, idxs=idxs.to(gpu), singular="zero")
    assert out.shape == (B, K, 2)
    for bi in (0, 5, 1023, 1024, 1029):                                   # both sides of the chunk boundary
        want = oracle.ransac_voting_layer_v3(_np(mask[bi:bi + 1]), _np(vertex[bi:bi + 1]), hn, 0.9, idxs=_np(idxs[bi:bi + 1]),
                                             singular="zero")
        tol.assert_means_close(_np(out[bi:bi + 1]), want)
    mean, cov = estimate_voting_distribution_with_mean(mask.to(gpu), vertex.to(gpu), out, 32, 32, inlier_thresh=0.9,
                                                       idxs=idxs.to(gpu))
    assert cov.shape == (B, K, 2, 2) and bool(torch.isfinite(cov).all())


def test_device_rng_is_keyed_by_global_image_index(synth, pkg, gpu):
    """pvv_problem.first_image: a batch cut into several calls with the same seed draws the same hypothesis pairs AND
    the same subsampling numbers as one call (device RNG, nothing injected) -- the split the Python layer makes beyond
    1024 images is invisible in the results."""
    from clean_pvnet_amd import ransac_voting as ext
    d = synth.make_batch(B=6, H=96, W=128, K=3, fg=0.15, sigma=0.05, seed=77, device=gpu)
    mask, vertex = d["mask"], d["vertex"]
    for max_num in (30000, 400):                                          # 400 < tn: the selection draws matter too
        whole = ext.ransac_voting_v3(mask, vertex, 64, 0.99, 5, max_num, None, None, 4242, ext.SINGULAR_ZERO)
        parts = [ext.ransac_voting_v3(mask[lo:hi], vertex[lo:hi], 64, 0.99, 5, max_num, None, None, 4242,
                                      ext.SINGULAR_ZERO, lo) for lo, hi in ((0, 1), (1, 4), (4, 6))]
        for i in range(3):                                                # keypoints, winning counts, tn
            assert torch.equal(whole[i], torch.cat([q[i] for q in parts]))
        other = ext.ransac_voting_v3(mask[1:4], vertex[1:4], 64, 0.99, 5, max_num, None, None, 4242, ext.SINGULAR_ZERO)
        assert not torch.equal(other[1], parts[1][1])                     # first_image = 0 there: different draws


def test_python_layers_are_invariant_to_sharding_with_a_common_seed(synth, pkg, gpu):
    """seed= / first_image= of the drop-in layers: voting on the shards of a batch (what N GPUs do) with the common seed
    gives, image by image, exactly the result of one call on the whole batch -- device RNG, nothing injected."""
    from clean_pvnet_amd.ransac_voting_gpu import estimate_voting_distribution_with_mean, ransac_voting_layer_v3
    d = synth.make_batch(B=5, H=96, W=128, K=3, fg=0.12, sigma=0.05, seed=88, device=gpu)
    mask, vertex = d["mask"], d["vertex"]
    whole = ransac_voting_layer_v3(mask, vertex, 64, inlier_thresh=0.99, seed=2024)
    parts = [ransac_voting_layer_v3(mask[lo:hi], vertex[lo:hi], 64, inlier_thresh=0.99, seed=2024, first_image=lo)
             for lo, hi in ((0, 3), (3, 5))]
    assert torch.equal(whole, torch.cat(parts))
    assert not torch.equal(whole, ransac_voting_layer_v3(mask, vertex, 64, inlier_thresh=0.99, seed=2025))
    _m, cov = estimate_voting_distribution_with_mean(mask, vertex, whole, 32, 64, seed=7)
    covs = [estimate_voting_distribution_with_mean(mask[lo:hi], vertex[lo:hi], whole[lo:hi], 32, 64, seed=7, first_image=lo)[1]
            for lo, hi in ((0, 2), (2, 5))]
    assert torch.equal(cov, torch.cat(covs))


def test_full_hd_frame_with_subsampling(oracle, synth, pkg, gpu):
    """1080x1920 (1013 compaction tiles, coordinates up to 1919), ~2 % foreground = 41 k pixels > max_num -> subsampled to
    ~30 k with injected draws; K = 9, 512 hypotheses; counts bit-exact, means within 1e-4."""
    from clean_pvnet_amd import ransac_voting as ext
    c = dict(B=1, H=1080, W=1920, K=9, fg=0.02, sigma=0.05)
    d = synth.make_batch(**c, seed=2025)
    mask, vertex = d["mask"], d["vertex"]
    selection = torch.rand(mask.shape, generator=torch.Generator().manual_seed(6))
    fg = mask.sum((1, 2)).float()
    assert float(fg[0]) > 30000
    keep = (mask != 0) & (selection < (torch.tensor(30000.0) / fg).view(-1, 1, 1))
    tn = [int(x) for x in keep.sum((1, 2))]
    idxs = synth.make_idxs(tn, 512, 9, seed=2025)
    out, win, tnn, _ws = ext.ransac_voting_v3(mask.to(gpu), vertex.to(gpu), 512, 0.99, 5, 30000, idxs.to(gpu),
                                              selection.to(gpu), 0, ext.SINGULAR_REFERENCE)
    assert _np(tnn).tolist() == tn
    _check_v3(oracle, out, win, tnn, mask, vertex, idxs, 512, 0.99, selection=selection)


def test_float_masks_and_strided_slices(oracle, synth, pkg, gpu):
    """mask as float (converted like the reference: .byte() for v3, == 1 for estimate) and mask / vertex given as strided
    slices of larger tensors (general-stride path of the mask scan and of the gather)."""
    from clean_pvnet_amd.ransac_voting_gpu import estimate_voting_distribution_with_mean, ransac_voting_layer_v3
    c = {**synth.CONFIGS["cfg1"], "B": 2}
    d = synth.make_batch(**c, seed=71)
    mask, vertex = d["mask"], d["vertex"]
    H, W, K = c["H"], c["W"], c["K"]
    tn = [int(x) for x in (mask != 0).sum((1, 2))]
    idxs = synth.make_idxs(tn, 64, K, seed=71)
    want = oracle.ransac_voting_layer_v3(_np(mask), _np(vertex), 64, 0.99, idxs=_np(idxs))
    # (a) float mask with a fractional value that .byte() truncates to 0 (torch semantics of :125)
    fm = mask.float()
    fm[0, 0, 0] = 0.7
    got = ransac_voting_layer_v3(fm.to(gpu), vertex.to(gpu), 64, inlier_thresh=0.99, idxs=idxs.to(gpu))
    tol.assert_means_close(_np(got), want)
    # (b) every second row/column of a 2x larger int32 mask, vertex as a slice of a padded tensor
    big_m = torch.zeros(2, 2 * H, 2 * W, dtype=torch.int32, device=gpu)
    big_m[:, ::2, ::2] = mask.to(gpu).int()
    mview = big_m[:, ::2, ::2]
    big_v = torch.randn(2, H + 3, W + 5, K + 2, 2, device=gpu)
    big_v[:, 1:H + 1, 2:W + 2, 1:K + 1] = vertex.to(gpu)
    vview = big_v[:, 1:H + 1, 2:W + 2, 1:K + 1]
    assert not mview.is_contiguous() and not vview.is_contiguous()
    got = ransac_voting_layer_v3(mview, vview, 64, inlier_thresh=0.99, idxs=idxs.to(gpu))
    tol.assert_means_close(_np(got), want)
    # (c) estimate with a float mask: only entries == 1.0 are foreground
    fm2 = mask.float() * 1.0
    fm2[1][fm2[1] != 0] = 1.5
    mean = d["kpt_2d"].clone()
    ii = synth.make_idxs([tn[0], 0], 128, K, seed=72)
    _m, wantc = oracle.estimate_voting_distribution_with_mean(_np(fm2), _np(vertex), _np(mean), 64, 128, idxs=_np(ii))
    _m2, cov = estimate_voting_distribution_with_mean(fm2.to(gpu), vertex.to(gpu), mean.to(gpu), 64, 128, idxs=ii.to(gpu))
    tol.assert_cov_close(_np(cov), wantc)


@pytest.mark.parametrize("thresh", [0.99, 0.999, 0.9])
def test_refit_band_prefilter_equals_the_exact_revote_on_adversarial_images(oracle, synth, pkg, gpu, thresh):
    """k_select_refit<BAND> (B*K >= 128): the winner's re-vote goes through the second-level guard-band test and only
    in-band pixels through the exact sequence.  Images whose hypotheses all meet in one point and whose other pixels are
    aimed at the threshold cone of THAT point (relative angle 1 +- 2e-6) or straight at it: thousands of pixels per keypoint
    sit in or beside the band of the winner.  Means and winner counts must equal those of PVV_COUNT_EXACT (whose refit is
    the exact sequence for every pixel) bit for bit, and the oracle's within the contract."""
    from clean_pvnet_amd import ransac_voting as ext
    K, hn, nimg = 4, 64, 8
    masks, verts, idxl = [], [], []
    ang0 = np.arccos(np.float64(np.float32(thresh)))
    for s in range(nimg):
        c = {**synth.CONFIGS["cfg1"], "B": 1, "K": K, "fg": 0.12}
        d = synth.make_batch(**c, seed=300 + s)
        mask, vertex = d["mask"], d["vertex"].clone()
        fg, coords, direct = oracle.compact_v3(_np(mask[0]), _np(vertex[0]))
        tn = coords.shape[0]
        rng = np.random.RandomState(300 + s)
        kp = _np(d["kpt_2d"][0]).astype(np.float64)                    # [K,2]
        used = rng.choice(tn, 40, replace=False)
        ys, xs = coords[:, 1].astype(int), coords[:, 0].astype(int)
        for ti in range(tn):
            for vi in range(K):
                dd = kp[vi] - coords[ti]
                base = np.arctan2(dd[1], dd[0])
                if ti in used or rng.rand() < 0.5:
                    off = 0.0                                          # straight at the keypoint
                else:
                    off = ang0 * (1 + rng.uniform(-2e-6, 2e-6)) * rng.choice([-1, 1])
                sc = rng.choice([1.0, 0.37, 12.5])
                vertex[0, ys[ti], xs[ti], vi, 0] = float(np.cos(base + off) * sc)
                vertex[0, ys[ti], xs[ti], vi, 1] = float(np.sin(base + off) * sc)
        masks.append(mask); verts.append(vertex)
        idxl.append(torch.from_numpy(used[rng.randint(0, 40, size=(hn, K, 2))].astype(np.int32))[None])
    mask = torch.cat(masks * 4); vertex = torch.cat(verts * 4); idxs = torch.cat(idxl * 4)      # B = 32: B*K = 128
    m, v, i = mask.to(gpu), vertex.to(gpu), idxs.to(gpu)
    res = {}
    for name, k in (("auto", ext.COUNT_AUTO), ("exact", ext.COUNT_EXACT)):
        out, win, tnn, _ws = ext.ransac_voting_v3(m, v, hn, thresh, 5, 30000, i, None, 0, ext.SINGULAR_ZERO, count_kernel=k)
        res[name] = (out.cpu(), win.cpu(), tnn.cpu())
    for a_, b_ in zip(res["auto"], res["exact"]):
        assert torch.equal(a_, b_)
    det = []
    want = oracle.ransac_voting_layer_v3(_np(mask[:nimg]), _np(vertex[:nimg]), hn, thresh, idxs=_np(idxs[:nimg]), singular="zero",
                                         details=det)
    np.testing.assert_array_equal(_np(res["auto"][1][:nimg]), np.stack([r["win_counts"] for r in det]))
    tol.assert_means_close(_np(res["auto"][0][:nimg]), want)
    # the construction works: the winners count about half of the pixels, and many others sit within 1e-5 of the threshold
    tn0 = int(res["auto"][2][0])
    assert 0.3 * tn0 < int(res["auto"][1][0].min()) and int(res["auto"][1][0].max()) < 0.8 * tn0
