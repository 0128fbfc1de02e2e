"""GPU parity of the round-3 additions to the count pass (ABI v6):

  * staged counting with exact elimination (k_count_bf16<first> + k_lead + k_count_bf16<filter>, ransac_voting_layer_v3 only) against the full
    pass, against the reference's own arithmetic (PVV_COUNT_EXACT) and against the oracle -- winners, winner counts and
    means bit for bit / within the one tolerance, on the randomized soak cases and at the BASELINE configs' OWN batch
    sizes, where the count kernel's scheduling branches flip (VERDICT r2 #2a: cfg3 at B = 8/16/24/32, cfg4 at B = 32,
    cfg5 at B = 16, the estimate with 4096 hypotheses at B = 1/8/16 -- every image's tn, all K*hn counts, means);
  * pvv_problem.d_status (list truncated at cap, image skipped, image subsampled);
  * pvv_problem.ev_marks and the streaming-read probe (measurement aids: they must run and make sense).
"""
import os

import numpy as np
import pytest
import torch

from tests import capi
from tests import tolerances as tol
from tests.test_gpu_parity import _mask_as, _np, _soak_cases

pytestmark = pytest.mark.gpu


def _modes(ext):
    return (("auto", ext.COUNT_AUTO), ("staged", ext.COUNT_STAGED), ("full", ext.COUNT_FULL), ("exact", ext.COUNT_EXACT))


@pytest.mark.parametrize("case", _soak_cases(), ids=lambda c: "H%dW%dK%dhn%dT%g" % (c["H"], c["W"], c["K"], c["hn"], c["thresh"]))
def test_staged_count_returns_the_winners_of_the_exact_pass_soak(oracle, synth, pkg, gpu, case):
    """Random shapes / K / hn / thresholds / noise (the soak sequence; PVV_SOAK_CASES=300 for the long run): the staged
    path forced at every size (PVV_COUNT_STAGED) returns the same keypoints and winner counts as the full matrix-core pass
    and as PVV_COUNT_EXACT, bit for bit, and the oracle's winners."""
    from clean_pvnet_amd import ransac_voting as ext
    c = dict(case)
    thresh, hn, seed, kind = c.pop("thresh"), c.pop("hn"), c.pop("seed"), c.pop("mask_kind")
    d = synth.make_batch(**c, seed=seed)
    mask, vertex = _mask_as(d["mask"], kind), d["vertex"]
    tn = [int(x) for x in (mask != 0).sum((1, 2))]
    if max(int(m.to(torch.uint8).long().sum()) for m in mask) > 30000:
        pytest.skip("subsampled case (needs injected selection draws): covered at full size below")
    idxs = synth.make_idxs(tn, hn, c["K"], seed=seed + 1)
    m, v, i = mask.to(gpu), vertex.to(gpu), idxs.to(gpu)
    res = {}
    for name, k in _modes(ext):
        out, win, t2, _ws = ext.ransac_voting_v3(m, v, hn, thresh, 5, 30000, i, None, 0, ext.SINGULAR_ZERO, count_kernel=k)
        res[name] = (out.cpu(), win.cpu(), t2.cpu())
    for name in ("auto", "staged", "full"):
        for a_, b_ in zip(res[name], res["exact"]):
            assert torch.equal(a_, b_), name
    det = []
    want = oracle.ransac_voting_layer_v3(_np(mask), _np(vertex), hn, thresh, idxs=_np(idxs), singular="zero", details=det)
    want_win = np.stack([r["win_counts"] if not r["skipped"] else np.zeros(c["K"], np.int32) for r in det])
    np.testing.assert_array_equal(_np(res["staged"][1]), want_win)
    tol.assert_means_close(_np(res["staged"][0]), want)


def _full_size_case(synth, cfg, B, seed):
    c = {**synth.CONFIGS[cfg], "B": B}
    d = synth.make_batch(**c, seed=seed)
    mask, vertex = d["mask"], d["vertex"]
    selection = torch.rand(mask.shape, generator=torch.Generator().manual_seed(seed + 3))
    fg = mask.sum((1, 2)).float()
    keep = (mask != 0) & ((fg <= 30000).view(-1, 1, 1) | (selection < (torch.tensor(30000.0) / fg).view(-1, 1, 1)))
    tn = [int(x) for x in keep.sum((1, 2))]
    need_sel = bool((fg > 30000).any())
    return c, mask, vertex, (selection if need_sel else None), tn


@pytest.mark.parametrize("cfg,B", [("cfg3", 8), ("cfg3", 16), ("cfg3", 24), ("cfg3", 32), ("cfg4", 32), ("cfg5", 16)])
def test_full_size_configs_at_their_own_batch_sizes_every_count(oracle, synth, pkg, gpu, cfg, B):
    """VERDICT r2 #2a.  The count kernel sizes its work items, its grid and its one-generation fallback from B, hn and the
    tn[] it finds (count_bf16.hpp: gpi / htpi halving against target_items, nblk, 5 / 15 / 48 blocks per CU), and from
    round 3 on ransac_voting_layer_v3 may count in stages -- none of which a reduced batch crosses.  Here every BASELINE
    config runs at the batch size it names (and config 3 at the shard sizes in between): every image's tn, ALL K*hn
    inlier counts (full pass, through the estimate entry of the C ABI), and winners + means of the v3 layer in every
    count mode against the oracle."""
    from clean_pvnet_amd import ransac_voting as ext
    c, mask, vertex, selection, tn = _full_size_case(synth, cfg, B, seed=900 + B)
    hn, K = c["hn"], c["K"]
    idxs = synth.make_idxs(tn, hn, K, seed=901 + B)
    m, v, i = mask.to(gpu), vertex.to(gpu), idxs.to(gpu)
    s = None if selection is None else selection.to(gpu)
    det = []
    want = oracle.ransac_voting_layer_v3(_np(mask), _np(vertex), hn, 0.99, idxs=_np(idxs), details=det,
                                         selection=None if selection is None else _np(selection))
    want_win = np.stack([r["win_counts"] for r in det])
    assert [r["tn"] for r in det] == tn
    # all K*hn counts of the full matrix-core pass
    cov, hyp, counts, tn2 = capi.estimate(m, v, torch.zeros(B, K, 2, device=gpu), hn, 0.99, idxs=i, selection=s)
    assert _np(tn2).tolist() == tn
    for bi in range(B):
        np.testing.assert_array_equal(_np(counts[bi]), det[bi]["counts"].T)
        np.testing.assert_array_equal(_np(hyp[bi]), det[bi]["hypo_pts"].transpose(1, 0, 2))
    # the v3 layer in every count mode (exact only where it is affordable)
    for name, k in _modes(ext):
        if name == "exact" and cfg == "cfg5":
            continue
        out, win, t3, _ws = ext.ransac_voting_v3(m, v, hn, 0.99, 5, 30000, i, s, 0, ext.SINGULAR_REFERENCE, count_kernel=k)
        assert _np(t3).tolist() == tn, name
        np.testing.assert_array_equal(_np(win), want_win, err_msg=name)
        tol.assert_means_close(_np(out), want)


@pytest.mark.parametrize("B", [1, 8, 16])
def test_estimate_with_4096_hypotheses_at_batch_sizes(oracle, synth, pkg, gpu, B):
    """The un_pnp path's estimate (resnet18.py:72: 16 x 256 hypotheses) at 480x640 for B = 1 / 8 / 16: every count of every
    image and the covariances (the count kernel walks 8 hypothesis groups per item here)."""
    from clean_pvnet_amd.ransac_voting_gpu import estimate_voting_distribution_with_mean
    K, hn_est = 9, 4096
    c = {**synth.CONFIGS["cfg2"], "B": B}
    d = synth.make_batch(**c, seed=640 + B)
    mask, vertex = d["mask"], d["vertex"]
    tn = [int(x) for x in (mask == 1).sum((1, 2))]
    idxs = synth.make_idxs(tn, hn_est, K, seed=641 + B)
    mean = d["kpt_2d"].float()
    _mm, cov, hyp, ratio = estimate_voting_distribution_with_mean(mask.to(gpu), vertex.to(gpu), mean.to(gpu),
                                                                  idxs=idxs.to(gpu), output_hyp=True)
    det = []
    _m2, want_cov = oracle.estimate_voting_distribution_with_mean(_np(mask), _np(vertex), _np(mean), idxs=_np(idxs), details=det)
    tol.assert_cov_close(_np(cov), want_cov)
    for bi in range(B):
        want_ratio = (det[bi]["counts"].astype(np.float32) / np.float32(det[bi]["tn"])).T
        np.testing.assert_array_equal(_np(ratio[bi]), want_ratio)
        np.testing.assert_array_equal(_np(hyp[bi]), det[bi]["hypo_pts"].transpose(1, 0, 2))


def test_staged_count_with_ties_empty_images_and_few_chunks(oracle, synth, pkg, gpu):
    """Corner cases of the elimination: an image without foreground and one below min_num beside normal ones, an image of
    a single 512-pixel chunk (fewer than n_min chunks: counted completely by the first launch), duplicated index pairs (tied hypotheses: the FIRST index must win, P:160) and a keypoint nobody votes for
    (all counts 0: winner stays (0,0), P:162-167)."""
    from clean_pvnet_amd import ransac_voting as ext
    c = {**synth.CONFIGS["cfg2"], "B": 5, "H": 240, "W": 320, "fg": 0.1}
    d = synth.make_batch(**c, seed=77)
    mask, vertex = d["mask"].clone(), d["vertex"].clone()
    mask[1] = 0
    mask[2] = 0
    mask[2, 5, 5:8] = 1                                   # 3 px < min_num
    keep = torch.zeros_like(mask[3])
    keep[100:110, 100:140] = 1
    mask[3] = mask[3] * keep                              # <= 400 px: one chunk
    vertex[4, :, :, 2, :] = 0.0                           # keypoint 2 of image 4: zero directions, never an inlier
    # non-finite and astronomically large directions on a tenth of image 0's foreground (keypoints 0 and 1): hypotheses
    # drawn from them are NaN / inf / far away -- the count kernels send their groups down the exact loop, k_lead gives such
    # a leader no sure inliers
    ys, xs = torch.nonzero(mask[0], as_tuple=True)
    pick = torch.arange(0, ys.numel(), 10)
    vertex[0, ys[pick[0::3]], xs[pick[0::3]], 0, 0] = float("nan")
    vertex[0, ys[pick[1::3]], xs[pick[1::3]], 0, 1] = float("inf")
    vertex[0, ys[pick[2::3]], xs[pick[2::3]], 1, :] *= 1e25
    tn = [int(x) for x in (mask != 0).sum((1, 2))]
    assert tn[1] == 0 and tn[2] == 3 and 0 < tn[3] <= 512 and tn[0] > 2048
    hn, K = 256, c["K"]
    idxs = synth.make_idxs(tn, hn, K, seed=78)
    idxs[:, 128:] = idxs[:, :128]                         # every hypothesis twice: ties everywhere
    m, v, i = mask.to(gpu), vertex.to(gpu), idxs.to(gpu)
    det = []
    want = oracle.ransac_voting_layer_v3(_np(mask), _np(vertex), hn, 0.99, idxs=_np(idxs), singular="zero", details=det)
    res = {}
    for name, k in _modes(ext):
        out, win, t2, _ws = ext.ransac_voting_v3(m, v, hn, 0.99, 5, 30000, i, None, 0, ext.SINGULAR_ZERO, count_kernel=k)
        res[name] = (out.cpu(), win.cpu(), t2.cpu())
    for name in ("auto", "staged", "full"):
        for a_, b_ in zip(res[name], res["exact"]):
            assert torch.equal(a_, b_), name
    want_win = np.stack([r["win_counts"] if not r["skipped"] else np.zeros(K, np.int32) for r in det])
    np.testing.assert_array_equal(_np(res["staged"][1]), want_win)
    tol.assert_means_close(_np(res["staged"][0]), want)
    assert (det[0]["win_idx"] < 128).all()                # the first of two tied hypotheses
    assert want_win[4, 2] == 0 and (want[4, 2] == 0).all()


@pytest.mark.parametrize("outlier,sigma", [((0.45, 0.6), 0.05), ((0.0, 0.0), 0.5), ((0.7, 0.8), 0.1)])
def test_staged_count_when_the_bound_barely_bites(oracle, synth, pkg, gpu, outlier, sigma):
    """Fields on which the winner explains half of the pixels or fewer (40-80 % outlier pixels, or 0.5 rad of direction
    noise) on masks large enough to be staged (12+ chunks): after a quarter of the pixels the elimination drops little or
    nothing -- the regime where a wrong bound would drop the winner.  Staged == full == exact, bit for bit, and the oracle's
    winners."""
    from clean_pvnet_amd import ransac_voting as ext
    c = {**synth.CONFIGS["cfg2"], "B": 3, "outlier": outlier, "sigma": sigma}
    d = synth.make_batch(**c, seed=321)
    mask, vertex = d["mask"], d["vertex"]
    tn = [int(x) for x in (mask != 0).sum((1, 2))]
    assert min(tn) >= 8 * 512
    hn, K = 512, c["K"]
    idxs = synth.make_idxs(tn, hn, K, seed=322)
    m, v, i = mask.to(gpu), vertex.to(gpu), idxs.to(gpu)
    res = {}
    for name, k in _modes(ext):
        out, win, t2, _ws = ext.ransac_voting_v3(m, v, hn, 0.99, 5, 30000, i, None, 0, ext.SINGULAR_ZERO, count_kernel=k)
        res[name] = (out.cpu(), win.cpu(), t2.cpu())
    for name in ("auto", "staged", "full"):
        for a_, b_ in zip(res[name], res["exact"]):
            assert torch.equal(a_, b_), name
    det = []
    want = oracle.ransac_voting_layer_v3(_np(mask), _np(vertex), hn, 0.99, idxs=_np(idxs), singular="zero", details=det)
    np.testing.assert_array_equal(_np(res["staged"][1]), np.stack([r["win_counts"] for r in det]))
    tol.assert_means_close(_np(res["staged"][0]), want)
    ratio = np.stack([r["win_counts"] for r in det]).max() / max(tn)
    assert ratio < 0.75                                  # the winners really are weak here


def test_staged_count_beyond_64_images(oracle, synth, pkg, gpu):
    """More images than one wavefront scans at a time (the item table and the staged / small bitmap of the count kernels
    are built 64 images per round): 70 images, some of them below 8 chunks (counted completely by the first launch) between
    staged ones, one empty."""
    from clean_pvnet_amd import ransac_voting as ext
    B, H, W, K, hn = 70, 96, 96, 2, 96
    d = synth.make_batch(B=B, H=H, W=W, K=K, fg=0.6, sigma=0.05, seed=4100)
    mask, vertex = d["mask"].clone(), d["vertex"]
    for bi in range(0, B, 7):                              # every seventh image small: <= 7 chunks
        keep = torch.zeros_like(mask[bi])
        keep[20:70, 20:80] = 1
        mask[bi] = mask[bi] * keep
    mask[33] = 0
    tn = [int(x) for x in (mask != 0).sum((1, 2))]
    assert max(tn) > 8 * 512 and 0 < tn[0] <= 7 * 512 and tn[33] == 0
    idxs = synth.make_idxs(tn, hn, K, seed=4101)
    m, v, i = mask.to(gpu), vertex.to(gpu), idxs.to(gpu)
    res = {}
    for name, k in _modes(ext):
        out, win, t2, _ws = ext.ransac_voting_v3(m, v, hn, 0.99, 5, 30000, i, None, 0, ext.SINGULAR_ZERO, count_kernel=k)
        res[name] = (out.cpu(), win.cpu(), t2.cpu())
    for name in ("auto", "staged", "full"):
        for a_, b_ in zip(res[name], res["exact"]):
            assert torch.equal(a_, b_), name
    det = []
    want = oracle.ransac_voting_layer_v3(_np(mask), _np(vertex), hn, 0.99, idxs=_np(idxs), singular="zero", details=det)
    want_win = np.stack([r["win_counts"] if not r["skipped"] else np.zeros(K, np.int32) for r in det])
    np.testing.assert_array_equal(_np(res["staged"][1]), want_win)
    tol.assert_means_close(_np(res["staged"][0]), want)


def test_status_flags_truncated_skipped_subsampled(synth, pkg, gpu):
    """pvv_problem.d_status (ABI v6): a list longer than the `cap` rows reserved is cut -- now
    reported --, an image below min_num is SKIPPED, an image above max_num SUBSAMPLED."""
    from clean_pvnet_amd import ransac_voting as ext
    c = {**synth.CONFIGS["cfg2"], "B": 3, "H": 240, "W": 320, "fg": 0.12}
    d = synth.make_batch(**c, seed=5)
    mask = d["mask"].clone()
    mask[1] = 0
    m, v = mask.to(gpu), d["vertex"].to(gpu)
    fg = int((mask[0] != 0).sum())
    assert fg > 5000
    st = torch.full((3,), -1, dtype=torch.int32, device=gpu)
    out, win, tn, _ws = ext.ransac_voting_v3(m, v, 64, 0.99, 5, 30000, None, None, 11, ext.SINGULAR_REFERENCE, status=st, cap=1000)
    assert st.tolist() == [ext.STATUS_TRUNCATED, ext.STATUS_SKIPPED, ext.STATUS_TRUNCATED] and tn.tolist() == [1000, 0, 1000]
    st.fill_(-1)
    ext.ransac_voting_v3(m, v, 64, 0.99, 5, 30000, None, None, 11, ext.SINGULAR_REFERENCE, status=st)
    assert st.tolist() == [0, ext.STATUS_SKIPPED, 0]
    # subsampled inside k_compact_hyp (max_num >= 1/16 of the image) and through k_tile_subsample (below)
    for max_num in (5000, 2000):
        st.fill_(-1)
        _o, _w, tn, _ws = ext.ransac_voting_v3(m, v, 64, 0.99, 5, max_num, None, None, 11, ext.SINGULAR_REFERENCE, status=st)
        assert st.tolist() == [ext.STATUS_SUBSAMPLED, ext.STATUS_SKIPPED, ext.STATUS_SUBSAMPLED], (max_num, st.tolist())
        assert abs(int(tn[0]) - max_num) < 6 * max_num ** 0.5
        # a cap well below the subsample: cut and reported
        st.fill_(-1)
        _o, _w, tn, _ws = ext.ransac_voting_v3(m, v, 64, 0.99, 5, max_num, None, None, 11, ext.SINGULAR_REFERENCE, status=st,
                                               cap=max_num // 2)
        assert st.tolist()[0] == ext.STATUS_SUBSAMPLED | ext.STATUS_TRUNCATED and int(tn[0]) == max_num // 2


def test_stage_marks_and_stream_probe(synth, pkg, gpu):
    from clean_pvnet_amd import ransac_voting as ext
    c = {**synth.CONFIGS["cfg2"], "B": 4}
    d = synth.make_batch(**c, seed=3, device=gpu)
    for k, staged in ((ext.COUNT_FULL, False), (ext.COUNT_STAGED, True)):
        ms = ext.stage_ms_in_pipeline([d["mask"]], [d["vertex"]], 512, 0.99, 5, 30000, 1, 6, k)
        assert len(ms) == 6 and all(len(r) == 7 for r in ms)
        assert all(min(r[i] for r in ms[2:]) < 5 for i in range(5)), ms   # scan, compact, count pass, select, finalize: ms (a
        for r in ms[2:]:                                             # box hiccup once put 80 ms into ONE repetition: bound the best)
            assert all(x > 0 for x in r[:5]), r
            if staged:
                assert 0 < r[5] < r[2] and 0 < r[6] < r[2], r    # first count launch and k_lead inside the count pass
            else:
                assert r[5] < 0 and r[6] < 0, r                  # not recorded
        ms2 = ext.stage_ms_in_pipeline([d["mask"]], [d["vertex"]], 512, 0.99, 5, 30000, 1, 4, k, False)
        assert all(r[5] < 0 and r[6] < 0 and r[2] > 0 for r in ms2) and min(r[2] for r in ms2) < 5   # no records inside the count pass on request
    buf = torch.empty(256 << 20, dtype=torch.uint8, device=gpu).random_(0, 255)
    sink = torch.zeros(1, dtype=torch.int32, device=gpu)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        ext.stream_read_probe(buf, sink)
    a.record()
    for _ in range(10):
        ext.stream_read_probe(buf, sink)
    b.record()
    torch.cuda.synchronize()
    gbs = 10 * buf.numel() / (a.elapsed_time(b) * 1e-3) / 1e9
    assert 500 < gbs < 20000, gbs                                # an MI355X streams a few TB/s
    with pytest.raises(RuntimeError, match="zero_counts"):
        _o, _w, _t, ws = ext.ransac_voting_v3(d["mask"], d["vertex"], 512, 0.99, 5, 30000, None, None, 1, ext.SINGULAR_REFERENCE,
                                              count_kernel=ext.COUNT_STAGED)
        ext.rerun_count_kernel(d["mask"], d["vertex"], 512, 0.99, 5, 30000, ws, False, ext.COUNT_STAGED)


def test_auto_stages_dense_crops_by_their_real_work(synth, pkg, gpu):
    """T-LESS votes on detector crops (256x256, a third of the pixels foreground): 16 of them are as many evaluations
    as 59 LINEMOD frames behind a B*K*hn*H*W of 4.8e9 -- below the 2e10 of the frame-calibrated proxy.  Once a call of the shape has
    reported its images' tn, AUTO decides on K * hn * sum(tn): the first call runs the full pass (no data: proxy), the following ones
    are staged; the same crops with 2 % foreground are not; results equal the full pass either way."""
    from clean_pvnet_amd import ransac_voting as ext
    hn = 512
    for fg, want_staged in ((0.35, True), (0.02, False)):
        d = synth.make_batch(B=16, H=256, W=256, K=9, fg=fg, sigma=0.05, seed=300 + int(fg * 100), device=gpu)
        d2 = synth.make_batch(B=3, H=128, W=256, K=9, fg=fg, sigma=0.05, seed=400, device=gpu)       # another shape (H): resets the hint
        ext.ransac_voting_v3(d2["mask"], d2["vertex"], hn, 0.99, 5, 30000, None, None, 5, ext.SINGULAR_REFERENCE, count_kernel=ext.COUNT_AUTO)

        def v3(k):
            return ext.ransac_voting_v3(d["mask"], d["vertex"], hn, 0.99, 5, 30000, None, None, 5, ext.SINGULAR_REFERENCE, count_kernel=k)

        ms = ext.stage_ms_in_pipeline([d["mask"]], [d["vertex"]], hn, 0.99, 5, 30000, 5, 1, ext.COUNT_AUTO)
        assert ms[0][5] < 0, ms                                       # first call of the shape: proxy 4.8e9 < 2e10, full pass
        torch.cuda.synchronize()
        valid, mean, thr = ext.stage_hint(d["mask"], d["vertex"], hn)
        assert valid and mean > 0.98
        assert (thr <= 0.995) == want_staged, thr                     # 2.0 = too little work
        ms = ext.stage_ms_in_pipeline([d["mask"]], [d["vertex"]], hn, 0.99, 5, 30000, 5, 3, ext.COUNT_AUTO)
        assert all((r[5] > 0) == want_staged for r in ms), ms
        ref, got = v3(ext.COUNT_FULL), v3(ext.COUNT_AUTO)
        assert all(torch.equal(a, b) for a, b in zip(got[:3], ref[:3]))
        # the batch size is not part of the shape that must match -- a detector hands over a different number of crops with every
        # frame --: fewer or more crops are decided at once on the last call's per-image figures (scaled to the new batch size)
        for Bn, first in ((11, want_staged), (21, want_staged)):
            dn = synth.make_batch(B=Bn, H=256, W=256, K=9, fg=fg, sigma=0.05, seed=500 + Bn, device=gpu)
            ms = [ext.stage_ms_in_pipeline([dn["mask"]], [dn["vertex"]], hn, 0.99, 5, 30000, 5, 1, ext.COUNT_AUTO)[0] for _ in range(2)]
            assert (ms[0][5] > 0) == first and (ms[1][5] > 0) == want_staged, (Bn, ms)     # (each call completes before the next decides)


def test_auto_follows_the_stage_hint(synth, pkg, gpu):
    """PVV_COUNT_AUTO stages a v3 call only if the winner ratios the last completed calls left in the per-device hint
    reach the problem's threshold (pvv_stage_hint_query): after calls on a clean field the next calls are staged, after
    calls on a field with 35 % outlier pixels they are not -- and either way the results are those of the full pass."""
    from clean_pvnet_amd import ransac_voting as ext
    c = {**synth.CONFIGS["cfg3"], "B": 16}                        # 2.3e10: large enough for AUTO to consider staging
    hn = c["hn"]
    clean = synth.make_batch(**c, seed=21, device=gpu)
    noisy = synth.make_batch(**{**c, "outlier": 0.35}, seed=22, device=gpu)

    def v3(d, k):
        return ext.ransac_voting_v3(d["mask"], d["vertex"], hn, 0.99, 5, 30000, None, None, 5, ext.SINGULAR_REFERENCE, count_kernel=k)

    for d, want_staged in ((clean, True), (noisy, False), (clean, True)):
        for _ in range(3):
            out, win, tn, _ws = v3(d, ext.COUNT_AUTO)
        torch.cuda.synchronize()
        valid, mean, thr = ext.stage_hint(d["mask"], d["vertex"], hn)
        ratio = float((win.double() / tn.double().view(-1, 1)).mean())
        assert valid and abs(mean - ratio) < 1e-4, (mean, ratio)
        assert 0.5 <= thr <= 0.995 and (mean >= thr) == want_staged, (mean, thr)
        ms = ext.stage_ms_in_pipeline([d["mask"]], [d["vertex"]], hn, 0.99, 5, 30000, 5, 3, ext.COUNT_AUTO)
        assert all((r[5] > 0) == want_staged for r in ms), ms     # the first count launch's mark: recorded iff staged
        ref = v3(d, ext.COUNT_FULL)
        got = v3(d, ext.COUNT_AUTO)
        assert all(torch.equal(a, b) for a, b in zip(got[:3], ref[:3]))
        # the explicit modes ignore the hint
        assert all(r[5] > 0 for r in ext.stage_ms_in_pipeline([d["mask"]], [d["vertex"]], hn, 0.99, 5, 30000, 5, 2, ext.COUNT_STAGED))
        assert all(r[5] < 0 for r in ext.stage_ms_in_pipeline([d["mask"]], [d["vertex"]], hn, 0.99, 5, 30000, 5, 2, ext.COUNT_FULL))


@pytest.mark.parametrize("B,H,W,K,hn,fg,outlier,what", [
    (40, 240, 320, 2, 1536, 0.30, 0.40, "three 512-hypothesis groups, most of them survive: several passes, runs of 2"),
    (96, 200, 320, 4, 512, (0.10, 0.45), 0.05, "ragged images (13 ... 57 chunks): long runs of unequal length"),
    (6, 480, 640, 3, 2048, 0.098, 0.0, "30 000 pixels per image, few items: runs of one chunk, four groups in one pass"),
    (12, 300, 400, 17, 700, 0.25, 0.15, "hn not a multiple of 32 or 512, 17 keypoints"),
    (112, 480, 640, 9, 512, 0.02, 0.0, "config 3 at B = 112: large enough for the EIGHTH first stage (mask 0x02)"),
    (2, 200, 320, 1, 40000, 0.30, 0.2, "40 000 hypotheses = 79 groups: more than one pass may span (63 groups)"),
])
def test_run_owning_filter_launch_branches(synth, pkg, gpu, B, H, W, K, hn, fg, outlier, what):
    """k_count_filter_runs (round 4: the staged pass's second launch -- runs of chunks, passes over hypothesis groups,
    cooperative progressive elimination through the shared miss counters): winners, winner counts, tn and keypoints of
    PVV_COUNT_STAGED equal those of the full pass bit for bit, on shapes that take its less-travelled branches; repeated
    calls give identical results (the elimination's timing may differ from call to call, its outcome may not)."""
    from clean_pvnet_amd import ransac_voting as ext
    d = synth.make_batch(B=B, H=H, W=W, K=K, fg=fg, sigma=0.05, outlier=outlier, seed=4000 + B, device=gpu)
    m, v = d["mask"], d["vertex"]
    full = ext.ransac_voting_v3(m, v, hn, 0.99, 5, 30000, None, None, 31, ext.SINGULAR_ZERO, count_kernel=ext.COUNT_FULL)
    tn = full[2].cpu()
    assert int((tn >= 8 * 512 - 511).sum()) >= B // 2, "the case should stage most images: %s" % tn.tolist()
    for rep in range(3):
        st = ext.ransac_voting_v3(m, v, hn, 0.99, 5, 30000, None, None, 31, ext.SINGULAR_ZERO, count_kernel=ext.COUNT_STAGED)
        for a_, b_, nm in zip(st[:3], full[:3], ("keypoints", "winner counts", "tn")):
            assert torch.equal(a_, b_), (what, nm, rep)
    ms = ext.stage_ms_in_pipeline([m], [v], hn, 0.99, 5, 30000, 31, 2, ext.COUNT_STAGED)
    assert all(r[5] > 0 for r in ms)                              # it WAS staged (the first launch's mark was recorded)


@pytest.mark.parametrize("B,H,W,K,hn,fg,outlier", [(3, 480, 640, 9, 4096, 0.02, 0.0), (20, 240, 320, 4, 1024, 0.12, 0.2),
                                                   (2, 540, 720, 5, 2048, 0.08, 0.05), (5, 200, 300, 3, 700, (0.02, 0.3), 0.0)])
def test_estimate_counted_in_stages_equals_the_full_pass(synth, pkg, gpu, B, H, W, K, hn, fg, outlier):
    """estimate_voting_distribution_with_mean zeroes every ratio below (max ratio - 0.1) in binary32 (P:262-264, k_covariance), so a
    count pass in stages may drop what provably falls below that window (stage_bound: L* - ceil(tn / 10) - margin).  Forced with
    PVV_COUNT_STAGED_ESTIMATE (ABI v8; AUTO takes it from ~6 LINEMOD frames on -- est_stage_auto and the stage hint, DESIGN.md 4.2 --
    and PVV_COUNT_STAGED stages v3 only): covariances and PnP weights equal the full pass bit for bit; a call that asks for the counts themselves is always
    counted in full."""
    from clean_pvnet_amd import ransac_voting as ext
    d = synth.make_batch(B=B, H=H, W=W, K=K, fg=fg, sigma=0.05, outlier=outlier, seed=5100 + B, device=gpu)
    m, v = d["mask"], d["vertex"]
    mean = (d["kpt_2d"] + 0.25).contiguous()
    full = ext.estimate_voting_distribution(m, v, mean, hn, 0.99, 5, 30000, None, None, 9, False, 0, ext.COUNT_FULL)
    for rep in range(2):
        st = ext.estimate_voting_distribution(m, v, mean, hn, 0.99, 5, 30000, None, None, 9, False, 0, ext.COUNT_STAGED_ESTIMATE)
        assert torch.equal(st[0], full[0]) and torch.equal(st[4], full[4]) and torch.equal(st[3], full[3]), rep
    # PVV_COUNT_STAGED is v3's alone again: the estimate under it runs -- and equals -- the full pass
    st = ext.estimate_voting_distribution(m, v, mean, hn, 0.99, 5, 30000, None, None, 9, False, 0, ext.COUNT_STAGED)
    assert torch.equal(st[0], full[0]) and torch.equal(st[4], full[4])
    # with the counts as an output every one of them is exact, whatever the mode
    a = ext.estimate_voting_distribution(m, v, mean, hn, 0.99, 5, 30000, None, None, 9, True, 0, ext.COUNT_STAGED_ESTIMATE)
    b = ext.estimate_voting_distribution(m, v, mean, hn, 0.99, 5, 30000, None, None, 9, True, 0, ext.COUNT_FULL)
    assert torch.equal(a[2], b[2]) and torch.equal(a[0], full[0])


def test_auto_counts_a_large_estimate_in_stages_and_equals_the_full_pass(synth, pkg, gpu):
    """Round 5: with the second launch at five blocks per CU the staged estimate wins from ~18 LINEMOD frames on, and AUTO takes it
    (pvv_estimate_counts_in_stages).  24 frames, 4096 hypotheses: covariances and PnP weights equal the full pass bit for bit, the
    stage marks show that the pass really ran in stages; 8 frames stay with the full pass UNTIL a v3 call on the same fields has
    reported clean winners (the stage hint of the shape, whatever its hn: with the nearest-first chunk order the staged pass wins
    from 6 clean frames on); decode_keypoint(un_pnp=True) is the one fused call at every size (which then counts its rows as two
    passes, see the next test) and equals the two separate calls."""
    from clean_pvnet_amd import decode_keypoint
    from clean_pvnet_amd import ransac_voting as ext
    ext.shutdown()                                                       # forget the stage hints other tests left on the device
    assert ext.estimate_counts_in_stages(24, 480, 640, 9, 4096) and not ext.estimate_counts_in_stages(8, 480, 640, 9, 4096)
    d = synth.make_batch(**{**synth.CONFIGS["cfg3"], "B": 24}, device=gpu)
    m, v = d["mask"], d["vertex"]
    mean = (d["kpt_2d"] + 0.25).contiguous()
    full = ext.estimate_voting_distribution(m, v, mean, 4096, 0.99, 5, 30000, None, None, 9, False, 0, ext.COUNT_FULL)
    auto = ext.estimate_voting_distribution(m, v, mean, 4096, 0.99, 5, 30000, None, None, 9, False, 0, ext.COUNT_AUTO)
    assert torch.equal(auto[0], full[0]) and torch.equal(auto[4], full[4])
    ms = ext.stage_ms_in_pipeline([m], [v], 4096, 0.99, 5, 30000, 3, 8, ext.COUNT_AUTO, True, True)
    assert all(r[5] > 0 for r in ms)                                   # the first-launch mark was recorded: staged
    ms8 = ext.stage_ms_in_pipeline([m[:8]], [v[:8]], 4096, 0.99, 5, 30000, 3, 8, ext.COUNT_AUTO, True, True)
    assert all(r[5] < 0 for r in ms8)                                  # 8 frames, no hint: the full pass
    for _ in range(2):                                                 # v3 on the same (clean) fields leaves the hint: ratios ~0.995, tn
        ext.ransac_voting_v3(m[:8], v[:8], 512, 0.99, 5, 30000, None, None, 1, ext.SINGULAR_REFERENCE)
        torch.cuda.synchronize()
    assert ext.estimate_counts_in_stages(8, 480, 640, 9, 4096)
    full8 = ext.estimate_voting_distribution(m[:8], v[:8], mean[:8].contiguous(), 4096, 0.99, 5, 30000, None, None, 9, False, 0, ext.COUNT_FULL)
    auto8 = ext.estimate_voting_distribution(m[:8], v[:8], mean[:8].contiguous(), 4096, 0.99, 5, 30000, None, None, 9, False, 0, ext.COUNT_AUTO)
    assert torch.equal(auto8[0], full8[0]) and torch.equal(auto8[4], full8[4])
    ms8 = ext.stage_ms_in_pipeline([m[:8]], [v[:8]], 4096, 0.99, 5, 30000, 3, 8, ext.COUNT_AUTO, True, True)
    assert all(r[5] > 0 for r in ms8)                                  # ... now in stages
    assert not ext.estimate_counts_in_stages(3, 480, 640, 9, 4096)     # (3 frames: 3.4e10 of work, below every bound)
    x = torch.empty(24, 2 + 18, 480, 640, device=gpu)
    x[:, 0] = 3.0 * (m == 0)
    x[:, 1] = 3.0 * (m != 0)
    x[:, 2:] = v.permute(0, 3, 4, 1, 2).reshape(24, 18, 480, 640)
    out = decode_keypoint({"seg": x[:, :2], "vertex": x[:, 2:]}, un_pnp=True, weights=True, seed=77)
    vtx = x[:, 2:].permute(0, 2, 3, 1).view(24, 480, 640, 9, 2)
    kpt, mask, var, wts, _w, _t = ext.decode_keypoint_un_pnp(x[:, :2], vtx, 512, 4096, 0.99, 5, 30000, None, None, None, 77,
                                                             ext.SINGULAR_REFERENCE, 0)
    assert torch.equal(out["kpt_2d"], kpt) and torch.equal(out["mask"], mask) and torch.equal(out["var"], var)
    assert torch.equal(out["var_weights"], wts)
    # ... and the two separate calls under the same seed (the fused call draws what they draw)
    kpt2, mask2, _w2, _t2 = ext.decode_keypoint_v3(x[:, :2], vtx, 512, 0.99, 5, 30000, None, None, 77, ext.SINGULAR_REFERENCE)
    est = ext.estimate_voting_distribution(mask2, vtx, kpt2, 4096, 0.99, 5, 30000, None, None, 77, False, 0, ext.COUNT_FULL)
    assert torch.equal(kpt, kpt2) and torch.equal(mask, mask2) and torch.equal(var, est[0]) and torch.equal(wts, est[4])


@pytest.mark.parametrize("B,H,W,K,hn,hn_est,fg,outlier", [(3, 480, 640, 9, 512, 4096, 0.02, 0.0), (6, 240, 320, 4, 200, 1100, 0.12, 0.2),
                                                          (2, 540, 720, 5, 128, 2048, 0.08, 0.05), (5, 200, 300, 3, 64, 700, (0.02, 0.3), 0.0)])
def test_fused_un_pnp_rows_counted_as_two_passes_equal_the_one_full_pass(synth, pkg, gpu, B, H, W, K, hn, hn_est, fg, outlier):
    """pvv_decode_keypoint_un_pnp keeps rows of hn + hn_est hypotheses from ONE compaction.  Where the estimate counts in stages
    (AUTO on large batches; forced here with PVV_COUNT_STAGED_ESTIMATE, which also forces v3's columns into stages) the rows are
    counted as two passes over column ranges -- [0, hn) against the arg-max bound, [hn, hn + hn_est) against the estimate's --
    with their own leader words: keypoints, winner counts, covariances and PnP weights equal the one full pass over all columns
    bit for bit, on repeated calls (the second set of leader words is re-zeroed per call), with injected index pairs too."""
    from clean_pvnet_amd import ransac_voting as ext
    d = synth.make_batch(B=B, H=H, W=W, K=K, fg=fg, sigma=0.05, outlier=outlier, seed=6100 + B, device=gpu)
    m, v = d["mask"], d["vertex"]
    x = torch.empty(B, 2 + 2 * K, H, W, device=gpu)
    x[:, 0] = 3.0 * (m == 0)
    x[:, 1] = 3.0 * (m != 0)
    x[:, 2:] = v.permute(0, 3, 4, 1, 2).reshape(B, 2 * K, H, W)
    seg, vtx = x[:, :2], x[:, 2:].permute(0, 2, 3, 1).view(B, H, W, K, 2)
    run = lambda ck, i0=None, i1=None: ext.decode_keypoint_un_pnp(seg, vtx, hn, hn_est, 0.99, 5, 30000, i0, i1, None, 31,   # noqa: E731
                                                                  ext.SINGULAR_REFERENCE, 0, ck)
    full = run(ext.COUNT_FULL)
    for rep in range(3):
        st = run(ext.COUNT_STAGED_ESTIMATE)
        for a_, b_, nm in zip(st, full, ("keypoints", "mask", "covariances", "weights", "winner counts", "tn")):
            assert torch.equal(a_, b_), (nm, rep)
    # PVV_COUNT_STAGED is v3's alone: the fused call then counts everything in one full pass, as before
    st = run(ext.COUNT_STAGED)
    assert all(torch.equal(a_, b_) for a_, b_ in zip(st, full))
    # injected index pairs for both parts
    g = torch.Generator().manual_seed(5)
    tn_min = int(full[5].min().item())
    if tn_min > 0:
        i0 = torch.randint(0, tn_min, (B, hn, K, 2), generator=g, dtype=torch.int32).to(gpu)
        i1 = torch.randint(0, tn_min, (B, hn_est, K, 2), generator=g, dtype=torch.int32).to(gpu)
        a, b = run(ext.COUNT_FULL, i0, i1), run(ext.COUNT_STAGED_ESTIMATE, i0, i1)
        assert all(torch.equal(a_, b_) for a_, b_ in zip(a, b))


@pytest.mark.parametrize("what", ["many_chunks", "nan_mean", "far_mean", "one_image", "mean_inside_each_band"])
def test_estimate_chunk_order_edge_cases(synth, pkg, gpu, what):
    """The estimate's second launch walks a run's chunks nearest (in y) to the keypoint first (count_filter_runs.hpp) -- any order
    counts the same pairs, so covariances and weights must equal the full pass bit for bit whatever the keypoints are: more than
    64 remaining chunks per image (the order falls back to row-major), NaN / infinite keypoints, keypoints far outside the image,
    a single image (many short runs per keypoint), and keypoints placed inside every band of the mask in turn."""
    from clean_pvnet_amd import ransac_voting as ext
    if what == "many_chunks":
        B, H, W, K, hn, fg, max_num = 2, 512, 512, 3, 1024, 0.3, 200000          # ~78 000 rows per image: 153 chunks, 115 remaining
    elif what == "one_image":
        B, H, W, K, hn, fg, max_num = 1, 480, 640, 9, 4096, 0.05, 30000
    else:
        B, H, W, K, hn, fg, max_num = 3, 480, 640, 5, 2048, 0.03, 30000
    d = synth.make_batch(B=B, H=H, W=W, K=K, fg=fg, sigma=0.05, outlier=0.05, seed=7300 + len(what), device=gpu)
    m, v = d["mask"], d["vertex"]
    means = [(d["kpt_2d"] + 0.25).contiguous()]
    if what == "nan_mean":
        mn = means[0].clone()
        mn[0, 0] = float("nan")
        mn[1, 1, 1] = float("inf")
        mn[2, 2, 0] = float("-inf")
        means = [mn]
    elif what == "far_mean":
        means = [means[0] + 1e7, means[0] * 0 - 5e4]
    elif what == "mean_inside_each_band":
        ys = torch.nonzero(m[0])[:, 0].float()
        means = []
        for q in (0.0, 0.1, 0.35, 0.5, 0.8, 1.0):
            mn = d["kpt_2d"].clone()
            mn[..., 1] = ys.min() + q * (ys.max() - ys.min())
            means.append(mn.contiguous())
    for mean in means:
        full = ext.estimate_voting_distribution(m, v, mean, hn, 0.99, 5, max_num, None, None, 9, False, 0, ext.COUNT_FULL)
        if what == "many_chunks":
            assert int(full[3].min().item()) > 64 * 512 * 8 // 6                 # enough rows for > 64 remaining chunks
        for rep in range(2):
            st = ext.estimate_voting_distribution(m, v, mean, hn, 0.99, 5, max_num, None, None, 9, False, 0, ext.COUNT_STAGED_ESTIMATE)
            same = lambda a_, b_: torch.equal(a_, b_) or bool(((a_ == b_) | (a_.isnan() & b_.isnan())).all())   # noqa: E731
            assert same(st[0], full[0]) and same(st[4], full[4]) and torch.equal(st[3], full[3]), (what, rep)


def test_staged_estimate_and_fused_two_pass_on_random_shapes(synth, pkg, gpu):
    """Twelve seeded random problems (image size, keypoints, hypothesis counts that are no multiples of 32, foreground, outliers):
    the estimate counted in stages and the fused un_pnp call with its rows counted as two passes against their full passes, bit
    for bit."""
    from clean_pvnet_amd import ransac_voting as ext
    rng = np.random.default_rng(20260926)
    for case in range(12):
        B = int(rng.integers(1, 7))
        H, W = int(rng.integers(96, 400)), int(rng.integers(96, 400))
        K = int(rng.integers(1, 10))
        hn, hn_est = int(rng.integers(33, 600)), int(rng.integers(130, 2500))
        fg = float(rng.uniform(0.05, 0.5))
        outlier = float(rng.choice([0.0, 0.05, 0.3]))
        d = synth.make_batch(B=B, H=H, W=W, K=K, fg=fg, sigma=0.05, outlier=outlier, seed=8800 + case, device=gpu)
        m, v = d["mask"], d["vertex"]
        tag = (case, B, H, W, K, hn, hn_est, round(fg, 2), outlier)
        mean = (d["kpt_2d"] + float(rng.uniform(-3, 3))).contiguous()
        full = ext.estimate_voting_distribution(m, v, mean, hn_est, 0.99, 5, 30000, None, None, case, False, 0, ext.COUNT_FULL)
        st = ext.estimate_voting_distribution(m, v, mean, hn_est, 0.99, 5, 30000, None, None, case, False, 0, ext.COUNT_STAGED_ESTIMATE)
        assert torch.equal(st[0], full[0]) and torch.equal(st[4], full[4]), tag
        x = torch.empty(B, 2 + 2 * K, H, W, device=gpu)
        x[:, 0] = 3.0 * (m == 0)
        x[:, 1] = 3.0 * (m != 0)
        x[:, 2:] = v.permute(0, 3, 4, 1, 2).reshape(B, 2 * K, H, W)
        seg, vtx = x[:, :2], x[:, 2:].permute(0, 2, 3, 1).view(B, H, W, K, 2)
        a = ext.decode_keypoint_un_pnp(seg, vtx, hn, hn_est, 0.99, 5, 30000, None, None, None, case, ext.SINGULAR_REFERENCE, 0, ext.COUNT_FULL)
        b = ext.decode_keypoint_un_pnp(seg, vtx, hn, hn_est, 0.99, 5, 30000, None, None, None, case, ext.SINGULAR_REFERENCE, 0, ext.COUNT_STAGED_ESTIMATE)
        for a_, b_, nm in zip(a, b, ("keypoints", "mask", "covariances", "weights", "winner counts", "tn")):
            assert torch.equal(a_, b_), (nm,) + tag
