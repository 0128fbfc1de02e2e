"""GPU: THE REFERENCE'S OWN PYTHON on the drop-in boundary (VERDICT r3 missing #2).

The code objects of ``/root/reference/lib/csrc/ransac_voting/ransac_voting_gpu.py`` and
``/root/reference/lib/networks/pvnet/resnet18.py`` (byte-compiled where they lie into oracle/_ref/*.pyc by
``make -C oracle _ref_py``; loaded by tests/refglue.py -- nothing of them is in the tree) run here on the MI355X with

  (i)  ``lib.csrc.ransac_voting.ransac_voting`` = the HIP extension: the reference's glue (P:6-274: torch compaction,
       ``random_``, its ``while True`` loop, ``torch.max``, its float32 refit and ``b_inv``) drives the product's four
       module-level kernels.  Every kernel call it makes is re-done by the oracle's restatement on the same arguments
       (bit-exact), its result is compared with the golden fixture the SAME glue produced on CPU (both are float32
       refits: they agree to float32 accumulation noise, reported) and with the product's fused layer on the same draws;
  (ii) ``lib.csrc.ransac_voting.ransac_voting_gpu`` = the product's layers: the unmodified ``Resnet18.decode_keypoint``
       (resnet18.py:65-76, both ``cfg.test.un_pnp`` branches) fills ``output`` with exactly what the fused
       ``clean_pvnet_amd.decode.decode_keypoint`` produces from the same network output and the same RNG key.

The numbers Weak #1 of VERDICT r3 asked for -- how far the product (binary64 refit) sits from what the reference's own
float32 glue returns -- are printed per fixture and for one 480x640 / K = 9 / 512-hypothesis image (``[ref-glue]`` lines).
"""
import os

import numpy as np
import pytest
import torch

from tests import refglue
from tests import tolerances as tol

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
torch11 = refglue.torch11_fixture()        # the torch-1.1 shims the reference's code needs: held for this module only, then restored


def _np(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def ref(pkg, gpu, torch11):
    # fails (not skips) when the byte code did not travel: this test IS the drop-in claim
    return refglue.load_glue()


def _check_kernel_calls_against_the_oracle(oracle, calls):
    n = {"generate_hypothesis": 0, "voting_for_hypothesis": 0}
    for c in calls:
        if c[0] == "generate_hypothesis":
            _n, direct, coords, idxs, hyp = c
            want = oracle.generate_hypothesis(_np(direct), _np(coords), _np(idxs))
            assert np.array_equal(_np(hyp).view(np.uint32), want.view(np.uint32)), "hypotheses differ from the oracle's kernel"
        else:
            _n, direct, coords, hyp, inl, thresh = c
            want = oracle.voting_for_hypothesis(_np(direct), _np(coords), _np(hyp), np.zeros(tuple(inl.shape), np.uint8), thresh)
            assert np.array_equal(_np(inl), want), "inlier bytes differ from the oracle's kernel"
        n[c[0]] += 1
    return n


V3_CASES = [("v3_basic", "ransac_voting_layer_v3", "reference"), ("v3_subsample", "ransac_voting_layer_v3", "reference"),
            ("v3_singular", "ransac_voting_layer_v3", "reference"), ("v3_bytemask", "ransac_voting_layer_v3", "reference"),
            ("v1_basic", "ransac_voting_layer", "image_zero")]


@pytest.mark.parametrize("name,fn,singular", V3_CASES)
def test_reference_glue_on_hip_kernels_equals_golden_and_the_fused_layer(oracle, pkg, gpu, ref, name, fn, singular):
    import clean_pvnet_amd.ransac_voting_gpu as product
    import lib.csrc.ransac_voting.ransac_voting as ext
    c = dict(np.load(os.path.join(GOLD, name + ".npz")))
    hn, thresh = int(c["hn"]), float(c["thresh"])
    kw = {"max_num": int(c["max_num"])} if "max_num" in c else {}
    n_img = c["mask"].shape[0]
    voted = [i for i in range(n_img) if c["idxs"][i].any() or n_img == 1]            # skipped images draw nothing (P:129-132)
    d = refglue.Draws(ext, idxs=[c["idxs"][i] for i in voted], selection=list(c["selection"]) if "selection" in c else None)
    ref.ransac_voting = d
    try:
        with d.patch_uniform():
            out_ref = getattr(ref, fn)(torch.from_numpy(c["mask"]).to(gpu).clone(), torch.from_numpy(c["vertex"]).to(gpu), hn,
                                       inlier_thresh=thresh, **kw)
    finally:
        ref.ransac_voting = ext
    assert not d.idxs and not d.selection                                            # every recorded draw was consumed
    n = _check_kernel_calls_against_the_oracle(oracle, d.calls)
    assert n["generate_hypothesis"] >= len(voted) and n["voting_for_hypothesis"] >= 2 * len(voted)
    # the product's fused layer on the same draws
    t = {k: torch.from_numpy(v).to(gpu) for k, v in c.items() if isinstance(v, np.ndarray) and v.ndim > 0}
    ours = getattr(product, fn)(t["mask"], t["vertex"], hn, inlier_thresh=thresh, idxs=t["idxs"], selection=t.get("selection"), **kw)
    exact = tol.exact_v3(oracle, c["mask"], c["vertex"], hn, thresh, c["idxs"], selection=c.get("selection"),
                         max_num=int(c.get("max_num", 30000)), singular=singular)
    out_ref, ours = _np(out_ref), _np(ours)
    dev = lambda a, b: float(np.abs(np.asarray(a, np.float64) - b).max())            # noqa: E731
    print("\n[ref-glue] %-13s max|ours-golden| = %.3g  max|ours-ref_on_hip| = %.3g  max|ref_on_hip-golden| = %.3g  "
          "vs exact: ours %.3g, golden %.3g, ref_on_hip %.3g px"
          % (name, dev(ours, c["out"]), dev(ours, out_ref), dev(out_ref, c["out"]), dev(ours, exact), dev(c["out"], exact),
             dev(out_ref, exact)))
    tol.assert_means_close(ours, exact)                                              # the product: within the contract of the exact answer
    # the reference's float32 glue, on CPU (golden) and on the GPU over the HIP kernels: same inlier sets (checked call by
    # call above), float32 accumulation in two different orders -- both within a few 1e-4 px of the exact solution
    for other, what in ((c["out"], "golden"), (out_ref, "reference glue on the HIP kernels")):
        tol.assert_means_close(ours, other, extra=np.abs(other - exact), what="ours vs " + what)
    tol.assert_means_close(out_ref, c["out"], extra=np.abs(c["out"] - exact) + np.abs(out_ref - exact),
                           what="reference glue on the HIP kernels vs golden")
    assert bool((np.abs(out_ref.astype(np.float64) - c["out"]) <= 1e-3 + 1e-6 * np.abs(c["out"])).all())   # (v3_singular returns ATb: ~3e4)


@pytest.mark.parametrize("name", ["estimate_basic", "estimate_subsample"])
def test_reference_estimate_on_hip_kernels_equals_golden_and_the_fused_layer(oracle, pkg, gpu, ref, name):
    import clean_pvnet_amd.ransac_voting_gpu as product
    import lib.csrc.ransac_voting.ransac_voting as ext
    c = dict(np.load(os.path.join(GOLD, name + ".npz")))
    rh, mh = int(c["round_hyp_num"]), int(c["min_hyp_num"])
    kw = {"max_num": int(c["max_num"])} if "max_num" in c else {}
    rounds = []
    for i in range(c["mask"].shape[0]):
        if (c["mask"][i] == 1).sum() >= 5:                                           # P:209-216: skipped images draw nothing
            rounds += [c["idxs"][i][r * rh:(r + 1) * rh] for r in range(mh // rh)]
    d = refglue.Draws(ext, idxs=rounds, selection=list(c["selection"]) if "selection" in c else None)
    ref.ransac_voting = d
    try:
        with d.patch_uniform():
            mean_ref, cov_ref = ref.estimate_voting_distribution_with_mean(
                torch.from_numpy(c["mask"]).to(gpu).clone(), torch.from_numpy(c["vertex"]).to(gpu), torch.from_numpy(c["mean"]).to(gpu),
                round_hyp_num=rh, min_hyp_num=mh, **kw)
    finally:
        ref.ransac_voting = ext
    assert not d.idxs and not d.selection
    _check_kernel_calls_against_the_oracle(oracle, d.calls)
    t = {k: torch.from_numpy(v).to(gpu) for k, v in c.items() if isinstance(v, np.ndarray) and v.ndim > 0}
    _m, ours = product.estimate_voting_distribution_with_mean(t["mask"], t["vertex"], t["mean"], rh, mh, idxs=t["idxs"],
                                                              selection=t.get("selection"), **kw)
    ours, cov_ref = _np(ours), _np(cov_ref)
    rel = lambda a, b: float((np.abs(a.astype(np.float64) - b) / (1e-4 + np.abs(b))).max())   # noqa: E731
    print("\n[ref-glue] %-18s cov: max|ours-golden| = %.3g (rel %.3g)  max|ours-ref_on_hip| = %.3g  max|ref_on_hip-golden| = %.3g"
          % (name, np.abs(ours - c["cov"]).max(), rel(ours, c["cov"]), np.abs(ours - cov_ref).max(), np.abs(cov_ref - c["cov"]).max()))
    tol.assert_cov_close(ours, c["cov"], rtol=tol.COV_RTOL_VS_REFERENCE_F32, what="cov vs golden")
    tol.assert_cov_close(ours, cov_ref, rtol=tol.COV_RTOL_VS_REFERENCE_F32, what="cov vs the reference's glue on the HIP kernels")
    tol.assert_cov_close(cov_ref, c["cov"], rtol=2 * tol.COV_RTOL_VS_REFERENCE_F32, what="reference glue on HIP vs golden")


# (config, image seed, what it stresses)
GLUE_CASES = [("cfg2", 4242, "LINEMOD size: tn ~ 6100"),
              ("cfg4", 4243, "Occlusion-LINEMOD: sparse occluded mask, 30-50 % outlier pixels, 1024 hypotheses"),
              ("cfg5", 4244, "T-LESS size: 540x720, K = 17, 2048 hypotheses, foreground above max_num -> the subsample of P:135-138 with injected draws, tn ~ 30000")]


@pytest.mark.parametrize("cfgname,seed,what", GLUE_CASES)
def test_reference_glue_per_config_vs_the_fused_layer(oracle, synth, pkg, gpu, ref, cfgname, seed, what):
    """One image of BASELINE configs 2 / 4 / 5 through THE REFERENCE'S OWN float32 glue (on the HIP kernels) and through the
    product on the same draws: the reference's glue draws its index pairs from torch's device generator (recorded, injected into
    the product), config 5's subsample draws are injected into both.  Counts and winners are identical; how far is the binary64
    refit from what the reference's float32 glue returns, and how far is each from the EXACT least-squares solution of the same
    inlier sets (rational arithmetic)?  The three numbers per config are printed, written to $PVV_EVIDENCE_DIR (tracked as
    profiles/r06_ref_glue_deviation.json) and quoted in INTEGRATION.md (a maintainer comparing against
    a CUDA run must find the deviation documented for the size they run)."""
    import json
    import clean_pvnet_amd.ransac_voting_gpu as product
    import lib.csrc.ransac_voting.ransac_voting as ext
    cfg = {**synth.CONFIGS[cfgname], "B": 1}
    hn = cfg.pop("hn")
    dd = synth.make_batch(**cfg, seed=seed)
    mask, vertex = dd["mask"].to(gpu), dd["vertex"].to(gpu)
    fg = int((dd["mask"] != 0).sum())
    sel = None
    if fg > 30000:                                                                    # P:134-138 will subsample: the same U(0,1) field for both
        sel = torch.rand(1, cfg["H"], cfg["W"], generator=torch.Generator().manual_seed(seed + 1))
    torch.manual_seed(7)
    d = refglue.Draws(ext, selection=None if sel is None else [sel[0]])
    ref.ransac_voting = d
    try:
        with d.patch_uniform():
            out_ref = ref.ransac_voting_layer_v3(mask, vertex, hn, inlier_thresh=0.99)
    finally:
        ref.ransac_voting = ext
    assert len(d.drawn) == 1 and not d.selection
    idxs = d.drawn[0][None]
    sel_g = None if sel is None else sel.to(gpu)
    ours = product.ransac_voting_layer_v3(mask, vertex, hn, inlier_thresh=0.99, idxs=idxs, selection=sel_g)
    # the winners the two paths refit are the same hypotheses with the same counts
    o2, win, tn, _ws = ext.ransac_voting_v3(mask, vertex, hn, 0.99, 5, 30000, idxs, sel_g, 0, ext.SINGULAR_REFERENCE)
    votes = [c for c in d.calls if c[0] == "voting_for_hypothesis"]
    assert votes[0][1].shape[0] == int(tn[0])                                         # the same foreground pixels survived
    counts_ref = votes[0][4].sum(2, dtype=torch.int32)                               # [hn,vn]: P:159
    assert torch.equal(counts_ref.max(0).values, win[0])
    assert torch.equal(votes[-1][4].sum(2, dtype=torch.int32)[0], win[0])            # the re-vote of P:183 counts the same inliers
    exact = tol.exact_v3(oracle, _np(mask), _np(vertex), hn, 0.99, _np(idxs), selection=None if sel is None else sel.numpy())
    out_ref, ours = _np(out_ref), _np(ours)
    row = {"config": cfgname, "what": what, "H": cfg["H"], "W": cfg["W"], "K": cfg["K"], "hn": hn, "tn": int(tn[0]),
           "winner_ratio_min": round(float(win[0].min()) / max(1, int(tn[0])), 4),
           "max_abs_ours_minus_reference_glue_px": float("%.3g" % np.abs(ours - out_ref).max()),
           "max_abs_ours_minus_exact_px": float("%.3g" % np.abs(ours - exact).max()),
           "max_abs_reference_glue_minus_exact_px": float("%.3g" % np.abs(out_ref - exact).max()),
           "counts_and_winners_identical": True}
    print("\n[ref-glue] %s %dx%d K=%d hn=%d tn=%d: max|ours-ref_on_hip| = %.3g px; vs the exact solution: ours %.3g, "
          "reference float32 glue %.3g px" % (cfgname, cfg["H"], cfg["W"], cfg["K"], hn, int(tn[0]), row["max_abs_ours_minus_reference_glue_px"],
                                              row["max_abs_ours_minus_exact_px"], row["max_abs_reference_glue_minus_exact_px"]))
    out_dir = os.environ.get("PVV_EVIDENCE_DIR")
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "ref_glue_deviation.jsonl"), "a") as f:
            f.write(json.dumps(row) + "\n")
    tol.assert_means_close(ours, exact)
    tol.assert_means_close(ours, out_ref, extra=np.abs(out_ref - exact), what="ours vs the reference's glue on the HIP kernels")
    assert np.abs(out_ref - exact).max() < 5e-2                                      # float32 sums of ~1e7..1e8-sized terms


@pytest.mark.parametrize("un_pnp", [False, True])
def test_unmodified_resnet18_decode_keypoint_runs_on_the_drop_in_layers(synth, pkg, gpu, un_pnp, monkeypatch):
    """resnet18.py:65-76 as the reference wrote it: ``output`` gets what the fused decode_keypoint produces."""
    import clean_pvnet_amd.decode as decode
    import clean_pvnet_amd.ransac_voting_gpu as product
    mod, cfg = refglue.load_resnet18(un_pnp)
    assert cfg.test.un_pnp == un_pnp
    B, H, W, K = 2, 240, 320, 9
    dd = synth.make_batch(B=B, H=H, W=W, K=K, fg=0.05, sigma=0.05, seed=77, planar=True)
    x = torch.empty(B, 2 + 2 * K, H, W)
    g = torch.Generator().manual_seed(3)
    x[:, :2] = torch.randn(B, 2, H, W, generator=g) * 0.1
    x[:, 0] += 1.0
    x[:, 1][dd["mask"] != 0] += 4.0
    x[:, 2:] = dd["vertex"].permute(0, 3, 4, 1, 2).reshape(B, 2 * K, H, W)
    x = x.to(gpu)
    # one RNG key for every layer call of both paths (the layers draw it from torch's CPU generator; the fused un_pnp
    # pass uses ONE key for v3 and the estimate where the reference's two calls would draw two)
    monkeypatch.setattr(product, "_next_seed", lambda: 20260926)
    monkeypatch.setattr(decode, "_next_seed", lambda: 20260926)
    out_ref = {"seg": x[:, :2], "vertex": x[:, 2:]}                                   # resnet18.py:93-96: channel slices of one tensor
    assert mod.Resnet18.decode_keypoint(None, out_ref) is None                       # updates `output` in place (resnet18.py:73,76)
    ours = decode.decode_keypoint({"seg": x[:, :2], "vertex": x[:, 2:]}, un_pnp=un_pnp)
    assert set(out_ref) == set(ours) == {"seg", "vertex", "mask", "kpt_2d"} | ({"var"} if un_pnp else set())
    assert torch.equal(out_ref["mask"], ours["mask"]) and out_ref["mask"].dtype == torch.int64
    assert torch.equal(out_ref["kpt_2d"], ours["kpt_2d"])                            # same kernels, same draws: bit-identical
    if un_pnp:
        assert torch.equal(out_ref["var"], ours["var"])
        assert np.abs(_np(ours["kpt_2d"]) - _np(dd["kpt_2d"])).max() < 3.0            # and it is the right answer
