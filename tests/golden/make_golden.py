#!/usr/bin/env python
"""Generate tests/golden/*.npz by running THE REFERENCE'S OWN Python glue on CPU.

The reference's voting layers (``/root/reference/lib/csrc/ransac_voting/ransac_voting_gpu.py``) are imported from
where they lie (never copied) and executed on CPU tensors.  They cannot run as they are here, so this script
supplies exactly three substitutions and records them:

  1. ``lib.csrc.ransac_voting.ransac_voting`` (the CUDA extension, unbuildable without nvcc) is replaced by a
     module whose ``generate_hypothesis`` / ``voting_for_hypothesis`` call the C oracle's line-by-line
     restatement of the two kernels (oracle/vote_oracle.c).  The kernels themselves are therefore NOT pinned
     by these fixtures; everything around them -- compaction order, index drawing, argmax, ratio update,
     confidence loop, refit, covariance -- is the reference's code.
  2. ``torch.solve`` (removed from torch 2.x; ``b_inv``:106 would silently fall into its bare ``except`` and
     return the identity) is provided with torch-1.1 semantics: ``(linalg.solve(A, B), None)``, raising
     RuntimeError for a singular batch member like the batched gesv did.
  3. ``Tensor.masked_select`` accepts the uint8 masks torch 1.1 accepted (:142, :227).

The fake extension records the ``idxs`` the glue drew with ``random_`` so the same draws can be injected into the
oracle / HIP path.  Run from the repository root in the build container:  python tests/golden/make_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference/lib/csrc/ransac_voting/ransac_voting_gpu.py"
OUT = os.path.dirname(os.path.abspath(__file__))

from oracle import vote_oracle  # noqa: E402

recorded = []


def _fake_extension():
    m = types.ModuleType("lib.csrc.ransac_voting.ransac_voting")

    def generate_hypothesis(direct, coords, idxs):
        recorded.append(idxs.numpy().copy())
        return torch.from_numpy(vote_oracle.generate_hypothesis(direct.numpy(), coords.numpy(), idxs.numpy()))

    def voting_for_hypothesis(direct, coords, hypo_pts, inliers, thresh):
        vote_oracle.voting_for_hypothesis(direct.numpy(), coords.numpy(), hypo_pts.numpy(), inliers.numpy(), thresh)

    m.generate_hypothesis = generate_hypothesis
    m.voting_for_hypothesis = voting_for_hypothesis
    return m


def load_reference():
    for name in ("lib", "lib.csrc", "lib.csrc.ransac_voting"):
        pkg = types.ModuleType(name)
        pkg.__path__ = []
        sys.modules[name] = pkg
    sys.modules["lib.csrc.ransac_voting.ransac_voting"] = _fake_extension()
    sys.modules["lib.csrc.ransac_voting"].ransac_voting = sys.modules["lib.csrc.ransac_voting.ransac_voting"]

    def solve(B, A):                       # torch 1.1: torch.solve(B, A) -> (X, LU), raises on a singular A
        X, info = torch.linalg.solve_ex(A, B)
        if (info != 0).any() or not torch.isfinite(X).all():
            raise RuntimeError("solve: U(i,i) is zero, singular U.")
        return X, None
    torch.solve = solve

    orig = torch.Tensor.masked_select

    def masked_select(self, mask):
        return orig(self, mask.bool() if mask.dtype == torch.uint8 else mask)
    torch.Tensor.masked_select = masked_select

    spec = importlib.util.spec_from_file_location("ref_ransac_voting_gpu", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    return ref


def field(seed, H, W, K, fg, sigma, zero_kpt=None):
    """Small synthetic {mask, vertex, kpt}; same recipe as clean-pvnet_amd/synth.py, kept local so the fixtures
    do not change when the synthetic generator does."""
    rng = np.random.RandomState(seed)
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    cx, cy = rng.uniform(0.35 * W, 0.65 * W), rng.uniform(0.35 * H, 0.65 * H)
    a = np.sqrt(fg * H * W / np.pi) * 1.2
    b = fg * H * W / (np.pi * a)
    mask = (((xs - cx) / a) ** 2 + ((ys - cy) / b) ** 2 <= 1.0)
    kpt = np.stack([rng.uniform(cx - 1.5 * a, cx + 1.5 * a, K), rng.uniform(cy - 1.5 * a, cy + 1.5 * a, K)], 1).astype(np.float32)
    v = kpt[None, None] - np.stack([xs, ys], -1)[:, :, None, :]
    n = np.linalg.norm(v, axis=-1, keepdims=True)
    n[n < 1e-3] += 1e-3
    v = v / n + sigma * rng.randn(H, W, K, 2)
    v = np.where(mask[:, :, None, None], v, rng.uniform(-1, 1, (H, W, K, 2))).astype(np.float32)
    if zero_kpt is not None:
        v[:, :, zero_kpt, :] = 0.0
    return mask.astype(np.int64), v, kpt


def main():
    ref = load_reference()
    cases = {}

    # ---- v3: two ordinary images + one below min_num, hn = 128, thresh 0.99
    torch.manual_seed(20260925)
    ms, vs, ks = zip(*[field(s, 96, 128, 5, 0.06, 0.03) for s in (1, 2)])
    m3 = np.zeros((96, 128), np.int64); m3[5, 5:8] = 1
    mask = np.stack(list(ms) + [m3]); vertex = np.stack(list(vs) + [vs[0]])
    recorded.clear()
    out = ref.ransac_voting_layer_v3(torch.from_numpy(mask), torch.from_numpy(vertex), 128, inlier_thresh=0.99)
    rounds = [len([r for r in recorded])]            # gen_hypothesis calls = confidence-loop iterations in total
    idx_per_image = []
    # the loop re-uses idxs (:145 is outside the while), so consecutive recordings repeat; keep the distinct ones in order
    for r in recorded:
        if not idx_per_image or not np.array_equal(idx_per_image[-1], r):
            idx_per_image.append(r)
    assert len(idx_per_image) == 2
    idxs = np.stack(idx_per_image + [np.zeros_like(idx_per_image[0])])
    cases["v3_basic"] = dict(mask=mask, vertex=vertex, idxs=idxs, hn=128, thresh=np.float32(0.99), out=out.numpy(),
                             kpt=np.stack(list(ks) + [ks[0]]), loop_calls=np.array(rounds))

    # ---- v3 with max_num subsampling: record the uniform_ draws by seeding identically and replaying
    torch.manual_seed(77)
    mask, vertex, kpt = field(3, 96, 128, 4, 0.25, 0.03)
    fg = int(mask.sum()); max_num = 1200
    assert fg > max_num
    state = torch.get_rng_state()
    recorded.clear()
    out = ref.ransac_voting_layer_v3(torch.from_numpy(mask[None]), torch.from_numpy(vertex[None]), 64, inlier_thresh=0.99,
                                     max_num=max_num)
    torch.set_rng_state(state)
    selection = torch.zeros(mask.shape, dtype=torch.float32).uniform_(0, 1).numpy()     # first draw of the call (:136)
    cases["v3_subsample"] = dict(mask=mask[None], vertex=vertex[None], idxs=recorded[0][None], hn=64, thresh=np.float32(0.99),
                                 out=out.numpy(), selection=selection[None], max_num=max_num, kpt=kpt[None])

    # ---- v3 singular: keypoint 2 has zero directions -> count 0 -> ATA = 0 -> b_inv falls back to identity
    torch.manual_seed(5)
    mask, vertex, kpt = field(4, 64, 96, 4, 0.08, 0.03, zero_kpt=2)
    recorded.clear()
    out = ref.ransac_voting_layer_v3(torch.from_numpy(mask[None]), torch.from_numpy(vertex[None]), 64, inlier_thresh=0.99)
    cases["v3_singular"] = dict(mask=mask[None], vertex=vertex[None], idxs=recorded[0][None], hn=64, thresh=np.float32(0.99),
                                out=out.numpy(), kpt=kpt[None])

    # ---- v1 layer on the basic case (torch.inverse; try/except -> zeros)
    torch.manual_seed(11)
    mask, vertex, kpt = field(6, 64, 96, 4, 0.08, 0.03)
    recorded.clear()
    out = ref.ransac_voting_layer(torch.from_numpy(mask[None]), torch.from_numpy(vertex[None]), 64, inlier_thresh=0.99)
    cases["v1_basic"] = dict(mask=mask[None], vertex=vertex[None], idxs=recorded[0][None], hn=64, thresh=np.float32(0.99),
                             out=out.numpy(), kpt=kpt[None])

    # ---- estimate_voting_distribution_with_mean: 4 rounds of 64, one image of class 2 (mask == 1 empty -> skipped)
    torch.manual_seed(99)
    ms, vs, ks = zip(*[field(s, 64, 96, 4, 0.08, 0.03) for s in (7, 8)])
    mask = np.stack(ms); vertex = np.stack(vs); mask[1] *= 2
    mean = (np.stack(ks) + 0.3).astype(np.float32)
    recorded.clear()
    rmean, cov = ref.estimate_voting_distribution_with_mean(torch.from_numpy(mask), torch.from_numpy(vertex),
                                                            torch.from_numpy(mean), round_hyp_num=64, min_hyp_num=256)
    assert len(recorded) == 4
    idxs = np.stack([np.concatenate(recorded, 0), np.zeros((256, 4, 2), np.int32)])
    cases["estimate_basic"] = dict(mask=mask, vertex=vertex, idxs=idxs, mean=mean, round_hyp_num=64, min_hyp_num=256,
                                   thresh=np.float32(0.99), cov=cov.numpy())

    # ---- v3 on a uint8 mask of 255s: foreground_num is the SUM of the byte values (:126), so ~740 pixels count as
    #      ~189 000 > max_num and are subsampled with probability max_num / (255 * pixels) (:135-138)
    torch.manual_seed(123)
    mask, vertex, kpt = field(9, 96, 128, 4, 0.06, 0.03)
    mask = (mask * 255).astype(np.uint8)
    state = torch.get_rng_state()
    recorded.clear()
    # (.byte() of a uint8 tensor is the tensor itself, so the reference's `cur_mask *= selected_mask` (:138) would
    #  subsample the CALLER's mask in place: hand it a copy)
    out = ref.ransac_voting_layer_v3(torch.from_numpy(mask[None].copy()), torch.from_numpy(vertex[None]), 64, inlier_thresh=0.99)
    torch.set_rng_state(state)
    selection = torch.zeros(mask.shape, dtype=torch.float32).uniform_(0, 1).numpy()
    assert int(mask.astype(np.int64).sum()) > 30000 and recorded[0].max() < 400
    cases["v3_bytemask"] = dict(mask=mask[None], vertex=vertex[None], idxs=recorded[0][None], hn=64, thresh=np.float32(0.99),
                                out=out.numpy(), selection=selection[None], max_num=30000, kpt=kpt[None])

    # ---- estimate with max_num subsampling (:219-223): foreground = count of (mask == 1), one uniform_ draw per image
    torch.manual_seed(321)
    mask, vertex, kpt = field(10, 96, 128, 4, 0.2, 0.03)
    max_num = 900
    assert int((mask == 1).sum()) > max_num
    mean = (kpt[None] + 0.25).astype(np.float32)
    state = torch.get_rng_state()
    recorded.clear()
    rmean, cov = ref.estimate_voting_distribution_with_mean(torch.from_numpy(mask[None].copy()), torch.from_numpy(vertex[None]),
                                                            torch.from_numpy(mean), round_hyp_num=64, min_hyp_num=128,
                                                            max_num=max_num)
    torch.set_rng_state(state)
    selection = torch.zeros(mask.shape, dtype=torch.float32).uniform_(0, 1).numpy()
    assert len(recorded) == 2
    cases["estimate_subsample"] = dict(mask=mask[None], vertex=vertex[None], idxs=np.concatenate(recorded, 0)[None], mean=mean,
                                       round_hyp_num=64, min_hyp_num=128, max_num=max_num, selection=selection[None],
                                       thresh=np.float32(0.99), cov=cov.numpy())

    for name, c in cases.items():
        path = os.path.join(OUT, name + ".npz")
        if os.path.exists(path) and "--force" not in sys.argv:       # committed fixtures are not rewritten (zip metadata churn)
            old = dict(np.load(path))
            same = all(np.array_equal(np.asarray(old[k]), np.asarray(v)) for k, v in c.items()) and set(old) == set(c)
            print(name, "exists,", "identical content" if same else "CONTENT DIFFERS (run with --force to rewrite)")
            continue
        np.savez_compressed(path, **c)
        print(name, {k: (v.shape if hasattr(v, "shape") else v) for k, v in c.items()})


if __name__ == "__main__":
    main()
