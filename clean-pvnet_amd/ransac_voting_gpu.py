"""Host-side mirror of ``lib/csrc/ransac_voting/ransac_voting_gpu.py`` of clean-pvnet.

Same function names, positional order, defaults and return types as the reference
(/root/reference/lib/csrc/ransac_voting/ransac_voting_gpu.py:6, :97, :112, :202), so
``Resnet18.decode_keypoint`` (lib/networks/pvnet/resnet18.py:65-76) runs unchanged.  What
differs is how the work is done: the reference loops over the batch in Python with ~40
torch launches and >=3 host syncs per image; here one call launches a fixed sequence of
HIP kernels for the whole batch on the current stream and never reads anything back.

Keyword-only additions (all default to reference behaviour):
  idxs       injected hypothesis index pairs ``[b,hn,vn,2]`` int32 (the ``random_`` draws of
             :145 / :235); None = counter-based RNG on the device
  selection  injected U(0,1) draws ``[b,h,w]`` float32 (the ``uniform_`` of :136 / :220)
  singular   "reference" | "zero" -- see ``ransac_voting_layer_v3``
  out        (v3 / v1 only) a ``[b,vn,2]`` float32 CUDA tensor the keypoints are written into and that is returned --
             e.g. this rank's rows of a persistent gather buffer (``clean_pvnet_amd.dist.GatherBuffer``), so that the
             exchange of a sharded batch needs no copy; None = a fresh tensor
  seed, first_image   key of the device RNG (used where nothing is injected): ``seed`` defaults to a draw from torch's
             CPU generator; the generator is keyed by ``(seed, first_image + b)``, so the shards of a batch -- one per
             GPU, ``first_image`` = index of the shard's first image -- draw exactly what one call on the whole batch
             draws with the same ``seed`` (``clean_pvnet_amd.dist.sharded_vote(..., seed=)`` does this)
"""
import numpy as np
import torch

try:
    from . import ransac_voting as _ext
except ImportError as e:  # no silent fallback: the HIP extension IS the implementation
    raise ImportError(
        "clean_pvnet_amd: the HIP extension (libpvnet_vote.so + ransac_voting.so) is not built; "
        "run `python __graft_entry__.py` (or `python clean-pvnet_amd/_build.py`) first. "
        "There is no CPU fallback. Original error: %s" % (e,)) from e

_MAX_BATCH = 1024  # images per launch (the count kernel keeps its item table in LDS)
_POLICY = {"reference": _ext.SINGULAR_REFERENCE, "zero": _ext.SINGULAR_ZERO, "image_zero": _ext.SINGULAR_IMAGE_ZERO}


def _next_seed():
    """63-bit key for the device RNG, drawn from torch's CPU generator so that
    ``torch.manual_seed`` makes voting reproducible (the reference draws from the CUDA
    generator, run.py:67)."""
    return int(torch.empty((), dtype=torch.int64).random_().item())


def _as_mask(mask, equal_one):
    """Integer / bool masks go to the kernels as they are (any strides).  Floating masks are
    converted exactly as the reference does: ``.byte()`` (:125) or ``== 1`` (:207)."""
    if mask.dtype == torch.bool:
        return mask.view(torch.uint8)
    if mask.dtype.is_floating_point:
        return (mask == 1).view(torch.uint8) if equal_one else mask.byte()
    return mask


def _chunks(b):
    return [(lo, min(b, lo + _MAX_BATCH)) for lo in range(0, b, _MAX_BATCH)]


def _vote_v3(mask, vertex, round_hyp_num, inlier_thresh, min_num, max_num, idxs, selection, policy, seed=None,
             first_image=0, out=None):
    b = vertex.shape[0]
    if b == 0:                      # an empty shard (dist.sharded_vote on a trailing rank): nothing to launch
        return vertex.new_zeros((0, vertex.shape[3], 2)) if out is None else out
    if out is not None and tuple(out.shape) != (b, vertex.shape[3], 2):
        raise ValueError("out must be [b,vn,2] = %r, got %r" % ((b, vertex.shape[3], 2), tuple(out.shape)))
    mask = _as_mask(mask, False)
    outs = []
    seed = _next_seed() if seed is None else int(seed)   # one key for the whole batch; the device RNG is keyed by (seed, image index), so the
    for lo, hi in _chunks(b):   # split below is invisible in the results
        o, _win, _tn, _ws = _ext.ransac_voting_v3(
            mask[lo:hi], vertex[lo:hi], int(round_hyp_num), float(inlier_thresh), int(min_num), int(max_num),
            None if idxs is None else idxs[lo:hi], None if selection is None else selection[lo:hi],
            seed, policy, int(first_image) + lo, out=None if out is None else out[lo:hi])
        outs.append(o)
    if out is not None:
        return out
    return outs[0] if len(outs) == 1 else torch.cat(outs)


def ransac_voting_layer_v3(mask, vertex, round_hyp_num, inlier_thresh=0.999, confidence=0.99, max_iter=20,
                           min_num=5, max_num=30000, *, idxs=None, selection=None, singular="reference", seed=None,
                           first_image=0, out=None):
    '''
    :param mask:      [b,h,w]
    :param vertex:    [b,h,w,vn,2]
    :param round_hyp_num:
    :param inlier_thresh:
    :return: [b,vn,2]

    ``confidence`` and ``max_iter`` are accepted and ignored: the reference draws ``idxs``
    once before its ``while True`` (:145 vs :150-174), so every further iteration recomputes
    the same hypotheses and the strict ``<`` of :165 never fires -- the result is the state
    after iteration 1.

    ``singular`` -- a keypoint whose 2x2 normal matrix is singular (e.g. no inlier at all):
      "reference"  what ``b_inv`` (:97-109) did under torch 1.1: the batched solve raises, the
                   bare ``except`` substitutes the identity for every keypoint of that image,
                   so the whole image returns ``ATb``;
      "zero"       only that keypoint is affected and returns (0,0).
    '''
    del confidence, max_iter
    return _vote_v3(mask, vertex, round_hyp_num, inlier_thresh, min_num, max_num, idxs, selection,
                    _POLICY[singular], seed, first_image, out)


def ransac_voting_layer(mask, vertex, round_hyp_num, inlier_thresh=0.999, confidence=0.99, max_iter=20,
                        min_num=5, max_num=30000, *, idxs=None, selection=None, seed=None, first_image=0, out=None):
    '''
    :param mask:      [b,h,w]
    :param vertex:    [b,h,w,vn,2]
    :param round_hyp_num:
    :param inlier_thresh:
    :return: [b,vn,2]

    The v1 layer (:6-95; imported by resnet18.py:5, never called there).  Identical to v3 except
    for the singular case: ``torch.inverse`` raising makes the whole image zeros (:86-91).
    '''
    del confidence, max_iter
    return _vote_v3(mask, vertex, round_hyp_num, inlier_thresh, min_num, max_num, idxs, selection,
                    _POLICY["image_zero"], seed, first_image, out)


def b_inv(b_mat):
    '''
    Batched inverse with the reference's fallback (:97-109): if ANY matrix of the batch is
    singular the result is the identity for the WHOLE batch.  Unlike the original this does not
    raise-and-catch on the host (no sync); it is not used by the fused layers above.
    :param b_mat: [...,n,n]
    :return: same shape
    '''
    eye = b_mat.new_ones(b_mat.size(-1)).diag().expand_as(b_mat)
    inv, info = torch.linalg.inv_ex(b_mat)
    bad = (info != 0).any() | ~torch.isfinite(inv).all()
    return torch.where(bad, eye, inv)


def estimate_voting_distribution_with_mean(mask, vertex, mean, round_hyp_num=256, min_hyp_num=4096, topk=128,
                                           inlier_thresh=0.99, min_num=5, max_num=30000, output_hyp=False, *,
                                           idxs=None, selection=None, return_weights=False, seed=None, first_image=0):
    '''
    :param mask:   [b,h,w]   foreground is ``mask == 1`` (:207)
    :param vertex: [b,h,w,vn,2]
    :param mean:   [b,vn,2]
    :return: mean [b,vn,2], cov [b,vn,2,2]   (and all_hyp_pts [b,vn,hn,2], all_inlier_ratio [b,vn,hn]
             when ``output_hyp`` -- the reference has that branch commented out, :271-272)

    ``ceil(min_hyp_num/round_hyp_num)`` rounds of ``round_hyp_num`` fresh hypotheses (:231-247)
    are evaluated as ONE pass of ``round_num*round_hyp_num`` hypotheses; ``idxs`` (if injected)
    holds the rounds concatenated in order.  ``topk`` is unused, as in the reference.

    ``return_weights=True`` appends ``weights [b,vn,3] = (wxx,wxy,wyy)`` of ``inv(sqrtm(cov))`` -- what the
    evaluators compute per keypoint on the host before calling uncertainty_pnp
    (lib/evaluators/linemod/pvnet.py:118-130) -- fused into the covariance kernel.
    '''
    del topk
    b = vertex.shape[0]
    if b == 0:
        empty = (mean, vertex.new_zeros((0, vertex.shape[3], 2, 2)))
        if output_hyp:
            hn0 = int(np.ceil(min_hyp_num / round_hyp_num)) * int(round_hyp_num)
            empty += (vertex.new_zeros((0, vertex.shape[3], hn0, 2)), vertex.new_zeros((0, vertex.shape[3], hn0)))
        return empty + ((vertex.new_zeros((0, vertex.shape[3], 3)),) if return_weights else ())
    hn_total = int(np.ceil(min_hyp_num / round_hyp_num)) * int(round_hyp_num)
    mask = _as_mask(mask, True)
    mean_c = mean.contiguous().float()
    covs, hyps, ratios, wts = [], [], [], []
    seed = _next_seed() if seed is None else int(seed)
    for lo, hi in _chunks(b):
        cov, hyp, counts, tn, w = _ext.estimate_voting_distribution(
            mask[lo:hi], vertex[lo:hi], mean_c[lo:hi], hn_total, float(inlier_thresh), int(min_num),
            int(max_num), None if idxs is None else idxs[lo:hi],
            None if selection is None else selection[lo:hi], seed, bool(output_hyp), int(first_image) + lo)
        covs.append(cov)
        wts.append(w)
        if output_hyp:
            hyps.append(hyp)
            tnf = tn.float().clamp(min=1).view(-1, 1, 1)
            ratio = counts.float() / tnf
            ratios.append(torch.where(tn.view(-1, 1, 1) > 0, ratio, torch.ones_like(ratio)))
    cov = covs[0] if len(covs) == 1 else torch.cat(covs)
    extra = (wts[0] if len(wts) == 1 else torch.cat(wts),) if return_weights else ()
    if output_hyp:
        return (mean, cov, torch.cat(hyps), torch.cat(ratios)) + extra
    return (mean, cov) + extra


def uncertainty_pnp_weights(var):
    '''
    The host loop of ``Evaluator.uncertainty_pnp`` (lib/evaluators/linemod/pvnet.py:118-128) as batched tensor
    math on the device: ``inv(sqrtm(var))`` per keypoint in closed form, zeros where ``var[...,0,0] < 1e-6``, any
    entry is NaN or the matrix is not positive definite.
    :param var: [...,2,2]
    :return: [...,3]  (wxx, wxy, wyy) -- ``cov_invs.reshape(-1,4)[:, (0,1,3)]`` of :127-128
    '''
    v = var.double()
    a, b, d = v[..., 0, 0], v[..., 0, 1], v[..., 1, 1]
    det = a * d - b * b
    ok = ~(var[..., 0, 0] < 1e-6) & ~torch.isnan(v).flatten(-2).any(-1) & (det > 0) & (a > 0)
    s = torch.sqrt(det.clamp(min=0))
    t = torch.sqrt((a + d + 2 * s).clamp(min=0))
    q = t / ((a + s) * (d + s) - b * b)
    w = torch.stack([q * (d + s), -q * b, q * (a + s)], -1)
    return torch.where(ok.unsqueeze(-1), w, torch.zeros_like(w)).to(var.dtype)
