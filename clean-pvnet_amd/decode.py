"""``Resnet18.decode_keypoint`` (/root/reference/lib/networks/pvnet/resnet18.py:65-76) with the argmax fused in.

The reference computes ``mask = torch.argmax(output['seg'], 1)`` (one pass over the logits, an int64 mask written),
then the voting layer reads that mask again.  Here the class argmax happens inside the voting path's mask scan: the
logits are read once, the int64 ``mask`` the evaluators expect (``output['mask']``, evaluators/linemod/pvnet.py:186)
is written once and never read back.  SURVEY.md section 8(f) rank 2.
"""
import torch

from . import ransac_voting as _ext
from .ransac_voting_gpu import _POLICY, _next_seed, estimate_voting_distribution_with_mean


_MAX_BATCH = 1024   # images per launch (the count kernel's item table)


def decode_keypoint(output, un_pnp=False, *, idxs=None, selection=None, singular="reference", idxs_est=None,
                    weights=False, seed=None, first_image=0):
    """In-place update of ``output`` exactly like the reference method: adds ``mask`` [b,h,w] int64, ``kpt_2d``
    [b,vn,2] and -- with ``un_pnp`` (``cfg.test.un_pnp``, config.py:75) -- ``var`` [b,vn,2,2].

    ``output['seg']`` is [b,c,h,w] float32 logits, ``output['vertex']`` [b,2*vn,h,w] float32; both may be channel
    slices of one network output tensor (resnet18.py:93-94), no copy is made.

    ``un_pnp`` with a two-class ``seg`` (PVNet's) runs ``ransac_voting_layer_v3`` and
    ``estimate_voting_distribution_with_mean`` (resnet18.py:71-72) as ONE call -- one mask scan, one compaction, one
    hypothesis launch for the 512 + 4096 hypotheses -- with results bit-identical to the two calls on the same draws.  Small
    batches count all 4608 hypotheses in one launch; on batches large enough for the library to count the estimate IN STAGES
    (``pvv_estimate_counts_in_stages``: from ~6 clean LINEMOD frames on once v3 has reported its winners, ~18 before) the same call counts the rows as two passes over the one
    compaction -- v3's 512 columns as the layer would, then the estimate's 4096 against its own bound -- and keeps the saved
    scan and compaction (round 5; before, the two separate calls were taken there).  ``idxs_est`` [b,4096,vn,2] injects the
    estimate's index pairs like ``idxs`` does for v3, and
    ``weights=True`` also stores ``var_weights`` [b,vn,3] = (wxx,wxy,wyy) of ``inv(sqrtm(var))``, what the evaluator
    feeds uncertainty_pnp (evaluators/linemod/pvnet.py:118-130).

    ``seed`` / ``first_image`` key the device RNG exactly as in the layers (``ransac_voting_gpu``): batches beyond 1024
    images are cut into several launches here, and a batch sharded over GPUs decodes to the same result as one call
    when every shard passes the common ``seed`` and the index of its first image.
    """
    seg = output["seg"]
    ver = output["vertex"]
    b, vn_2, h, w = ver.shape
    vertex = ver.permute(0, 2, 3, 1).view(b, h, w, vn_2 // 2, 2)              # resnet18.py:66-68, a strided view
    seed = _next_seed() if seed is None else int(seed)
    segf = seg.float()
    fused = un_pnp and seg.shape[1] == 2
    if un_pnp:
        hn, max_num = 512, 30000                                             # resnet18.py:71
    else:
        hn, max_num = 128, 100                                               # resnet18.py:75
    parts = {"mask": [], "kpt_2d": [], "var": [], "var_weights": []}
    for lo in range(0, max(b, 1), _MAX_BATCH) if b else []:
        hi = min(b, lo + _MAX_BATCH)
        sl = slice(lo, hi)
        cut = lambda t: None if t is None else t[sl]                          # noqa: E731
        first = int(first_image) + lo
        if fused:
            # resnet18.py:71-72 fused; the estimate's defaults (P:202): ceil(4096 / 256) rounds of 256 hypotheses
            kpt, mask, var, wts, _win, _tn = _ext.decode_keypoint_un_pnp(segf[sl], vertex[sl], 512, 4096, 0.99, 5, 30000,
                                                                         cut(idxs), cut(idxs_est), cut(selection), seed,
                                                                         _POLICY[singular], first)
        else:
            kpt, mask, _win, _tn = _ext.decode_keypoint_v3(segf[sl], vertex[sl], hn, 0.99, 5, max_num, cut(idxs),
                                                           cut(selection), seed, _POLICY[singular], first)
            var = wts = None
            if un_pnp:                                                        # resnet18.py:72
                res = estimate_voting_distribution_with_mean(mask, vertex[sl], kpt, idxs=cut(idxs_est), return_weights=weights,
                                                             seed=seed, first_image=first)
                kpt, var = res[0], res[1]
                wts = res[2] if weights else None
        for k, t in (("mask", mask), ("kpt_2d", kpt), ("var", var), ("var_weights", wts)):
            if t is not None:
                parts[k].append(t)
    if b == 0:
        parts = {"mask": [seg.new_zeros((0, h, w), dtype=torch.int64)], "kpt_2d": [ver.new_zeros((0, vn_2 // 2, 2))],
                 "var": [ver.new_zeros((0, vn_2 // 2, 2, 2))], "var_weights": [ver.new_zeros((0, vn_2 // 2, 3))]}
    cat = lambda ts: ts[0] if len(ts) == 1 else torch.cat(ts)                 # noqa: E731
    output.update({"mask": cat(parts["mask"]), "kpt_2d": cat(parts["kpt_2d"])})
    if un_pnp:
        output["var"] = cat(parts["var"])
        if weights:
            output["var_weights"] = cat(parts["var_weights"])
    return output
