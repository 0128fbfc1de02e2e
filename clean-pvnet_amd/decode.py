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
                    weights=False):
    """In-place update of ``output`` exactly like the reference method: adds ``mask`` [b,h,w] int64, ``kpt_2d``
    [b,vn,2] and -- with ``un_pnp`` (``cfg.test.un_pnp``, config.py:75) -- ``var`` [b,vn,2,2].

    ``output['seg']`` is [b,c,h,w] float32 logits, ``output['vertex']`` [b,2*vn,h,w] float32; both may be channel
    slices of one network output tensor (resnet18.py:93-94), no copy is made.

    ``un_pnp`` with a two-class ``seg`` (PVNet's) runs ``ransac_voting_layer_v3`` and
    ``estimate_voting_distribution_with_mean`` (resnet18.py:71-72) as ONE pass -- one mask scan, one compaction, one
    hypothesis and one inlier-count launch for the 512 + 4096 hypotheses -- with results bit-identical to the two calls
    on the same draws; ``idxs_est`` [b,4096,vn,2] injects the estimate's index pairs like ``idxs`` does for v3, and
    ``weights=True`` also stores ``var_weights`` [b,vn,3] = (wxx,wxy,wyy) of ``inv(sqrtm(var))``, what the evaluator
    feeds uncertainty_pnp (evaluators/linemod/pvnet.py:118-130).
    """
    seg = output["seg"]
    ver = output["vertex"]
    b, vn_2, h, w = ver.shape
    vertex = ver.permute(0, 2, 3, 1).view(b, h, w, vn_2 // 2, 2)              # resnet18.py:66-68, a strided view
    if un_pnp and seg.shape[1] == 2 and b <= _MAX_BATCH:
        # resnet18.py:71-72 fused; the estimate's defaults (P:202): ceil(4096 / 256) rounds of 256 hypotheses
        kpt, mask, var, w, _win, _tn = _ext.decode_keypoint_un_pnp(seg.float(), vertex, 512, 4096, 0.99, 5, 30000, idxs,
                                                                   idxs_est, selection, _next_seed(), _POLICY[singular])
        output.update({"mask": mask, "kpt_2d": kpt, "var": var})
        if weights:
            output["var_weights"] = w
        return output
    if un_pnp:
        hn, max_num = 512, 30000                                             # resnet18.py:71
    else:
        hn, max_num = 128, 100                                               # resnet18.py:75
    kpt, mask, _win, _tn = _ext.decode_keypoint_v3(seg.float(), vertex, hn, 0.99, 5, max_num, idxs, selection,
                                                   _next_seed(), _POLICY[singular])
    if un_pnp:
        res = estimate_voting_distribution_with_mean(mask, vertex, kpt, idxs=idxs_est, return_weights=weights)   # resnet18.py:72
        output.update({"mask": mask, "kpt_2d": res[0], "var": res[1]})
        if weights:
            output["var_weights"] = res[2]
    else:
        output.update({"mask": mask, "kpt_2d": kpt})
    return output
