"""GPU: the HIP path against the committed golden fixtures produced by the reference's own Python glue
(tests/golden/make_golden.py).  Same tolerances as tests/test_oracle.py uses for the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name, gpu):
    c = dict(np.load(os.path.join(GOLD, name + ".npz")))
    t = {k: torch.from_numpy(v).to(gpu) for k, v in c.items() if isinstance(v, np.ndarray) and v.ndim > 0}
    return c, t


def test_v3_basic(pkg, gpu):
    from clean_pvnet_amd.ransac_voting_gpu import ransac_voting_layer_v3
    c, t = gold("v3_basic", gpu)
    out = ransac_voting_layer_v3(t["mask"], t["vertex"], int(c["hn"]), inlier_thresh=float(c["thresh"]), idxs=t["idxs"])
    np.testing.assert_allclose(out.cpu().numpy(), c["out"], rtol=0, atol=2e-4)
    assert (out[2] == 0).all()


def test_v3_subsample(pkg, gpu):
    from clean_pvnet_amd.ransac_voting_gpu import ransac_voting_layer_v3
    c, t = gold("v3_subsample", gpu)
    out = ransac_voting_layer_v3(t["mask"], t["vertex"], int(c["hn"]), inlier_thresh=float(c["thresh"]),
                                 max_num=int(c["max_num"]), idxs=t["idxs"], selection=t["selection"])
    np.testing.assert_allclose(out.cpu().numpy(), c["out"], rtol=0, atol=5e-4)


def test_v3_singular_reference_policy_is_default(pkg, gpu):
    from clean_pvnet_amd.ransac_voting_gpu import ransac_voting_layer_v3
    c, t = gold("v3_singular", gpu)
    out = ransac_voting_layer_v3(t["mask"], t["vertex"], int(c["hn"]), inlier_thresh=float(c["thresh"]), idxs=t["idxs"])
    np.testing.assert_allclose(out.cpu().numpy(), c["out"], rtol=2e-6, atol=1e-3)


def test_v1_layer(pkg, gpu):
    from clean_pvnet_amd.ransac_voting_gpu import ransac_voting_layer
    c, t = gold("v1_basic", gpu)
    out = ransac_voting_layer(t["mask"], t["vertex"], int(c["hn"]), inlier_thresh=float(c["thresh"]), idxs=t["idxs"])
    np.testing.assert_allclose(out.cpu().numpy(), c["out"], rtol=0, atol=2e-4)


def test_estimate(pkg, gpu):
    from clean_pvnet_amd.ransac_voting_gpu import estimate_voting_distribution_with_mean
    c, t = gold("estimate_basic", gpu)
    mean, cov = estimate_voting_distribution_with_mean(t["mask"], t["vertex"], t["mean"], int(c["round_hyp_num"]),
                                                       int(c["min_hyp_num"]), idxs=t["idxs"])
    np.testing.assert_allclose(cov.cpu().numpy(), c["cov"], rtol=2e-5, atol=1e-4)
    assert mean is t["mean"]
