"""``lib.csrc.ransac_voting.ransac_voting_gpu`` -- the module ``resnet18.py:5`` imports."""
from lib import _register_clean_pvnet_amd

_register_clean_pvnet_amd()
from clean_pvnet_amd.ransac_voting_gpu import (b_inv, estimate_voting_distribution_with_mean,  # noqa: E402,F401
                                               ransac_voting_layer, ransac_voting_layer_v3)
