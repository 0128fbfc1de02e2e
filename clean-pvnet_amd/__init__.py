"""clean-pvnet's RANSAC voting hot path, MI355X-native (gfx950 HIP kernels behind the
reference's own Python / extension-module interface).

The directory name contains a hyphen, so it is imported under the module name
``clean_pvnet_amd`` (see ``lib/__init__.py`` at the repository root, which registers it
and provides the reference's import path ``lib.csrc.ransac_voting.ransac_voting_gpu``).

There is no CPU implementation in this package: importing the voting layers without the
built extension (``python __graft_entry__.py`` builds it) raises ImportError, and calling
them with CPU tensors raises RuntimeError.
"""
from .ransac_voting_gpu import (b_inv, estimate_voting_distribution_with_mean,  # noqa: F401
                                ransac_voting_layer, ransac_voting_layer_v3, uncertainty_pnp_weights)

from .decode import decode_keypoint  # noqa: F401,E402

__all__ = ["ransac_voting_layer", "ransac_voting_layer_v3", "estimate_voting_distribution_with_mean", "b_inv",
           "decode_keypoint", "uncertainty_pnp_weights"]
