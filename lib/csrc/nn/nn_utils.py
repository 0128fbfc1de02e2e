"""``lib.csrc.nn.nn_utils`` -- the module ``lib/evaluators/linemod/pvnet.py:20`` imports for the ADD-S metric."""
from lib import _register_clean_pvnet_amd

_register_clean_pvnet_amd()
from clean_pvnet_amd.nn_utils import find_nearest_point_idx  # noqa: E402,F401
