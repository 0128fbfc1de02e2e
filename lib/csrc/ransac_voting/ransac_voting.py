"""``lib.csrc.ransac_voting.ransac_voting`` -- the extension module of the reference
(built in place there by setup.py, ransac_voting.cpp:102-107).  Here it re-exports the
pybind11 module built from ``clean-pvnet_amd/csrc`` (same four functions, same signatures)."""
from lib import _register_clean_pvnet_amd

_register_clean_pvnet_amd()
from clean_pvnet_amd.ransac_voting import *  # noqa: E402,F401,F403
from clean_pvnet_amd.ransac_voting import (generate_hypothesis, generate_hypothesis_vanishing_point,  # noqa: E402,F401
                                           voting_for_hypothesis, voting_for_hypothesis_vanishing_point)
