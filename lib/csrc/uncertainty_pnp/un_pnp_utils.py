"""``lib.csrc.uncertainty_pnp.un_pnp_utils`` -- the module ``lib/evaluators/linemod/pvnet.py`` imports for
``cfg.test.un_pnp`` (``un_pnp_utils.uncertainty_pnp``, evaluators/linemod/pvnet.py:130)."""
from lib import _register_clean_pvnet_amd

_register_clean_pvnet_amd()
from clean_pvnet_amd.un_pnp_utils import (uncertainty_pnp, uncertainty_pnp_batched, uncertainty_pnp_v2)  # noqa: E402,F401
