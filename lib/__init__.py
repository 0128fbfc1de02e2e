"""Drop-in import path of the reference: ``lib.csrc.ransac_voting.ransac_voting_gpu``.

clean-pvnet imports the voting layers as
``from lib.csrc.ransac_voting.ransac_voting_gpu import ransac_voting_layer, ransac_voting_layer_v3,
estimate_voting_distribution_with_mean`` (lib/networks/pvnet/resnet18.py:5) and the extension as
``import lib.csrc.ransac_voting.ransac_voting`` (lib/csrc/ransac_voting/ransac_voting_gpu.py:2).
Copying this ``lib/csrc/ransac_voting`` directory over the reference's (next to
``clean-pvnet_amd/``) keeps both imports working; nothing else of ``lib`` is provided here.

The implementation lives in ``clean-pvnet_amd/`` -- a directory name Python cannot import
directly -- so it is registered once under the module name ``clean_pvnet_amd``.
"""
import importlib.util
import os
import sys


def _register_clean_pvnet_amd():
    if "clean_pvnet_amd" in sys.modules:
        return sys.modules["clean_pvnet_amd"]
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "clean-pvnet_amd")
    spec = importlib.util.spec_from_file_location("clean_pvnet_amd", os.path.join(root, "__init__.py"),
                                                  submodule_search_locations=[root])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["clean_pvnet_amd"] = mod
    try:
        spec.loader.exec_module(mod)
    except BaseException:
        del sys.modules["clean_pvnet_amd"]
        raise
    return mod
