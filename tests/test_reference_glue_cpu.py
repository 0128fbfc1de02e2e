"""CPU: the reference's own Python (byte code under oracle/_ref, tests/refglue.py) loads and runs -- here on CPU tensors
with the oracle's two kernels underneath, which must reproduce the committed golden fixtures bit for bit (they were
generated exactly so, from the SOURCE files, by tests/golden/make_golden.py).  The GPU counterpart
(tests/test_gpu_reference_glue.py) runs the same code objects on top of the HIP extension."""
import os
import types

import numpy as np
import pytest
import torch

from tests import refglue

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
torch11 = refglue.torch11_fixture()        # the torch-1.1 shims the reference's code needs: held for this module only, then restored
needs_pyc = pytest.mark.skipif(not refglue.available(), reason="oracle/_ref/*.pyc not built (make -C oracle _ref_py needs /root/reference)")


def oracle_extension(oracle):
    m = types.ModuleType("lib.csrc.ransac_voting.ransac_voting")
    m.generate_hypothesis = lambda direct, coords, idxs: torch.from_numpy(
        oracle.generate_hypothesis(direct.numpy(), coords.numpy(), idxs.numpy()))
    m.voting_for_hypothesis = lambda direct, coords, hyp, inl, thresh: oracle.voting_for_hypothesis(
        direct.numpy(), coords.numpy(), hyp.numpy(), inl.numpy(), thresh)
    return m


@needs_pyc
def test_reference_glue_bytecode_reproduces_the_golden_fixtures_on_cpu(oracle):
    ref = refglue.load_glue(extension=oracle_extension(oracle))
    c = dict(np.load(os.path.join(GOLD, "v3_basic.npz")))
    d = refglue.Draws(ref.ransac_voting, idxs=list(c["idxs"][:2]))
    ref.ransac_voting = d
    out = ref.ransac_voting_layer_v3(torch.from_numpy(c["mask"]), torch.from_numpy(c["vertex"]), int(c["hn"]),
                                     inlier_thresh=float(c["thresh"]))
    assert np.array_equal(out.numpy(), c["out"])
    assert len(d.drawn) == 2 and not d.idxs
    c = dict(np.load(os.path.join(GOLD, "v3_subsample.npz")))
    d = refglue.Draws(d.ext, idxs=list(c["idxs"]), selection=list(c["selection"]))
    ref.ransac_voting = d
    with d.patch_uniform():
        out = ref.ransac_voting_layer_v3(torch.from_numpy(c["mask"]), torch.from_numpy(c["vertex"]), int(c["hn"]),
                                         inlier_thresh=float(c["thresh"]), max_num=int(c["max_num"]))
    assert np.array_equal(out.numpy(), c["out"])


@needs_pyc
def test_reference_resnet18_bytecode_loads_with_stubbed_config_and_backbone():
    mod, cfg = refglue.load_resnet18(un_pnp=True)
    fn = mod.Resnet18.decode_keypoint
    assert fn.__code__.co_filename.endswith("lib/networks/pvnet/resnet18.py")      # the reference's file, not a mirror
    assert {"ransac_voting_layer_v3", "estimate_voting_distribution_with_mean", "argmax"} <= set(fn.__code__.co_names)
    # its voting layers are whatever lib.csrc.ransac_voting.ransac_voting_gpu exports: the product's
    import lib.csrc.ransac_voting.ransac_voting_gpu as drop_in
    assert mod.ransac_voting_layer_v3 is drop_in.ransac_voting_layer_v3
    assert mod.estimate_voting_distribution_with_mean is drop_in.estimate_voting_distribution_with_mean
    assert mod.cfg is cfg


def test_torch11_shims_are_restored_behind_the_modules_that_need_them():
    """the shims used to stay patched into torch for the rest of the session.  Inside this module they are held by the
    module fixture; a nested install / remove pair leaves them in place, and the pyc check refuses byte code of another Python."""
    assert refglue._shim_state["depth"] >= 1 and hasattr(torch, "solve")
    before = torch.Tensor.masked_select
    refglue.install_torch11_shims()
    refglue.remove_torch11_shims()
    assert torch.Tensor.masked_select is before and refglue._shim_state["depth"] >= 1
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".pyc") as f:
        f.write(b"\x00\x00\x00\x00rest")
        f.flush()
        assert not refglue._pyc_matches_this_python(f.name)
    assert not refglue._pyc_matches_this_python("/nonexistent.pyc")
