"""Test infrastructure: THE REFERENCE'S OWN Python for the voting path, executed from byte code.

``make -C oracle _ref_py`` byte-compiles ``/root/reference/lib/csrc/ransac_voting/ransac_voting_gpu.py`` and
``/root/reference/lib/networks/pvnet/resnet18.py`` where they lie into ``oracle/_ref/*.pyc`` (artefacts: git-ignored,
shipped to the GPU box like the ``.so`` files; no reference source enters the tree).  This module loads those code objects

  * ``load_glue()``     -- the reference's ``ransac_voting_gpu`` module.  Its line 2, ``import
                           lib.csrc.ransac_voting.ransac_voting as ransac_voting``, resolves to whatever that import path
                           holds: the product's HIP extension (this repository's ``lib/`` shim) unless the caller puts a
                           fake module there first (``extension=``: the CPU tests use the oracle's kernels);
  * ``load_resnet18()`` -- the reference's ``lib.networks.pvnet.resnet18`` module with ``lib.config.cfg`` and the
                           backbone import (``from .resnet import resnet18``) stubbed; ``Resnet18.decode_keypoint`` is
                           the unmodified code object (resnet18.py:65-76) and imports the voting layers from
                           ``lib.csrc.ransac_voting.ransac_voting_gpu`` -- the product's.

and supplies what torch 2.10 no longer has (the same two shims as tests/golden/make_golden.py):
``torch.solve`` with torch-1.1 semantics and ``masked_select`` with a uint8 mask.

``Draws`` replays recorded random draws inside the reference's glue: the ``random_`` index pairs of P:145 / P:235 are
overwritten in place when the glue hands them to ``generate_hypothesis``, the ``uniform_`` field of P:136 / P:220 is
served from the fixture -- so the reference's code, the oracle and the product all see the same draws.
"""
import importlib.machinery
import importlib.util
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
GLUE_PYC = os.path.join(REF_DIR, "ref_ransac_voting_gpu.pyc")
RESNET_PYC = os.path.join(REF_DIR, "ref_resnet18.pyc")


def _pyc_matches_this_python(path):
    """byte code is bound to the interpreter's minor version: its first four bytes are importlib.util.MAGIC_NUMBER"""
    try:
        with open(path, "rb") as f:
            return f.read(4) == importlib.util.MAGIC_NUMBER
    except OSError:
        return False


def available():
    """Both .pyc files exist AND were compiled by this Python's minor version (a mismatch used to surface as an
    ImportError inside the tests instead of a skip)."""
    return _pyc_matches_this_python(GLUE_PYC) and _pyc_matches_this_python(RESNET_PYC)


_shim_state = {"depth": 0, "saved": None}


def install_torch11_shims():
    """torch.solve with torch-1.1 semantics and masked_select with a uint8 mask -- what the reference's glue calls at RUN time.
    Nested installs are counted; ``remove_torch11_shims`` restores the originals when the last one leaves (the
    patches used to stay for the whole pytest session and every later test ran against a patched torch).  The test modules
    that run the reference's code hold them through the module-scoped ``torch11`` fixture below."""
    if _shim_state["depth"] == 0:
        missing = object()
        _shim_state["saved"] = (getattr(torch, "solve", missing), torch.Tensor.masked_select, missing)

        def solve(B, A):                           # torch 1.1: torch.solve(B, A) -> (X, LU), raises on a singular A
            X, info = torch.linalg.solve_ex(A, B)
            if (info != 0).any() or not torch.isfinite(X).all():
                raise RuntimeError("solve: U(i,i) is zero, singular U.")
            return X, None
        torch.solve = solve
        orig = torch.Tensor.masked_select

        def masked_select(self, mask):
            return orig(self, mask.bool() if mask.dtype == torch.uint8 else mask)
        torch.Tensor.masked_select = masked_select
    _shim_state["depth"] += 1


def remove_torch11_shims():
    if _shim_state["depth"] == 0:
        return
    _shim_state["depth"] -= 1
    if _shim_state["depth"] == 0:
        solve0, ms0, missing = _shim_state["saved"]
        if solve0 is missing:
            del torch.solve
        else:
            torch.solve = solve0
        torch.Tensor.masked_select = ms0
        _shim_state["saved"] = None


def torch11_fixture():
    """-> a module-scoped autouse pytest fixture: shims installed for the module's tests, originals restored behind them."""
    import pytest

    @pytest.fixture(scope="module", autouse=True)
    def torch11():
        install_torch11_shims()
        yield
        remove_torch11_shims()
    return torch11


def _load_pyc(name, path, package=None):
    loader = importlib.machinery.SourcelessFileLoader(name, path)
    spec = importlib.util.spec_from_loader(name, loader)
    mod = importlib.util.module_from_spec(spec)
    if package:
        mod.__package__ = package
    loader.exec_module(mod)
    return mod


class Draws:
    """Replays recorded draws inside the reference's glue and records what its kernels were called with."""

    def __init__(self, ext, idxs=None, selection=None):
        self.ext = ext
        self.idxs = list(idxs) if idxs is not None else None      # one [hn,vn,2] int32 array per generate_hypothesis with a NEW idxs tensor
        self.selection = list(selection) if selection is not None else None
        self.calls = []                                           # (name, args..., result) of every kernel call
        self.drawn = []                                           # the idxs tensors the glue ended up using, in order
        self._last_ptr = None

    # -- the extension module surface the reference's glue uses (ransac_voting_gpu.py:152, :156, :183, :237, :241)
    def generate_hypothesis(self, direct, coords, idxs):
        if idxs.data_ptr() != self._last_ptr:                     # a fresh draw (the v3 loop re-uses its tensor, P:145 vs P:150)
            self._last_ptr = idxs.data_ptr()
            if self.idxs is not None:
                idxs.copy_(torch.as_tensor(self.idxs.pop(0)).to(idxs.device))
            self.drawn.append(idxs.clone())
        hyp = self.ext.generate_hypothesis(direct, coords, idxs)
        self.calls.append(("generate_hypothesis", direct, coords, idxs.clone(), hyp))
        return hyp

    def voting_for_hypothesis(self, direct, coords, hypo_pts, inliers, thresh):
        r = self.ext.voting_for_hypothesis(direct, coords, hypo_pts, inliers, thresh)
        self.calls.append(("voting_for_hypothesis", direct, coords, hypo_pts, inliers, thresh))
        return r

    def __getattr__(self, name):
        return getattr(self.ext, name)

    # -- uniform_ of P:136 / P:220
    def patch_uniform(self):
        draws = self
        orig = torch.Tensor.uniform_

        def uniform_(t, *a, **k):
            if draws.selection:
                return t.copy_(torch.as_tensor(draws.selection.pop(0)).to(t.device))
            return orig(t, *a, **k)

        class _Ctx:
            def __enter__(self_c):
                torch.Tensor.uniform_ = uniform_

            def __exit__(self_c, *exc):
                torch.Tensor.uniform_ = orig
        return _Ctx()


def load_glue(extension=None):
    """The reference's ransac_voting_gpu module, its kernels = ``extension`` (default: lib.csrc.ransac_voting.ransac_voting,
    i.e. the product's HIP module).  The module global ``ransac_voting`` may be replaced afterwards (``Draws``)."""
    if not os.path.exists(GLUE_PYC):
        raise FileNotFoundError("%s is missing: run `make -C oracle _ref_py` where /root/reference exists" % GLUE_PYC)
    assert _shim_state["depth"] > 0, "hold the torch-1.1 shims while the reference's code runs (refglue.torch11_fixture / install_torch11_shims)"
    if extension is None:
        import lib.csrc.ransac_voting.ransac_voting as extension   # noqa: F811  (this repository's shim -> the HIP module)
        mod = _load_pyc("ref_ransac_voting_gpu", GLUE_PYC)
        assert mod.ransac_voting is extension
    else:
        import lib.csrc.ransac_voting as pkg
        key = "lib.csrc.ransac_voting.ransac_voting"
        saved, saved_attr = sys.modules.get(key), getattr(pkg, "ransac_voting", None)
        sys.modules[key] = extension
        pkg.ransac_voting = extension
        try:
            mod = _load_pyc("ref_ransac_voting_gpu", GLUE_PYC)
        finally:
            if saved is not None:
                sys.modules[key] = saved
            else:
                del sys.modules[key]
            if saved_attr is not None:
                pkg.ransac_voting = saved_attr
            else:
                del pkg.ransac_voting
        assert mod.ransac_voting is extension
    return mod


def load_resnet18(un_pnp):
    """The reference's lib/networks/pvnet/resnet18.py; returns (module, cfg).  ``cfg.test.un_pnp`` is a live attribute."""
    if not os.path.exists(RESNET_PYC):
        raise FileNotFoundError("%s is missing: run `make -C oracle _ref_py` where /root/reference exists" % RESNET_PYC)
    assert _shim_state["depth"] > 0, "hold the torch-1.1 shims while the reference's code runs (refglue.torch11_fixture / install_torch11_shims)"
    import lib                                                   # this repository's lib/ (csrc only)
    cfg = types.SimpleNamespace(test=types.SimpleNamespace(un_pnp=bool(un_pnp)))
    stubs = {}

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__path__ = []
        for k, v in attrs.items():
            setattr(m, k, v)
        stubs[name] = m
        return m

    def _no_backbone(*a, **k):
        raise RuntimeError("the backbone is not part of this repository (SURVEY 2: out of scope)")
    stub("lib.config", cfg=cfg)                                   # lib/config needs yacs + open3d
    stub("lib.networks")
    stub("lib.networks.pvnet")
    stub("lib.networks.pvnet.resnet", resnet18=_no_backbone)      # `from .resnet import resnet18`: only the constructor uses it
    saved = {k: sys.modules.get(k) for k in stubs}
    sys.modules.update(stubs)
    try:
        mod = _load_pyc("lib.networks.pvnet.resnet18", RESNET_PYC, package="lib.networks.pvnet")
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    del lib
    return mod, cfg
