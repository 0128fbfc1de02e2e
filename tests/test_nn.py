"""the ADD-S nearest-neighbour search (lib/csrc/nn).  CPU: the oracle against a numpy brute force and
the library's exported symbols; GPU: the HIP kernel against the oracle, bit-exact indices (first minimum wins)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NNLIB = os.path.join(ROOT, "clean-pvnet_amd", "libpvnet_nn.so")


def _np_nearest(ref, que, exclude_self=False):
    f = np.float32
    d = np.zeros((que.shape[0], ref.shape[0]), f)
    for k in range(ref.shape[1]):                                  # ((dx^2 + dy^2) + dz^2), one binary32 rounding per op
        diff = ref[None, :, k].astype(f) - que[:, None, k].astype(f)
        d = d + diff * diff if k else diff * diff
    if exclude_self:
        n = min(d.shape)
        d[np.arange(n), np.arange(n)] = np.inf
    return np.argmin(d, 1).astype(np.int32)                        # argmin returns the first minimum


def _clouds(dim, pn1, pn2, seed):
    rng = np.random.RandomState(seed)
    ref = rng.randn(pn1, dim).astype(np.float32) * 0.05
    que = (ref[rng.randint(0, pn1, pn2)] + rng.randn(pn2, dim).astype(np.float32) * 0.002).astype(np.float32)
    if pn1 > 10:
        ref[7] = ref[3]                                            # exact duplicates: ties -> the lower index wins
        que[0] = ref[3]
    return ref, que


@pytest.mark.parametrize("dim,pn1,pn2", [(3, 1, 5), (3, 500, 300), (2, 1000, 257), (3, 2000, 1)])
def test_oracle_matches_numpy_brute_force(oracle, dim, pn1, pn2):
    ref, que = _clouds(dim, pn1, pn2, 1)
    np.testing.assert_array_equal(oracle.find_nearest_point_idx(ref, que), _np_nearest(ref, que))
    if pn1 > 10:
        assert oracle.find_nearest_point_idx(ref, que)[0] == 3


def test_nn_library_exports_what_the_header_declares():
    txt = open(os.path.join(ROOT, "include", "pvnet_nn.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set(re.findall(r"\b(findNearestPointIdxLauncher|pvv_nn_[a-z_]+)\s*\(", txt))
    assert names == {"findNearestPointIdxLauncher", "pvv_nn_find_nearest"}
    L = ctypes.CDLL(NNLIB)
    for n in names:
        assert hasattr(L, n)
    # the reference's cffi declares exactly this prototype (lib/csrc/nn/src/ext.h)
    import lib.csrc.nn.nn_utils as drop_in
    assert callable(drop_in.find_nearest_point_idx)


def test_no_cpu_fallback_without_a_gpu(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from clean_pvnet_amd.nn_utils import find_nearest_point_idx
    ref, que = _clouds(3, 50, 20, 2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        find_nearest_point_idx(ref, que)


@pytest.mark.gpu
@pytest.mark.parametrize("dim,pn1,pn2", [(3, 1, 5), (3, 5841, 5841), (2, 1000, 257), (3, 3000, 1), (2, 4097, 1025)])
def test_hip_nearest_neighbour_bit_exact(oracle, pkg, gpu, dim, pn1, pn2):
    from lib.csrc.nn.nn_utils import find_nearest_point_idx       # the evaluator's import path
    ref, que = _clouds(dim, pn1, pn2, 3)
    got = find_nearest_point_idx(ref, que)
    assert got.dtype == np.int32 and got.shape == (pn2,)
    np.testing.assert_array_equal(got, oracle.find_nearest_point_idx(ref, que))


@pytest.mark.gpu
def test_hip_nearest_neighbour_device_pointers_batched_exclude_self(oracle, pkg, gpu):
    import torch
    L = ctypes.CDLL(NNLIB)
    L.pvv_nn_find_nearest.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 5 + [ctypes.c_void_p]
    b, pn, dim = 3, 777, 3
    rng = np.random.RandomState(4)
    pts = rng.randn(b, pn, dim).astype(np.float32)
    d = torch.from_numpy(pts).to(gpu)
    idx = torch.full((b, pn), -1, dtype=torch.int32, device=gpu)
    rc = L.pvv_nn_find_nearest(d.data_ptr(), d.data_ptr(), idx.data_ptr(), b, pn, pn, dim, 1,
                               torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    got = idx.cpu().numpy()
    for bi in range(b):
        want = oracle.find_nearest_point_idx(pts[bi], pts[bi], exclude_self=True)
        np.testing.assert_array_equal(got[bi], want)
        assert (got[bi] != np.arange(pn)).all()                    # a point is never its own neighbour


@pytest.mark.gpu
@pytest.mark.parametrize("dim,b,pn1,pn2,excl", [(3, 1, 5841, 5841, 0), (3, 2, 700, 300, 0), (2, 1, 1500, 1500, 1), (3, 1, 1, 9, 0),
                                                (2, 3, 333, 1025, 0)])
def test_reference_nn_kernel_pins_oracle_and_product(oracle, pkg, gpu, dim, b, pn1, pn2, excl):
    """oracle/_ref/libref_nn.so = the reference's own lib/csrc/nn/src/nearest_neighborhood.cu, compiled where it lies with
    hipcc through oracle/ref_shim/ and run on the MI355X through its own launcher (host pointers).  Indices of the
    reference == oracle == product (same C symbol, same arguments), ties and duplicates included."""
    path = os.path.join(ROOT, "oracle", "_ref", "libref_nn.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libref_nn.so is built only where /root/reference is mounted (see tests/test_host.py)")
    R, P = ctypes.CDLL(path), ctypes.CDLL(NNLIB)
    args = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 5
    R.ref_findNearestPointIdxLauncher.argtypes = args
    P.findNearestPointIdxLauncher.argtypes = args
    rng = np.random.RandomState(9)
    ref = (rng.randn(b, pn1, dim) * 0.05).astype(np.float32)
    if excl:
        que = ref.copy()
    else:
        que = (ref[:, rng.randint(0, pn1, pn2)] + rng.randn(b, pn2, dim).astype(np.float32) * 0.002).astype(np.float32)
    if pn1 > 10:
        ref[:, 7] = ref[:, 3]                                      # exact duplicates: the lower index must win
        if not excl:
            que[:, 0] = ref[:, 3]
    got_ref = np.full((b, pn2), -1, np.int32)
    got_new = np.full((b, pn2), -2, np.int32)
    R.ref_findNearestPointIdxLauncher(ref.ctypes.data, que.ctypes.data, got_ref.ctypes.data, b, pn1, pn2, dim, excl)
    P.findNearestPointIdxLauncher(ref.ctypes.data, que.ctypes.data, got_new.ctypes.data, b, pn1, pn2, dim, excl)
    np.testing.assert_array_equal(got_new, got_ref)
    for bi in range(b):
        np.testing.assert_array_equal(oracle.find_nearest_point_idx(ref[bi], que[bi], exclude_self=bool(excl)), got_ref[bi])
