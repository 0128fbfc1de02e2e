"""Host-side mirror of ``lib/csrc/nn/nn_utils.py`` of clean-pvnet (the ADD-S nearest-neighbour search).

The reference binds ``findNearestPointIdxLauncher`` of its CUDA library through cffi (nn_utils.py:1,18); cffi is not
needed here: the same C symbol, exported by ``libpvnet_nn.so`` (HIP, gfx950), is bound with ctypes.  Same function name,
arguments (numpy arrays on the host) and return value as the reference; there is no CPU fallback.
"""
import ctypes
import os

import numpy as np

_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpvnet_nn.so")
try:
    _lib = ctypes.CDLL(_LIB)
except OSError as e:
    raise ImportError("clean_pvnet_amd.nn_utils: libpvnet_nn.so is not built (run `python __graft_entry__.py`); "
                      "there is no CPU fallback. Original error: %s" % (e,)) from e
_lib.findNearestPointIdxLauncher.restype = None
_lib.findNearestPointIdxLauncher.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int, ctypes.c_int]


def find_nearest_point_idx(ref_pts, que_pts):
    """nn_utils.py:5-20: ``ref_pts [pn1,dim]``, ``que_pts [pn2,dim]`` (dim 2 or 3) -> ``idxs [pn2]`` int32, the index of the
    nearest reference point of every query point (first minimum wins)."""
    assert(ref_pts.shape[1] == que_pts.shape[1] and 1 < que_pts.shape[1] <= 3)
    pn1 = ref_pts.shape[0]
    pn2 = que_pts.shape[0]
    dim = ref_pts.shape[1]
    ref_pts = np.ascontiguousarray(ref_pts[None, :, :], np.float32)
    que_pts = np.ascontiguousarray(que_pts[None, :, :], np.float32)
    idxs = np.full([1, pn2], -1, np.int32)
    _lib.findNearestPointIdxLauncher(ref_pts.ctypes.data, que_pts.ctypes.data, idxs.ctypes.data, 1, pn1, pn2, dim, 0)
    if pn2 and (idxs < 0).any():
        raise RuntimeError("findNearestPointIdxLauncher failed (no GPU? see stderr); there is no CPU fallback")
    return idxs[0]
