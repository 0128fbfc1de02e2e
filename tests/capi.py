"""ctypes view of the C ABI declared in include/pvnet_vote.h -- what a non-Python host would bind.
The GPU parity tests call the library through this (raw device pointers + the current HIP stream)
as well as through the pybind11 module."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pvnet_vote.h")
# PVV_LIBPATH: tools/variant_time.py points the same binding at an experimental build of the library
LIBPATH = os.environ.get("PVV_LIBPATH") or os.path.join(ROOT, "clean-pvnet_amd", "libpvnet_vote.so")

c_i32, c_i64, c_u64, c_f32, vp, sz = (ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64, ctypes.c_float,
                                       ctypes.c_void_p, ctypes.c_size_t)


class Problem(ctypes.Structure):
    _fields_ = [("B", c_i32), ("H", c_i32), ("W", c_i32), ("K", c_i32), ("hn", c_i32),
                ("mask_elem_size", c_i32), ("min_num", c_i32), ("max_num", c_i32), ("cap", c_i32),
                ("singular_policy", c_i32), ("inlier_thresh", c_f32),
                ("mask_stride", c_i64 * 3), ("vertex_stride", c_i64 * 5), ("seed", c_u64),
                ("seg_classes", c_i32), ("first_image", c_i32), ("seg_stride", c_i64 * 4),
                ("count_kernel", c_i32), ("flags", c_i32), ("d_draws_out", vp), ("ev_count_begin", vp), ("ev_count_end", vp),
                ("d_status", vp), ("ev_marks", vp)]


def declared_symbols():
    """Every function the header declares (name, followed by an opening parenthesis at top level)."""
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pvv_[a-z0-9_]+)\s*\(", txt)))


_lib = None


def load():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(LIBPATH)
        L.pvv_last_error.restype = ctypes.c_char_p
        L.pvv_workspace_bytes.restype = sz
        L.pvv_workspace_bytes.argtypes = [ctypes.POINTER(Problem)]
        L.pvv_default_cap.restype = c_i32
        L.pvv_default_cap.argtypes = [c_i32, c_i32, c_i32]
        legacy = [vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.pvv_generate_hypothesis.argtypes = legacy + [vp]
        L.pvv_generate_hypothesis_vanishing_point.argtypes = legacy + [vp]
        L.pvv_voting_for_hypothesis.argtypes = legacy + [c_f32, vp]
        L.pvv_voting_for_hypothesis_vanishing_point.argtypes = legacy + [c_f32, vp]
        L.pvv_count_inliers.argtypes = legacy + [c_f32, vp]
        L.pvv_ransac_voting_v3.argtypes = [ctypes.POINTER(Problem), vp, vp, vp, vp, vp, sz, vp, vp, vp, vp]
        L.pvv_decode_keypoint_v3.argtypes = [ctypes.POINTER(Problem), vp, vp, vp, vp, vp, sz, vp, vp, vp, vp, vp]
        L.pvv_estimate_voting_distribution.argtypes = [ctypes.POINTER(Problem), vp, vp, vp, vp, vp, vp, sz,
                                                       vp, vp, vp, vp, vp, vp]
        L.pvv_rerun_count_kernel.argtypes = [ctypes.POINTER(Problem), vp, sz, ctypes.c_int, vp]
        L.pvv_stream_read_probe.argtypes = [vp, sz, vp, vp]
        L.pvv_stage_hint_query.argtypes = [ctypes.POINTER(c_f32), ctypes.POINTER(c_f32), ctypes.POINTER(Problem), vp]
        L.pvv_shutdown.argtypes = []
        L.pvv_estimate_counts_in_stages.argtypes = [ctypes.POINTER(Problem)]
        _lib = L
    return _lib


def problem(mask, vertex, hn, thresh, min_num=5, max_num=30000, policy=0, seed=0, count_kernel=0, draws_out=None,
            first_image=0, cap=None, status=None, flags=0):
    L = load()
    p = Problem()
    p.B, p.H, p.W, p.K, _ = vertex.shape
    p.hn = hn
    p.mask_elem_size = mask.element_size()
    p.min_num, p.max_num = min_num, max_num
    p.cap = L.pvv_default_cap(p.H, p.W, max_num) if cap is None else cap
    p.count_kernel = count_kernel
    p.flags = flags
    p.first_image = first_image
    p.d_draws_out = None if draws_out is None else draws_out.data_ptr()
    p.d_status = None if status is None else status.data_ptr()
    p.singular_policy = policy
    p.inlier_thresh = thresh
    p.mask_stride[:] = mask.stride()
    p.vertex_stride[:] = vertex.stride()
    p.seed = seed
    return p


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def check(code):
    assert code == 0, "pvv call failed (%d): %s" % (code, load().pvv_last_error().decode())


def v3(mask, vertex, hn, thresh, idxs=None, selection=None, **kw):
    """pvv_ransac_voting_v3 on torch CUDA tensors -> (out [B,K,2], win_counts [B,K], tn [B])."""
    import torch
    L = load()
    if selection is not None:      # injected draws may keep any number of pixels: reserve the whole image (as the pybind shim does)
        kw.setdefault("cap", vertex.shape[1] * vertex.shape[2])
    p = problem(mask, vertex, hn, thresh, **kw)
    n = L.pvv_workspace_bytes(ctypes.byref(p))
    assert n > 0, L.pvv_last_error()
    dev = vertex.device
    ws = torch.empty(n, dtype=torch.uint8, device=dev)
    out = torch.empty(p.B, p.K, 2, device=dev)
    win = torch.empty(p.B, p.K, dtype=torch.int32, device=dev)
    tn = torch.empty(p.B, dtype=torch.int32, device=dev)
    check(L.pvv_ransac_voting_v3(ctypes.byref(p), ptr(mask), ptr(vertex), ptr(idxs), ptr(selection), ptr(ws), n,
                                 ptr(out), ptr(win), ptr(tn), stream()))
    return out, win, tn


def estimate(mask, vertex, mean, hn_total, thresh, idxs=None, selection=None, want_counts=True, **kw):
    """want_counts=False: no hypothesis / count outputs (NULL) -- only then may the pass count in stages."""
    import torch
    L = load()
    if selection is not None:
        kw.setdefault("cap", vertex.shape[1] * vertex.shape[2])
    p = problem(mask, vertex, hn_total, thresh, **kw)
    n = L.pvv_workspace_bytes(ctypes.byref(p))
    assert n > 0, L.pvv_last_error()
    dev = vertex.device
    ws = torch.empty(n, dtype=torch.uint8, device=dev)
    cov = torch.empty(p.B, p.K, 2, 2, device=dev)
    hyp = torch.empty(p.B, p.K, hn_total, 2, device=dev)
    counts = torch.empty(p.B, p.K, hn_total, dtype=torch.int32, device=dev)
    tn = torch.empty(p.B, dtype=torch.int32, device=dev)
    check(L.pvv_estimate_voting_distribution(ctypes.byref(p), ptr(mask), ptr(vertex), ptr(idxs), ptr(selection),
                                             ptr(mean), ptr(ws), n, ptr(cov), ptr(hyp) if want_counts else None,
                                             ptr(counts) if want_counts else None, ptr(tn), None, stream()))
    return cov, hyp, counts, tn
