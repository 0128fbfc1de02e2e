"""CPU oracle for clean-pvnet's RANSAC voting path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module, and only as the checker / timed CPU baseline.  The
product (``clean-pvnet_amd/``) never imports it and has no CPU fallback.

PARITY STATUS: pinned against outputs of the reference itself (it has no tests
of its own for this path):

* the kernels (C in ``vote_oracle.c``, numpy twins below, cross-checked against
  each other) are checked bit for bit against ``oracle/_ref`` -- the reference's
  own ``lib/csrc/ransac_voting/src/ransac_voting_kernel.cu``, compiled where it
  lies with hipcc for gfx950 through ``oracle/ref_shim/`` and run on the MI355X
  through its own launchers (``oracle/ref_build.hip``, ``tests/test_ref_pin.py``);
* the Python glue (select / refit / covariance) is checked against the
  reference's own ``ransac_voting_gpu.py`` executed on CPU tensors with these
  kernels substituted for the CUDA extension (``tests/golden/make_golden.py``
  -> ``tests/golden/*.npz``);
* ``compute_vertex`` known-answer fields (``lib/utils/pvnet/pvnet_data_utils.py:30-44``).

Arithmetic contract: IEEE binary32, one rounding per operation, no FMA.

Layouts follow the reference: ``direct [tn,vn,2]``, ``coords [tn,2]`` (x,y),
``idxs [hn,vn,2]`` int32, ``hypo_pts [hn,vn,2]``, ``inliers [hn,vn,tn]`` uint8.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvote_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_i32p = ctypes.POINTER(ctypes.c_int32)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    """Compile ``vote_oracle.c`` -> ``libvote_oracle.so`` (gcc, see Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "vote_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libvote_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


# --------------------------------------------------------------------------
# C oracle wrappers (kernel.cu restatements)
# --------------------------------------------------------------------------
def generate_hypothesis(direct, coords, idxs):
    """kernel.cu:11-86 -> zero-initialised ``[hn,vn,2]``."""
    direct, coords, idxs = _c(direct, np.float32), _c(coords, np.float32), _c(idxs, np.int32)
    tn, vn, _ = direct.shape
    hn = idxs.shape[0]
    out = np.empty((hn, vn, 2), np.float32)
    lib().orc_generate_hypothesis(_p(direct, _f32p), _p(coords, _f32p), _p(idxs, _i32p),
                                  _p(out, _f32p), tn, vn, hn)
    return out


def voting_for_hypothesis(direct, coords, hypo_pts, inliers, thresh):
    """kernel.cu:88-167 -- in place, writes 1 only (caller pre-zeroes)."""
    direct, coords, hypo_pts = _c(direct, np.float32), _c(coords, np.float32), _c(hypo_pts, np.float32)
    assert inliers.dtype == np.uint8 and inliers.flags.c_contiguous
    tn, vn, _ = direct.shape
    hn = hypo_pts.shape[0]
    assert inliers.shape == (hn, vn, tn)
    lib().orc_voting_for_hypothesis(_p(direct, _f32p), _p(coords, _f32p), _p(hypo_pts, _f32p),
                                    _p(inliers, _u8p), tn, vn, hn, ctypes.c_float(thresh))
    return inliers


def count_inliers(direct, coords, hypo_pts, thresh):
    """voting_for_hypothesis + sum over tn (ransac_voting_gpu.py:156-159) -> ``[hn,vn]`` int32."""
    direct, coords, hypo_pts = _c(direct, np.float32), _c(coords, np.float32), _c(hypo_pts, np.float32)
    tn, vn, _ = direct.shape
    hn = hypo_pts.shape[0]
    counts = np.empty((hn, vn), np.int32)
    lib().orc_count_inliers(_p(direct, _f32p), _p(coords, _f32p), _p(hypo_pts, _f32p),
                            _p(counts, _i32p), tn, vn, hn, ctypes.c_float(thresh))
    return counts


def generate_hypothesis_vanishing_point(direct, coords, idxs):
    """kernel.cu:170-266 -> ``[hn,vn,3]``."""
    direct, coords, idxs = _c(direct, np.float32), _c(coords, np.float32), _c(idxs, np.int32)
    tn, vn, _ = direct.shape
    hn = idxs.shape[0]
    out = np.zeros((hn, vn, 3), np.float32)
    lib().orc_generate_hypothesis_vanishing_point(_p(direct, _f32p), _p(coords, _f32p),
                                                  _p(idxs, _i32p), _p(out, _f32p), tn, vn, hn)
    return out


def voting_for_hypothesis_vanishing_point(direct, coords, hypo_pts, inliers, thresh):
    """kernel.cu:268-350 -- in place, writes 1 only."""
    direct, coords, hypo_pts = _c(direct, np.float32), _c(coords, np.float32), _c(hypo_pts, np.float32)
    assert inliers.dtype == np.uint8 and inliers.flags.c_contiguous
    tn, vn, _ = direct.shape
    hn = hypo_pts.shape[0]
    lib().orc_voting_for_hypothesis_vanishing_point(
        _p(direct, _f32p), _p(coords, _f32p), _p(hypo_pts, _f32p), _p(inliers, _u8p),
        tn, vn, hn, ctypes.c_float(thresh))
    return inliers


# --------------------------------------------------------------------------
# numpy twins of the two kernels on the named path (readable restatement;
# every arithmetic op is a separate binary32 ufunc call => one rounding each)
# --------------------------------------------------------------------------
def np_generate_hypothesis(direct, coords, idxs):
    """Appendix A.1 of SURVEY.md; kernel.cu:22-48."""
    f = np.float32
    direct, coords = np.asarray(direct, f), np.asarray(coords, f)
    hn, vn, _ = idxs.shape
    vi = np.arange(vn)[None, :]
    t0, t1 = idxs[..., 0], idxs[..., 1]
    nx0, ny0 = direct[t0, vi, 1], -direct[t0, vi, 0]
    nx1, ny1 = direct[t1, vi, 1], -direct[t1, vi, 0]
    cx0, cy0 = coords[t0, 0], coords[t0, 1]
    cx1, cy1 = coords[t1, 0], coords[t1, 1]
    den_y = nx1 * ny0 - nx0 * ny1
    den_x = ny1 * nx0 - ny0 * nx1
    ok = ~(np.abs(den_y).astype(np.float64) < 1e-6) & ~(np.abs(den_x).astype(np.float64) < 1e-6)
    with np.errstate(all="ignore"):
        b0 = nx0 * cx0 + ny0 * cy0
        b1 = nx1 * cx1 + ny1 * cy1
        y = (nx1 * b0 - nx0 * b1) / den_y
        x = (ny1 * b0 - ny0 * b1) / den_x
    out = np.zeros((hn, vn, 2), f)
    out[..., 0] = np.where(ok, x, f(0))
    out[..., 1] = np.where(ok, y, f(0))
    return out


def np_vote_mask(direct, coords, hypo_pts, thresh):
    """Appendix A.2; kernel.cu:100-125 -> bool ``[hn,vn,tn]``."""
    f = np.float32
    direct, coords, hypo_pts = np.asarray(direct, f), np.asarray(coords, f), np.asarray(hypo_pts, f)
    nx = direct[:, :, 0].T[None]            # [1,vn,tn]
    ny = direct[:, :, 1].T[None]
    cx = coords[:, 0][None, None]
    cy = coords[:, 1][None, None]
    hx = hypo_pts[:, :, 0][:, :, None]      # [hn,vn,1]
    hy = hypo_pts[:, :, 1][:, :, None]
    with np.errstate(all="ignore"):
        dx = hx - cx
        dy = hy - cy
        norm1 = np.sqrt(nx * nx + ny * ny)
        norm2 = np.sqrt(dx * dx + dy * dy)
        bad = (norm1.astype(np.float64) < 1e-6) | (norm2.astype(np.float64) < 1e-6)
        angle = (dx * nx + dy * ny) / (norm1 * norm2)
        return (angle > f(thresh)) & ~bad


# --------------------------------------------------------------------------
# Python glue restated (ransac_voting_gpu.py)
# --------------------------------------------------------------------------
def compact_v3(mask2d, vertex_hwk2, max_num=30000, selection=None):
    """ransac_voting_gpu.py:125-144.

    ``mask.byte()`` wraps integers modulo 256; ``foreground_num`` is the SUM of
    the byte values (:126), not the count.  Returns ``(fg_sum, coords, direct)``;
    ``selection`` is the injected U(0,1) tensor of :136 (required iff fg_sum > max_num).
    """
    cur = np.asarray(mask2d)
    cur = cur.astype(np.uint8) if cur.dtype != np.bool_ else cur.astype(np.uint8)
    fg = int(cur.sum(dtype=np.int64))
    if fg > max_num:
        assert selection is not None, "subsampling active: inject `selection`"
        p = np.float32(max_num) / np.float32(fg)                       # :137 binary32
        cur = cur * (np.asarray(selection, np.float32) < p).astype(np.uint8)
    ys, xs = np.nonzero(cur)                                           # row-major, :140
    coords = np.stack([xs, ys], 1).astype(np.float32)                  # :141 (x,y)
    direct = np.ascontiguousarray(np.asarray(vertex_hwk2, np.float32)[ys, xs])  # :142-143 [tn,vn,2]
    return fg, coords, direct


def compact_estimate(mask2d, vertex_hwk2, max_num=30000, selection=None):
    """ransac_voting_gpu.py:207-229: foreground is ``mask == 1``; counts are counts."""
    cur = (np.asarray(mask2d) == 1)
    fg = int(cur.sum())
    fg0 = fg
    if fg > max_num:
        assert selection is not None, "subsampling active: inject `selection`"
        p = np.float32(max_num) / np.float32(fg)
        cur = cur & (np.asarray(selection, np.float32) < p)
        fg = int(cur.sum())                                            # :223
    ys, xs = np.nonzero(cur)
    coords = np.stack([xs, ys], 1).astype(np.float32)
    direct = np.ascontiguousarray(np.asarray(vertex_hwk2, np.float32)[ys, xs])
    return fg0, coords, direct


def compact_v3_c(mask2d, vertex_hwk2, max_num=30000, selection=None):
    """``compact_v3`` in C (``orc_compact_v3``): the same ``(fg_sum, coords, direct)``, no numpy pass over the image --
    what bench.py's cpu_baseline leg times.  tests/test_oracle.py checks it against the numpy restatement above."""
    m = np.ascontiguousarray(mask2d)
    if m.dtype == np.bool_:
        m = m.view(np.uint8)
    assert m.dtype.kind in "iu", m.dtype
    v = _c(vertex_hwk2, np.float32)
    H, W, vn, _ = v.shape
    sel = None if selection is None else _c(selection, np.float32)
    rows = int(np.count_nonzero(m)) if sel is not None else H * W
    rows = min(H * W, max(1, rows))
    # (row capacity: every pixel of the image at most; the count above is only taken when draws are injected)
    coords = np.empty((H * W if sel is None else rows, 2), np.float32)
    direct = np.empty((coords.shape[0], vn, 2), np.float32)
    tn = ctypes.c_int(0)
    f = lib().orc_compact_v3
    f.restype = ctypes.c_longlong
    fg = f(m.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(m.dtype.itemsize), _p(v, _f32p), H, W, vn,
           None if sel is None else _p(sel, _f32p), int(max_num), _p(coords, _f32p), _p(direct, _f32p), coords.shape[0],
           ctypes.byref(tn))
    assert tn.value >= 0, "subsampling active: inject `selection`"
    return int(fg), coords[: tn.value], direct[: tn.value]


def v3_image(direct, coords, idxs, thresh):
    """Select + refit of one compacted image (C oracle). Returns a dict of everything."""
    direct, coords, idxs = _c(direct, np.float32), _c(coords, np.float32), _c(idxs, np.int32)
    tn, vn, _ = direct.shape
    hn = idxs.shape[0]
    r = dict(hypo_pts=np.empty((hn, vn, 2), np.float32), counts=np.empty((hn, vn), np.int32),
             win_pts=np.empty((vn, 2), np.float32), win_counts=np.empty(vn, np.int32),
             win_idx=np.empty(vn, np.int32), ATA=np.empty((vn, 3), np.float64),
             ATb=np.empty((vn, 2), np.float64), singular=np.empty(vn, np.int32),
             pts=np.empty((vn, 2), np.float32))
    lib().orc_v3_image(_p(direct, _f32p), _p(coords, _f32p), _p(idxs, _i32p), tn, vn, hn,
                       ctypes.c_float(thresh), _p(r["hypo_pts"], _f32p), _p(r["counts"], _i32p),
                       _p(r["win_pts"], _f32p), _p(r["win_counts"], _i32p), _p(r["win_idx"], _i32p),
                       _p(r["ATA"], _f64p), _p(r["ATb"], _f64p), _p(r["singular"], _i32p),
                       _p(r["pts"], _f32p))
    r["tn"] = tn
    return r


def ransac_voting_layer_v3(mask, vertex, round_hyp_num, inlier_thresh=0.999, confidence=0.99,
                           max_iter=20, min_num=5, max_num=30000, *, idxs, selection=None,
                           singular="reference", details=None, compact_in_c=False):
    """ransac_voting_gpu.py:112-199 on numpy arrays.

    ``idxs``: per-image injected index pairs, ``[B,hn,vn,2]`` (entries for skipped
    images are ignored).  ``confidence``/``max_iter`` have no effect on the output
    (appendix A.3) and are accepted for signature parity only.

    ``singular``: what happens when a keypoint's 2x2 normal matrix is singular.
      "reference": bug-compatible with ``b_inv`` (:97-109) under torch 1.1 -- the batched
                   solve raises for the whole image, every keypoint of that image gets
                   ``inverse = identity`` i.e. ``x = ATb``.
      "zero":      only the singular keypoint is affected and becomes (0,0).
    """
    del confidence, max_iter
    mask, vertex = np.asarray(mask), np.asarray(vertex, np.float32)
    b, h, w, vn, _ = vertex.shape
    out = np.zeros((b, vn, 2), np.float32)
    for bi in range(b):
        fg, coords, direct = (compact_v3_c if compact_in_c else compact_v3)(mask[bi], vertex[bi], max_num,
                                                                            None if selection is None else selection[bi])
        if fg < min_num:                                               # :129-132
            if details is not None:
                details.append(dict(tn=0, skipped=True))
            continue
        r = v3_image(direct, coords, idxs[bi], inlier_thresh)
        if singular == "reference" and r["singular"].any():
            out[bi] = r["ATb"].astype(np.float32)
        else:
            out[bi] = r["pts"]
        if details is not None:
            r["skipped"] = False
            details.append(r)
    return out


def v3_batch(mask, vertex, round_hyp_num, inlier_thresh, idxs, min_num=5, max_num=30000, total=None):
    """``ransac_voting_layer_v3`` (singular="reference") over ``total`` images cycling through the ``n`` samples given, IMAGE-PARALLEL in C
    (``orc_v3_batch``: one OpenMP thread per image, serial inside -- the form that uses a host's cores; what bench.py's cpu_baseline leg
    times beside the hypothesis-parallel single-image form).  -> (out [n,vn,2], win_counts [n,vn], -1 rows for skipped images)."""
    m = np.ascontiguousarray(mask)
    if m.dtype == np.bool_:
        m = m.view(np.uint8)
    assert m.dtype.kind in "iu" and m.ndim == 3, (m.dtype, m.shape)
    v = _c(vertex, np.float32)
    n, H, W, vn, _ = v.shape
    ix = _c(idxs, np.int32)
    assert ix.shape == (n, round_hyp_num, vn, 2) and m.shape == (n, H, W)
    out = np.zeros((n, vn, 2), np.float32)
    win = np.zeros((n, vn), np.int32)
    f = lib().orc_v3_batch
    f.restype = ctypes.c_int
    rc = f(m.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(m.dtype.itemsize), _p(v, _f32p), _p(ix, _i32p), n, H, W, vn,
           int(round_hyp_num), ctypes.c_float(inlier_thresh), int(min_num), int(max_num), int(n if total is None else total),
           _p(out, _f32p), _p(win, _i32p))
    assert rc >= 0, "orc_v3_batch failed: %d (-1 allocation, -2 an image exceeds max_num: subsampling is not supported here)" % rc
    return out, win


def estimate_voting_distribution_with_mean(mask, vertex, mean, round_hyp_num=256, min_hyp_num=4096,
                                           topk=128, inlier_thresh=0.99, min_num=5, max_num=30000,
                                           output_hyp=False, *, idxs, selection=None, details=None):
    """ransac_voting_gpu.py:202-274 on numpy arrays.

    ``idxs``: ``[B, round_num*round_hyp_num, vn, 2]`` -- the fresh draws of every round
    (:235) concatenated in round order.
    """
    del topk, output_hyp
    mask, vertex, mean = np.asarray(mask), np.asarray(vertex, np.float32), np.asarray(mean, np.float32)
    b, h, w, vn, _ = vertex.shape
    round_num = int(np.ceil(min_hyp_num / round_hyp_num))              # :231
    hn_total = round_num * round_hyp_num
    cov = np.zeros((b, vn, 2, 2), np.float32)
    for bi in range(b):
        fg, coords, direct = compact_estimate(mask[bi], vertex[bi], max_num,
                                              None if selection is None else selection[bi])
        if fg < min_num:
            # :211-216 hyps = zeros[min_hyp_num], ratios = ones -> cov = mean mean^T * n/(n+1e-3)
            m = mean[bi].astype(np.float64)
            d = (np.float32(0) - mean[bi]).astype(np.float64)          # :266 binary32 diff
            n = float(min_hyp_num)
            c = np.einsum("ki,kj->kij", d, d) * n / (n + 1e-3)
            cov[bi] = c.astype(np.float32)
            del m
            if details is not None:
                details.append(dict(tn=0, skipped=True))
            continue
        tn = coords.shape[0]
        idx = _c(idxs[bi], np.int32)
        assert idx.shape == (hn_total, vn, 2)
        hyp = np.empty((hn_total, vn, 2), np.float32)
        counts = np.empty((hn_total, vn), np.int32)
        c = np.empty((vn, 2, 2), np.float32)
        m = _c(mean[bi], np.float32)
        lib().orc_estimate_image(_p(direct, _f32p), _p(coords, _f32p), _p(idx, _i32p), tn, vn,
                                 hn_total, ctypes.c_float(inlier_thresh), _p(m, _f32p),
                                 _p(hyp, _f32p), _p(counts, _i32p), _p(c, _f32p))
        cov[bi] = c
        if details is not None:
            details.append(dict(tn=tn, skipped=False, hypo_pts=hyp, counts=counts))
    return mean, cov


def find_nearest_point_idx(ref_pts, que_pts, exclude_self=False):
    """nearest_neighborhood.cu:48-117 through nn_utils.find_nearest_point_idx's interface: ref [pn1,dim], que [pn2,dim]
    -> int32 [pn2]."""
    ref, que = _c(ref_pts, np.float32), _c(que_pts, np.float32)
    assert ref.shape[1] == que.shape[1] and 1 < que.shape[1] <= 3
    idxs = np.zeros(que.shape[0], np.int32)
    lib().orc_find_nearest(_p(ref, _f32p), _p(que, _f32p), _p(idxs, _i32p), 1, ref.shape[0], que.shape[0], ref.shape[1],
                           int(bool(exclude_self)))
    return idxs


def num_threads():
    return int(lib().orc_num_threads())


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))
