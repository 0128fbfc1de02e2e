"""Host-side mirror of ``lib/csrc/uncertainty_pnp/un_pnp_utils.py`` of clean-pvnet (uncertainty-weighted PnP).

The reference binds ``uncertainty_pnp`` of a Ceres-based host library through cffi (un_pnp_utils.py:1,47) and calls it
once per image from the evaluator (lib/evaluators/linemod/pvnet.py:118-132).  Here the same C symbol is exported by
``libpvnet_pnp.so`` (HIP, gfx950) and bound with ctypes; ``uncertainty_pnp`` / ``uncertainty_pnp_v2`` keep the reference's
names, arguments (numpy arrays on the host) and return value ([3,4] ``Rt``).  ``uncertainty_pnp_batched`` is what the GPU
is for: keypoints and weights of a whole batch stay on the device (``decode_keypoint(..., weights=True)`` produces them),
one launch refines every pose.  There is no CPU fallback.

Initial pose.  The reference starts from ``cv2.solvePnP(..., SOLVEPNP_P3P)`` on the four best-weighted keypoints
(un_pnp_utils.py:28-32).  With OpenCV importable that call is made exactly so; without it (this image has no cv2) the
start is a DLT over all keypoints (>= 6, not coplanar -- PVNet's 8 surface points + centre), orthogonalised: a different
start inside the same basin; the refinement converges to the same minimum (tests/test_pnp.py).
"""
import ctypes
import os

import numpy as np

_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpvnet_pnp.so")
try:
    _lib = ctypes.CDLL(_LIB)
except OSError as e:
    raise ImportError("clean_pvnet_amd.un_pnp_utils: libpvnet_pnp.so is not built (run `python __graft_entry__.py`); "
                      "there is no CPU fallback. Original error: %s" % (e,)) from e
_dp = ctypes.POINTER(ctypes.c_double)
_lib.uncertainty_pnp.restype = None
_lib.uncertainty_pnp.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int]
_lib.pvp_uncertainty_pnp_batched.restype = ctypes.c_int
_lib.pvp_uncertainty_pnp_batched.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_int] * 5 + [ctypes.c_double, ctypes.c_void_p]


def rodrigues(w):
    """``cv2.Rodrigues(w)[0]``: rotation matrix of an angle-axis vector (host, numpy)."""
    w = np.asarray(w, np.float64).reshape(3)
    th = float(np.linalg.norm(w))
    Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + Kx
    return np.eye(3) + np.sin(th) / th * Kx + (1 - np.cos(th)) / th ** 2 * (Kx @ Kx)


def rotation_to_angle_axis(R):
    """Inverse of ``rodrigues`` (host, numpy)."""
    c = np.clip((np.trace(R) - 1) / 2, -1, 1)
    th = float(np.arccos(c))
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    if th < 1e-9:
        return v / 2
    if np.pi - th < 1e-6:                       # near pi: axis from the symmetric part
        A = (R + np.eye(3)) / 2
        a = np.sqrt(np.clip(np.diag(A), 0, None))
        k = int(np.argmax(a))
        a = A[k] / a[k]
        if v @ a < 0:
            a = -a
        return a / np.linalg.norm(a) * th
    return v / (2 * np.sin(th)) * th


def initial_pose_dlt(points_3d, points_2d, camera_matrix, order_key=None):
    """Pose from >= 6 non-coplanar correspondences by the direct linear transform, rotation orthogonalised by SVD.
    -> rt [6] (angle-axis, translation).  Used when OpenCV's P3P is not importable.

    ``order_key`` [pn] (the reference's ``weights_2d[:,0] + weights_2d[:,1]``, un_pnp_utils.py:26) makes the start follow
    the reference's selection -- it seeds from its best-weighted keypoints only --: keypoints without weight (occluded /
    high-variance votes, weight 0) are left out as long as six remain, and the rows of the others are scaled by the square
    root of their relative key, so that one gross outlier with no weight cannot push the start out of the basin of the
    refinement."""
    P = np.asarray(points_3d, np.float64)
    p = np.asarray(points_2d, np.float64)
    pn = P.shape[0]
    if pn < 6:
        raise NotImplementedError("the DLT start needs >= 6 keypoints; with 4-5 install OpenCV (SOLVEPNP_P3P, as the reference)")
    row_w = np.ones(pn)
    if order_key is not None:
        key = np.where(np.isfinite(order_key), np.asarray(order_key, np.float64), 0.0)
        key = np.clip(key, 0.0, None)
        keep = key > 0
        if keep.sum() < 6:                                      # too few weighted keypoints: the six best-weighted
            keep = np.zeros(pn, bool)
            keep[np.argsort(-key, kind="stable")[:6]] = True
            key = np.where(key > 0, key, 0.0) + 1e-12
        P, p, key = P[keep], p[keep], key[keep]
        pn = P.shape[0]
        row_w = np.sqrt(key / key.max())
        row_w = np.maximum(row_w, 1e-3)
    Kinv = np.linalg.inv(np.asarray(camera_matrix, np.float64))
    n = (Kinv @ np.concatenate([p, np.ones((pn, 1))], 1).T).T                 # normalised image points
    c = P.mean(0)
    s = np.sqrt(((P - c) ** 2).sum(1).mean()) + 1e-30
    Q = (P - c) / s                                                            # conditioned object points
    A = np.zeros((2 * pn, 12))
    Qh = np.concatenate([Q, np.ones((pn, 1))], 1)
    A[0::2, 0:4] = Qh
    A[0::2, 8:12] = -n[:, :1] * Qh
    A[1::2, 4:8] = Qh
    A[1::2, 8:12] = -n[:, 1:2] * Qh
    A *= np.repeat(row_w, 2)[:, None]
    M = np.linalg.svd(A)[2][-1].reshape(3, 4)
    if np.linalg.det(M[:, :3]) < 0:
        M = -M
    U, S, Vt = np.linalg.svd(M[:, :3])
    R = U @ Vt
    scale = S.mean()
    t = M[:, 3] / scale
    # undo the conditioning: X = R ((P - c)/s) + t  =>  X' = s X = R P + (s t - R c)
    t = s * t - R @ c
    return np.concatenate([rotation_to_angle_axis(R), t])


def p3p_depths(f, P):
    """Grunert's perspective-three-point solution: unit bearings ``f`` [3,3] of the object points ``P`` [3,3] -> the (up
    to four) depth triples (s1, s2, s3) with ``|s_i f_i - s_j f_j| = |P_i - P_j|``.  With s2 = u s1, s3 = v s1 the three
    cosine-law equations reduce to a quartic in v (Haralick et al., "Review and analysis of solutions of the three point
    perspective pose estimation problem", 1994, eqs. 9-11)."""
    a2 = float(((P[1] - P[2]) ** 2).sum()); b2 = float(((P[0] - P[2]) ** 2).sum()); c2 = float(((P[0] - P[1]) ** 2).sum())
    if min(a2, b2, c2) <= 0:
        return []
    ca, cb, cg = float(f[1] @ f[2]), float(f[0] @ f[2]), float(f[0] @ f[1])
    q, r = (a2 - c2) / b2, (a2 + c2) / b2
    coef = [(q - 1) ** 2 - 4 * c2 / b2 * ca ** 2,
            4 * (q * (1 - q) * cb - (1 - r) * ca * cg + 2 * c2 / b2 * ca ** 2 * cb),
            2 * (q ** 2 - 1 + 2 * q ** 2 * cb ** 2 + 2 * ((b2 - c2) / b2) * ca ** 2 - 4 * r * ca * cb * cg + 2 * ((b2 - a2) / b2) * cg ** 2),
            4 * (-q * (1 + q) * cb + 2 * a2 / b2 * cg ** 2 * cb - (1 - r) * ca * cg),
            (1 + q) ** 2 - 4 * a2 / b2 * cg ** 2]
    if not np.isfinite(coef).all():
        return []
    sols = []
    for v in np.roots(coef):
        if abs(v.imag) > 1e-6 * max(1.0, abs(v)) or v.real <= 0:
            continue
        v = float(v.real)
        den = 2 * (cg - v * ca)
        if abs(den) < 1e-14:
            continue
        u = ((q - 1) * v * v - 2 * q * cb * v + 1 + q) / den
        d = 1 + u * u - 2 * u * cg
        if u <= 0 or d <= 0:
            continue
        s1 = np.sqrt(c2 / d)
        sols.append((s1, u * s1, v * s1))
    return sols


def initial_pose_p3p(points_3d, points_2d, camera_matrix, order_key):
    """The reference's start (un_pnp_utils.py:26-32: ``cv2.solvePnP(..., SOLVEPNP_P3P)`` on the four best-weighted
    keypoints) without OpenCV: P3P on the first three of those four, the fourth picks among the (up to four) solutions by
    its reprojection error -- what OpenCV's P3P does with its fourth point.  -> rt [6] or None when no solution exists
    (collinear points, a keypoint behind the camera)."""
    P = np.asarray(points_3d, np.float64)
    p = np.asarray(points_2d, np.float64)
    Kc = np.asarray(camera_matrix, np.float64)
    key = np.where(np.isfinite(order_key), np.asarray(order_key, np.float64), -np.inf)
    idxs = np.argsort(key)[-4:]                                                # un_pnp_utils.py:26 / :88
    P4, p4 = P[idxs], p[idxs]
    n = (np.linalg.inv(Kc) @ np.concatenate([p4, np.ones((4, 1))], 1).T).T
    f = n / np.linalg.norm(n, axis=1, keepdims=True)
    best = None
    for s in p3p_depths(f[:3], P4[:3]):
        X = f[:3] * np.asarray(s)[:, None]                                     # the three points in the camera frame
        cP, cX = P4[:3].mean(0), X.mean(0)
        # absolute orientation of three points: rotation that maps the (centred) object triangle onto the camera one
        H = (P4[:3] - cP).T @ (X - cX)
        U, _S, Vt = np.linalg.svd(H)
        D = np.diag([1.0, 1.0, np.sign(np.linalg.det(Vt.T @ U.T)) or 1.0])
        R = Vt.T @ D @ U.T
        t = cX - R @ cP
        x4 = R @ P4[3] + t
        if x4[2] <= 0:
            continue
        proj = Kc @ (x4 / x4[2])
        err = float(((proj[:2] - p4[3]) ** 2).sum())
        if best is None or err < best[0]:
            best = (err, R, t)
    if best is None:
        return None
    return np.concatenate([rotation_to_angle_axis(best[1]), best[2]])


def _initial_pose(points_3d, points_2d, camera_matrix, order_key):
    try:
        import cv2
    except ImportError:
        # no OpenCV in this image: the same selection and the same solver family in numpy; the DLT only as the fallback
        rt = initial_pose_p3p(points_3d, points_2d, camera_matrix, order_key)
        if rt is not None:
            return rt, (rt[:3].reshape(3, 1), rt[3:].reshape(3, 1))
        return initial_pose_dlt(points_3d, points_2d, camera_matrix, order_key), None
    try:
        dist_coeffs = uncertainty_pnp.dist_coeffs
    except AttributeError:
        dist_coeffs = np.zeros(shape=[8, 1], dtype=np.float64)
    idxs = np.argsort(order_key)[-4:]                                          # un_pnp_utils.py:26 / :88
    _, R_exp, t = cv2.solvePnP(np.expand_dims(points_3d[idxs, :], 0), np.expand_dims(points_2d[idxs, :], 0),
                               camera_matrix, dist_coeffs, None, None, False, flags=cv2.SOLVEPNP_P3P)
    return np.concatenate([R_exp.reshape(3), t.reshape(3)]), (R_exp, t)


def _refine(points_2d, weights_2d, points_3d, camera_matrix, init_rt):
    a = [np.ascontiguousarray(v, np.float64) for v in (points_2d, points_3d, weights_2d, camera_matrix, init_rt)]
    result_rt = np.full([6], np.nan, np.float64)
    _lib.uncertainty_pnp(a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, a[3].ctypes.data, a[4].ctypes.data,
                         result_rt.ctypes.data, a[0].shape[0])
    if not np.isfinite(result_rt).all():
        raise RuntimeError("uncertainty_pnp failed (no GPU? see stderr); there is no CPU fallback")
    return result_rt


def uncertainty_pnp(points_2d, weights_2d, points_3d, camera_matrix):
    '''
    :param points_2d:           [pn,2]
    :param weights_2d:          [pn,3] wxx,wxy,wyy
    :param points_3d:           [pn,3]
    :param camera_matrix:       [3,3]
    :return:                    [3,4] Rt
    '''
    pn = points_2d.shape[0]
    assert(points_3d.shape[0] == pn and pn >= 4)
    points_3d = points_3d.astype(np.float64)
    points_2d = points_2d.astype(np.float64)
    weights_2d = weights_2d.astype(np.float64)
    camera_matrix = camera_matrix.astype(np.float64)
    init_rt, p3p = _initial_pose(points_3d, points_2d, camera_matrix, weights_2d[:, 0] + weights_2d[:, 1])
    if pn == 4 and p3p is not None:
        # no other points (un_pnp_utils.py:34-38)
        return np.concatenate([rodrigues(p3p[0]), p3p[1].reshape(3, 1)], axis=-1)
    result_rt = _refine(points_2d, weights_2d, points_3d, camera_matrix, init_rt)
    return np.concatenate([rodrigues(result_rt[:3]), result_rt[3:, None]], axis=-1)


def uncertainty_pnp_v2(points_2d, covars, points_3d, camera_matrix, type='single'):
    '''
    :param points_2d:           [pn,2]
    :param covars:              [pn,2,2]
    :param points_3d:           [pn,3]
    :param camera_matrix:       [3,3]
    :return:                    [3,4] Rt
    Isotropic weights 1 / lambda_max(cov) (0 where cov[0,0] < 1e-5), un_pnp_utils.py:60-83.
    '''
    del type
    pn = points_2d.shape[0]
    assert(points_3d.shape[0] == pn and pn >= 4 and covars.shape[0] == pn)
    points_3d = points_3d.astype(np.float64)
    points_2d = points_2d.astype(np.float64)
    camera_matrix = camera_matrix.astype(np.float64)
    w = np.array([0.0 if covars[pi, 0, 0] < 1e-5 else 1.0 / np.max(np.linalg.eigvals(covars[pi]).real)
                  for pi in range(pn)], np.float64)
    init_rt, p3p = _initial_pose(points_3d, points_2d, camera_matrix, w)
    if pn == 4 and p3p is not None:
        return np.concatenate([rodrigues(p3p[0]), p3p[1].reshape(3, 1)], axis=-1)
    weights_2d = np.stack([w, np.zeros(pn), w], 1)
    result_rt = _refine(points_2d, weights_2d, points_3d, camera_matrix, init_rt)
    return np.concatenate([rodrigues(result_rt[:3]), result_rt[3:, None]], axis=-1)


def uncertainty_pnp_batched(points_2d, weights_2d, points_3d, camera_matrix, init_rt, max_iterations=0,
                            function_tolerance=0.0, return_info=False):
    """The refinement for a whole batch on the device, one launch on the current stream, nothing read back.
    :param points_2d:      [b,pn,2] CUDA tensor (any float dtype; e.g. ``output['kpt_2d']``)
    :param weights_2d:     [b,pn,3] (wxx,wxy,wyy), e.g. ``output['var_weights']``
    :param points_3d:      [pn,3] (one object model) or [b,pn,3]
    :param camera_matrix:  [3,3] or [b,3,3]
    :param init_rt:        [b,6] angle-axis + translation
    :return:               rt [b,6] float64 (and info [b,4]: initial cost, final cost, iterations, termination)
    """
    import torch
    dev = points_2d.device
    assert dev.type == "cuda", "uncertainty_pnp_batched needs CUDA tensors (no CPU path exists)"
    f64 = lambda t: t.to(device=dev, dtype=torch.float64).contiguous()          # noqa: E731
    p2, w2, p3, Km, rt0 = f64(points_2d), f64(weights_2d), f64(points_3d), f64(camera_matrix), f64(init_rt)
    b, pn = p2.shape[0], p2.shape[1]
    assert p2.shape == (b, pn, 2) and w2.shape == (b, pn, 3) and rt0.shape == (b, 6)
    assert p3.shape in ((pn, 3), (b, pn, 3)) and Km.shape in ((3, 3), (b, 3, 3))
    out = torch.empty(b, 6, dtype=torch.float64, device=dev)
    info = torch.empty(b, 4, dtype=torch.float64, device=dev) if return_info else None
    if b == 0:
        return (out, info) if return_info else out
    with torch.cuda.device(dev):
        rc = _lib.pvp_uncertainty_pnp_batched(p2.data_ptr(), p3.data_ptr(), w2.data_ptr(), Km.data_ptr(), rt0.data_ptr(),
                                              out.data_ptr(), info.data_ptr() if return_info else None, b, pn,
                                              1 if p3.dim() == 3 else 0, 1 if Km.dim() == 3 else 0, int(max_iterations),
                                              float(function_tolerance), torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        raise RuntimeError("pvp_uncertainty_pnp_batched failed (%d)" % rc)
    return (out, info) if return_info else out
