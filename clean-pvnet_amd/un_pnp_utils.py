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


def initial_pose_dlt(points_3d, points_2d, camera_matrix):
    """Pose from >= 6 non-coplanar correspondences by the direct linear transform, rotation orthogonalised by SVD.
    -> rt [6] (angle-axis, translation).  Used when OpenCV's P3P is not importable."""
    P = np.asarray(points_3d, np.float64)
    p = np.asarray(points_2d, np.float64)
    pn = P.shape[0]
    if pn < 6:
        raise NotImplementedError("the DLT start needs >= 6 keypoints; with 4-5 install OpenCV (SOLVEPNP_P3P, as the reference)")
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


def _initial_pose(points_3d, points_2d, camera_matrix, order_key):
    try:
        import cv2
    except ImportError:
        return initial_pose_dlt(points_3d, points_2d, camera_matrix), None
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
