"""CPU oracle of the uncertainty-weighted PnP refinement -- TEST INFRASTRUCTURE ONLY (tests/, bench cpu legs,
__graft_entry__.smoke may import it; the product must not).

Restates, in numpy binary64:
  residuals   ReprojectionErrorArray::operator()  /root/reference/lib/csrc/uncertainty_pnp/src/uncertainty_pnp.cpp:19-38
  rotation    ceres::AngleAxisRotatePoint         .../uncertainty_pnp/include/ceres/rotation.h:563-622
and provides two minimisers of that function from a given start:
  solve_lm      Levenberg-Marquardt with Ceres' documented default schedule -- the numpy twin of what the HIP kernel
                (clean-pvnet_amd/csrc/pvnet_pnp.hip) and the shim Solve() of oracle/ref_build_pnp.cpp run;
  solve_scipy   scipy.optimize.least_squares (MINPACK lmder), tolerances at machine precision -- an independent
                minimiser: the reference point for "same minimum".
How it is pinned: ``ref()`` loads oracle/_ref/libref_uncertainty_pnp.so = the reference's OWN uncertainty_pnp.cpp compiled
where it lies against a shim ceres.h (oracle/ref_shim_pnp): its functor evaluated through the reference's own vendored
ceres/jet.h gives residuals and Jacobians that tests/test_pnp.py compares with this file's to 1e-12.  What stays unpinned:
the iterate path of the real Ceres library (libceres.so cannot be linked here: libglog / libspqr / libcholmod / liblapack
are missing); parity is therefore defined on the minimum (cost within 1e-9 relative, pose within 1e-6), not on iterates.
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
EPS = np.finfo(np.float64).eps


def angle_axis_rotate_point(w, P, jac=False):
    """rotation.h:563-622; with ``jac`` also d result / d w [3,3] (the derivative of that very expression)."""
    w, P = np.asarray(w, np.float64), np.asarray(P, np.float64)
    theta2 = float(w @ w)
    if theta2 > EPS:
        theta = np.sqrt(theta2)
        c, s = np.cos(theta), np.sin(theta)
        a = w / theta
        axP = np.cross(a, P)
        adP = float(a @ P)
        X = P * c + axP * s + a * (adP * (1.0 - c))
        if not jac:
            return X
        base = -s * P + c * axP + s * adP * a
        J = np.zeros((3, 3))
        for j in range(3):
            da = (np.eye(3)[j] - a * a[j]) / theta
            J[:, j] = a[j] * base + s * np.cross(da, P) + (1.0 - c) * (da * adP + a * float(da @ P))
        return X, J
    X = P + np.cross(w, P)
    if not jac:
        return X
    J = np.array([[0.0, P[2], -P[1]], [-P[2], 0.0, P[0]], [P[1], -P[0], 0.0]])
    return X, J


def residuals(rt, pts2d, pts3d, wgt2d, K, jac=False):
    """uncertainty_pnp.cpp:19-38 for all keypoints -> r [pn,2] (and J [pn,2,6])."""
    rt = np.asarray(rt, np.float64)
    fx, fy, px, py = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    pn = pts2d.shape[0]
    r = np.zeros((pn, 2))
    J = np.zeros((pn, 2, 6))
    for i in range(pn):
        if jac:
            X, dX = angle_axis_rotate_point(rt[:3], pts3d[i], True)
        else:
            X = angle_axis_rotate_point(rt[:3], pts3d[i])
        X = X + rt[3:]
        dx = fx * X[0] / X[2] + px - pts2d[i, 0]
        dy = fy * X[1] / X[2] + py - pts2d[i, 1]
        wxx, wxy, wyy = wgt2d[i]
        r[i] = (wxx * dx + wxy * dy, wxy * dx + wyy * dy)
        if jac:
            G = np.concatenate([dX, np.eye(3)], 1)                                  # dX/dparams [3,6]
            du = fx / X[2] * G[0] - fx * X[0] / X[2] ** 2 * G[2]
            dv = fy / X[2] * G[1] - fy * X[1] / X[2] ** 2 * G[2]
            J[i, 0] = wxx * du + wxy * dv
            J[i, 1] = wxy * du + wyy * dv
    return (r, J) if jac else r


def cost(rt, pts2d, pts3d, wgt2d, K):
    r = residuals(rt, pts2d, pts3d, wgt2d, K)
    return 0.5 * float((r * r).sum())


def solve_lm(init_rt, pts2d, pts3d, wgt2d, K, max_iterations=50):
    """Ceres' default trust-region / Levenberg-Marquardt schedule (see oracle/ref_shim_pnp/ceres/ceres.h).
    -> (rt, info dict)."""
    x = np.array(init_rt, np.float64)
    radius, decrease = 1e4, 2.0

    def ev(q):
        r, J = residuals(q, pts2d, pts3d, wgt2d, K, True)
        J = J.reshape(-1, 6)
        r = r.reshape(-1)
        return 0.5 * float(r @ r), J.T @ J, J.T @ r
    c, A, g = ev(x)
    # Ceres: Jacobi scaling 1/(1 + |J_i|) from the initial point; LM diagonal = squared norm of the scaled column clamped to
    # [min_lm_diagonal, max_lm_diagonal] = [1e-6, 1e32] (levenberg_marquardt_strategy.cc)
    s2 = (1.0 / (1.0 + np.sqrt(np.diag(A)))) ** 2
    info = dict(initial_cost=c, termination=0)
    it = 0
    while it < max_iterations:
        if np.abs(g).max() <= 1e-10:
            info["termination"] = 1
            break
        d = np.clip(np.diag(A) * s2, 1e-6, 1e32) / (s2 * radius)
        try:
            step = np.linalg.solve(A + np.diag(d), -g)
            ok = bool(np.all(np.linalg.eigvalsh(A + np.diag(d)) > 0))
        except np.linalg.LinAlgError:
            ok = False
        rho, new_c = -1.0, c
        if ok:
            if np.linalg.norm(step) <= 1e-8 * (np.linalg.norm(x) + 1e-8):
                info["termination"] = 2
                break
            model = -float(step @ (g + 0.5 * A @ step))
            if model > 0:
                new_c = cost(x + step, pts2d, pts3d, wgt2d, K)
                rho = (c - new_c) / model
        if rho > 1e-3 and np.isfinite(new_c):
            change, old = c - new_c, c
            x = x + step
            t = 2.0 * rho - 1.0
            radius = min(radius / max(1.0 / 3.0, 1.0 - t ** 3), 1e16)
            decrease = 2.0
            c, A, g = ev(x)
            if abs(change) <= 1e-6 * old:
                info["termination"] = 3
                it += 1
                break
        else:
            radius /= decrease
            decrease *= 2.0
            if radius < 1e-32:
                info["termination"] = 4
                break
        it += 1
    info.update(final_cost=c, iterations=it)
    return x, info


def solve_scipy(init_rt, pts2d, pts3d, wgt2d, K):
    """An independent minimiser (MINPACK's Levenberg-Marquardt through scipy) run to machine precision."""
    from scipy.optimize import least_squares
    f = lambda q: residuals(q, pts2d, pts3d, wgt2d, K).reshape(-1)                # noqa: E731
    j = lambda q: residuals(q, pts2d, pts3d, wgt2d, K, True)[1].reshape(-1, 6)    # noqa: E731
    s = least_squares(f, np.asarray(init_rt, np.float64), jac=j, method="lm", xtol=1e-15, ftol=1e-15, gtol=1e-15,
                      max_nfev=2000)
    return s.x, dict(final_cost=float(s.cost), nfev=int(s.nfev))


def rodrigues(w):
    """cv2.Rodrigues(w)[0] (un_pnp_utils.py:55): the rotation matrix of an angle-axis vector."""
    return np.stack([angle_axis_rotate_point(w, e) if float(np.dot(w, w)) > EPS else e + np.cross(w, e)
                     for e in np.eye(3)], 1)


# ---------------------------------------------------------------------------------------------------------------
_ref = None


def ref():
    """oracle/_ref/libref_uncertainty_pnp.so (the reference's own source + shim ceres) or None when not built."""
    global _ref
    if _ref is None:
        path = os.path.join(HERE, "_ref", "libref_uncertainty_pnp.so")
        if not os.path.exists(path):
            return None
        L = ctypes.CDLL(path)
        dp = ctypes.POINTER(ctypes.c_double)
        L.refpnp_eval.restype = ctypes.c_double
        L.refpnp_eval.argtypes = [dp, dp, dp, dp, dp, ctypes.c_int, dp, dp]
        L.uncertainty_pnp.restype = None
        L.uncertainty_pnp.argtypes = [dp, dp, dp, dp, dp, dp, ctypes.c_int]
        L.refpnp_set_max_iterations.argtypes = [ctypes.c_int]
        L.refpnp_last_summary.argtypes = [dp, dp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        _ref = L
    return _ref


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def ref_eval(rt, pts2d, pts3d, wgt2d, K):
    """Residuals [pn,2], Jacobian [pn,2,6] and cost of the REFERENCE'S functor (through its vendored Jets)."""
    L = ref()
    a = [np.ascontiguousarray(v, np.float64) for v in (pts2d, pts3d, wgt2d, K, rt)]
    pn = a[0].shape[0]
    res, jac = np.zeros((pn, 2)), np.zeros((pn, 2, 6))
    c = L.refpnp_eval(_dp(a[0]), _dp(a[1]), _dp(a[2]), _dp(a[3]), _dp(a[4]), pn, _dp(res), _dp(jac))
    return res, jac, float(c)


def ref_solve(init_rt, pts2d, pts3d, wgt2d, K):
    """The reference's own C entry point uncertainty_pnp() (its functor, the shim's Levenberg-Marquardt)."""
    L = ref()
    a = [np.ascontiguousarray(v, np.float64) for v in (pts2d, pts3d, wgt2d, K, init_rt)]
    out = np.zeros(6)
    L.refpnp_set_max_iterations(-1)
    L.uncertainty_pnp(_dp(a[0]), _dp(a[1]), _dp(a[2]), _dp(a[3]), _dp(a[4]), _dp(out), a[0].shape[0])
    ic, fc = ctypes.c_double(), ctypes.c_double()
    it, tm = ctypes.c_int(), ctypes.c_int()
    L.refpnp_last_summary(ctypes.byref(ic), ctypes.byref(fc), ctypes.byref(it), ctypes.byref(tm))
    return out, dict(initial_cost=ic.value, final_cost=fc.value, iterations=it.value, termination=tm.value)
