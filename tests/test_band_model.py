"""CPU model of the guard bands: numpy emulations of the two fast inlier tests on adversarial
samples concentrated around the threshold.  Claim checked: whenever the fast path says "outside the band"
(|t| - beta*a > eps) its decision sign(t) equals the exact binary32 decision of ransac_voting_kernel.cu:100-125.
The constants are computed exactly as clean-pvnet_amd/csrc/pvnet_vote.hip does (fast_consts / bf16_consts)."""
import numpy as np
import pytest

f32 = np.float32
U = 2.0 ** -24


def exact_decision(cx, cy, hx, hy, nx, ny, T):
    """K:100-125 in binary32, one rounding per operation (numpy ufuncs on float32 arrays)."""
    with np.errstate(all="ignore"):
        dx, dy = hx - cx, hy - cy
        norm1 = np.sqrt(nx * nx + ny * ny)
        norm2 = np.sqrt(dx * dx + dy * dy)
        ang = (dx * nx + dy * ny) / (norm1 * norm2)
        return (ang > f32(T)) & ~(norm1 <= f32(1e-6)) & ~(norm2 <= f32(1e-6))


def fma32(a, b, c):
    """binary32 fma emulated in binary64 (product exact; the double rounding of the sum is immaterial here)."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def samples(n, T, seed, spread):
    """pixels c, hypotheses h and directions n whose angle to (h - c) is acos(T) * (1 +- spread)."""
    rng = np.random.RandomState(seed)
    cx = rng.randint(0, 640, n).astype(f32)
    cy = rng.randint(0, 480, n).astype(f32)
    r = np.exp(rng.uniform(np.log(0.3), np.log(3000.0), n))
    phi = rng.uniform(0, 2 * np.pi, n)
    hx = (cx + r * np.cos(phi)).astype(f32)
    hy = (cy + r * np.sin(phi)).astype(f32)
    d = np.stack([hx.astype(np.float64) - cx, hy.astype(np.float64) - cy], 1)
    base = np.arctan2(d[:, 1], d[:, 0])
    off = np.arccos(np.float64(f32(T))) * (1 + rng.uniform(-spread, spread, n)) * rng.choice([-1, 1], n)
    s = np.exp(rng.uniform(np.log(1e-3), np.log(50.0), n))
    nx = (np.cos(base + off) * s).astype(f32)
    ny = (np.sin(base + off) * s).astype(f32)
    return cx, cy, hx, hy, nx, ny


def rsq_unit_normal(nx, ny, seed):
    """The kernels' unit normal: dd = fl(nx^2 + ny^2) in binary32, r = v_rsq_f32(dd) modelled as the correctly rounded
    1/sqrt perturbed by an adversarial +-1 ulp (measured on MI355X over all normal inputs: <= 0.8633 ulp,
    tools/microbench/rsq_accuracy.hip), n = (fl(nx r), fl(ny r))."""
    rng = np.random.RandomState(seed)
    with np.errstate(all="ignore"):
        dd = nx * nx + ny * ny
        r = (1.0 / np.sqrt(dd.astype(np.float64))).astype(f32)
        r = np.nextafter(r, np.where(rng.rand(len(r)) < 0.5, f32(np.inf), f32(-np.inf)).astype(f32))
        return nx * r, ny * r, dd


DD_ALIVE = np.array([0x2b8cbcce], dtype=np.uint32).view(f32)[0]     # count_bf16.hpp: kDdAlive


def test_dead_pixel_threshold_is_the_exact_reject():
    """K:121 rejects a pixel when (double)sqrtf(dd) < 1e-6; sqrtf is monotone, so that is a threshold on dd: kDdAlive must
    be the smallest binary32 the exact test lets through (and the constant in the source must be this one)."""
    import os
    import re
    src = open(os.path.join(os.path.dirname(__file__), "..", "clean-pvnet_amd", "csrc", "count_bf16.hpp")).read()
    m = re.search(r"#define kDdAlive __uint_as_float\(0x([0-9a-f]+)u\)", src)
    assert m and int(m.group(1), 16) == 0x2b8cbcce
    below = np.nextafter(DD_ALIVE, f32(0))
    assert not (np.float64(np.sqrt(DD_ALIVE)) < 1e-6) and (np.float64(np.sqrt(below)) < 1e-6)
    xs = np.arange(0x2b8cbcce - 5000, 0x2b8cbcce + 5000, dtype=np.uint32).view(f32)
    assert np.array_equal(xs >= DD_ALIVE, ~(np.sqrt(xs).astype(np.float64) < 1e-6))


@pytest.mark.parametrize("T", [0.9, 0.99, 0.999])
def test_second_level_band_with_rsq_normals(T):
    """k_count_bf16's second level as the kernel runs it from round 3 on: d = fl(h - c), the unit normal from v_rsq_f32
    (components within 4u), B = kappa perp(n) in f32, one fma each; band beta2 = 1.25 (8 (1+kappa) + 8/(1-T^2)) u / T,
    eps0.  Outside the band the decision is the exact one."""
    n = 2_000_000
    Td = np.float64(f32(T)); s2 = 1 - Td * Td; kappa = Td / np.sqrt(s2)
    beta = f32(1.25 * (8 * (1 + kappa) + 8 / s2) * U / Td)
    eps = f32(1.5e-6 * (1 + kappa))
    kf = f32(kappa)
    band_angle = float(beta) * Td * np.sqrt(s2)
    cx, cy, hx, hy, nx, ny = samples(n, T, 31, spread=5 * band_angle / np.arccos(Td))
    ux, uy, _dd = rsq_unit_normal(nx, ny, 32)
    bx, by = -kf * uy, kf * ux
    dx, dy = hx - cx, hy - cy
    a = fma32(dx, ux, dy * uy)
    b = fma32(dx, bx, dy * by)
    t = a - np.abs(b)
    outside = fma32(np.full(n, -beta, f32), a, np.abs(t)) > eps
    ex = exact_decision(cx, cy, hx, hy, nx, ny, T)
    assert (outside.mean() > 0.1) and ((~outside).mean() > 0.005)
    assert np.array_equal((t > 0)[outside], ex[outside])
    assert ((t > 0) != ex).sum() > 0


@pytest.mark.parametrize("T", [0.9, 0.99, 0.999])
def test_packed_valu_band(T):
    """The sqrt/divide-free decision on f32 operands with binary64 quotients rounded once as normals (round 1's
    k_count_fast; the second level of k_count_bf16 as it runs today is test_second_level_band_with_rsq_normals above):
    d = fl(h-c); a = fma(dx,nhx, dy*nhy); t = a - |b'|."""
    n = 2_000_000
    Td = np.float64(f32(T)); s2 = 1 - Td * Td; kappa = Td / np.sqrt(s2)
    beta = f32(1.25 * (3 * (1 + kappa) + 8 / s2) * U / Td)
    eps = f32(1.5e-6 * (1 + kappa))
    band_angle = float(beta) * Td * np.sqrt(s2)                           # |t| <= beta a  <=>  |dtheta| <= beta T s
    cx, cy, hx, hy, nx, ny = samples(n, T, 11, spread=5 * band_angle / np.arccos(Td))
    N1 = np.sqrt(nx.astype(np.float64) ** 2 + ny.astype(np.float64) ** 2)
    nhx, nhy = (nx / N1).astype(f32), (ny / N1).astype(f32)
    Bx, By = (-kappa * ny / N1).astype(f32), (kappa * nx / N1).astype(f32)
    dx, dy = hx - cx, hy - cy
    a = fma32(dx, nhx, dy * nhy)
    b = fma32(dx, Bx, dy * By)
    t = a - np.abs(b)
    z = fma32(np.full(n, -beta, f32), a, np.abs(t))
    outside = z > eps
    ex = exact_decision(cx, cy, hx, hy, nx, ny, T)
    assert (outside.mean() > 0.1) and ((~outside).mean() > 0.005)         # the sample really straddles the band
    assert np.array_equal((t > 0)[outside], ex[outside])
    # and the band is not vacuous: ignoring it DOES produce wrong decisions on this sample
    assert ((t > 0) != ex).sum() > 0


def split3(x):
    def bf(v):                                                            # round-to-nearest-even to 8 significant bits
        u = v.astype(f32).view(np.uint32).astype(np.uint64)
        u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
        return u.astype(np.uint32).view(f32)
    p0 = bf(x); r = x - p0
    p1 = bf(r); r = r - p1
    p2 = bf(r)
    return p0, p1, p2


@pytest.mark.parametrize("T", [0.9, 0.99, 0.999])
def test_split_bf16_prefilter_band(T):
    """k_count_bf16: origin-translated operands split into three bf16 pieces, 15 piece products accumulated with the
    WORST-CASE model of the MFMA (every partial sum rounded to binary32, adversarial order = as listed), band from
    bf16_consts.  C1 = the block extent."""
    n = 1_000_000
    Td0 = np.float64(f32(T)); s20 = 1 - Td0 * Td0; k0 = Td0 / np.sqrt(s20)
    beta0 = 1.25 * (33 * (1 + k0) + 8 / s20) * U / Td0
    cx, cy, hx, hy, nx, ny = samples(n, T, 12, spread=5 * beta0 * Td0 * np.sqrt(s20) / np.arccos(Td0))
    rng = np.random.RandomState(5)
    ox = (cx - rng.randint(0, 120, n)).astype(f32)                         # block origin: within ~120 px, integer
    oy = (cy - rng.randint(0, 4, n)).astype(f32)
    C1 = f32(124.0)
    Td = np.float64(f32(T)); s2 = 1 - Td * Td; kappa = Td / np.sqrt(s2)
    beta = f32(1.25 * (33 * (1 + kappa) + 8 / s2) * U / Td)
    eps = f32(1.25 * (1 + kappa) * 35 * U) * C1 + f32(1.5e-6 * (1 + kappa))
    kf = f32(kappa)
    ux, uy, _dd = rsq_unit_normal(nx, ny, 13)                              # binary32 with v_rsq_f32, as the kernel
    bx, by = -kf * uy, kf * ux
    cpx, cpy = cx - ox, cy - oy
    cn = -(cpx * ux + cpy * uy)
    cb = -(cpx * bx + cpy * by)
    hpx, hpy = hx - ox, hy - oy

    def dot15(vx, vy, cv):
        qx, qy = split3(hpx), split3(hpy)
        px, py, pc = split3(vx), split3(vy), split3(cv)
        terms = [px[0] * qx[0], px[0] * qx[1], px[0] * qx[2], px[1] * qx[0], px[1] * qx[1], px[2] * qx[0],
                 py[0] * qy[0], py[0] * qy[1], py[0] * qy[2], py[1] * qy[0], py[1] * qy[1], py[2] * qy[0],
                 pc[0], pc[1], pc[2]]
        acc = np.zeros(n, f32)
        for tm in terms:                                                   # products of bf16 pairs are exact in f32
            acc = acc + tm.astype(f32)                                     # one binary32 rounding per add
        return acc
    a = dot15(ux, uy, cn)
    b = dot15(bx, by, cb)
    t = a - np.abs(b)
    z = fma32(np.full(n, -beta, f32), a, np.abs(t))
    outside = z > eps
    ex = exact_decision(cx, cy, hx, hy, nx, ny, T)
    assert outside.mean() > 0.1
    assert np.array_equal((t > 0)[outside], ex[outside])


@pytest.mark.parametrize("T", [0.6, 0.9, 0.99, 0.999, 0.9999])
def test_lead_sure_inlier_test_is_a_lower_bound(T):
    """k_lead (count_prune.hpp): a pixel is counted for a leader only if it is an inlier BEYOND DOUBT -- the second-level
    test with the unit normal from v_rsq_f32 and twice the second level's band.  Claim: sure => the exact binary32 vote
    (K:100-125) accepts it, so the sum is a lower bound of the leader's exact count.  rsq is modelled as the correctly
    rounded 1/sqrt perturbed by an adversarial +-1 ulp (its documented accuracy); samples straddle the threshold far more
    closely than the band."""
    n = 2_000_000
    Td = np.float64(f32(T)); s2 = 1 - Td * Td; kappa = Td / np.sqrt(s2)
    beta2 = 1.25 * (8 * (1 + kappa) + 8 / s2) * U / Td
    beta, eps = f32(2 * f32(beta2)), f32(2 * f32(1.5e-6 * (1 + kappa)))
    kf = f32(kappa)
    cx, cy, hx, hy, nx, ny = samples(n, T, 21, spread=6 * 2 * beta2 * Td * np.sqrt(s2) / np.arccos(Td))
    rng = np.random.RandomState(22)
    with np.errstate(all="ignore"):
        dd = nx * nx + ny * ny                                              # binary32, as the kernel
        r = (1.0 / np.sqrt(dd.astype(np.float64))).astype(f32)
        r = np.nextafter(r, np.where(rng.rand(n) < 0.5, f32(np.inf), f32(-np.inf)).astype(f32))   # +-1 ulp
        ok = (dd > f32(4e-12)) & (dd < f32(np.inf))
        ux, uy = nx * r, ny * r
        bx, by = -kf * uy, kf * ux
        dx, dy = hx - cx, hy - cy
        a = fma32(dx, ux, dy * uy)
        b = fma32(dx, bx, dy * by)
        t = a - np.abs(b)
        sure = ok & (t > 0) & (fma32(np.full(n, -beta, f32), a, t) > eps)
    ex = exact_decision(cx, cy, hx, hy, nx, ny, T)
    assert sure.mean() > 0.1 and (ex & ~sure).mean() > 0.005                # plenty of sure ones, plenty left in doubt
    assert not (sure & ~ex).any()                                          # never counts what the exact vote rejects
    # tiny and huge directions: what the exact vote rejects outright (norm1 < 1e-6) is never sure
    tiny = f32(rng.uniform(0, 1.5e-6, n))
    ang = rng.uniform(0, 2 * np.pi, n)
    nx2, ny2 = (tiny * np.cos(ang)).astype(f32), (tiny * np.sin(ang)).astype(f32)
    with np.errstate(all="ignore"):
        dd2 = nx2 * nx2 + ny2 * ny2
    ok2 = (dd2 > f32(4e-12)) & (dd2 < f32(np.inf))
    ex2 = exact_decision(cx, cy, hx, hy, nx2, ny2, T)
    assert not (ok2 & ~(np.sqrt(nx2 * nx2 + ny2 * ny2) > f32(1e-6))).any() and not (ex2 & ~ok2).sum() > ex2.sum()  # ok2 => norm1 > 1e-6


def test_staged_elimination_never_drops_a_winner():
    """The elimination rule itself on random count tables: partial counts over a part S of the
    pixels, L* = ANY lower bound of a leader's full count (here: partial + a random share of its true rest), keep h iff
    partial(h) + R >= L*.  Every hypothesis with the maximal full count is kept -- in particular the first one -- and a
    dropped hypothesis' partial count is below the maximum."""
    rng = np.random.RandomState(3)
    for trial in range(2000):
        hn = int(rng.choice([32, 100, 512]))
        tn = int(rng.randint(600, 8000))
        nS = int(tn * rng.uniform(0.1, 0.6))
        rho = rng.uniform(0.05, 1.0, hn) ** rng.choice([0.3, 1.0, 3.0])     # true inlier ratios, varied shapes
        if trial % 3 == 0:
            rho[rng.randint(hn, size=3)] = rho.max()                        # ties at the top
        partial = rng.binomial(nS, rho)
        rest = rng.binomial(tn - nS, rho)
        full = partial + rest
        if trial % 3 == 0:                                                  # make the ties exact: the same full count, split
            top = np.flatnonzero(rho == rho.max())                          # differently between the sample and the rest
            ft = int(full[top].min())
            for h in top:
                lo, hi = max(0, ft - (tn - nS)), min(nS, ft)
                partial[h] = rng.randint(lo, hi + 1)
                rest[h] = ft - partial[h]
            full = partial + rest
        lead = np.argsort(-partial, kind="stable")[:2]
        lstar = max(int(partial[h] + rng.randint(0, rest[h] + 1)) for h in lead)      # any lower bound of a full count
        keep = partial + (tn - nS) >= lstar
        winners = np.flatnonzero(full == full.max())
        assert keep[winners].all()
        assert (partial[~keep] < full.max()).all()
        # the arg-max over what the counters hold afterwards (full for kept, partial for dropped) is the arg-max of full
        after = np.where(keep, full, partial)
        assert int(np.argmax(after)) == int(np.argmax(full)) and after.max() == full.max()
