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


@pytest.mark.parametrize("T", [0.9, 0.99, 0.999])
def test_packed_valu_band(T):
    """The sqrt/divide-free decision on f32 operands (round 1's k_count_fast, now the model of k_count_bf16's second level,
    whose f32-computed unit normals get the wider beta2): d = fl(h-c); nh, B binary64 quotients rounded once;
    a = fma(dx,nhx, dy*nhy); t = a - |b'|."""
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
    beta0 = 1.25 * (32 * (1 + k0) + 8 / s20) * U / Td0
    cx, cy, hx, hy, nx, ny = samples(n, T, 12, spread=5 * beta0 * Td0 * np.sqrt(s20) / np.arccos(Td0))
    rng = np.random.RandomState(5)
    ox = (cx - rng.randint(0, 120, n)).astype(f32)                         # block origin: within ~120 px, integer
    oy = (cy - rng.randint(0, 4, n)).astype(f32)
    C1 = f32(124.0)
    Td = np.float64(f32(T)); s2 = 1 - Td * Td; kappa = Td / np.sqrt(s2)
    beta = f32(1.25 * (32 * (1 + kappa) + 8 / s2) * U / Td)
    eps = f32(1.25 * (1 + kappa) * 34 * U) * C1 + f32(1.5e-6 * (1 + kappa))
    kf = f32(kappa)
    norm1 = np.sqrt(nx * nx + ny * ny)
    ux, uy = nx / norm1, ny / norm1                                        # binary32, as the kernel
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
