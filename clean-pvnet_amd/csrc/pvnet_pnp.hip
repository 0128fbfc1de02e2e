// pvnet_pnp.hip -- uncertainty-weighted PnP refinement of clean-pvnet, batched, native HIP for gfx950.
// Replaces lib/csrc/uncertainty_pnp/src/uncertainty_pnp.cpp (Ceres, one image per call on the host):
//   residual functor  ReprojectionErrorArray::operator()   uncertainty_pnp.cpp:19-38
//   rotation          ceres::AngleAxisRotatePoint           include/ceres/rotation.h:563-622 (Rodrigues, first-order branch
//                                                            for theta^2 <= DBL_EPSILON)
//   solve             ceres::Solve, default options          uncertainty_pnp.cpp:71-89
// One wavefront per image (4 images per block): lane i owns keypoint i (PVNet: 9), computes its two residuals and their
// 2x6 Jacobian analytically in binary64, the 28 sums (JtJ upper triangle, Jt r, cost) are reduced through LDS, and every
// lane solves the damped 6x6 system by Cholesky itself (no broadcast).  Nothing here is throughput: it is a latency chain
// of ~10 iterations, which is why it lives on the GPU at all -- the keypoints and weights are already there (the voting
// layers produced them), and a batch of images costs what one image costs.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>

#include "pvnet_pnp.h"

#define PVP_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kSums = 28;   // 21 (JtJ upper triangle) + 6 (Jt r) + 1 (cost)

struct Cam { double fx, fy, px, py; };

// X = R(w) P (rotation.h:563-622) and, when dX != nullptr, dX[r][c] = d X_r / d w_c -- the derivative of exactly that
// expression (what Ceres' Jets compute), including the first-order branch near zero.
__device__ __forceinline__ void rotate_point(const double w[3], const double P[3], double X[3], double (*dX)[3])
{
    const double theta2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    if (theta2 > 2.220446049250313e-16) {
        const double theta = sqrt(theta2), c = cos(theta), s = sin(theta), ti = 1.0 / theta;
        const double a[3] = {w[0] * ti, w[1] * ti, w[2] * ti};
        const double axP[3] = {a[1] * P[2] - a[2] * P[1], a[2] * P[0] - a[0] * P[2], a[0] * P[1] - a[1] * P[0]};
        const double adP = a[0] * P[0] + a[1] * P[1] + a[2] * P[2];
        const double tmp = adP * (1.0 - c);
        for (int r = 0; r < 3; ++r) X[r] = P[r] * c + axP[r] * s + a[r] * tmp;
        if (dX) {
            // d theta / d w_j = a_j;  d a / d w_j = (e_j - a a_j) / theta
            double base[3];                                   // d X / d theta at fixed axis
            for (int r = 0; r < 3; ++r) base[r] = -s * P[r] + c * axP[r] + s * adP * a[r];
            for (int j = 0; j < 3; ++j) {
                double da[3];
                for (int r = 0; r < 3; ++r) da[r] = ((r == j ? 1.0 : 0.0) - a[r] * a[j]) * ti;
                const double daxP[3] = {da[1] * P[2] - da[2] * P[1], da[2] * P[0] - da[0] * P[2], da[0] * P[1] - da[1] * P[0]};
                const double dadP = da[0] * P[0] + da[1] * P[1] + da[2] * P[2];
                for (int r = 0; r < 3; ++r)
                    dX[r][j] = a[j] * base[r] + s * daxP[r] + (1.0 - c) * (da[r] * adP + a[r] * dadP);
            }
        }
    } else {
        X[0] = P[0] + (w[1] * P[2] - w[2] * P[1]);
        X[1] = P[1] + (w[2] * P[0] - w[0] * P[2]);
        X[2] = P[2] + (w[0] * P[1] - w[1] * P[0]);
        if (dX) {                                             // d (w x P) / d w_j = e_j x P
            dX[0][0] = 0.0;   dX[0][1] = P[2];  dX[0][2] = -P[1];
            dX[1][0] = -P[2]; dX[1][1] = 0.0;   dX[1][2] = P[0];
            dX[2][0] = P[1];  dX[2][1] = -P[0]; dX[2][2] = 0.0;
        }
    }
}

// residuals of one keypoint at pose x (uncertainty_pnp.cpp:19-38); J[k][6] when J != nullptr
__device__ __forceinline__ void point_residual(const double x[6], const double P[3], const double p2[2], const double wg[3],
                                               const Cam &cam, double r[2], double (*J)[6])
{
    double X[3], dX[3][3];
    rotate_point(x, P, X, J ? dX : nullptr);
    X[0] += x[3]; X[1] += x[4]; X[2] += x[5];
    const double iz = 1.0 / X[2];
    const double dx = cam.fx * X[0] * iz + cam.px - p2[0];
    const double dy = cam.fy * X[1] * iz + cam.py - p2[1];
    r[0] = wg[0] * dx + wg[1] * dy;
    r[1] = wg[1] * dx + wg[2] * dy;
    if (J) {
        // d(u,v)/dX
        const double ux = cam.fx * iz, uz = -cam.fx * X[0] * iz * iz, vy = cam.fy * iz, vz = -cam.fy * X[1] * iz * iz;
        for (int j = 0; j < 6; ++j) {
            double gx, gy, gz;                                // dX/dparam_j
            if (j < 3) { gx = dX[0][j]; gy = dX[1][j]; gz = dX[2][j]; }
            else { gx = j == 3 ? 1.0 : 0.0; gy = j == 4 ? 1.0 : 0.0; gz = j == 5 ? 1.0 : 0.0; }
            const double du = ux * gx + uz * gz, dv = vy * gy + vz * gz;
            J[0][j] = wg[0] * du + wg[1] * dv;
            J[1][j] = wg[1] * du + wg[2] * dv;
        }
    }
}

// (A + diag(d)) s = b for the symmetric 6x6 A given by its upper triangle (row-major packed, 21 entries); false when
// the damped matrix is not positive definite
__device__ __forceinline__ bool solve6(const double *Au, const double d[6], const double b[6], double s[6])
{
    double L[6][6];
    int idx = 0;
    double A[6][6];
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j) { A[i][j] = Au[idx]; A[j][i] = Au[idx]; ++idx; }
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j <= i; ++j) {
            double v = A[i][j] + (i == j ? d[i] : 0.0);
            for (int k = 0; k < j; ++k) v -= L[i][k] * L[j][k];
            if (i == j) {
                if (!(v > 0.0)) return false;
                L[i][i] = sqrt(v);
            } else {
                L[i][j] = v / L[j][j];
            }
        }
    double y[6];
    for (int i = 0; i < 6; ++i) {
        double v = b[i];
        for (int k = 0; k < i; ++k) v -= L[i][k] * y[k];
        y[i] = v / L[i][i];
    }
    for (int i = 5; i >= 0; --i) {
        double v = y[i];
        for (int k = i + 1; k < 6; ++k) v -= L[k][i] * s[k];
        s[i] = v / L[i][i];
    }
    return true;
}

__global__ __launch_bounds__(64 * kWavesPerBlock) void k_uncertainty_pnp(
    const double *__restrict__ pts2d, const double *__restrict__ pts3d, const double *__restrict__ wgt2d,
    const double *__restrict__ Kmat, const double *__restrict__ init_rt, double *__restrict__ result_rt,
    double *__restrict__ info, int B, int pn, int pts3d_batched, int K_batched, int max_iter, double ftol)
{
    __shared__ double s_part[kWavesPerBlock][64][kSums + 1];   // per-lane contributions (one row per lane, padded)
    __shared__ double s_sum[kWavesPerBlock][kSums];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * kWavesPerBlock + wave;
    if (b >= B) return;                                        // whole wave: no block-level barrier below
    const double *p2 = pts2d + (size_t)b * pn * 2;
    const double *p3 = pts3d + (pts3d_batched ? (size_t)b * pn * 3 : 0);
    const double *wg = wgt2d + (size_t)b * pn * 3;
    const double *Kb = Kmat + (K_batched ? (size_t)b * 9 : 0);
    const Cam cam = {Kb[0], Kb[4], Kb[2], Kb[5]};              // uncertainty_pnp.cpp:79
    double x[6];
    for (int i = 0; i < 6; ++i) x[i] = init_rt[(size_t)b * 6 + i];

    // sums over the keypoints at pose q: with_jac -> all 28, else the cost only (slot 27).  Wave-synchronous: the LDS
    // rows are private to the wave and its lanes run in lockstep; the fences order the LDS traffic.
    auto accumulate = [&](const double q[6], bool with_jac) {
        double acc[kSums];
        for (int k = 0; k < kSums; ++k) acc[k] = 0.0;
        for (int i = lane; i < pn; i += 64) {
            const double P[3] = {p3[i * 3], p3[i * 3 + 1], p3[i * 3 + 2]};
            const double pp[2] = {p2[i * 2], p2[i * 2 + 1]};
            const double ww[3] = {wg[i * 3], wg[i * 3 + 1], wg[i * 3 + 2]};
            double r[2], J[2][6];
            point_residual(q, P, pp, ww, cam, r, with_jac ? J : nullptr);
            acc[27] += 0.5 * (r[0] * r[0] + r[1] * r[1]);
            if (with_jac) {
                int idx = 0;
                for (int u = 0; u < 6; ++u) {
                    for (int w2 = u; w2 < 6; ++w2) acc[idx++] += J[0][u] * J[0][w2] + J[1][u] * J[1][w2];
                    acc[21 + u] += J[0][u] * r[0] + J[1][u] * r[1];
                }
            }
        }
        const int first = with_jac ? 0 : 27;
        for (int k = first; k < kSums; ++k) s_part[wave][lane][k] = acc[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int nl = pn < 64 ? pn : 64;
        if (lane >= first && lane < kSums) {
            double t = 0.0;
            for (int l = 0; l < nl; ++l) t += s_part[wave][l][lane];     // fixed order: deterministic
            s_sum[wave][lane] = t;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    double radius = 1e4, decrease = 2.0;
    accumulate(x, true);
    double JtJ[21], g[6], cost;
    for (int k = 0; k < 21; ++k) JtJ[k] = s_sum[wave][k];
    for (int k = 0; k < 6; ++k) g[k] = s_sum[wave][21 + k];
    cost = s_sum[wave][27];
    const double initial_cost = cost;
    // Ceres' Jacobi scaling: column i is scaled by 1 / (1 + |J_i|) with the norms of the INITIAL point (trust_region_minimizer);
    // the LM diagonal is the squared norm of the SCALED column clamped to [min_lm_diagonal, max_lm_diagonal] = [1e-6, 1e32]
    // (levenberg_marquardt_strategy).  In unscaled coordinates: damping_i = clamp(JtJ_ii s_i^2) / (s_i^2 radius)  (ADVICE r2).
    double jscale2[6];
    {
        int idx = 0;
        for (int u = 0; u < 6; ++u) {
            const double sc = 1.0 / (1.0 + sqrt(JtJ[idx]));
            jscale2[u] = sc * sc;
            idx += 6 - u;
        }
    }
    const int limit = max_iter > 0 ? max_iter : 50;
    int it = 0, term = 0;
    for (; it < limit; ++it) {
        double gmax = 0.0;
        for (int k = 0; k < 6; ++k) gmax = fmax(gmax, fabs(g[k]));
        if (gmax <= 1e-10) { term = 1; break; }
        double d[6], ng[6], step[6];
        {
            int idx = 0;
            for (int u = 0; u < 6; ++u) {
                const double di = fmin(fmax(JtJ[idx] * jscale2[u], 1e-6), 1e32);
                d[u] = di / (jscale2[u] * radius);
                ng[u] = -g[u];
                idx += 6 - u;
            }
        }
        const bool ok = solve6(JtJ, d, ng, step);
        double xn = 0.0, sn = 0.0, model = 0.0;
        if (ok) {
            double A[6][6];
            int idx = 0;
            for (int i = 0; i < 6; ++i)
                for (int j = i; j < 6; ++j) { A[i][j] = JtJ[idx]; A[j][i] = JtJ[idx]; ++idx; }
            for (int i = 0; i < 6; ++i) {
                xn += x[i] * x[i]; sn += step[i] * step[i];
                double Js = 0.0;
                for (int j = 0; j < 6; ++j) Js += A[i][j] * step[j];
                model -= step[i] * (g[i] + 0.5 * Js);
            }
            if (sqrt(sn) <= 1e-8 * (sqrt(xn) + 1e-8)) { term = 2; break; }
        }
        double cand[6], new_cost = cost, rho = -1.0;
        if (ok && model > 0.0) {
            for (int i = 0; i < 6; ++i) cand[i] = x[i] + step[i];
            accumulate(cand, false);
            new_cost = s_sum[wave][27];
            rho = (cost - new_cost) / model;
        }
        if (rho > 1e-3 && isfinite(new_cost)) {
            const double change = cost - new_cost, old = cost;
            for (int i = 0; i < 6; ++i) x[i] = cand[i];
            const double t = 2.0 * rho - 1.0;
            radius = fmin(radius / fmax(1.0 / 3.0, 1.0 - t * t * t), 1e16);
            decrease = 2.0;
            accumulate(x, true);
            for (int k = 0; k < 21; ++k) JtJ[k] = s_sum[wave][k];
            for (int k = 0; k < 6; ++k) g[k] = s_sum[wave][21 + k];
            cost = s_sum[wave][27];
            if (fabs(change) <= ftol * old) { term = 3; ++it; break; }
        } else {
            radius /= decrease;
            decrease *= 2.0;
            if (radius < 1e-32) { term = 4; break; }
        }
    }
    if (lane < 6) result_rt[(size_t)b * 6 + lane] = x[lane];
    if (info && lane == 0) {
        info[(size_t)b * 4] = initial_cost; info[(size_t)b * 4 + 1] = cost;
        info[(size_t)b * 4 + 2] = (double)it; info[(size_t)b * 4 + 3] = (double)term;
    }
}

int launch(const double *pts2d, const double *pts3d, const double *wgt2d, const double *K, const double *init_rt,
           double *result_rt, double *info, int B, int pn, int pts3d_batched, int K_batched, int max_iter, double ftol,
           hipStream_t st)
{
    if (!pts2d || !pts3d || !wgt2d || !K || !init_rt || !result_rt || B <= 0 || pn < 1 || pn > 4096) return -1;
    hipLaunchKernelGGL(k_uncertainty_pnp, dim3((B + kWavesPerBlock - 1) / kWavesPerBlock), dim3(64 * kWavesPerBlock), 0, st,
                       pts2d, pts3d, wgt2d, K, init_rt, result_rt, info, B, pn, pts3d_batched, K_batched, max_iter,
                       ftol > 0.0 ? ftol : 1e-6);
    return (int)hipGetLastError();
}

}  // namespace

PVP_EXPORT int pvp_uncertainty_pnp_batched(const double *d_pts2d, const double *d_pts3d, const double *d_wgt2d,
                                           const double *d_K, const double *d_init_rt, double *d_result_rt, double *d_info,
                                           int B, int pn, int pts3d_batched, int K_batched, int max_iterations,
                                           double function_tolerance, void *stream)
{
    return launch(d_pts2d, d_pts3d, d_wgt2d, d_K, d_init_rt, d_result_rt, d_info, B, pn, pts3d_batched, K_batched,
                  max_iterations, function_tolerance, (hipStream_t)stream);
}

PVP_EXPORT void uncertainty_pnp(double *pts2d, double *pts3d, double *wgt2d, double *K, double *init_rt, double *result_rt,
                                int pn)
{
    for (int i = 0; i < 6; ++i) result_rt[i] = init_rt[i];
    if (pn < 1) return;
    const size_t n2 = sizeof(double) * 2 * pn, n3 = sizeof(double) * 3 * pn;
    double *d = nullptr;                                       // one allocation: pts2d | pts3d | wgt2d | K | init | result
    const size_t total = n2 + n3 + n3 + sizeof(double) * (9 + 6 + 6);
    hipError_t e = hipMalloc(&d, total);
    if (e != hipSuccess) { fprintf(stderr, "uncertainty_pnp: %s\n", hipGetErrorString(e)); return; }
    double *d2 = d, *d3 = d2 + 2 * pn, *dw = d3 + 3 * pn, *dK = dw + 3 * pn, *di = dK + 9, *dr = di + 6;
    auto ok = [&](hipError_t r) { if (e == hipSuccess) e = r; return r == hipSuccess; };
    if (ok(hipMemcpy(d2, pts2d, n2, hipMemcpyHostToDevice)) && ok(hipMemcpy(d3, pts3d, n3, hipMemcpyHostToDevice)) &&
        ok(hipMemcpy(dw, wgt2d, n3, hipMemcpyHostToDevice)) && ok(hipMemcpy(dK, K, sizeof(double) * 9, hipMemcpyHostToDevice)) &&
        ok(hipMemcpy(di, init_rt, sizeof(double) * 6, hipMemcpyHostToDevice))) {
        const int rc = launch(d2, d3, dw, dK, di, dr, nullptr, 1, pn, 0, 0, 0, 0.0, nullptr);
        if (rc == 0) ok(hipMemcpy(result_rt, dr, sizeof(double) * 6, hipMemcpyDeviceToHost));
        else fprintf(stderr, "uncertainty_pnp: bad arguments or launch failure (%d)\n", rc);
    }
    if (e != hipSuccess) fprintf(stderr, "uncertainty_pnp: %s\n", hipGetErrorString(e));
    (void)hipFree(d);
}
