/*
 * pvnet_pnp.h -- C ABI of libpvnet_pnp.so: clean-pvnet's uncertainty-weighted PnP refinement
 * (lib/csrc/uncertainty_pnp/src/uncertainty_pnp.cpp, called per image by lib/evaluators/linemod/pvnet.py:118-132
 * through lib/csrc/uncertainty_pnp/un_pnp_utils.py:6-57) as a batched HIP kernel for gfx950.
 * SURVEY.md section 8(f) rank 3, second half.
 *
 * The problem (uncertainty_pnp.cpp:19-38): pose = (angle-axis w[3], t[3]); for every keypoint i
 *     X = R(w) P_i + t,   (u, v) = (fx X/Z + px, fy Y/Z + py),   d = (u, v) - p_i,
 *     r_i = [[wxx, wxy], [wxy, wyy]] d,          minimise 1/2 sum |r_i|^2 over the 6 parameters.
 * The reference hands this to Ceres (trust region, Levenberg-Marquardt, DENSE_SCHUR, default options).  Here one
 * wavefront per image runs Levenberg-Marquardt in binary64 with an analytic Jacobian and Ceres' documented default
 * schedule (radius 1e4, accept rho > 1e-3, radius /= max(1/3, 1-(2rho-1)^3) | halved with a doubling factor, tolerances
 * 1e-6 / 1e-10 / 1e-8, 50 iterations).  Same minimum as any correct minimiser from the same start; Ceres' own
 * iterate path is not reproduced (its library cannot be built or linked in this environment).
 */
#ifndef PVNET_PNP_H_
#define PVNET_PNP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The reference's exported symbol, same signature (uncertainty_pnp.cpp:61-92, declared in src/ext.h, bound by cffi in
 * un_pnp_utils.py:1,47): HOST pointers, one image; allocates, copies and frees device memory per call.
 *   pts2d [pn,2], pts3d [pn,3], wgt2d [pn,3] = (wxx,wxy,wyy), K [3,3] row-major, init_rt [6], result_rt [6].
 * On a HIP error prints to stderr and returns init_rt in result_rt. */
void uncertainty_pnp(double *pts2d, double *pts3d, double *wgt2d, double *K, double *init_rt, double *result_rt,
                     int pn);

/* Batched form on DEVICE pointers, launched on `stream` (hipStream_t as void*), no allocation, no synchronisation.
 *   d_pts2d  [B,pn,2]       d_wgt2d [B,pn,3]       d_init_rt [B,6]      d_result_rt [B,6]
 *   d_pts3d  [pn,3] shared by all images when pts3d_batched == 0 (one object model), else [B,pn,3]
 *   d_K      [9]    shared when K_batched == 0, else [B,9]
 *   d_info   [B,4] f64 or NULL: initial cost, final cost, iterations, termination (1 gradient, 2 parameter,
 *            3 function tolerance, 4 radius underflow, 0 iteration limit)
 *   max_iterations <= 0 selects Ceres' default 50, function_tolerance <= 0 its default 1e-6 (the relative cost
 *   decrease below which an accepted step ends the iteration; pass e.g. 1e-15 to run to the minimum).  pn in [1, 4096].
 * Returns 0, -1 (bad arguments) or a hipError_t. */
int pvp_uncertainty_pnp_batched(const double *d_pts2d, const double *d_pts3d, const double *d_wgt2d, const double *d_K,
                                const double *d_init_rt, double *d_result_rt, double *d_info, int B, int pn,
                                int pts3d_batched, int K_batched, int max_iterations, double function_tolerance,
                                void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PVNET_PNP_H_ */
