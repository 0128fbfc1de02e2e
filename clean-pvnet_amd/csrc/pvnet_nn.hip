// pvnet_nn.hip -- brute-force nearest-neighbour index search of clean-pvnet's ADD-S metric, native HIP for gfx950.
// Restates lib/csrc/nn/src/nearest_neighborhood.cu:48-117 (one thread per query, linear scan of the references,
// `dist < min_dist` => first minimum) with the references staged through LDS in tiles, so a block of 256 queries
// reads each reference once from global memory instead of 256 times.  Distances are binary32, one rounding per
// operation, no FMA (-ffp-contract=off), in the reference's operand order -- indices are bit-exact against the
// oracle (oracle/vote_oracle.c: orc_find_nearest).
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdio>

#include "pvnet_nn.h"

#pragma clang fp contract(off)

#define PVV_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

constexpr int kBlock = 256;
constexpr int kTile = 1024;   // reference points per LDS tile (12 KB for dim 3)

template <int DIM>
__global__ __launch_bounds__(kBlock) void k_find_nearest(const float *__restrict__ ref, const float *__restrict__ que,
                                                         int *__restrict__ idxs, int pn1, int pn2, int exclude_self)
{
    __shared__ float tile[kTile * DIM];
    const int bi = blockIdx.y;
    const int p2i = blockIdx.x * kBlock + threadIdx.x;
    const float *rb = ref + (size_t)bi * pn1 * DIM;
    float q[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) q[k] = p2i < pn2 ? que[((size_t)bi * pn2 + p2i) * DIM + k] : 0.f;
    float min_dist = FLT_MAX;
    int min_idx = 0;
    for (int t0 = 0; t0 < pn1; t0 += kTile) {
        const int n = min(kTile, pn1 - t0);
        __syncthreads();
        for (int i = threadIdx.x; i < n * DIM; i += kBlock) tile[i] = rb[(size_t)t0 * DIM + i];   // coalesced
        __syncthreads();
        for (int j = 0; j < n; ++j) {                       // every lane reads the same LDS address: broadcast
            const int p1i = t0 + j;
            if (exclude_self && p1i == p2i) continue;
            float dist = 0.f;
            if (DIM == 3) {
                const float dx = tile[j * 3] - q[0], dy = tile[j * 3 + 1] - q[1], dz = tile[j * 3 + 2] - q[2];
                dist = dx * dx + dy * dy + dz * dz;         // (x1-x2)^2 + (y1-y2)^2 + (z1-z2)^2, left to right (:74)
            } else {
                const float dx = tile[j * 2] - q[0], dy = tile[j * 2 + 1] - q[1];
                dist = dx * dx + dy * dy;                   // :107
            }
            if (dist < min_dist) { min_dist = dist; min_idx = p1i; }
        }
    }
    if (p2i < pn2) idxs[(size_t)bi * pn2 + p2i] = min_idx;
}

int launch(const float *d_ref, const float *d_que, int *d_idxs, int b, int pn1, int pn2, int dim, int exclude_self,
           hipStream_t st)
{
    if (!d_ref || !d_que || !d_idxs || b <= 0 || pn1 < 0 || pn2 <= 0 || (dim != 2 && dim != 3) || b > 65535) return -1;
    dim3 grid((pn2 + kBlock - 1) / kBlock, b);
    if (dim == 3)
        hipLaunchKernelGGL(k_find_nearest<3>, grid, dim3(kBlock), 0, st, d_ref, d_que, d_idxs, pn1, pn2, exclude_self);
    else
        hipLaunchKernelGGL(k_find_nearest<2>, grid, dim3(kBlock), 0, st, d_ref, d_que, d_idxs, pn1, pn2, exclude_self);
    return (int)hipGetLastError();
}

}  // namespace

PVV_EXPORT int pvv_nn_find_nearest(const float *d_ref_pts, const float *d_que_pts, int *d_idxs, int b, int pn1,
                                   int pn2, int dim, int exclude_self, void *stream)
{
    return launch(d_ref_pts, d_que_pts, d_idxs, b, pn1, pn2, dim, exclude_self, (hipStream_t)stream);
}

PVV_EXPORT void findNearestPointIdxLauncher(float *ref_pts, float *que_pts, int *idxs, int b, int pn1, int pn2, int dim,
                                            int exclude_self)
{
    float *d_ref = nullptr, *d_que = nullptr;
    int *d_idx = nullptr;
    const size_t nr = (size_t)b * pn1 * dim * sizeof(float), nq = (size_t)b * pn2 * dim * sizeof(float),
                 ni = (size_t)b * pn2 * sizeof(int);
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t r) { if (e == hipSuccess) e = r; return r == hipSuccess; };
    if (ok(hipMalloc(&d_ref, nr ? nr : 4)) && ok(hipMalloc(&d_que, nq ? nq : 4)) && ok(hipMalloc(&d_idx, ni ? ni : 4)) &&
        ok(hipMemcpy(d_ref, ref_pts, nr, hipMemcpyHostToDevice)) && ok(hipMemcpy(d_que, que_pts, nq, hipMemcpyHostToDevice))) {
        int rc = launch(d_ref, d_que, d_idx, b, pn1, pn2, dim, exclude_self, nullptr);
        if (rc == 0) ok(hipMemcpy(idxs, d_idx, ni, hipMemcpyDeviceToHost));
        else fprintf(stderr, "findNearestPointIdxLauncher: bad arguments or launch failure (%d)\n", rc);
    }
    if (e != hipSuccess) fprintf(stderr, "findNearestPointIdxLauncher: %s\n", hipGetErrorString(e));
    (void)hipFree(d_ref); (void)hipFree(d_que); (void)hipFree(d_idx);
}
