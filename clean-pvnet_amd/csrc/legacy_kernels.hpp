// legacy_kernels.hpp -- legacy-layout kernels of the extension-module surface.
// Part of the single translation unit pvnet_vote.hip (included inside its anonymous namespace); see that file
// for the numerical contract and the reference citations (K = ransac_voting_kernel.cu, P = ransac_voting_gpu.py).
#pragma once

// ---------------------------------------------------------------------------------------------
// Legacy-layout kernels (module-level drop-in, reference layouts [tn,vn,2] / [hn,vn,*]).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_legacy_gen(const float *__restrict__ direct,
                                                       const float *__restrict__ coords,
                                                       const int32_t *__restrict__ idxs,
                                                       float *__restrict__ hypo, int tn, int vn, int hn)
{
    int hvi = blockIdx.x * kBlock + threadIdx.x;
    if (hvi >= hn * vn) return;
    int vi = hvi % vn;
    int t0 = idxs[hvi * 2], t1 = idxs[hvi * 2 + 1];
    const float2 *d = (const float2 *)direct;
    const float2 *c = (const float2 *)coords;
    float2 d0 = d[(size_t)t0 * vn + vi], d1 = d[(size_t)t1 * vn + vi];
    float2 c0 = c[t0], c1 = c[t1];
    float2 h = hypothesis_exact(d0.x, d0.y, c0.x, c0.y, d1.x, d1.y, c1.x, c1.y);
    ((float2 *)hypo)[hvi] = h;
}

// K:88-126.  Thread = (ti, vi); loops a slab of hypotheses so the pixel is loaded once and the
// byte stores of a wave are contiguous in ti.
__global__ __launch_bounds__(kBlock) void k_legacy_vote(const float *__restrict__ direct,
                                                        const float *__restrict__ coords,
                                                        const float *__restrict__ hypo,
                                                        uint8_t *__restrict__ inliers, int tn, int vn,
                                                        int hn, int h_per_block, float thresh)
{
    int ti = blockIdx.x * kBlock + threadIdx.x;
    int vi = blockIdx.y;
    int h0 = blockIdx.z * h_per_block;
    int h1 = min(hn, h0 + h_per_block);
    if (ti >= tn) return;
    float2 d = ((const float2 *)direct)[(size_t)ti * vn + vi];
    float2 c = ((const float2 *)coords)[ti];
    for (int hi = h0; hi < h1; ++hi) {
        float2 h = ((const float2 *)hypo)[hi * vn + vi];  // wave-uniform
        if (vote_exact(c.x, c.y, h.x, h.y, d.x, d.y, thresh))
            inliers[((size_t)hi * vn + vi) * tn + ti] = 1;
    }
}

// K:170-229
__global__ __launch_bounds__(kBlock) void k_legacy_gen_vp(const float *__restrict__ direct,
                                                          const float *__restrict__ coords,
                                                          const int32_t *__restrict__ idxs,
                                                          float *__restrict__ hypo, int tn, int vn, int hn)
{
    int hvi = blockIdx.x * kBlock + threadIdx.x;
    if (hvi >= hn * vn) return;
    int vi = hvi % vn;
    int id0 = idxs[hvi * 2], id1 = idxs[hvi * 2 + 1];
    const float2 *d = (const float2 *)direct;
    const float2 *c = (const float2 *)coords;
    float2 d0 = d[(size_t)id0 * vn + vi], d1 = d[(size_t)id1 * vn + vi];
    float2 c0 = c[id0], c1 = c[id1];
    float dx0 = d0.x, dy0 = d0.y, cx0 = c0.x, cy0 = c0.y;
    float dx1 = d1.x, dy1 = d1.y, cx1 = c1.x, cy1 = c1.y;

    float lx0 = dy0, ly0 = -dx0, lz0 = cy0 * dx0 - cx0 * dy0;
    float lx1 = dy1, ly1 = -dx1, lz1 = cy1 * dx1 - cx1 * dy1;

    float x = ly0 * lz1 - lz0 * ly1;
    float y = lz0 * lx1 - lx0 * lz1;
    float z = lx0 * ly1 - ly0 * lx1;

    float val_x0 = dx0 * (x - z * cx0);
    float val_x1 = dx1 * (x - z * cx1);
    float val_y0 = dy0 * (y - z * cy0);
    float val_y1 = dy1 * (y - z * cy1);

    if (val_x0 < 0 && val_x1 < 0 && val_y0 < 0 && val_y1 < 0) { z = -z; x = -x; y = -y; }
    if (val_x0 * val_x1 < 0 || val_y0 * val_y1 < 0) { x = 0.f; y = 0.f; z = 0.f; }

    hypo[hvi * 3] = x;
    hypo[hvi * 3 + 1] = y;
    hypo[hvi * 3 + 2] = z;
}

// K:268-310
__global__ __launch_bounds__(kBlock) void k_legacy_vote_vp(const float *__restrict__ direct,
                                                           const float *__restrict__ coords,
                                                           const float *__restrict__ hypo,
                                                           uint8_t *__restrict__ inliers, int tn, int vn,
                                                           int hn, int h_per_block, float thresh)
{
    int ti = blockIdx.x * kBlock + threadIdx.x;
    int vi = blockIdx.y;
    int h0 = blockIdx.z * h_per_block;
    int h1 = min(hn, h0 + h_per_block);
    if (ti >= tn) return;
    float2 d = ((const float2 *)direct)[(size_t)ti * vn + vi];
    float2 c = ((const float2 *)coords)[ti];
    float norm1 = sqrtf(d.x * d.x + d.y * d.y);
    for (int hi = h0; hi < h1; ++hi) {
        const float *h = hypo + ((size_t)hi * vn + vi) * 3;
        float hx = h[0], hy = h[1], hz = h[2];
        float diff_x = hx - c.x * hz;
        float diff_y = hy - c.y * hz;
        float norm2 = sqrtf(diff_x * diff_x + diff_y * diff_y);
        if (lt_1e6(norm1) || lt_1e6(norm2)) continue;
        float angle_dist = (d.x * diff_x + d.y * diff_y) / (norm1 * norm2);
        float val_x = diff_x * d.x;
        float val_y = diff_y * d.y;
        if (val_x < 0 || val_y < 0) continue;
        if (fabsf(angle_dist) > thresh) inliers[((size_t)hi * vn + vi) * tn + ti] = 1;
    }
}
