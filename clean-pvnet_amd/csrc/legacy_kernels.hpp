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

// K:170-229 -- the intersection of two pixel rays in homogeneous coordinates.  A ray through pixel p with direction v is the
// line  L = (v.y, -v.x, p.y v.x - p.x v.y);  two lines meet at  M = L_a x L_b  (a point at infinity when the rays are parallel:
// m.w = 0).  M is oriented so that it lies AHEAD of both pixels along their directions (the four sign probes below), and
// dropped -- (0,0,0) -- when the two pixels disagree about the side.  Every product and difference is the reference's, in the
// reference's order (bit-exactness against oracle/_ref: tests/test_ref_pin.py); only the formulation and the names are ours.
struct Line3 { float a, b, c; };
__device__ __forceinline__ Line3 ray_line(float2 p, float2 v) { return Line3{v.y, -v.x, p.y * v.x - p.x * v.y}; }

__global__ __launch_bounds__(kBlock) void k_legacy_gen_vp(const float *__restrict__ direct,
                                                          const float *__restrict__ coords,
                                                          const int32_t *__restrict__ idxs,
                                                          float *__restrict__ hypo, int tn, int vn, int hn)
{
    const int pair = blockIdx.x * kBlock + threadIdx.x;          // (hypothesis, keypoint) flattened
    if (pair >= hn * vn) return;
    const int vi = pair % vn;
    const int ia = idxs[pair * 2], ib = idxs[pair * 2 + 1];
    const float2 va = ((const float2 *)direct)[(size_t)ia * vn + vi], vb = ((const float2 *)direct)[(size_t)ib * vn + vi];
    const float2 pa = ((const float2 *)coords)[ia], pb = ((const float2 *)coords)[ib];
    const Line3 la = ray_line(pa, va), lb = ray_line(pb, vb);
    float3 m = make_float3(la.b * lb.c - la.c * lb.b, la.c * lb.a - la.a * lb.c, la.a * lb.b - la.b * lb.a);   // la x lb
    // is M ahead of pixel a / b, per axis?  v . (m.xy - m.w p), component-wise
    const float ahead_ax = va.x * (m.x - m.z * pa.x), ahead_bx = vb.x * (m.x - m.z * pb.x);
    const float ahead_ay = va.y * (m.y - m.z * pa.y), ahead_by = vb.y * (m.y - m.z * pb.y);
    if (ahead_ax < 0 && ahead_bx < 0 && ahead_ay < 0 && ahead_by < 0) m = make_float3(-m.x, -m.y, -m.z);   // behind both: flip (z first in K: same values)
    if (ahead_ax * ahead_bx < 0 || ahead_ay * ahead_by < 0) m = make_float3(0.f, 0.f, 0.f);             // the pixels disagree
    float *out = hypo + (size_t)pair * 3;
    out[0] = m.x; out[1] = m.y; out[2] = m.z;
}

// K:268-310 -- a pixel votes for the homogeneous point M when the ray from the pixel towards M (M.xy - M.w p) runs along its
// direction field within the threshold, on the positive side in both axes.
__global__ __launch_bounds__(kBlock) void k_legacy_vote_vp(const float *__restrict__ direct,
                                                           const float *__restrict__ coords,
                                                           const float *__restrict__ hypo,
                                                           uint8_t *__restrict__ inliers, int tn, int vn,
                                                           int hn, int h_per_block, float thresh)
{
    const int ti = blockIdx.x * kBlock + threadIdx.x;
    const int vi = blockIdx.y;
    const int h0 = blockIdx.z * h_per_block, h1 = min(hn, h0 + h_per_block);   // this block's slab of hypotheses
    if (ti >= tn) return;
    const float2 v = ((const float2 *)direct)[(size_t)ti * vn + vi];
    const float2 p = ((const float2 *)coords)[ti];
    const float len_v = sqrtf(v.x * v.x + v.y * v.y);
    for (int hi = h0; hi < h1; ++hi) {
        const float *mp = hypo + ((size_t)hi * vn + vi) * 3;
        const float2 to_m = make_float2(mp[0] - p.x * mp[2], mp[1] - p.y * mp[2]);
        const float len_m = sqrtf(to_m.x * to_m.x + to_m.y * to_m.y);
        if (lt_1e6(len_v) || lt_1e6(len_m)) continue;
        const float cosine = (v.x * to_m.x + v.y * to_m.y) / (len_v * len_m);
        if (to_m.x * v.x < 0 || to_m.y * v.y < 0) continue;
        if (fabsf(cosine) > thresh) inliers[((size_t)hi * vn + vi) * tn + ti] = 1;
    }
}
