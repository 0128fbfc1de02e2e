// covariance.hpp -- stage 4 (estimate): ratio threshold, weighted covariance, PnP weights.
// Part of the single translation unit pvnet_vote.hip (included inside its anonymous namespace); see that file
// for the numerical contract and the reference citations (K = ransac_voting_kernel.cu, P = ransac_voting_gpu.py).
#pragma once

// ---------------------------------------------------------------------------------------------
// Stage 4 (estimate): ratio threshold + weighted covariance about `mean` (P:244, P:262-269).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_covariance(
    const int *__restrict__ tn_arr, const float2 *__restrict__ hyps, const int *__restrict__ counts,
    const float2 *__restrict__ mean, float *__restrict__ cov /*[B,K,2,2]*/,
    float2 *__restrict__ hyp_out /*[B,K,hn] or null*/, int *__restrict__ counts_out,
    float *__restrict__ weights /*[B,K,3] or null*/, int K, int hn, int hstride /*row length of hyps / counts*/,
    int hoff /*first hypothesis of the row that belongs to this estimate*/)
{
    __shared__ int redi[4];
    __shared__ double redd[4];
    const int vi = blockIdx.x, b = blockIdx.y;
    const int bk = b * K + vi;
    const int tn = tn_arr[b];
    const float2 m = mean[bk];
    const float2 *hp = hyps + (size_t)bk * hstride + hoff;
    const int *cp = counts + (size_t)bk * hstride + hoff;
    if (hyp_out || counts_out)
        for (int h = threadIdx.x; h < hn; h += kBlock) {
            if (hyp_out) hyp_out[(size_t)bk * hn + h] = tn > 0 ? hp[h] : make_float2(0.f, 0.f);
            if (counts_out) counts_out[(size_t)bk * hn + h] = tn > 0 ? cp[h] : 0;
        }
    double sxx = 0, sxy = 0, syy = 0, sw = 0;
    if (tn <= 0) {
        // P:211-216: hypotheses are zeros, ratios are ones
        if (threadIdx.x == 0) {
            double dx = (double)(0.f - m.x), dy = (double)(0.f - m.y);
            sxx = dx * dx * hn; sxy = dx * dy * hn; syy = dy * dy * hn; sw = (double)hn;
        }
    } else {
        int mx = 0;
        for (int h = threadIdx.x; h < hn; h += kBlock) mx = max(mx, cp[h]);
        mx = wave_max(mx);
        __syncthreads();
        if (lane_id() == 0) redi[threadIdx.x >> 6] = mx;
        __syncthreads();
        mx = max(max(redi[0], redi[1]), max(redi[2], redi[3]));
        const float ftn = (float)tn;
        const float thr = (float)mx / ftn - 0.1f;            // P:244, P:262 (binary32)
        for (int h = threadIdx.x; h < hn; h += kBlock) {
            float r = (float)cp[h] / ftn;
            if (r < thr) r = 0.f;                             // P:263
            float2 q = hp[h];
            double dx = (double)(q.x - m.x), dy = (double)(q.y - m.y);  // P:266 binary32 diff
            sxx += (double)r * dx * dx; sxy += (double)r * dx * dy; syy += (double)r * dy * dy;
            sw += (double)r;
        }
    }
    sxx = block_sum(sxx, redd); sxy = block_sum(sxy, redd);
    syy = block_sum(syy, redd); sw = block_sum(sw, redd);
    if (threadIdx.x == 0) {
        double den = sw + 1e-3;                               // P:269
        float *c = cov + (size_t)bk * 4;
        c[0] = (float)(sxx / den); c[1] = (float)(sxy / den);
        c[2] = (float)(sxy / den); c[3] = (float)(syy / den);
        if (weights) {
            // evaluators/linemod/pvnet.py:118-128: inv(sqrtm(var)) per keypoint, zeros when var[0,0] < 1e-6 or NaN.
            // Closed form for a 2x2 SPD matrix A: sqrtm(A) = (A + s I)/t, s = sqrt(det A), t = sqrt(tr A + 2 s).
            const double a = (double)c[0], b = (double)c[1], d = (double)c[3];
            double wxx = 0.0, wxy = 0.0, wyy = 0.0;
            const double det = a * d - b * b;
            if (!(c[0] < 1e-6f) && a == a && b == b && d == d && det > 0.0 && a > 0.0) {
                const double s = sqrt(det), t = sqrt(a + d + 2.0 * s);
                const double q = t / ((a + s) * (d + s) - b * b);
                wxx = q * (d + s); wxy = -q * b; wyy = q * (a + s);
            }
            float *w = weights + (size_t)bk * 3;
            w[0] = (float)wxx; w[1] = (float)wxy; w[2] = (float)wyy;
        }
    }
}
