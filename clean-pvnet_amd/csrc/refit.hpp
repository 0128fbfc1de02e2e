// refit.hpp -- stage 4 (v3): winner selection, least-squares refit, singular policy.
// Part of the single translation unit pvnet_vote.hip (included inside its anonymous namespace); see that file
// for the numerical contract and the reference citations (K = ransac_voting_kernel.cu, P = ransac_voting_gpu.py).
#pragma once

// ---------------------------------------------------------------------------------------------
// Stage 4 (v3): winner selection + least-squares refit (P:159-167 and P:176-196).
// One block per (keypoint, image).
// ---------------------------------------------------------------------------------------------
// nsplit (1..kRefitSplitMax) blocks per (keypoint, image), each over its share of the pixels: the exact re-vote is a
// long dependent chain (two sqrt, one divide) and B*K blocks alone leave the SIMDs latency-bound, while more than
// about one wave of blocks only adds dispatch time; the host picks nsplit from B*K (refit_split).  Partial sums go to
// sums[b,vi,split,5] and are merged in a fixed order by k_finalize_v3 (deterministic for a given launch shape).
// (Finalising in the last block of an image to arrive was tried: the device-scope release/acquire it needs writes
// back and invalidates the XCD's L2 on gfx950 and cost 30 % of the whole call.)
constexpr int kRefitSplitMax = 8;

// BAND: the re-vote through the guard-band prefilter (large grids: VALU-bound) or the exact sequence alone (small ones:
// a latency chain, where the prefilter's extra instructions only add to it: +1.2 % per call at B = 1)
template <bool BAND>
__global__ __launch_bounds__(kBlock) void k_select_refit(
    const int *__restrict__ tn_arr, const float2 *__restrict__ coords,
    const float2 *__restrict__ dirs, const float2 *__restrict__ hyps,
    const int *__restrict__ counts, double *__restrict__ sums /*[B,K,nsplit,5]*/,
    int *__restrict__ win_counts /*[B,K] or null*/, int K, int hn, int hstride /*row length of hyps / counts (>= hn)*/,
    int cap, float thresh, int nsplit, float *__restrict__ win_ratio /*[B,K]: winner count / tn (k_finalize_v3's stage hint)*/,
    Bf16Consts fc /*kappa, beta2, eps0: the count kernel's second-level test, here the prefilter of the re-vote*/)
{
    __shared__ int s_cnt[4], s_idx[4];
    __shared__ double red5[20];
    const int vi = blockIdx.x / nsplit, split = blockIdx.x - vi * nsplit, b = blockIdx.y;
    const int bk = b * K + vi;
    const int tn = tn_arr[b];
    double *part = sums + ((size_t)bk * nsplit + split) * 5;
    if (tn <= 0) {
        if (threadIdx.x == 0) {
            for (int i = 0; i < 5; ++i) part[i] = 0.0;
            if (win_counts && split == 0) win_counts[bk] = 0;
            if (split == 0) win_ratio[bk] = -1.f;
        }
        return;
    }
    // torch.max(counts, 0): maximal count, FIRST index among ties (P:160)
    const int *cp = counts + (size_t)bk * hstride;
    int best = -1, besti = 0x7fffffff;
    for (int h = threadIdx.x; h < hn; h += kBlock) {
        int c = cp[h];
        if (c > best) { best = c; besti = h; }
    }
    wave_argmax_first(best, besti);
    if (lane_id() == 0) { s_cnt[threadIdx.x >> 6] = best; s_idx[threadIdx.x >> 6] = besti; }
    __syncthreads();
    best = s_cnt[0]; besti = s_idx[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
        if (s_cnt[w] > best || (s_cnt[w] == best && s_idx[w] < besti)) { best = s_cnt[w]; besti = s_idx[w]; }

    // P:162-167: all_win_ratio (0) < count/tn  <=>  count > 0; otherwise the winner stays (0,0)
    float2 win = make_float2(0.f, 0.f);
    if (best > 0) win = hyps[(size_t)bk * hstride + besti];

    const bool win_near = fabsf(win.x) < 1e15f && fabsf(win.y) < 1e15f;   // (beyond: the exact vote's squares overflow, the test's do not)
    // P:176-191: re-vote the winner (hn = 1) and accumulate the normal equations in binary64
    const float2 *dp = dirs + (size_t)bk * cap;
    const float2 *cq = coords + (size_t)b * cap;
    double xx = 0, xy = 0, yy = 0, bx = 0, by = 0;
    // this block's share of the pixels; four pixels per trip so that the loads of a trip overlap
    const int per = (tn + nsplit - 1) / nsplit;
    const int tbeg = split * per, tend = min(tn, tbeg + per);
    for (int t0 = tbeg + threadIdx.x; t0 < tend; t0 += 4 * kBlock) {
        float2 d[4], c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ti = t0 + u * kBlock;
            d[u] = ti < tend ? dp[ti] : make_float2(0.f, 0.f);  // zero direction: norm1 < 1e-6, never an inlier
            c[u] = ti < tend ? cq[ti] : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            // The re-vote decides like the count kernel's second level (count_bf16.hpp): t = a - kappa |d x n|
            // with the unit direction from v_rsq_f32, taken when it lies outside the guard band (beta2, eps0) -- ~15 VALU
            // instructions -- and the exact sequence of K:100-125 (two square roots, a division: ~50) only inside the band
            // (4e-5 of the pixels), for a dead direction and for a winner beyond 1e15 px.  Same inlier set, bit for bit.
            bool inlier;
            if constexpr (BAND) {
                const float dd = d[u].x * d[u].x + d[u].y * d[u].y;
                const float rinv = __builtin_amdgcn_rsqf(dd);
                const float ux = d[u].x * rinv, uy = d[u].y * rinv;
                const float dx = win.x - c[u].x, dy = win.y - c[u].y;
                const float a2 = __builtin_fmaf(dx, ux, dy * uy);
                const float b2 = __builtin_fmaf(dx, -fc.kappa * uy, dy * (fc.kappa * ux));
                const float t2 = a2 - fabsf(b2);
                inlier = t2 > 0.f;
                const bool unsure = !win_near || !(dd >= kDdAlive && dd < INFINITY) ||
                                    !(__builtin_fmaf(-fc.beta2, a2, fabsf(t2)) > fc.eps0);
                if (__any(unsure)) {
                    if (unsure) inlier = vote_exact(c[u].x, c[u].y, win.x, win.y, d[u].x, d[u].y, thresh);
                }
            } else {
                inlier = vote_exact(c[u].x, c[u].y, win.x, win.y, d[u].x, d[u].y, thresh);
            }
            if (!inlier) continue;
            double nx = (double)d[u].y, ny = -(double)d[u].x;          // P:178-179
            double bb = nx * (double)c[u].x + ny * (double)c[u].y;     // P:189
            xx += nx * nx; xy += nx * ny; yy += ny * ny;               // P:190
            bx += nx * bb; by += ny * bb;                              // P:191
        }
    }
    // one reduction for all five sums: five independent shuffle chains interleave, a single barrier
    double v[5] = {xx, xy, yy, bx, by};
#pragma unroll
    for (int i = 0; i < 5; ++i) v[i] = wave_total(v[i]);
    if (lane_id() == 0) {
#pragma unroll
        for (int i = 0; i < 5; ++i) red5[(threadIdx.x >> 6) * 5 + i] = v[i];
    }
    __syncthreads();
    if (threadIdx.x < 5) part[threadIdx.x] = red5[threadIdx.x] + red5[5 + threadIdx.x] + red5[10 + threadIdx.x] + red5[15 + threadIdx.x];
    if (threadIdx.x == 0 && split == 0) {
        if (win_counts) win_counts[bk] = best;
        win_ratio[bk] = (float)best / (float)tn;
    }
}

// Merge the partial normal equations, solve the 2x2 systems (P:193, closed form in binary64) and apply the
// singular-matrix policy across the keypoints of an image (b_inv, P:97-109).  One block per image.
// Also reports the image's mean winner ratio (winner count / tn over its keypoints; -1: image skipped) to hint[b] and its
// tn to hint[hint_stride + b] -- host-visible memory the NEXT calls read to decide whether staged counting pays (stage_hint_allows, pvnet_vote.hip);
// hint may be null.
__global__ __launch_bounds__(64) void k_finalize_v3(const int *__restrict__ tn_arr, const double *__restrict__ sums,
                                                    float2 *__restrict__ out, int K, int policy, int nsplit,
                                                    const float *__restrict__ win_ratio, float *hint, int hint_stride)
{
    __shared__ int any_singular;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) any_singular = 0;
    __syncthreads();
    // skipped image (tn <= 0): k_select_refit left win_ratio = -1 and zero partial sums -- read with the sums, in ONE round trip
    // (round 5: tn_arr[b] first was a dependent load in a kernel that is nothing but its latency chain)
    const bool skipped = win_ratio[(size_t)b * K] < -0.5f;
    if (hint) {
        float r = 0.f;
        for (int vi = threadIdx.x; vi < K; vi += 64) r += win_ratio[(size_t)b * K + vi];
        r = wave_total(r);                                         // (a hint: the order of the additions is irrelevant)
        if (threadIdx.x == 0) {
            hint[b] = skipped ? -1.f : r / (float)K;
            hint[hint_stride + b] = (float)tn_arr[b];              // (the host also learns whether any image is large enough to stage)
        }
    }
    for (int v0 = 0; v0 < K; v0 += 64) {          // K <= 64 in every real use: one trip
        const int vi = v0 + threadIdx.x;
        float2 o = make_float2(0.f, 0.f);
        double bx = 0, by = 0;
        bool sing = false;
        if (vi < K && !skipped) {
            const double *q = sums + ((size_t)b * K + vi) * nsplit * 5;
            double xx = 0, xy = 0, yy = 0;
            for (int sp = 0; sp < nsplit; ++sp) {
                xx += q[sp * 5]; xy += q[sp * 5 + 1]; yy += q[sp * 5 + 2]; bx += q[sp * 5 + 3]; by += q[sp * 5 + 4];
            }
            const double det = xx * yy - xy * xy;
            sing = !(det != 0.0) || !isfinite(det);
            if (!sing) {
                o.x = (float)((yy * bx - xy * by) / det);
                o.y = (float)((xx * by - xy * bx) / det);
            }
        }
        if (K > 64) {   // generic path: the policy needs every keypoint's flag first
            if (sing) atomicOr(&any_singular, 1);
            continue;
        }
        if (sing) any_singular = 1;
        __syncthreads();
        if (vi < K) {
            if (!skipped && any_singular && policy != PVV_SINGULAR_ZERO) {
                if (policy == PVV_SINGULAR_REFERENCE) o = make_float2((float)bx, (float)by);   // inverse := identity => x = ATb
                else o = make_float2(0.f, 0.f);                                                // v1: the whole image becomes zeros
            }
            out[(size_t)b * K + vi] = o;
        }
        return;
    }
    // K > 64: second pass now that any_singular is complete
    __syncthreads();
    for (int vi = threadIdx.x; vi < K; vi += 64) {
        float2 o = make_float2(0.f, 0.f);
        if (!skipped) {
            const double *q = sums + ((size_t)b * K + vi) * nsplit * 5;
            double xx = 0, xy = 0, yy = 0, bx = 0, by = 0;
            for (int sp = 0; sp < nsplit; ++sp) {
                xx += q[sp * 5]; xy += q[sp * 5 + 1]; yy += q[sp * 5 + 2]; bx += q[sp * 5 + 3]; by += q[sp * 5 + 4];
            }
            const double det = xx * yy - xy * xy;
            const bool sing = !(det != 0.0) || !isfinite(det);
            if (!sing) {
                o.x = (float)((yy * bx - xy * by) / det);
                o.y = (float)((xx * by - xy * bx) / det);
            }
            if (any_singular && policy != PVV_SINGULAR_ZERO)
                o = policy == PVV_SINGULAR_REFERENCE ? make_float2((float)bx, (float)by) : make_float2(0.f, 0.f);
        }
        out[(size_t)b * K + vi] = o;
    }
}
