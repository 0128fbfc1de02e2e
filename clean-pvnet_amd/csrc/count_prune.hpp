// count_prune.hpp -- stage 3, between the two launches of the staged count: a lower bound L* of the winner's count
// (ransac_voting_layer_v3 only).
// Part of the single translation unit pvnet_vote.hip (included inside its anonymous namespace); see that file
// for the numerical contract and the reference citations (K = ransac_voting_kernel.cu, P = ransac_voting_gpu.py).
#pragma once

// ---------------------------------------------------------------------------------------------
// ransac_voting_layer_v3 keeps, per (image, keypoint), the FIRST hypothesis with the maximal inlier count and that count
// (torch.max over hn, P:160; ratio update P:162-167); the other hn - 1 counts never leave the layer.  After
// k_count_bf16<kCountFirst> has counted every hypothesis over a spread quarter of an image's 512-pixel chunks, this kernel
// takes kLead = 2 leaders per (image, keypoint) -- the best two of the four per-wavefront leaders (maximal PARTIAL count
// among the wave's share of the hypotheses, first index among ties): the overall partial leader is one of them -- and
// counts their SURE inliers over every pixel the first launch did not count (see the loop below: a lower bound of the exact
// count, without its square roots and divisions).  lead[b,k,0..3] = the leaders' partial counts (-1: none),
// lead[b,k,4..7] = their sure inliers in the rest: lower bounds of two FULL counts, whose maximum L* bounds the winner's
// count from below.  The second launch (k_count_bf16<kCountFilter>) then only counts hypotheses with  partial + R >= L*
// (count_bf16.hpp).
//
// Grid (K * nsplit, B): the remaining pixels of an (image, keypoint) are cut into nsplit shares so that the whole batch is
// ONE generation of blocks (<= 8 per CU); a thread sees ~10 pixels, both leaders per pixel, and all its loads are in
// flight together; the shares add their sums to lead[..4..7] with one atomic each (zeroed by k_compact_hyp).  One block per
// (image, keypoint) that also compacted the survivors -- the first form of round 3 -- took 36 us at B = 64 (18 dependent
// memory round trips per block), 4608 short blocks of 2-3 pixels per thread 28 us (two and a half generations of blocks,
// five round trips each); one generation with the EXACT vote for four / two leaders 24.7 / 15.0 us (VALU-bound: ~50
// instructions per pixel and leader), with the sure-inlier test below 11 us.
// ---------------------------------------------------------------------------------------------
struct LeadArgs {
    const int *tn_arr;
    const float2 *coords;    // [B,cap]
    const float2 *dirs;      // [B,K,cap]
    const float2 *hyps;      // [B,K,hn]
    const int *counts;       // [B,K,hn] partial counts (of the chunks in done_mask)
    int *lead;               // [B,K,8]
    const int *any_staged;   // one word written by k_count_bf16<kCountFirst>: 0 = no image of the batch is staged
    int K, hn, cap;
    int hstride;             // row length of hyps / counts (>= hn; StageArgs.hstride)
    float kappa, beta, eps;  // the sure-inlier test below: kappa = T/sqrt(1-T^2), band = 2 x the count kernel's second level
    int nsplit;
};

constexpr int kLead = 2;     // leaders per (image, keypoint), <= 4

template <uint32_t FIRST>
__global__ __launch_bounds__(kBlock) void k_lead(LeadArgs a)
{
    constexpr uint32_t REST = stage_rest_of(FIRST);
    __shared__ int s_cnt[4], s_idx[4], s_sum[4][4];
    const int vi = blockIdx.x / a.nsplit, split = blockIdx.x - vi * a.nsplit, b = blockIdx.y, bk = b * a.K + vi;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    constexpr int PC = 4 * kBfPixPerWave;                          // the count kernel's chunk: 512 pixels
    if (*a.any_staged == 0) return;
    const int tn = a.tn_arr[b];
    const int nch = (tn + PC - 1) / PC;
    if (tn <= 0 || nch < kStageMinChunks) return;                          // skipped (P:129-132) or counted completely by the first launch
    // ---- the pixels of the chunks not counted yet, in order: r-th remaining pixel -> chunk stage_chunk_at(rem, r / 512).
    //      The first trip's loads (kTrip pixels per thread) are issued BEFORE the leader search: they depend on tn alone,
    //      so the counts, the pixels and then the leaders' hypotheses are three memory round trips, not five.
    constexpr int kTrip = 10;
    const int rem_px = stage_pixels<REST>(tn, nch, PC);
    const int per = (rem_px + a.nsplit - 1) / a.nsplit;
    const int r1 = min(rem_px, (split + 1) * per);
    const float2 *crd = a.coords + (size_t)b * a.cap;
    const float2 *dir = a.dirs + (size_t)bk * a.cap;
    float2 cc[kTrip], dd_[kTrip];
    auto load_trip = [&](int r0) {
#pragma unroll
        for (int u = 0; u < kTrip; ++u) {
            const int r = r0 + u * kBlock;
            cc[u] = dd_[u] = make_float2(0.f, 0.f);                // zero direction: never an inlier
            if (r < r1) {
                const int p = stage_chunk_at<REST>(r / PC) * PC + (r & (PC - 1));
                cc[u] = crd[p];
                dd_[u] = dir[p];
            }
        }
    };
    int r0 = split * per + threadIdx.x;
    load_trip(r0);
    // ---- this wave's leader: maximal partial count, first index among ties
    const int *cp = a.counts + (size_t)bk * a.hstride;
    int best = -1, besth = 0x7fffffff;
    for (int h = threadIdx.x; h < a.hn; h += kBlock) {
        const int c = cp[h];
        if (c > best) { best = c; besth = h; }
    }
    wave_argmax_first(best, besth);
    if (lane == 0) { s_cnt[wave] = best; s_idx[wave] = besth; }
    __syncthreads();
    // the best kLead of the four per-wave leaders (count descending, index ascending); the bound L* comes from the overall
    // partial leader in nearly every case, the second one is insurance on noisy fields
    int lc[4], li[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) { lc[w] = s_cnt[w]; li[w] = s_idx[w]; }
#pragma unroll
    for (int i = 0; i < kLead; ++i)
#pragma unroll
        for (int j = 3; j > i; --j)
            if (lc[j] > lc[j - 1] || (lc[j] == lc[j - 1] && li[j] < li[j - 1])) {
                const int tc = lc[j], ti = li[j];
                lc[j] = lc[j - 1]; li[j] = li[j - 1]; lc[j - 1] = tc; li[j - 1] = ti;
            }
    float2 ld[kLead];
#pragma unroll
    for (int w = 0; w < kLead; ++w) ld[w] = lc[w] >= 0 ? a.hyps[(size_t)bk * a.hstride + li[w]] : make_float2(0.f, 0.f);
    if (split == 0 && threadIdx.x < 4) {
        int v = -1;
#pragma unroll
        for (int w = 0; w < kLead; ++w) if ((int)threadIdx.x == w) v = lc[w];
        a.lead[(size_t)bk * 8 + threadIdx.x] = v;                  // slots kLead..3: no leader
    }
    // ---- SURE inliers only.  L* has to be a LOWER bound of the winner's count, nothing more: a leader's partial count plus
    //      the pixels of the rest that are inliers BEYOND DOUBT is one (<= its exact full count <= the maximum), and it needs
    //      neither the correctly rounded square roots and divisions of the exact vote (~50 VALU instructions per pixel and
    //      leader: two leaders made this kernel VALU-bound at 15 us) nor a fallback.  The test is the count kernel's second
    //      level (count_bf16.hpp: t = a - kappa |d x nh| against a guard band) with the unit normal from
    //      v_rsq_f32 (components within 4u, what that band assumes of an f32 normal: count_bf16.hpp) and TWICE its band: |t| - 2 beta2 a > 2 eps0
    //      and t > 0.  Pixels the exact vote rejects outright (K:121: norm1 < 1e-6, non-finite directions) have dd <= 4e-12
    //      or dd = inf/NaN and are not counted; what the band excludes (a few 1e-5 of the pixels) only lowers the bound.
    int inl[kLead];
    bool lnear[kLead];       // a leader beyond 1e15 px (or non-finite) gets no sure inliers: the exact vote's squares overflow
#pragma unroll               // out there (norm2 = inf: never an inlier) and this test's do not -- its partial count alone is a bound
    for (int w = 0; w < kLead; ++w) { inl[w] = 0; lnear[w] = lc[w] >= 0 && fabsf(ld[w].x) < 1e15f && fabsf(ld[w].y) < 1e15f; }   // (no leader in the slot: nothing to count)
    for (;;) {
#pragma unroll
        for (int u = 0; u < kTrip; ++u) {
            const float dd = dd_[u].x * dd_[u].x + dd_[u].y * dd_[u].y;
            const bool ok = dd > 4e-12f && dd < INFINITY;
            const float r = __builtin_amdgcn_rsqf(dd);
            const float nx = dd_[u].x * r, ny = dd_[u].y * r;
            const float bx = -a.kappa * ny, by = a.kappa * nx;
#pragma unroll
            for (int w = 0; w < kLead; ++w) {
                const float dx = ld[w].x - cc[u].x, dy = ld[w].y - cc[u].y;
                const float av = __builtin_fmaf(dx, nx, dy * ny);
                const float bv = __builtin_fmaf(dx, bx, dy * by);
                const float t = av - fabsf(bv);
                inl[w] += (ok && lnear[w] && t > 0.f && __builtin_fmaf(-a.beta, av, t) > a.eps) ? 1 : 0;
            }
        }
        r0 += kTrip * kBlock;
        if (r0 - (int)threadIdx.x >= r1) break;                    // block-uniform
        load_trip(r0);
    }
#pragma unroll
    for (int w = 0; w < kLead; ++w) {
        const int s = wave_total(inl[w]);
        if (lane == 0) s_sum[wave][w] = s;
    }
    __syncthreads();
    if ((int)threadIdx.x < kLead) {
        const int s = s_sum[0][threadIdx.x] + s_sum[1][threadIdx.x] + s_sum[2][threadIdx.x] + s_sum[3][threadIdx.x];
        if (s) atomicAdd(&a.lead[(size_t)bk * 8 + 4 + threadIdx.x], s);
    }
}
