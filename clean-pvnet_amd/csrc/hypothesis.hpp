// hypothesis.hpp -- stage 2: hypothesis generation.
// Part of the single translation unit pvnet_vote.hip (included inside its anonymous namespace); see that file
// for the numerical contract and the reference citations (K = ransac_voting_kernel.cu, P = ransac_voting_gpu.py).
#pragma once

// ---------------------------------------------------------------------------------------------
// Stage 2: hypotheses (replaces random_ P:145/P:235 + generate_hypothesis K:11-86), and zeroes
// the inlier counters of the same (b,vi,hi).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_gen_hypothesis(
    const int32_t *__restrict__ idxs /*[B,hn_first,K,2] or null*/, const int32_t *__restrict__ idxs2 /*[B,hn-hn_first,K,2] or null*/,
    int hn_first /*hypotheses [0,hn_first) draw from idxs, the rest from idxs2 (a layer call: hn_first = hn)*/,
    const int *__restrict__ tn_arr,
    const float2 *__restrict__ coords, const float2 *__restrict__ dirs, float2 *__restrict__ hyps,
    int *__restrict__ counts, int B, int K, int hn, int cap, uint64_t seed, int b0)
{
    // B*K*hn < 2^31 (validate()): 32-bit index arithmetic, the 64-bit divides cost ~2 us of this 12 us kernel
    const unsigned gid = blockIdx.x * (unsigned)kBlock + threadIdx.x;
    if (gid >= (unsigned)B * (unsigned)K * (unsigned)hn) return;
    const unsigned bk = gid / (unsigned)hn;
    const int hi = (int)(gid - bk * (unsigned)hn);
    const int b = (int)(bk / (unsigned)K);
    const int vi = (int)(bk - (unsigned)b * (unsigned)K);
    counts[gid] = 0;
    const int tn = tn_arr[b];
    if (tn <= 0) {
        hyps[gid] = make_float2(0.f, 0.f);
        return;
    }
    int t0, t1;
    const int32_t *src = hi < hn_first ? idxs : idxs2;
    if (src) {
        const int32_t *ip = hi < hn_first ? idxs + (((size_t)b * hn_first + hi) * K + vi) * 2
                                          : idxs2 + (((size_t)b * (hn - hn_first) + (hi - hn_first)) * K + vi) * 2;
        t0 = ip[0];
        t1 = ip[1];
        // the reference reads out of bounds here; clamp instead of faulting
        t0 = t0 < 0 ? 0 : (t0 >= tn ? tn - 1 : t0);
        t1 = t1 < 0 ? 0 : (t1 >= tn ? tn - 1 : t1);
    } else {
        uint32_t c = (uint32_t)(hi * K + vi) * 2u;
        t0 = (int)(rng_u32(seed, 1u, (uint32_t)(b0 + b), c) % (uint32_t)tn);
        t1 = (int)(rng_u32(seed, 1u, (uint32_t)(b0 + b), c + 1u) % (uint32_t)tn);
    }
    const float2 *dp = dirs + ((size_t)b * K + vi) * cap;
    const float2 *cp = coords + (size_t)b * cap;
    float2 d0 = dp[t0], d1 = dp[t1], c0 = cp[t0], c1 = cp[t1];
    hyps[gid] = hypothesis_exact(d0.x, d0.y, c0.x, c0.y, d1.x, d1.y, c1.x, c1.y);
}
