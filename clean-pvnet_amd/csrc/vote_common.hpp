// vote_common.hpp -- exact binary32 building blocks, wave64/block helpers, counter-based RNG.
// Part of the single translation unit pvnet_vote.hip (included inside its anonymous namespace); see that file
// for the numerical contract and the reference citations (K = ransac_voting_kernel.cu, P = ransac_voting_gpu.py).
#pragma once

// ---------------------------------------------------------------------------------------------
// Exact binary32 building blocks (shared by every kernel that takes an inlier decision).
// ---------------------------------------------------------------------------------------------

// (double)f < 1e-6  <=>  f <= fl32(1e-6): fl32(1e-6) = 9.99999997e-07 is the largest binary32
// below the double literal of K:42-43,121.  False for NaN, like the original compare.
__device__ __forceinline__ bool lt_1e6(float f) { return f <= 1e-6f; }

// K:100-125, one (hi,vi,ti) thread.
__device__ __forceinline__ bool vote_exact(float cx, float cy, float hx, float hy, float nx,
                                           float ny, float thresh)
{
    float dx = hx - cx;
    float dy = hy - cy;
    float norm1 = sqrtf(nx * nx + ny * ny);
    float norm2 = sqrtf(dx * dx + dy * dy);
    if (lt_1e6(norm1) || lt_1e6(norm2)) return false;
    float angle_dist = (dx * nx + dy * ny) / (norm1 * norm2);
    return angle_dist > thresh;
}

// K:22-48, one (hi,vi) thread; (0,0) when degenerate (K:42-43 + at::zeros K:75).
__device__ __forceinline__ float2 hypothesis_exact(float dx0, float dy0, float cx0, float cy0,
                                                   float dx1, float dy1, float cx1, float cy1)
{
    float nx0 = dy0, ny0 = -dx0;
    float nx1 = dy1, ny1 = -dx1;
    float den_y = nx1 * ny0 - nx0 * ny1;
    float den_x = ny1 * nx0 - ny0 * nx1;
    if (lt_1e6(fabsf(den_y))) return make_float2(0.f, 0.f);
    if (lt_1e6(fabsf(den_x))) return make_float2(0.f, 0.f);
    float y = (nx1 * (nx0 * cx0 + ny0 * cy0) - nx0 * (nx1 * cx1 + ny1 * cy1)) / den_y;
    float x = (ny1 * (nx0 * cx0 + ny0 * cy0) - ny0 * (nx1 * cx1 + ny1 * cy1)) / den_x;
    return make_float2(x, y);
}

// ---------------------------------------------------------------------------------------------
// wave64 / block helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }

__device__ __forceinline__ float bcast(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Wave-wide scans and reductions on DPP row shifts and broadcasts (row_shr:1/2/4/8 inside each row of 16 lanes, then
// row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3: lane 63 ends with the combination of all 64 lanes).
// No LDS crossbar: the six ds_bpermute + wait pairs of a __shfl scan are ~350 cycles of latency, these VALU instructions
// ~40.  Used where the result is exact whatever the order (integer sums, maxima, arg-max with a first-index tie rule);
// floating-point SUMS keep their butterfly (the order of the additions is part of the result).
#define PVV_DPP(old, v, ctrl, rmask, bc) __builtin_amdgcn_update_dpp((old), (v), (ctrl), (rmask), 0xf, (bc))

// inclusive prefix sum; lane 63 holds the wave's total
__device__ __forceinline__ int wave_incl_scan(int v)
{
    v += PVV_DPP(0, v, 0x111, 0xf, true);
    v += PVV_DPP(0, v, 0x112, 0xf, true);
    v += PVV_DPP(0, v, 0x114, 0xf, true);
    v += PVV_DPP(0, v, 0x118, 0xf, true);
    v += PVV_DPP(0, v, 0x142, 0xa, false);
    v += PVV_DPP(0, v, 0x143, 0xc, false);
    return v;
}
__device__ __forceinline__ int wave_total(int v) { return __builtin_amdgcn_readlane(wave_incl_scan(v), 63); }

__device__ __forceinline__ long long wave_total(long long v)
{
    auto step = [&](auto shift) {
        const int lo = (int)(unsigned long long)v, hi = (int)((unsigned long long)v >> 32);
        return (long long)(((unsigned long long)(unsigned)shift(hi) << 32) | (unsigned)shift(lo));
    };
    v += step([](int x) { return PVV_DPP(0, x, 0x111, 0xf, true); });
    v += step([](int x) { return PVV_DPP(0, x, 0x112, 0xf, true); });
    v += step([](int x) { return PVV_DPP(0, x, 0x114, 0xf, true); });
    v += step([](int x) { return PVV_DPP(0, x, 0x118, 0xf, true); });
    v += step([](int x) { return PVV_DPP(0, x, 0x142, 0xa, false); });
    v += step([](int x) { return PVV_DPP(0, x, 0x143, 0xc, false); });
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned long long)v, 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)((unsigned long long)v >> 32), 63);
    return (long long)(((unsigned long long)hi << 32) | lo);
}

// maximum over the wave (lanes without a partner in a step combine with themselves)
__device__ __forceinline__ float wave_max(float v)
{
#define PVV_STEP(ctrl, rmask) v = fmaxf(v, __int_as_float(PVV_DPP(__float_as_int(v), __float_as_int(v), ctrl, rmask, false)))
    PVV_STEP(0x111, 0xf); PVV_STEP(0x112, 0xf); PVV_STEP(0x114, 0xf); PVV_STEP(0x118, 0xf); PVV_STEP(0x142, 0xa); PVV_STEP(0x143, 0xc);
#undef PVV_STEP
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// sum of a double over the wave in the order of the row shifts (lane 63's chain); deterministic, but NOT the butterfly's
// order: the last bits differ from a __shfl_xor tree
__device__ __forceinline__ double wave_total(double v)
{
    auto step = [&](auto shift) {
        const unsigned long long u = (unsigned long long)__double_as_longlong(v);
        const unsigned long long r = ((unsigned long long)(unsigned)shift((int)(u >> 32)) << 32) | (unsigned)shift((int)u);
        return __longlong_as_double((long long)r);
    };
    v += step([](int x) { return PVV_DPP(0, x, 0x111, 0xf, true); });
    v += step([](int x) { return PVV_DPP(0, x, 0x112, 0xf, true); });
    v += step([](int x) { return PVV_DPP(0, x, 0x114, 0xf, true); });
    v += step([](int x) { return PVV_DPP(0, x, 0x118, 0xf, true); });
    v += step([](int x) { return PVV_DPP(0, x, 0x142, 0xa, false); });
    v += step([](int x) { return PVV_DPP(0, x, 0x143, 0xc, false); });
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)u, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(u >> 32), 63);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

__device__ __forceinline__ int wave_max(int v)
{
#define PVV_STEP(ctrl, rmask) v = max(v, PVV_DPP(v, v, ctrl, rmask, false))
    PVV_STEP(0x111, 0xf); PVV_STEP(0x112, 0xf); PVV_STEP(0x114, 0xf); PVV_STEP(0x118, 0xf); PVV_STEP(0x142, 0xa); PVV_STEP(0x143, 0xc);
#undef PVV_STEP
    return __builtin_amdgcn_readlane(v, 63);
}

// sum of a float over the wave (order of the row shifts): only where the last bits do not matter
__device__ __forceinline__ float wave_total(float v)
{
#define PVV_STEP(ctrl, rmask, bc) v += __int_as_float(PVV_DPP(0, __float_as_int(v), ctrl, rmask, bc))
    PVV_STEP(0x111, 0xf, true); PVV_STEP(0x112, 0xf, true); PVV_STEP(0x114, 0xf, true); PVV_STEP(0x118, 0xf, true);
    PVV_STEP(0x142, 0xa, false); PVV_STEP(0x143, 0xc, false);
#undef PVV_STEP
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// arg-max over the wave with torch.max's tie rule (the FIRST index among equal counts): best/idx of all lanes, in every lane
__device__ __forceinline__ void wave_argmax_first(int &best, int &idx)
{
#define PVV_STEP(ctrl, rmask) { const int oc = PVV_DPP(best, best, ctrl, rmask, false), oi = PVV_DPP(idx, idx, ctrl, rmask, false); \
                                if (oc > best || (oc == best && oi < idx)) { best = oc; idx = oi; } }
    PVV_STEP(0x111, 0xf) PVV_STEP(0x112, 0xf) PVV_STEP(0x114, 0xf) PVV_STEP(0x118, 0xf) PVV_STEP(0x142, 0xa) PVV_STEP(0x143, 0xc)
#undef PVV_STEP
    best = __builtin_amdgcn_readlane(best, 63);
    idx = __builtin_amdgcn_readlane(idx, 63);
}

// Sum over the 256-thread block; result valid in every thread.  `red` holds >= 4 T.
template <typename T>
__device__ __forceinline__ T block_sum(T v, T *red)
{
    v = wave_total(v);
    __syncthreads();
    if (lane_id() == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// Counter-based RNG: splitmix64 finaliser over (seed, stream, a, b).  Used when no
// idxs / selection tensors are injected; statistical parity with torch's Philox only.
__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint32_t rng_u32(uint64_t seed, uint32_t stream, uint32_t a, uint32_t b)
{
    uint64_t k = mix64(seed + 0x9E3779B97F4A7C15ull * (uint64_t)(stream + 1));
    return (uint32_t)(mix64(k ^ (((uint64_t)a << 32) | b)) >> 32);
}
