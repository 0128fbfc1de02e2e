// pvnet_vote.hip -- RANSAC voting hot path of clean-pvnet as native HIP for gfx950 (MI355X).
//
// Written for CDNA4 only: 64-lane wavefronts are assumed everywhere (ballot masks are 64 bit,
// lane broadcasts use v_readlane), there is no CUDA path and no CPU fallback.
//
// Arithmetic contract (see oracle/vote_oracle.c): the inlier decision and the hypothesis
// intersection are IEEE binary32 with one rounding per source-level operation and NO fused
// multiply-add, so inlier counts are bit-exact against the oracle.  This file must be built
// with -ffp-contract=off; the pragma below enforces it even if the flag is forgotten.
//
// Reference behaviour restated (paths relative to /root/reference):
//   K = lib/csrc/ransac_voting/src/ransac_voting_kernel.cu
//   P = lib/csrc/ransac_voting/ransac_voting_gpu.py
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "pvnet_vote.h"

#pragma clang fp contract(off)

#define PVV_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

constexpr int kBlock = 256;          // 4 wavefronts
constexpr int kTileSteps = 8;
constexpr int kTile = kBlock * kTileSteps;  // pixels per compaction tile
constexpr int kPixPerWave = 64;      // pixels one wave walks per work item of the count kernel

thread_local char g_err[512] = "";

int fail(int code, const char *msg)
{
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return PVV_OK;
}

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

// Sum over the 256-thread block; result valid in every thread.  `red` holds >= 4 T.
template <typename T>
__device__ __forceinline__ T block_sum(T v, T *red)
{
    v = wave_sum(v);
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

// ---------------------------------------------------------------------------------------------
// Stage 1: foreground compaction (replaces sum / nonzero / masked_select / uniform_ of
// P:125-144 and P:207-229), order = row-major order of torch.nonzero.
// ---------------------------------------------------------------------------------------------
struct MaskArgs {
    const void *mask;
    const float *selection;  // [B,H,W] injected U(0,1) or nullptr
    int64_t sb, sh, sw;      // element strides
    int es;                  // element size in bytes
    int contig;              // sh == W && sw == 1
    int mode;                // 0: v3 (low byte != 0, weight = low byte)  1: estimate (== 1)
    int W, HW, T;
    int min_num, max_num, cap;
    uint64_t seed;
    // fused argmax (decode_keypoint): when seg != nullptr the mask value is argmax_c seg[b,c,y,x]
    const float *seg;
    long long *mask_out;     // [B,H,W] int64 or nullptr
    int64_t gb, gc, gh, gw;  // element strides of seg
    int C;
};

template <int ES>
__device__ __forceinline__ uint64_t load_elem(const void *base, int64_t off)
{
    if (ES == 1) return ((const uint8_t *)base)[off];
    if (ES == 2) return ((const uint16_t *)base)[off];
    if (ES == 4) return ((const uint32_t *)base)[off];
    return ((const uint64_t *)base)[off];
}

// weight of pixel p of image b: 0 = background; v3: low byte (P:125-126 sums the bytes),
// estimate: 1 (P:207-208).
// torch.argmax over the class axis: first maximal index, a NaN beats everything (and the first NaN wins)
__device__ __forceinline__ int argmax_class(const MaskArgs &a, int b, int p)
{
    const int y = p / a.W;
    const int x = p - y * a.W;
    const float *q = a.seg + (int64_t)b * a.gb + (int64_t)y * a.gh + (int64_t)x * a.gw;
    float best = q[0];
    int idx = 0;
    for (int c = 1; c < a.C; ++c) {
        const float v = q[(int64_t)c * a.gc];
        if (v > best || (v != v && best == best)) { best = v; idx = c; }
    }
    return idx;
}

template <int ES>
__device__ __forceinline__ int mask_weight(const MaskArgs &a, int b, int p)
{
    if (a.seg) {
        const int idx = argmax_class(a, b, p);
        if (a.mask_out) a.mask_out[(int64_t)b * a.HW + p] = idx;
        return a.mode == 0 ? (idx & 0xFF) : (idx == 1 ? 1 : 0);
    }
    int64_t off;
    if (a.contig) {
        off = (int64_t)b * a.sb + p;
    } else {
        int y = p / a.W;
        int x = p - y * a.W;
        off = (int64_t)b * a.sb + (int64_t)y * a.sh + (int64_t)x * a.sw;
    }
    uint64_t v = load_elem<ES>(a.mask, off);
    if (a.mode == 0) return (int)(v & 0xFF);
    return v == 1 ? 1 : 0;
}

// U(0,1) draw of P:136 / P:220 for pixel p of image b.
__device__ __forceinline__ float selection_draw(const MaskArgs &a, int b, int p)
{
    if (a.selection) return a.selection[(int64_t)b * a.HW + p];
    return (float)(rng_u32(a.seed, 0u, (uint32_t)b, (uint32_t)p) >> 8) * 0x1p-24f;
}

// Pass 1 -- the ONLY pass that reads the mask: per tile of 2048 pixels the foreground count, the weight sum
// (foreground_num of P:126 sums byte VALUES) and a 2048-bit foreground map (one wave64 ballot per 64 pixels,
// word s*4+w = step s, wave w).  Later passes work from the bit map.
template <int ES>
__global__ __launch_bounds__(kBlock) void k_tile_count(MaskArgs a, int *__restrict__ tile_nz,
                                                       int *__restrict__ tile_sum,
                                                       unsigned long long *__restrict__ bits)
{
    __shared__ int red[4];
    const int t = blockIdx.x, b = blockIdx.y;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    unsigned long long *wb = bits + ((size_t)b * a.T + t) * (kTileSteps * 4);
    int nz = 0, sum = 0;
#pragma unroll
    for (int s = 0; s < kTileSteps; ++s) {
        int p = t * kTile + s * kBlock + threadIdx.x;
        int w = 0;
        if (p < a.HW) w = mask_weight<ES>(a, b, p);
        unsigned long long m = __ballot(w != 0);
        if (lane == 0) wb[s * 4 + wave] = m;
        nz += (w != 0);
        sum += w;
    }
    nz = block_sum(nz, red);
    sum = block_sum(sum, red);
    if (threadIdx.x == 0) {
        tile_nz[b * a.T + t] = nz;
        tile_sum[b * a.T + t] = sum;
    }
}

// foreground_num of P:126 / P:208 from the per-tile partial sums.
__device__ __forceinline__ long long image_fg(const int *__restrict__ tile_sum, int b, int T,
                                              long long *red)
{
    long long s = 0;
    for (int i = threadIdx.x; i < T; i += kBlock) s += tile_sum[b * T + i];
    return block_sum(s, red);
}

// P:135-138 / P:219-223: when foreground_num > max_num every foreground pixel survives with
// probability max_num/foreground_num (binary32 quotient).  Clears the dropped pixels in the bit map and
// recounts the tile.  Images that are not subsampled exit at once.
__global__ __launch_bounds__(kBlock) void k_tile_subsample(MaskArgs a, int *__restrict__ tile_nz,
                                                           const int *__restrict__ tile_sum,
                                                           unsigned long long *__restrict__ bits)
{
    __shared__ long long redl[4];
    __shared__ int red[4];
    const int t = blockIdx.x, b = blockIdx.y;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    long long fg = image_fg(tile_sum, b, a.T, redl);
    if (fg <= (long long)a.max_num) return;
    const float prob = (float)a.max_num / (float)fg;
    unsigned long long *wb = bits + ((size_t)b * a.T + t) * (kTileSteps * 4);
    int nz = 0;
#pragma unroll
    for (int s = 0; s < kTileSteps; ++s) {
        int p = t * kTile + s * kBlock + threadIdx.x;
        bool f = (wb[s * 4 + wave] >> lane) & 1ull;
        if (f) f = selection_draw(a, b, p) < prob;
        unsigned long long m = __ballot(f);
        if (lane == 0) wb[s * 4 + wave] = m;
        nz += f ? 1 : 0;
    }
    nz = block_sum(nz, red);
    if (threadIdx.x == 0) tile_nz[b * a.T + t] = nz;
}

struct VertexArgs {
    const float *vertex;
    int64_t sb, sh, sw, sk, sc;
    int K;
    int vec2;  // sc == 1 and every other stride even: (x,y) is one aligned 8-byte load
    double kappa;  // thresh / sqrt(1 - thresh^2) for the fast-path records, 0 = no records
};

// Per (image, keypoint, compacted pixel) record of the fast inlier test, 32 bytes = one
// s_load_dwordx8 in the count kernel:
//   lo = (cx, cy, nhx, nhy)   nh = n / |n|  (binary64 quotient rounded once)
//   hi = (Bx, By, nx, ny)     B  = kappa * perp(nh); (nx,ny) raw, for the exact fallback
// A pixel the exact test can never accept (K:121: norm1 < 1e-6, or a non-finite norm1) gets
// cx = +inf, nh = (1,0), B = (1,0): then a = b' = -inf, t = a - |b'| = -inf (never an inlier) and the
// ambiguity measure is +inf (never flagged).
struct __attribute__((aligned(32))) PixelRec {
    float4 lo, hi;
};

__device__ __forceinline__ PixelRec make_record(float cx, float cy, float nx, float ny, double kappa)
{
    PixelRec r;
    float norm1 = sqrtf(nx * nx + ny * ny);           // the exact path's own norm1 (K:116)
    bool ok = !lt_1e6(norm1) && norm1 < INFINITY && norm1 == norm1;
    if (ok) {
        double N1 = sqrt((double)nx * (double)nx + (double)ny * (double)ny);
        double ux = (double)nx / N1, uy = (double)ny / N1;
        r.lo = make_float4(cx, cy, (float)ux, (float)uy);
        r.hi = make_float4((float)(-kappa * uy), (float)(kappa * ux), nx, ny);
    } else {
        r.lo = make_float4(INFINITY, 0.f, 1.f, 0.f);
        r.hi = make_float4(1.f, 0.f, nx, ny);
    }
    return r;
}

// Ordered scatter: pixel -> row r of the image's compacted list; writes coords[b][r] = (x,y)
// (P:140-141) and dirs[b][vi][r] = vertex[b,y,x,vi,:] (P:142-143, stored planar per keypoint so
// that the count kernel's loads are unit-stride).
__global__ __launch_bounds__(kBlock) void k_compact(MaskArgs a, VertexArgs v,
                                                    const int *__restrict__ tile_nz,
                                                    const int *__restrict__ tile_sum,
                                                    const unsigned long long *__restrict__ bits,
                                                    int *__restrict__ tn_out,
                                                    float2 *__restrict__ coords,
                                                    float2 *__restrict__ dirs,
                                                    PixelRec *__restrict__ recs)
{
    __shared__ long long redl[4];
    __shared__ int red[4];
    __shared__ int seg[kTileSteps * 4 + 1];
    __shared__ unsigned short list[kTile];
    const int t = blockIdx.x, b = blockIdx.y;
    const int lane = lane_id(), wave = threadIdx.x >> 6;

    if (t != 0 && tile_nz[b * a.T + t] == 0) return;  // background-only tile: nothing to scatter

    const long long fg = image_fg(tile_sum, b, a.T, redl);
    if (fg < (long long)a.min_num) {  // P:129-132 / P:211-216: image skipped
        if (t == 0 && threadIdx.x == 0) tn_out[b] = 0;
        return;
    }

    int before = 0, total = 0;
    for (int i = threadIdx.x; i < a.T; i += kBlock) {
        int c = tile_nz[b * a.T + i];
        total += c;
        if (i < t) before += c;
    }
    before = block_sum(before, red);
    total = block_sum(total, red);
    if (t == 0 && threadIdx.x == 0) tn_out[b] = total < a.cap ? total : a.cap;

    const unsigned long long *wb = bits + ((size_t)b * a.T + t) * (kTileSteps * 4);
    unsigned long long word[kTileSteps];
#pragma unroll
    for (int s = 0; s < kTileSteps; ++s) {
        word[s] = wb[s * 4 + wave];                       // wave-uniform
        if (lane == 0) seg[s * 4 + wave] = __popcll(word[s]);
    }
    __syncthreads();
    if (threadIdx.x < 64) {  // wave 0: exclusive scan of the 32 (step,wave) segment counts
        int c = threadIdx.x < kTileSteps * 4 ? seg[threadIdx.x] : 0;
        int inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int n = __shfl_up(inc, o, 64);
            if (lane >= o) inc += n;
        }
        if (threadIdx.x < kTileSteps * 4) seg[threadIdx.x] = inc - c;
    }
    __syncthreads();

    // foreground pixels of the tile -> LDS list (in rank order), so that the K-fold gather below is spread over
    // all 256 threads instead of looping inside the few lanes that own a foreground pixel
    int nfg = 0;
#pragma unroll
    for (int s = 0; s < kTileSteps; ++s) nfg += __popcll(word[s]);      // this wave's pixels ...
    __syncthreads();
#pragma unroll
    for (int s = 0; s < kTileSteps; ++s) {
        const unsigned long long m = word[s];
        if (!((m >> lane) & 1ull)) continue;
        const int lr = seg[s * 4 + wave] + __popcll(m & ((1ull << lane) - 1ull));   // rank within the tile
        list[lr] = (unsigned short)(s * kBlock + threadIdx.x);
    }
    __syncthreads();
    const int tile_n = tile_nz[b * a.T + t];
    const int room = a.cap - before;                                     // rows left in the image's list
    const int n = tile_n < room ? tile_n : (room > 0 ? room : 0);
    (void)nfg;
    for (int i = threadIdx.x; i < n; i += kBlock) {
        const int p = t * kTile + list[i];
        const int y = p / a.W;
        coords[(size_t)b * a.cap + before + i] = make_float2((float)(p - y * a.W), (float)y);
    }
    for (int i = threadIdx.x; i < n * v.K; i += kBlock) {
        const int vi = i / n, li = i - vi * n;                           // consecutive threads -> consecutive rows
        const int p = t * kTile + list[li];
        const int y = p / a.W;
        const int x = p - y * a.W;
        const float *src = v.vertex + (int64_t)b * v.sb + (int64_t)y * v.sh + (int64_t)x * v.sw + (int64_t)vi * v.sk;
        float2 d;
        if (v.vec2) {
            d = *(const float2 *)src;
        } else {
            d.x = src[0];
            d.y = src[v.sc];
        }
        const size_t row = ((size_t)b * v.K + vi) * a.cap + before + li;
        dirs[row] = d;
        if (v.kappa != 0.0) recs[row] = make_record((float)x, (float)y, d.x, d.y, v.kappa);
    }
}

// ---------------------------------------------------------------------------------------------
// Stage 2: hypotheses (replaces random_ P:145/P:235 + generate_hypothesis K:11-86), and zeroes
// the inlier counters of the same (b,vi,hi).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_gen_hypothesis(
    const int32_t *__restrict__ idxs /*[B,hn,K,2] or null*/, const int *__restrict__ tn_arr,
    const float2 *__restrict__ coords, const float2 *__restrict__ dirs, float2 *__restrict__ hyps,
    int *__restrict__ counts, int B, int K, int hn, int cap, uint64_t seed)
{
    const long long gid = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= (long long)B * K * hn) return;
    const int hi = (int)(gid % hn);
    const int vi = (int)((gid / hn) % K);
    const int b = (int)(gid / ((long long)hn * K));
    counts[gid] = 0;
    const int tn = tn_arr[b];
    if (tn <= 0) {
        hyps[gid] = make_float2(0.f, 0.f);
        return;
    }
    int t0, t1;
    if (idxs) {
        const int32_t *ip = idxs + (((size_t)b * hn + hi) * K + vi) * 2;
        t0 = ip[0];
        t1 = ip[1];
        // the reference reads out of bounds here; clamp instead of faulting
        t0 = t0 < 0 ? 0 : (t0 >= tn ? tn - 1 : t0);
        t1 = t1 < 0 ? 0 : (t1 >= tn ? tn - 1 : t1);
    } else {
        uint32_t c = (uint32_t)(hi * K + vi) * 2u;
        t0 = (int)(rng_u32(seed, 1u, (uint32_t)b, c) % (uint32_t)tn);
        t1 = (int)(rng_u32(seed, 1u, (uint32_t)b, c + 1u) % (uint32_t)tn);
    }
    const float2 *dp = dirs + ((size_t)b * K + vi) * cap;
    const float2 *cp = coords + (size_t)b * cap;
    float2 d0 = dp[t0], d1 = dp[t1], c0 = cp[t0], c1 = cp[t1];
    hyps[gid] = hypothesis_exact(d0.x, d0.y, c0.x, c0.y, d1.x, d1.y, c1.x, c1.y);
}

// ---------------------------------------------------------------------------------------------
// Stage 3: inlier counting -- the hot kernel.  Replaces voting_for_hypothesis (K:88-167) +
// torch.sum(inlier, 2) (P:159 / P:243) without the [hn,vn,tn] byte scratch.
//
// Mapping (wave64): LANES ARE HYPOTHESES.  Each lane keeps R hypotheses of one keypoint and their
// R counters in VGPRs; the wave loads 64 compacted pixels with one coalesced load per array,
// then walks them one by one, broadcasting a pixel's (cx,cy,nx,ny,norm1) to SGPRs with
// v_readlane so that every evaluation is VGPR(hypothesis) x SGPR(pixel) arithmetic.  Counters
// are private per lane: no cross-lane reduction in the loop; one atomicAdd per (lane, r) per
// work item at the end (integer adds => order independent => bit-exact counts).
//
// Work item = (image b, keypoint vi, hypothesis tile of 64*R, pixel chunk of 4 waves x 64 px).
// The number of items depends on tn[b], which only the device knows, so the grid is persistent
// and every block derives the item list from tn[] itself (no host sync, no empty blocks).
// ---------------------------------------------------------------------------------------------
// ---- work-item table of the persistent count kernels ---------------------------------------------------------
// The number of pixel chunks of an image depends on tn[b], which only the device knows, so every block builds the
// same table itself: item_end[b] = inclusive prefix of (chunks of image b) * items_per_chunk.  No host sync, no
// empty blocks.  Returns the total number of items (valid in every thread after the barrier inside).
constexpr int kMaxBatchLds = 1024;  // images per launch (the table lives in LDS)

__device__ __forceinline__ int build_item_table(int *item_end, const int *__restrict__ tn_arr, int tn_fixed, int B,
                                                int pixels_per_chunk, int items_per_chunk)
{
    const int lane = lane_id();
    if (wave_id() == 0) {
        int carry = 0;
        for (int b0 = 0; b0 < B; b0 += 64) {
            const int b = b0 + lane;
            int n = 0;
            if (b < B) {
                const int tn = tn_arr ? tn_arr[b] : tn_fixed;
                n = ((tn + pixels_per_chunk - 1) / pixels_per_chunk) * items_per_chunk;
            }
            int inc = n;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int m = __shfl_up(inc, o, 64);
                if (lane >= o) inc += m;
            }
            inc += carry;
            if (b < B) item_end[b] = inc;
            carry = __builtin_amdgcn_readlane(inc, 63);
        }
    }
    __syncthreads();
    return item_end[B - 1];
}

// image of work item `item` (first b with item_end[b] > item) and the item's index within that image
__device__ __forceinline__ int locate_item(const int *item_end, int B, int item, int *local)
{
    int lo = 0, hi = B - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (item_end[mid] > item) hi = mid; else lo = mid + 1;
    }
    const int b = __builtin_amdgcn_readfirstlane(lo);
    *local = __builtin_amdgcn_readfirstlane(item - (b ? item_end[b - 1] : 0));
    return b;
}

struct CountArgs {
    const float2 *coords;  // pixel p of image b: coords[b*c_b + p]
    const float2 *dirs;    // dirs[b*d_b + vi*d_v + p*d_p]
    const float2 *hyps;    // hyps[b*h_b + vi*h_v + hi*h_h]
    int *counts;           // counts[b*h_b + vi*h_v + hi*h_h]
    const int *tn_arr;     // per image, or nullptr -> tn_fixed
    long long c_b, d_b, d_v, d_p, h_b, h_v, h_h;
    int tn_fixed;
    int B, K, hn;
    float thresh;
};

template <int R>
__global__ __launch_bounds__(kBlock) void k_count_inliers(CountArgs a)
{
    __shared__ int item_end[kMaxBatchLds];  // inclusive prefix of items per image
    const int lane = lane_id(), wave = wave_id();
    constexpr int HT = 64 * R;
    constexpr int PC = 4 * kPixPerWave;
    const int nht = (a.hn + HT - 1) / HT;
    const int per_chunk = a.K * nht;

    const int total = build_item_table(item_end, a.tn_arr, a.tn_fixed, a.B, PC, per_chunk);

    for (int item = blockIdx.x; item < total; item += gridDim.x) {
        int local;
        const int b = locate_item(item_end, a.B, item, &local);
        const int chunk = local / per_chunk;
        const int rem = local - chunk * per_chunk;
        const int vi = rem / nht;
        const int ht = rem - vi * nht;
        const int tn = __builtin_amdgcn_readfirstlane(a.tn_arr ? a.tn_arr[b] : a.tn_fixed);

        // this lane's R hypotheses (NaN => never an inlier => padding)
        float hx[R], hy[R];
        int cnt[R];
        const long long hbase = (long long)b * a.h_b + (long long)vi * a.h_v;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int h = ht * HT + r * 64 + lane;
            float2 hp = make_float2(NAN, NAN);
            if (h < a.hn) hp = a.hyps[hbase + (long long)h * a.h_h];
            hx[r] = hp.x;
            hy[r] = hp.y;
            cnt[r] = 0;
        }

        const int p0 = chunk * PC + wave * kPixPerWave;
        const int nvalid = min(kPixPerWave, tn - p0);  // wave-uniform
        if (nvalid > 0) {
            float cx = 0.f, cy = 0.f, nx = 0.f, ny = 0.f;
            if (lane < nvalid) {
                float2 c = a.coords[(long long)b * a.c_b + p0 + lane];
                float2 d = a.dirs[(long long)b * a.d_b + (long long)vi * a.d_v +
                                  (long long)(p0 + lane) * a.d_p];
                cx = c.x; cy = c.y; nx = d.x; ny = d.y;
            }
            float norm1 = sqrtf(nx * nx + ny * ny);
            if (lt_1e6(norm1)) norm1 = NAN;  // K:121 reject, folded into the quotient below

            for (int j = 0; j < nvalid; ++j) {
                const float scx = bcast(cx, j), scy = bcast(cy, j);
                const float snx = bcast(nx, j), sny = bcast(ny, j);
                const float sn1 = bcast(norm1, j);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    float dx = hx[r] - scx;
                    float dy = hy[r] - scy;
                    float norm2 = sqrtf(dx * dx + dy * dy);
                    float angle = (dx * snx + dy * sny) / (sn1 * norm2);
                    cnt[r] += (!lt_1e6(norm2) && angle > a.thresh) ? 1 : 0;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int h = ht * HT + r * 64 + lane;
            if (h < a.hn && cnt[r] != 0) atomicAdd(&a.counts[hbase + (long long)h * a.h_h], cnt[r]);
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Stage 3, fast form.  Same mapping (lanes are hypotheses, R per lane) but:
//   * the pixel arrives as a 32-byte PixelRec through a SCALAR load (s_load_dwordx8): zero VALU
//     cycles, the operands of every evaluation are VGPR(hypothesis) x SGPR(pixel);
//   * hypotheses are processed in PAIRS with packed fp32 (v_pk_add/mul/fma_f32: two evaluations per
//     issue slot -- measured 4.2 cycles per packed wave-instruction vs 4.1 for a scalar v_fma_f32 on
//     gfx950, tools/microbench);
//   * no sqrt, no divide.  With d = h - c (the SAME rounded subtraction as the exact path),
//     nh = n/|n| and kappa = T/sqrt(1-T^2):
//         a  = d . nh            = |d| cos(theta)
//         b' = kappa * d x nh    = kappa |d| sin(theta)
//         cos(theta) > T  <=>  t := a - |b'| > 0
//   * the decision is taken from t only when it is OUTSIDE a guard band,  |t| - beta*a > eps_abs;
//     inside it (about 3e-6 of all evaluations) the pixel is re-evaluated with the exact binary32
//     sequence of K:100-125 and the counters are corrected.  Derivation of beta:
//     the exact path's computed cosine deviates from the true one by <= 8u (u = 2^-24), which is
//     |d| 8u/(1-T^2) in t; the fast path's t deviates by <= 3u(1+kappa)|d|; a ~ T|d| in the band.
//     eps_abs covers the exact path's norm2 < 1e-6 reject (K:121): |d| <= 1e-6 => |t| <= (1+kappa)|d|.
//   * hypotheses that are not finite or beyond 1e15 px (where the exact path's squares overflow and
//     the bounds above stop holding) send the whole work item down the exact loop.
// Inlier counts stay bit-exact against the oracle; tests/test_gpu_parity.py hammers the band.
// ---------------------------------------------------------------------------------------------
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float8v __attribute__((ext_vector_type(8)));

struct FastConsts {
    float beta;     // relative half-width of the guard band (in units of a)
    float eps_abs;  // absolute floor of the band, px
};

// (a, b') of one hypothesis pair against the pixel held in SGPRs.  cxy=(cx,cy), nh=(nhx,nhy), Bv=(Bx,By).
__device__ __forceinline__ void pk_project(float2v hx2, float2v hy2, float2v cxy, float2v nh, float2v Bv,
                                           float2v &a2, float2v &b2)
{
    float2v dx2, dy2, p2, q2;
    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dx2) : "v"(hx2), "s"(cxy));
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dy2) : "v"(hy2), "s"(cxy));
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(p2) : "v"(dy2), "s"(nh));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(a2) : "v"(dx2), "s"(nh), "v"(p2));
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(q2) : "v"(dy2), "s"(Bv));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(b2) : "v"(dx2), "s"(Bv), "v"(q2));
}

// t = a - |b|: one v_sub_f32 with an abs source modifier (f32 add/sub issue in 2.4 cycles on gfx950, every
// other VALU op in ~4.2).  Kept as asm so the SLP vectoriser cannot turn two of them into
// v_and + v_and + v_pk_add.
__device__ __forceinline__ float sub_abs(float a, float b)
{
    float t;
    asm("v_sub_f32 %0, %1, |%2|" : "=v"(t) : "v"(a), "v"(b));
    return t;
}

template <int R>
__global__ __launch_bounds__(kBlock) void k_count_fast(
    const float8v *__restrict__ recs /*[B,K,cap]*/, const float2 *__restrict__ hyps /*[B,K,hn]*/,
    int *__restrict__ counts /*[B,K,hn]*/, const int *__restrict__ tn_arr, int B, int K, int hn, int cap,
    float thresh, FastConsts fc, int max_pix_per_wave, int target_items)
{
    static_assert(R % 2 == 0, "hypotheses are processed in pairs");
    __shared__ int item_end[kMaxBatchLds];
    __shared__ int s_ppw;
    const int lane = lane_id(), wave = wave_id();
    constexpr int HT = 64 * R;
    const int nht = (hn + HT - 1) / HT;
    const int per_chunk = K * nht;

    // Pixels one wave walks per work item: as many as max_pix_per_wave (amortises the hypothesis loads and
    // the final atomics) but few enough that the batch still splits into >= target_items items -- a
    // single 480x640 image must spread over 256 CUs too.  Every block derives the same value from tn[].
    if (wave == 0) {
        long long px = 0;
        for (int b = lane; b < B; b += 64) px += tn_arr[b];
        px = wave_sum(px) * per_chunk;
        long long want = px / (4ll * target_items);
        int ppw = (int)(want < 16 ? 16 : (want > max_pix_per_wave ? max_pix_per_wave : want));
        if (lane == 0) s_ppw = ppw;
    }
    __syncthreads();
    const int pix_per_wave = __builtin_amdgcn_readfirstlane(s_ppw);
    const int PC = 4 * pix_per_wave;

    const int total = build_item_table(item_end, tn_arr, 0, B, PC, per_chunk);

    for (int item = blockIdx.x; item < total; item += gridDim.x) {
        int local;
        const int b = locate_item(item_end, B, item, &local);
        const int chunk = local / per_chunk;
        const int rem = local - chunk * per_chunk;
        const int vi = rem / nht;
        const int ht = rem - vi * nht;
        const int tn = __builtin_amdgcn_readfirstlane(tn_arr[b]);
        const int bk = b * K + vi;

        // this lane's R hypotheses; lanes past hn get (0,0) and never write their counters
        float2v hx2[R / 2], hy2[R / 2];
        int cnt[R];
        bool far = false;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int h = ht * HT + r * 64 + lane;
            float2 hp = make_float2(0.f, 0.f);
            if (h < hn) hp = hyps[(size_t)bk * hn + h];
            hx2[r / 2][r & 1] = hp.x;
            hy2[r / 2][r & 1] = hp.y;
            cnt[r] = 0;
            far |= !(fabsf(hp.x) < 1e15f && fabsf(hp.y) < 1e15f);
        }
        const int p0 = chunk * PC + wave * pix_per_wave;
        const int p1 = min(tn, p0 + pix_per_wave);
        const float8v *rp = recs + (size_t)bk * cap;

        if (__builtin_expect(__any(far), 0)) {
            // exact loop (K:100-125) for the whole work item
            for (int p = p0; p < p1; ++p) {
                const float8v rec = rp[p];
                const float cx = rec[0], cy = rec[1], nx = rec[6], ny = rec[7];
#pragma unroll
                for (int r = 0; r < R; ++r)
                    cnt[r] += vote_exact(cx, cy, hx2[r / 2][r & 1], hy2[r / 2][r & 1], nx, ny, thresh) ? 1 : 0;
            }
        } else if (p1 > p0) {
            // Counting by sign bit: t < 0 (not an inlier) shifts a 1 into a per-hypothesis bit queue
            // (one v_alignbit_b32 per evaluation); every 32 pixels the queue is popcounted.  t = +0 is
            // always inside the guard band, so "sign bit clear" == "fast path says inlier".
            int neg[R];
#pragma unroll
            for (int r = 0; r < R; ++r) neg[r] = 0;
            for (int pp = p0; pp < p1; pp += 32) {
                unsigned acc[R];
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] = 0u;
                const int pe = min(p1, pp + 32);
                for (int p = pp; p < pe; ++p) {
                    const float8v rec = rp[p];     // wave-uniform address -> scalar load
                    const float2v cxy = {rec[0], rec[1]}, nh = {rec[2], rec[3]}, Bv = {rec[4], rec[5]};
                    float zmin = INFINITY;
#pragma unroll
                    for (int q = 0; q < R / 2; ++q) {
                        float2v a2, b2;
                        pk_project(hx2[q], hy2[q], cxy, nh, Bv, a2, b2);
                        const float t0 = sub_abs(a2[0], b2[0]);
                        const float t1 = sub_abs(a2[1], b2[1]);
                        acc[2 * q] = __builtin_amdgcn_alignbit(acc[2 * q], __float_as_uint(t0), 31);
                        acc[2 * q + 1] = __builtin_amdgcn_alignbit(acc[2 * q + 1], __float_as_uint(t1), 31);
                        const float z0 = __builtin_fmaf(-fc.beta, a2[0], fabsf(t0));
                        const float z1 = __builtin_fmaf(-fc.beta, a2[1], fabsf(t1));
                        zmin = fminf(fminf(zmin, z0), z1);     // one v_min3_f32
                    }
                    if (__builtin_expect(__any(zmin <= fc.eps_abs), 0)) {
                        // some evaluation of this pixel sits inside the guard band: replace the fast
                        // decisions of the pixel by the exact ones
                        const float cx = rec[0], cy = rec[1], nx = rec[6], ny = rec[7];
#pragma unroll
                        for (int q = 0; q < R / 2; ++q) {
                            float2v a2, b2;
                            pk_project(hx2[q], hy2[q], cxy, nh, Bv, a2, b2);
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                const float t = sub_abs(a2[e], b2[e]);
                                const int fast = (__float_as_uint(t) >> 31) ? 0 : 1;
                                const int exact = vote_exact(cx, cy, hx2[q][e], hy2[q][e], nx, ny, thresh) ? 1 : 0;
                                cnt[2 * q + e] += exact - fast;
                            }
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < R; ++r) neg[r] += __popc(acc[r]);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) cnt[r] += (p1 - p0) - neg[r];
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int h = ht * HT + r * 64 + lane;
            if (h < hn && cnt[r] != 0) atomicAdd(&counts[(size_t)bk * hn + h], cnt[r]);
        }
    }
}



// ---------------------------------------------------------------------------------------------
// Stage 3, matrix-core prefilter form (k_count_bf16).
//
// k_count_fast is bound by VALU issue (7.35 VALU instructions per evaluation, 84 % VALU busy); 4 of the 6.5
// useful ones are the two dot products a = d.nh and b' = kappa d x nh.  Those are bilinear in (hx,hy,1) and the
// pixel's (nh, -c.nh) / (B, -c.B): a rank-3 form.  The f32 MFMA shares the fp32 VALU datapath on gfx950
// (tools/microbench/mfma_valu_overlap.hip: no overlap), but the bf16 matrix core is a separate pipe that does
// overlap (tools/microbench/bf16_mfma_overlap.hip).  So every fp32 operand is split EXACTLY into three bf16
// pieces x = x0 + x1 + x2 (+ <= 2^-27 |x|), the six leading piece products of hx*nhx and of hy*nhy plus the three
// pieces of the constant are the 15 terms of a K=16 dot product, and ONE v_mfma_f32_32x32x16_bf16 delivers a
// (rows 0-15) and b' (rows 16-31) for 16 pixels x 32 hypotheses.  The VALU keeps t = a - |b'|, the sign-bit
// queue and the guard-band measure: 29 instructions per 512 evaluations instead of 56 x 4.
//
// The MFMA result is only a PREFILTER: the decision is taken from it when |t| - beta*a > eps, otherwise that
// evaluation is redone with the exact binary32 sequence (K:100-125).  Bound, with u = 2^-24, d = fl(h-c) as the
// exact path sees it, o = the block's integer origin, c' = c-o (exact), h' = fl(h-o), C1 = max |c'|_1 of the block:
//     piece residuals and dropped piece products          <= 0.5 u S,   S = |hx' nhx| + |hy' nhy| + |c'.nh|
//     bf16 MFMA accumulation (products exact in f32; 15 f32 roundings in any order)  <= 15 u S
//     (measured on MI355X: 3.9 u S including the split, bf16_mfma_overlap.hip)
//     fl(h-o), the exact path's fl(h-c), f32 unit normal (3u), f32 c'.nh                 
//  => |a_mfma - a_true| <= u (30 |d| + 34 C1),  kappa times that for b', and
//     beta = 1.25 (30 (1+kappa) + 8/(1-T^2)) u / T,     eps = 1.25 (1+kappa) 34 u C1 + eps_abs.
// Inlier counts stay bit-exact (tests/test_gpu_parity.py, every parity test runs through this kernel by default).
//
// Layout (MI355X_MICROARCH / verified in the microbenchmark): A operand lane l = row l%32, k = 8*(l/32)..+7;
// B operand lane l = column l%32, same k; D register r of lane l = row 4*(l/32) + r%4 + 8*(r/4), column l%32.
// Rows = (form, pixel) of a 16-pixel tile, columns = 32 hypotheses: every lane owns ONE hypothesis and 8 of the
// 16 pixels; lanes l and l^32 share a hypothesis and are merged in LDS.
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

constexpr int kBfPixPerWave = 128;   // 8 tiles of 16 pixels, A operands live in 32 VGPRs
constexpr int kBfMaxHt = 16;         // 32-hypothesis tiles per work item (512 hypotheses)

struct Bf16Consts {
    float beta;    // relative half-width of the guard band (units of a)
    float eps_c;   // absolute half-width per pixel of block extent C1
    float eps0;    // absolute floor (covers the exact path's norm2 < 1e-6 reject)
    float kappa;
    float beta2;   // band of the second-level (f32, un-translated) test used on flagged evaluations
};

// x = p[0] + p[1] + p[2] + r, |r| <= 2^-27 |x|; every piece is a bf16 value (round to nearest even)
__device__ __forceinline__ void split3(float x, __bf16 (&p)[3])
{
    p[0] = (__bf16)x;
    float r = x - (float)p[0];
    p[1] = (__bf16)r;
    r = r - (float)p[1];
    p[2] = (__bf16)r;
}

__global__ __launch_bounds__(kBlock) void k_count_bf16(
    const float2 *__restrict__ coords /*[B,cap]*/, const float2 *__restrict__ dirs /*[B,K,cap]*/,
    const float2 *__restrict__ hyps /*[B,K,hn]*/, int *__restrict__ counts /*[B,K,hn]*/,
    const int *__restrict__ tn_arr, int B, int K, int hn, int cap, float thresh, Bf16Consts fc, int target_items)
{
    __shared__ int item_end[kMaxBatchLds];
    __shared__ int s_htpi;
    __shared__ bf16x8 sB[kBfMaxHt * 64];            // B operands of the item's hypothesis tiles (16 KB)
    __shared__ float4 sP[4 * kBfPixPerWave * 2];    // per pixel: (nhx, nhy, cn', -) and (Bx, By, cB', -)  (16 KB)
    __shared__ int sCnt[kBfMaxHt * 32];
    __shared__ float sRed[4];
    const int lane = lane_id(), wave = wave_id();
    constexpr int PC = 4 * kBfPixPerWave;
    const int nt = (hn + 31) >> 5;                  // 32-hypothesis tiles per keypoint

    // hypothesis tiles per work item: up to 16, fewer when the batch is too small to fill the chip
    if (wave == 0) {
        long long chunks = 0;
        for (int b = lane; b < B; b += 64) chunks += (tn_arr[b] + PC - 1) / PC;
        chunks = wave_sum(chunks) * K;
        int htpi = min(nt, kBfMaxHt);
        while (htpi > 2 && chunks * ((nt + htpi - 1) / htpi) < target_items) htpi = (htpi + 1) >> 1;
        if (lane == 0) s_htpi = htpi;
    }
    __syncthreads();
    const int htpi = __builtin_amdgcn_readfirstlane(s_htpi);
    const int nhg = (nt + htpi - 1) / htpi;
    const int per_chunk = K * nhg;

    const int total = build_item_table(item_end, tn_arr, 0, B, PC, per_chunk);
    const int col = lane & 31, kslice = lane >> 5;

    for (int item = blockIdx.x; item < total; item += gridDim.x) {
        int local;
        const int b = locate_item(item_end, B, item, &local);
        const int chunk = local / per_chunk;
        const int rem = local - chunk * per_chunk;
        const int vi = rem / nhg;
        const int hg = rem - vi * nhg;
        const int tn = __builtin_amdgcn_readfirstlane(tn_arr[b]);
        const int bk = b * K + vi;
        const int ht0 = hg * htpi;
        const int nht = min(nt, ht0 + htpi) - ht0;
        const float2 *hyp_k = hyps + (size_t)bk * hn;
        const float2 *crd = coords + (size_t)b * cap;
        const float2 *dir_k = dirs + (size_t)bk * cap;
        const int pb = chunk * PC;                              // first pixel of the block's chunk (< tn)

        __syncthreads();                                        // previous item's LDS fully consumed
        const float2 org = crd[pb];                             // integer origin: the chunk's first pixel

        // ---- per-pixel operands (two pixels per thread): unit normal, its kappa-scaled perpendicular, and the
        //      constants -(c-o).nh, -(c-o).B; a pixel the exact test can never accept (K:121 norm1 < 1e-6, a
        //      non-finite norm1) or beyond tn gets nh = B = 0, constant -1e30: a = -1e30, b' = 0, t < 0.
        float c1 = 0.f;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int pl = threadIdx.x + q * kBlock, p = pb + pl;
            float4 fa = make_float4(0.f, 0.f, -1e30f, 0.f), fb = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p < tn) {
                const float2 c = crd[p], d = dir_k[p];
                const float cx = c.x - org.x, cy = c.y - org.y;  // exact (integers)
                c1 = fmaxf(c1, fabsf(cx) + fabsf(cy));
                const float norm1 = sqrtf(d.x * d.x + d.y * d.y);
                if (!lt_1e6(norm1) && norm1 < INFINITY && norm1 == norm1) {
                    const float ux = d.x / norm1, uy = d.y / norm1;
                    const float bx = -fc.kappa * uy, by = fc.kappa * ux;
                    fa = make_float4(ux, uy, -(cx * ux + cy * uy), cx);   // .w: c' (for the second-level test)
                    fb = make_float4(bx, by, -(cx * bx + cy * by), cy);
                }
            }
            sP[pl * 2] = fa;
            sP[pl * 2 + 1] = fb;
        }
        // ---- B operands: lane l of tile ht holds column l%32, k = 8*(l/32)..+7 of
        //      (qx0,qx1,qx2,qx0,qx1,qx0, qy0,qy1 | qy2,qy0,qy1,qy0, 1,1,1,0),  q = pieces of h' = fl(h - o)
        int far = 0;
        for (int i = threadIdx.x; i < nht * 32; i += kBlock) {
            const int h = (ht0 + (i >> 5)) * 32 + (i & 31);
            float2 hp = make_float2(0.f, 0.f);
            if (h < hn) hp = hyp_k[h];
            far |= !(fabsf(hp.x) < 1e15f && fabsf(hp.y) < 1e15f);
            __bf16 qx[3], qy[3];
            split3(hp.x - org.x, qx);
            split3(hp.y - org.y, qy);
            const __bf16 one = (__bf16)1.f, zero = (__bf16)0.f;
            const bf16x8 lo8 = {qx[0], qx[1], qx[2], qx[0], qx[1], qx[0], qy[0], qy[1]};
            const bf16x8 hi8 = {qy[2], qy[0], qy[1], qy[0], one, one, one, zero};
            sB[(i >> 5) * 64 + (i & 31)] = lo8;
            sB[(i >> 5) * 64 + 32 + (i & 31)] = hi8;
        }
        for (int i = threadIdx.x; i < nht * 32; i += kBlock) sCnt[i] = 0;
        c1 = fmaxf(c1, __shfl_xor(c1, 32, 64));
        c1 = fmaxf(c1, __shfl_xor(c1, 16, 64));
        c1 = fmaxf(c1, __shfl_xor(c1, 8, 64));
        c1 = fmaxf(c1, __shfl_xor(c1, 4, 64));
        c1 = fmaxf(c1, __shfl_xor(c1, 2, 64));
        c1 = fmaxf(c1, __shfl_xor(c1, 1, 64));
        if (lane == 0) sRed[wave] = c1;
        far = __syncthreads_or(far);
        const float C1 = fmaxf(fmaxf(sRed[0], sRed[1]), fmaxf(sRed[2], sRed[3]));
        const float eps = fc.eps0 + fc.eps_c * C1;

        const int p0 = pb + wave * kBfPixPerWave;               // this wave's 128 pixels
        const int npix = min(tn - p0, kBfPixPerWave);           // may be <= 0

        if (__builtin_expect(far, 0)) {
            // some hypothesis of the item is non-finite / astronomically far: exact loop (K:100-125)
            for (int ht = 0; ht < nht; ++ht) {
                const int h = (ht0 + ht) * 32 + col;
                if (h >= hn) continue;
                const float2 hp = hyp_k[h];
                int inl = 0;
                for (int p = p0 + kslice; p < p0 + npix; p += 2) {
                    const float2 c = crd[p], d = dir_k[p];
                    inl += vote_exact(c.x, c.y, hp.x, hp.y, d.x, d.y, thresh) ? 1 : 0;
                }
                if (inl) atomicAdd(&sCnt[ht * 32 + col], inl);
            }
        } else if (npix > 0) {
            // ---- A operands: lane l = row l%32 (form = row/16, pixel = row%16), k = 8*(l/32)..+7 of
            //      (v0,v0,v0,v1,v1,v2 of vx | vy0,vy0 || vy0,vy1,vy1,vy2 | cv0,cv1,cv2, 0)
            bf16x8 A[8];
            const int form = (lane >> 4) & 1, prow = lane & 15;
            auto make_A = [&](int j) -> bf16x8 {
                const float4 v = sP[(wave * kBfPixPerWave + j * 16 + prow) * 2 + form];
                __bf16 vx[3], vy[3], cv[3];
                split3(v.x, vx);
                split3(v.y, vy);
                split3(v.z, cv);
                const __bf16 zero = (__bf16)0.f;
                const bf16x8 lo8 = {vx[0], vx[0], vx[0], vx[1], vx[1], vx[2], vy[0], vy[0]};
                const bf16x8 hi8 = {vy[0], vy[1], vy[1], vy[2], cv[0], cv[1], cv[2], zero};
                return kslice ? hi8 : lo8;
            };
#pragma unroll
            for (int j = 0; j < 8; ++j) A[j] = make_A(j);
            const int ebase = kslice * 4;                        // this lane's pixels: ebase + e%4 + 8*(e/4)
            const float16v zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int ht = 0; ht < nht; ++ht) {
                const bf16x8 Bop = sB[ht * 64 + lane];
                int inl = 0;
                unsigned flagged = 0u;                            // wave-uniform: tiles with an evaluation in the band
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    unsigned q = 0u;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int j = half * 4 + jj;
                        const float16v acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[j], Bop, zero16, 0, 0, 0);
                        // conservative band test per tile: min |t|  vs  beta * max a + eps  (|t| - beta a <= eps for some
                        // evaluation implies it); the per-evaluation measure is only formed in the rare path below
                        float tmin = INFINITY, amax = 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float t = acc[e] - fabsf(acc[8 + e]);
                            q = __builtin_amdgcn_alignbit(q, __float_as_uint(t), 31);
                            tmin = fminf(tmin, fabsf(t));
                            amax = fmaxf(amax, acc[e]);
                        }
                        flagged |= __ballot(tmin <= __builtin_fmaf(fc.beta, amax, eps)) ? (1u << j) : 0u;
                    }
                    inl += 32 - __popc(q);                        // 4 tiles x 8 evaluations, sign bit set = not an inlier
                }
                while (__builtin_expect(flagged != 0u, 0)) {
                    // rare: tile j holds an evaluation inside the guard band.  Re-derive its operands, repeat the
                    // MFMA (bitwise the same result) and re-decide the flagged evaluations exactly (K:100-125).
                    const int j = __builtin_ctz(flagged);
                    flagged &= flagged - 1;
                    const float16v acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(make_A(j), Bop, zero16, 0, 0, 0);
                    const int h = (ht0 + ht) * 32 + col;
                    const float2 hp = h < hn ? hyp_k[h] : make_float2(0.f, 0.f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float t = acc[e] - fabsf(acc[8 + e]);
                        const float z = __builtin_fmaf(-fc.beta, acc[e], fabsf(t));
                        if (!__any(z <= eps)) continue;
                        const int prow = j * 16 + ebase + (e & 3) + 8 * (e >> 2);
                        const int p = p0 + prow;
                        const int fast = (__float_as_uint(t) >> 31) ? 0 : 1;
                        // second level: the sqrt/divide-free test of k_count_fast on d = fl(h - c) (the exact path's own
                        // d) with the f32 unit normal from LDS; its band (beta2, eps0) is ~10x narrower than the MFMA's
                        const float4 ra = sP[(wave * kBfPixPerWave + prow) * 2], rb = sP[(wave * kBfPixPerWave + prow) * 2 + 1];
                        const float dx = hp.x - (ra.w + org.x), dy = hp.y - (rb.w + org.y);
                        const float a2 = __builtin_fmaf(dx, ra.x, dy * ra.y);
                        const float b2 = __builtin_fmaf(dx, rb.x, dy * rb.y);
                        const float t2 = a2 - fabsf(b2);
                        int decided = t2 > 0.f ? 1 : 0;
                        const bool unsure = !(__builtin_fmaf(-fc.beta2, a2, fabsf(t2)) > fc.eps0) || ra.z <= -1e29f;
                        if (__any(unsure)) {
                            int exact = 0;
                            if (p < tn) {
                                const float2 c = crd[p], d = dir_k[p];
                                exact = vote_exact(c.x, c.y, hp.x, hp.y, d.x, d.y, thresh) ? 1 : 0;
                            }
                            if (unsure) decided = exact;
                        }
                        if (p >= tn) decided = 0;
                        inl += decided - fast;
                    }
                }
                if (inl) atomicAdd(&sCnt[ht * 32 + col], inl);    // LDS: 2 lanes x 4 waves per hypothesis
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nht * 32; i += kBlock) {
            const int h = ht0 * 32 + i;
            const int c = sCnt[i];
            if (h < hn && c != 0) atomicAdd(&counts[(size_t)bk * hn + h], c);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Stage 4 (v3): winner selection + least-squares refit (P:159-167 and P:176-196).
// One block per (keypoint, image).
// ---------------------------------------------------------------------------------------------
// kRefitSplit blocks per (keypoint, image), each over a quarter of the pixels: the exact re-vote is a long
// dependent chain (two sqrt, one divide) and 576 blocks alone leave the SIMDs latency-bound.  Partial sums go to
// sums[b,vi,split,5] and are merged in a fixed order by k_finalize_v3 (deterministic).
constexpr int kRefitSplit = 4;

__global__ __launch_bounds__(kBlock) void k_select_refit(
    const int *__restrict__ tn_arr, const float2 *__restrict__ coords,
    const float2 *__restrict__ dirs, const float2 *__restrict__ hyps,
    const int *__restrict__ counts, double *__restrict__ sums /*[B,K,kRefitSplit,5]*/,
    int *__restrict__ win_counts /*[B,K] or null*/, int K, int hn, int cap, float thresh)
{
    __shared__ int s_cnt[4], s_idx[4];
    __shared__ double red5[20];
    const int vi = blockIdx.x / kRefitSplit, split = blockIdx.x % kRefitSplit, b = blockIdx.y;
    const int bk = b * K + vi;
    const int tn = tn_arr[b];
    double *part = sums + ((size_t)bk * kRefitSplit + split) * 5;
    if (tn <= 0) {
        if (threadIdx.x == 0) {
            for (int i = 0; i < 5; ++i) part[i] = 0.0;
            if (win_counts && split == 0) win_counts[bk] = 0;
        }
        return;
    }
    // torch.max(counts, 0): maximal count, FIRST index among ties (P:160)
    const int *cp = counts + (size_t)bk * hn;
    int best = -1, besti = 0x7fffffff;
    for (int h = threadIdx.x; h < hn; h += kBlock) {
        int c = cp[h];
        if (c > best) { best = c; besti = h; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        int oc = __shfl_xor(best, o, 64), oi = __shfl_xor(besti, o, 64);
        if (oc > best || (oc == best && oi < besti)) { best = oc; besti = oi; }
    }
    if (lane_id() == 0) { s_cnt[threadIdx.x >> 6] = best; s_idx[threadIdx.x >> 6] = besti; }
    __syncthreads();
    best = s_cnt[0]; besti = s_idx[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
        if (s_cnt[w] > best || (s_cnt[w] == best && s_idx[w] < besti)) { best = s_cnt[w]; besti = s_idx[w]; }

    // P:162-167: all_win_ratio (0) < count/tn  <=>  count > 0; otherwise the winner stays (0,0)
    float2 win = make_float2(0.f, 0.f);
    if (best > 0) win = hyps[(size_t)bk * hn + besti];

    // P:176-191: re-vote the winner (hn = 1) and accumulate the normal equations in binary64
    const float2 *dp = dirs + (size_t)bk * cap;
    const float2 *cq = coords + (size_t)b * cap;
    double xx = 0, xy = 0, yy = 0, bx = 0, by = 0;
    // this block's quarter of the pixels; four pixels per trip so that the loads of a trip overlap
    const int per = (tn + kRefitSplit - 1) / kRefitSplit;
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
            if (!vote_exact(c[u].x, c[u].y, win.x, win.y, d[u].x, d[u].y, thresh)) continue;
            double nx = (double)d[u].y, ny = -(double)d[u].x;          // P:178-179
            double bb = nx * (double)c[u].x + ny * (double)c[u].y;     // P:189
            xx += nx * nx; xy += nx * ny; yy += ny * ny;               // P:190
            bx += nx * bb; by += ny * bb;                              // P:191
        }
    }
    // one reduction for all five sums: five independent shuffle chains interleave, a single barrier
    double v[5] = {xx, xy, yy, bx, by};
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int i = 0; i < 5; ++i) v[i] += __shfl_xor(v[i], o, 64);
    }
    if (lane_id() == 0) {
#pragma unroll
        for (int i = 0; i < 5; ++i) red5[(threadIdx.x >> 6) * 5 + i] = v[i];
    }
    __syncthreads();
    if (threadIdx.x < 5) part[threadIdx.x] = red5[threadIdx.x] + red5[5 + threadIdx.x] + red5[10 + threadIdx.x] + red5[15 + threadIdx.x];
    if (threadIdx.x == 0 && win_counts && split == 0) win_counts[bk] = best;
}

// Merge the partial normal equations, solve the 2x2 systems (P:193, closed form in binary64) and apply the
// singular-matrix policy across the keypoints of an image (b_inv, P:97-109).  One block per image.
__global__ __launch_bounds__(64) void k_finalize_v3(const int *__restrict__ tn_arr, const double *__restrict__ sums,
                                                    float2 *__restrict__ out, int K, int policy)
{
    __shared__ int any_singular;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) any_singular = 0;
    __syncthreads();
    const bool skipped = tn_arr[b] <= 0;
    for (int v0 = 0; v0 < K; v0 += 64) {          // K <= 64 in every real use: one trip
        const int vi = v0 + threadIdx.x;
        float2 o = make_float2(0.f, 0.f);
        double bx = 0, by = 0;
        bool sing = false;
        if (vi < K && !skipped) {
            const double *q = sums + ((size_t)b * K + vi) * kRefitSplit * 5;
            double xx = 0, xy = 0, yy = 0;
            for (int sp = 0; sp < kRefitSplit; ++sp) {
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
            const double *q = sums + ((size_t)b * K + vi) * kRefitSplit * 5;
            double xx = 0, xy = 0, yy = 0, bx = 0, by = 0;
            for (int sp = 0; sp < kRefitSplit; ++sp) {
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

// ---------------------------------------------------------------------------------------------
// Stage 4 (estimate): ratio threshold + weighted covariance about `mean` (P:244, P:262-269).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_covariance(
    const int *__restrict__ tn_arr, const float2 *__restrict__ hyps, const int *__restrict__ counts,
    const float2 *__restrict__ mean, float *__restrict__ cov /*[B,K,2,2]*/,
    float2 *__restrict__ hyp_out /*[B,K,hn] or null*/, int *__restrict__ counts_out,
    float *__restrict__ weights /*[B,K,3] or null*/, int K, int hn)
{
    __shared__ int redi[4];
    __shared__ double redd[4];
    const int vi = blockIdx.x, b = blockIdx.y;
    const int bk = b * K + vi;
    const int tn = tn_arr[b];
    const float2 m = mean[bk];
    const float2 *hp = hyps + (size_t)bk * hn;
    const int *cp = counts + (size_t)bk * hn;
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
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
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

// ---------------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------------
size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

struct Layout {
    int T;
    size_t tile_nz, tile_sum, bits, tn, coords, dirs, recs, hyps, counts, sums, total;
};

Layout make_layout(const pvv_problem *p)
{
    Layout L;
    const size_t HW = (size_t)p->H * p->W;
    L.T = (int)((HW + kTile - 1) / kTile);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes); return o; };
    L.tile_nz = take(sizeof(int) * (size_t)p->B * L.T);
    L.tile_sum = take(sizeof(int) * (size_t)p->B * L.T);
    L.bits = take(sizeof(unsigned long long) * (size_t)p->B * L.T * kTileSteps * 4);
    L.tn = take(sizeof(int) * (size_t)p->B);
    L.coords = take(sizeof(float2) * (size_t)p->B * p->cap);
    L.dirs = take(sizeof(float2) * (size_t)p->B * p->K * p->cap);
    L.recs = take(sizeof(PixelRec) * (size_t)p->B * p->K * p->cap);
    L.hyps = take(sizeof(float2) * (size_t)p->B * p->K * p->hn);
    L.counts = take(sizeof(int) * (size_t)p->B * p->K * p->hn);
    L.sums = take(sizeof(double) * (size_t)p->B * p->K * kRefitSplit * 5);
    L.total = off;
    return L;
}

int validate(const pvv_problem *p)
{
    if (!p) return fail(PVV_E_ARG, "problem is NULL");
    if (p->B <= 0 || p->H <= 0 || p->W <= 0 || p->K <= 0 || p->hn <= 0)
        return fail(PVV_E_ARG, "B, H, W, K, hn must be positive");
    if (p->B > kMaxBatchLds) return fail(PVV_E_ARG, "B > 1024: split the batch");
    if ((long long)p->H * p->W >= (1ll << 31)) return fail(PVV_E_ARG, "H*W must be < 2^31");
    if (p->mask_elem_size != 1 && p->mask_elem_size != 2 && p->mask_elem_size != 4 &&
        p->mask_elem_size != 8)
        return fail(PVV_E_ARG, "mask_elem_size must be 1, 2, 4 or 8");
    if (p->cap <= 0 || (long long)p->cap > (long long)p->H * p->W)
        return fail(PVV_E_ARG, "cap must be in [1, H*W]");
    if ((long long)p->B * p->K * p->hn >= (1ll << 31) || (long long)p->B * p->K * p->cap >= (1ll << 40))
        return fail(PVV_E_ARG, "problem too large for one call");
    if (p->singular_policy < PVV_SINGULAR_REFERENCE || p->singular_policy > PVV_SINGULAR_IMAGE_ZERO)
        return fail(PVV_E_ARG, "unknown singular_policy");
    return PVV_OK;
}

int num_cus()
{
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

int launch_count(const CountArgs &a, hipStream_t st)
{
    const int grid = num_cus() * 8;
    if (a.hn <= 64)
        hipLaunchKernelGGL(k_count_inliers<1>, dim3(grid), dim3(kBlock), 0, st, a);
    else if (a.hn <= 128)
        hipLaunchKernelGGL(k_count_inliers<2>, dim3(grid), dim3(kBlock), 0, st, a);
    else if (a.hn <= 256)
        hipLaunchKernelGGL(k_count_inliers<4>, dim3(grid), dim3(kBlock), 0, st, a);
    else
        hipLaunchKernelGGL(k_count_inliers<8>, dim3(grid), dim3(kBlock), 0, st, a);
    return check_launch("k_count_inliers");
}

// The fast test needs 0 < T < 1 with a sane kappa; outside [0.5, 0.99995] (and when PVV_COUNT_KERNEL=exact
// is set, for A/B runs) the exact kernel is used.
// PVV_COUNT_KERNEL = exact | fast | bf16 selects the inlier-count kernel (A/B runs; the tests exercise all three).
int count_kernel_choice()
{
    static int choice = -1;   // 0 exact (sqrt/divide), 1 fast (packed VALU), 2 bf16 (matrix-core prefilter)
    if (choice < 0) {
        const char *e = getenv("PVV_COUNT_KERNEL");
        choice = 2;
        if (e && !strcmp(e, "exact")) choice = 0;
        else if (e && !strcmp(e, "fast")) choice = 1;
    }
    return choice;
}

bool use_fast_count(float thresh)
{
    return count_kernel_choice() != 0 && thresh >= 0.5f && thresh <= 0.99995f;
}

bool use_bf16_count(const pvv_problem *p)
{
    // block extents enter the guard band; keep pixel coordinates well inside f32/bf16-split integer range
    return count_kernel_choice() == 2 && use_fast_count(p->inlier_thresh) && p->H <= 16384 && p->W <= 16384;
}

Bf16Consts bf16_consts(float thresh)
{
    const double T = (double)thresh, s2 = 1.0 - T * T, kappa = T / std::sqrt(s2);
    const double u = 0x1p-24;
    Bf16Consts fc;
    fc.beta = (float)(1.25 * (30.0 * (1.0 + kappa) + 8.0 / s2) * u / T);
    fc.eps_c = (float)(1.25 * (1.0 + kappa) * 34.0 * u);
    fc.eps0 = (float)(1.5e-6 * (1.0 + kappa));
    fc.kappa = (float)kappa;
    // second level: d exact-path's own, nh/B computed in f32 (<= 2u / 3u relative), one fma each => 6u(1+kappa)|d|
    fc.beta2 = (float)(1.25 * (6.0 * (1.0 + kappa) + 8.0 / s2) * u / T);
    // PVV_DEBUG_BAND_SCALE (timing experiments only; != 1 voids the exactness guarantee): scales the guard band
    static const char *dbg = getenv("PVV_DEBUG_BAND_SCALE");
    if (dbg && *dbg) {
        const float k = (float)atof(dbg);
        fc.beta *= k; fc.eps_c *= k; fc.eps0 *= k;
    }
    return fc;
}

double fast_kappa_value(float thresh)
{
    const double T = (double)thresh;
    return T / std::sqrt(1.0 - T * T);
}

FastConsts fast_consts(float thresh)
{
    const double T = (double)thresh, s2 = 1.0 - T * T, kappa = T / std::sqrt(s2);
    const double u = 0x1p-24;
    FastConsts fc;
    fc.beta = (float)(1.25 * (3.0 * (1.0 + kappa) + 8.0 / s2) * u / T);
    fc.eps_abs = (float)(1.5e-6 * (1.0 + kappa));
    return fc;
}

// Persistent grid = exactly the blocks that are co-resident (VGPR-limited), so that no block starts late
// and drags a tail behind the others.
template <typename Kern>
int resident_blocks(Kern kern)
{
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, kBlock, 0) != hipSuccess || per_cu <= 0)
        per_cu = 4;
    if (per_cu > 8) per_cu = 8;
    return per_cu * num_cus();
}

int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e && *e ? atoi(e) : dflt;
}

int launch_count_fast(const pvv_problem *p, const Layout &L, char *ws, hipStream_t st)
{
    // tuning knobs (defaults chosen from sweeps on MI355X)
    static const int per_cu = env_int("PVV_GRID_PER_CU", 24);
    static const int ppw = env_int("PVV_PIX_PER_WAVE", 128);
    static const int items_per_cu = env_int("PVV_ITEMS_PER_CU", 6);
    const int grid2 = per_cu * num_cus(), grid4 = grid2, grid8 = grid2;
    const float8v *recs = (const float8v *)(ws + L.recs);
    const float2 *hyps = (const float2 *)(ws + L.hyps);
    int *counts = (int *)(ws + L.counts);
    const int *tn = (const int *)(ws + L.tn);
    const FastConsts fc = fast_consts(p->inlier_thresh);
#define PVV_LAUNCH_FAST(R)                                                                                   \
    hipLaunchKernelGGL(k_count_fast<R>, dim3(grid##R), dim3(kBlock), 0, st, recs, hyps, counts, tn, p->B, \
                       p->K, p->hn, p->cap, p->inlier_thresh, fc, ppw, items_per_cu * num_cus())
    if (p->hn <= 128) PVV_LAUNCH_FAST(2);
    else if (p->hn <= 256) PVV_LAUNCH_FAST(4);
    else PVV_LAUNCH_FAST(8);
#undef PVV_LAUNCH_FAST
    return check_launch("k_count_fast");
}

int launch_count_bf16(const pvv_problem *p, const Layout &L, char *ws, hipStream_t st)
{
    static const int per_cu = env_int("PVV_GRID_PER_CU", 24);
    static const int items_per_cu = env_int("PVV_ITEMS_PER_CU", 6);
    hipLaunchKernelGGL(k_count_bf16, dim3(per_cu * num_cus()), dim3(kBlock), 0, st, (const float2 *)(ws + L.coords),
                       (const float2 *)(ws + L.dirs), (const float2 *)(ws + L.hyps), (int *)(ws + L.counts),
                       (const int *)(ws + L.tn), p->B, p->K, p->hn, p->cap, p->inlier_thresh,
                       bf16_consts(p->inlier_thresh), items_per_cu * num_cus());
    return check_launch("k_count_bf16");
}

int launch_count_any(const pvv_problem *p, const Layout &L, char *ws, hipStream_t st);

CountArgs planar_count_args(const pvv_problem *p, const Layout &L, char *ws)
{
    CountArgs a;
    a.coords = (const float2 *)(ws + L.coords);
    a.dirs = (const float2 *)(ws + L.dirs);
    a.hyps = (const float2 *)(ws + L.hyps);
    a.counts = (int *)(ws + L.counts);
    a.tn_arr = (const int *)(ws + L.tn);
    a.c_b = p->cap;
    a.d_b = (long long)p->K * p->cap; a.d_v = p->cap; a.d_p = 1;
    a.h_b = (long long)p->K * p->hn; a.h_v = p->hn; a.h_h = 1;
    a.tn_fixed = 0;
    a.B = p->B; a.K = p->K; a.hn = p->hn;
    a.thresh = p->inlier_thresh;
    return a;
}

int launch_count_any(const pvv_problem *p, const Layout &L, char *ws, hipStream_t st)
{
    if (use_bf16_count(p)) return launch_count_bf16(p, L, ws, st);
    if (use_fast_count(p->inlier_thresh)) return launch_count_fast(p, L, ws, st);
    return launch_count(planar_count_args(p, L, ws), st);
}

template <int ES>
int launch_compaction(const MaskArgs &m, const VertexArgs &v, const Layout &L, char *ws, int B,
                      hipStream_t st)
{
    dim3 grid(L.T, B), block(kBlock);
    int *tile_nz = (int *)(ws + L.tile_nz), *tile_sum = (int *)(ws + L.tile_sum);
    unsigned long long *bits = (unsigned long long *)(ws + L.bits);
    hipLaunchKernelGGL(k_tile_count<ES>, grid, block, 0, st, m, tile_nz, tile_sum, bits);
    if (int e = check_launch("k_tile_count")) return e;
    hipLaunchKernelGGL(k_tile_subsample, grid, block, 0, st, m, tile_nz, (const int *)tile_sum, bits);
    if (int e = check_launch("k_tile_subsample")) return e;
    hipLaunchKernelGGL(k_compact, grid, block, 0, st, m, v, (const int *)tile_nz, (const int *)tile_sum,
                       (const unsigned long long *)bits, (int *)(ws + L.tn), (float2 *)(ws + L.coords),
                       (float2 *)(ws + L.dirs), (PixelRec *)(ws + L.recs));
    return check_launch("k_compact");
}

// compaction + hypotheses + counting, shared by both layers
int run_front(const pvv_problem *p, int mode, const void *d_mask, const float *d_vertex,
              const int32_t *d_idxs, const float *d_selection, char *ws, const Layout &L,
              hipStream_t st, const float *d_seg = nullptr, int64_t *d_mask_out = nullptr)
{
    MaskArgs m;
    m.mask = d_mask;
    m.seg = d_seg;
    m.mask_out = (long long *)d_mask_out;
    m.gb = p->seg_stride[0]; m.gc = p->seg_stride[1]; m.gh = p->seg_stride[2]; m.gw = p->seg_stride[3];
    m.C = p->seg_classes;
    m.selection = d_selection;
    m.sb = p->mask_stride[0]; m.sh = p->mask_stride[1]; m.sw = p->mask_stride[2];
    m.es = p->mask_elem_size;
    m.contig = (m.sw == 1 && m.sh == p->W) ? 1 : 0;
    m.mode = mode;
    m.W = p->W; m.HW = p->H * p->W; m.T = L.T;
    m.min_num = p->min_num; m.max_num = p->max_num; m.cap = p->cap;
    m.seed = p->seed;
    VertexArgs v;
    v.vertex = d_vertex;
    v.sb = p->vertex_stride[0]; v.sh = p->vertex_stride[1]; v.sw = p->vertex_stride[2];
    v.sk = p->vertex_stride[3]; v.sc = p->vertex_stride[4];
    v.K = p->K;
    v.kappa = use_bf16_count(p) ? 0.0 : (use_fast_count(p->inlier_thresh) ? fast_kappa_value(p->inlier_thresh) : 0.0);
    v.vec2 = (v.sc == 1 && !(v.sb & 1) && !(v.sh & 1) && !(v.sw & 1) && !(v.sk & 1) &&
              ((uintptr_t)d_vertex % 8 == 0)) ? 1 : 0;
    int e;
    switch (m.es) {
    case 1: e = launch_compaction<1>(m, v, L, ws, p->B, st); break;
    case 2: e = launch_compaction<2>(m, v, L, ws, p->B, st); break;
    case 4: e = launch_compaction<4>(m, v, L, ws, p->B, st); break;
    default: e = launch_compaction<8>(m, v, L, ws, p->B, st); break;
    }
    if (e) return e;

    const long long nh = (long long)p->B * p->K * p->hn;
    hipLaunchKernelGGL(k_gen_hypothesis, dim3((unsigned)((nh + kBlock - 1) / kBlock)), dim3(kBlock),
                       0, st, d_idxs, (const int *)(ws + L.tn), (const float2 *)(ws + L.coords),
                       (const float2 *)(ws + L.dirs), (float2 *)(ws + L.hyps), (int *)(ws + L.counts),
                       p->B, p->K, p->hn, p->cap, p->seed);
    if ((e = check_launch("k_gen_hypothesis"))) return e;
    return launch_count_any(p, L, ws, st);
}

int check_ptrs(const pvv_problem *p, const void *mask, const void *vertex, void *ws, size_t ws_bytes,
               Layout *L)
{
    if (int e = validate(p)) return e;
    if (!mask || !vertex || !ws) return fail(PVV_E_ARG, "mask, vertex and workspace must be non-NULL");
    *L = make_layout(p);
    if (ws_bytes < L->total) return fail(PVV_E_WORKSPACE, "workspace too small");
    return PVV_OK;
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
PVV_EXPORT int pvv_abi_version(void) { return PVV_ABI_VERSION; }
PVV_EXPORT const char *pvv_last_error(void) { return g_err; }

PVV_EXPORT int32_t pvv_default_cap(int32_t H, int32_t W, int32_t max_num)
{
    long long hw = (long long)H * W;
    if (max_num < 0) max_num = 0;
    long long cap = (long long)max_num + (long long)(8.0 * std::sqrt((double)max_num)) + 64;
    return (int32_t)(cap < hw ? cap : hw);
}

PVV_EXPORT size_t pvv_workspace_bytes(const pvv_problem *p)
{
    if (validate(p)) return 0;
    return make_layout(p).total;
}

static int finish_v3(const pvv_problem *p, const Layout &L, char *ws, float *d_out, int32_t *d_win_counts,
                     int32_t *d_tn, hipStream_t st)
{
    hipLaunchKernelGGL(k_select_refit, dim3(p->K * kRefitSplit, p->B), dim3(kBlock), 0, st,
                       (const int *)(ws + L.tn), (const float2 *)(ws + L.coords),
                       (const float2 *)(ws + L.dirs), (const float2 *)(ws + L.hyps),
                       (const int *)(ws + L.counts), (double *)(ws + L.sums), d_win_counts, p->K, p->hn,
                       p->cap, p->inlier_thresh);
    if (int e = check_launch("k_select_refit")) return e;
    hipLaunchKernelGGL(k_finalize_v3, dim3(p->B), dim3(64), 0, st, (const int *)(ws + L.tn),
                       (const double *)(ws + L.sums), (float2 *)d_out, p->K, p->singular_policy);
    if (int e = check_launch("k_finalize_v3")) return e;
    if (d_tn) {
        hipError_t e = hipMemcpyAsync(d_tn, ws + L.tn, sizeof(int) * p->B, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    }
    return PVV_OK;
}

PVV_EXPORT int pvv_ransac_voting_v3(const pvv_problem *p, const void *d_mask, const float *d_vertex,
                                    const int32_t *d_idxs, const float *d_selection,
                                    void *d_workspace, size_t workspace_bytes, float *d_out,
                                    int32_t *d_win_counts, int32_t *d_tn, void *stream)
{
    Layout L;
    if (int e = check_ptrs(p, d_mask, d_vertex, d_workspace, workspace_bytes, &L)) return e;
    if (!d_out) return fail(PVV_E_ARG, "d_out is NULL");
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)d_workspace;
    if (int e = run_front(p, 0, d_mask, d_vertex, d_idxs, d_selection, ws, L, st)) return e;
    return finish_v3(p, L, ws, d_out, d_win_counts, d_tn, st);
}

PVV_EXPORT int pvv_decode_keypoint_v3(const pvv_problem *p, const float *d_seg, const float *d_vertex,
                                      const int32_t *d_idxs, const float *d_selection, void *d_workspace,
                                      size_t workspace_bytes, int64_t *d_mask_out, float *d_out,
                                      int32_t *d_win_counts, int32_t *d_tn, void *stream)
{
    Layout L;
    if (p && p->mask_elem_size == 0) return fail(PVV_E_ARG, "set mask_elem_size = 8 (the int64 mask this call emits)");
    if (int e = check_ptrs(p, d_seg, d_vertex, d_workspace, workspace_bytes, &L)) return e;
    if (!d_out) return fail(PVV_E_ARG, "d_out is NULL");
    if (p->seg_classes < 1 || p->seg_classes > 256) return fail(PVV_E_ARG, "seg_classes must be in [1, 256]");
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)d_workspace;
    if (int e = run_front(p, 0, nullptr, d_vertex, d_idxs, d_selection, ws, L, st, d_seg, d_mask_out)) return e;
    return finish_v3(p, L, ws, d_out, d_win_counts, d_tn, st);
}

PVV_EXPORT int pvv_estimate_voting_distribution(const pvv_problem *p, const void *d_mask,
                                                const float *d_vertex, const int32_t *d_idxs,
                                                const float *d_selection, const float *d_mean,
                                                void *d_workspace, size_t workspace_bytes,
                                                float *d_cov, float *d_hyp, int32_t *d_counts,
                                                int32_t *d_tn, float *d_weights, void *stream)
{
    Layout L;
    if (int e = check_ptrs(p, d_mask, d_vertex, d_workspace, workspace_bytes, &L)) return e;
    if (!d_mean || !d_cov) return fail(PVV_E_ARG, "d_mean / d_cov is NULL");
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)d_workspace;
    if (int e = run_front(p, 1, d_mask, d_vertex, d_idxs, d_selection, ws, L, st)) return e;
    hipLaunchKernelGGL(k_covariance, dim3(p->K, p->B), dim3(kBlock), 0, st,
                       (const int *)(ws + L.tn), (const float2 *)(ws + L.hyps),
                       (const int *)(ws + L.counts), (const float2 *)d_mean, d_cov, (float2 *)d_hyp,
                       d_counts, d_weights, p->K, p->hn);
    if (int e = check_launch("k_covariance")) return e;
    if (d_tn) {
        hipError_t e = hipMemcpyAsync(d_tn, ws + L.tn, sizeof(int) * p->B, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    }
    return PVV_OK;
}

PVV_EXPORT int pvv_rerun_count_kernel(const pvv_problem *p, void *d_workspace, size_t workspace_bytes,
                                      int zero_counts, void *stream)
{
    if (int e = validate(p)) return e;
    if (!d_workspace) return fail(PVV_E_ARG, "workspace is NULL");
    Layout L = make_layout(p);
    if (workspace_bytes < L.total) return fail(PVV_E_WORKSPACE, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)d_workspace;
    if (zero_counts) {
        hipError_t e = hipMemsetAsync(ws + L.counts, 0, sizeof(int) * (size_t)p->B * p->K * p->hn, st);
        if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    }
    return launch_count_any(p, L, ws, st);
}

// ---- legacy module surface ---------------------------------------------------------------
static int check_legacy(const void *a, const void *b, const void *c, const void *d, int tn, int vn,
                        int hn)
{
    if (!a || !b || !c || !d) return fail(PVV_E_ARG, "NULL device pointer");
    if (tn < 0 || vn <= 0 || hn < 0) return fail(PVV_E_ARG, "tn, hn must be >= 0 and vn > 0");
    if ((long long)hn * vn * (long long)(tn > 0 ? tn : 1) >= (1ll << 40))
        return fail(PVV_E_ARG, "problem too large");
    return PVV_OK;
}

PVV_EXPORT int pvv_generate_hypothesis(const float *d_direct, const float *d_coords,
                                       const int32_t *d_idxs, float *d_hypo_pts, int tn, int vn,
                                       int hn, void *stream)
{
    if (int e = check_legacy(d_direct, d_coords, d_idxs, d_hypo_pts, tn, vn, hn)) return e;
    if (hn == 0) return PVV_OK;
    hipLaunchKernelGGL(k_legacy_gen, dim3((hn * vn + kBlock - 1) / kBlock), dim3(kBlock), 0,
                       (hipStream_t)stream, d_direct, d_coords, d_idxs, d_hypo_pts, tn, vn, hn);
    return check_launch("k_legacy_gen");
}

PVV_EXPORT int pvv_generate_hypothesis_vanishing_point(const float *d_direct, const float *d_coords,
                                                       const int32_t *d_idxs, float *d_hypo_pts,
                                                       int tn, int vn, int hn, void *stream)
{
    if (int e = check_legacy(d_direct, d_coords, d_idxs, d_hypo_pts, tn, vn, hn)) return e;
    if (hn == 0) return PVV_OK;
    hipLaunchKernelGGL(k_legacy_gen_vp, dim3((hn * vn + kBlock - 1) / kBlock), dim3(kBlock), 0,
                       (hipStream_t)stream, d_direct, d_coords, d_idxs, d_hypo_pts, tn, vn, hn);
    return check_launch("k_legacy_gen_vp");
}

static void legacy_vote_shape(int tn, int vn, int hn, dim3 *grid, int *h_per_block)
{
    // enough blocks to fill 256 CUs, few enough that each thread amortises its pixel load
    int tiles = (tn + kBlock - 1) / kBlock;
    int hz = 1;
    while ((long long)tiles * vn * hz < 4096 && hz < hn) hz <<= 1;
    if (hz > hn) hz = hn;
    if (hz > 65535) hz = 65535;
    *h_per_block = (hn + hz - 1) / hz;
    *grid = dim3(tiles, vn, (hn + *h_per_block - 1) / *h_per_block);
}

PVV_EXPORT int pvv_voting_for_hypothesis(const float *d_direct, const float *d_coords,
                                         const float *d_hypo_pts, uint8_t *d_inliers, int tn, int vn,
                                         int hn, float inlier_thresh, void *stream)
{
    if (int e = check_legacy(d_direct, d_coords, d_hypo_pts, d_inliers, tn, vn, hn)) return e;
    if (vn > 65535) return fail(PVV_E_ARG, "vn > 65535");
    if (hn == 0 || tn == 0) return PVV_OK;
    dim3 grid; int hpb;
    legacy_vote_shape(tn, vn, hn, &grid, &hpb);
    hipLaunchKernelGGL(k_legacy_vote, grid, dim3(kBlock), 0, (hipStream_t)stream, d_direct, d_coords,
                       d_hypo_pts, d_inliers, tn, vn, hn, hpb, inlier_thresh);
    return check_launch("k_legacy_vote");
}

PVV_EXPORT int pvv_voting_for_hypothesis_vanishing_point(const float *d_direct, const float *d_coords,
                                                         const float *d_hypo_pts, uint8_t *d_inliers,
                                                         int tn, int vn, int hn, float inlier_thresh,
                                                         void *stream)
{
    if (int e = check_legacy(d_direct, d_coords, d_hypo_pts, d_inliers, tn, vn, hn)) return e;
    if (vn > 65535) return fail(PVV_E_ARG, "vn > 65535");
    if (hn == 0 || tn == 0) return PVV_OK;
    dim3 grid; int hpb;
    legacy_vote_shape(tn, vn, hn, &grid, &hpb);
    hipLaunchKernelGGL(k_legacy_vote_vp, grid, dim3(kBlock), 0, (hipStream_t)stream, d_direct,
                       d_coords, d_hypo_pts, d_inliers, tn, vn, hn, hpb, inlier_thresh);
    return check_launch("k_legacy_vote_vp");
}

PVV_EXPORT int pvv_count_inliers(const float *d_direct, const float *d_coords, const float *d_hypo_pts,
                                 int32_t *d_counts, int tn, int vn, int hn, float inlier_thresh,
                                 void *stream)
{
    if (int e = check_legacy(d_direct, d_coords, d_hypo_pts, d_counts, tn, vn, hn)) return e;
    if (hn == 0) return PVV_OK;
    hipStream_t st = (hipStream_t)stream;
    hipError_t me = hipMemsetAsync(d_counts, 0, sizeof(int) * (size_t)hn * vn, st);
    if (me != hipSuccess) return fail((int)me, hipGetErrorString(me));
    if (tn == 0) return PVV_OK;
    CountArgs a;
    a.coords = (const float2 *)d_coords;
    a.dirs = (const float2 *)d_direct;
    a.hyps = (const float2 *)d_hypo_pts;
    a.counts = d_counts;
    a.tn_arr = nullptr;
    a.c_b = 0;
    a.d_b = 0; a.d_v = 1; a.d_p = vn;      // direct[ti,vi]
    a.h_b = 0; a.h_v = 1; a.h_h = vn;      // hypo[hi,vi], counts[hi,vi]
    a.tn_fixed = tn;
    a.B = 1; a.K = vn; a.hn = hn;
    a.thresh = inlier_thresh;
    return launch_count(a, st);
}
