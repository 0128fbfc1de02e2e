// compaction.hpp -- stages 1 and 2: mask scan, subsample, ordered compaction, hypothesis generation.
// Part of the single translation unit pvnet_vote.hip (included inside its anonymous namespace); see that file
// for the numerical contract and the reference citations (K = ransac_voting_kernel.cu, P = ransac_voting_gpu.py).
#pragma once

// ---------------------------------------------------------------------------------------------
// Stage 1: foreground compaction (replaces sum / nonzero / masked_select / uniform_ of
// P:125-144 and P:207-229), order = row-major order of torch.nonzero.
//
// The image is cut into tiles of kTile = 2048 consecutive pixels.  k_tile_scan -- the ONLY pass over the mask --
// leaves per tile (i) one packed word: foreground pixels | weight sum << 12 (foreground_num of P:126 sums byte
// VALUES), (ii) the tile's foreground pixels as an ordered list of 16-bit offsets and -- when subsampling is possible
// -- (iii) each listed pixel's U(0,1) draw of P:136 / P:220 (injected or from the counter RNG keyed by (image, pixel)),
// so that no later stage evaluates the generator or touches the selection tensor again.  Everything downstream works from
// those lists: the subsample (k_tile_subsample, or fused into k_compact_hyp) filters them, the compaction blocks of
// k_compact_hyp gather the vertex field through them, and its hypothesis blocks pick the t-th foreground pixel of an
// image by a search in the tile prefix + ONE list read -- so hypotheses need not wait for the compaction and the
// separate hypothesis launch of round 1 is gone.
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
    int b0;                  // first_image: RNG key offset of image 0
    int *tn_user;            // the caller's tn[B] (or nullptr): written beside the workspace copy, no D2D copy later
    int *status;             // the caller's status[B] (or nullptr): PVV_STATUS_* bits, written with tn
    int fuse_sub;            // 1: k_compact_hyp applies the subsampling itself (no k_tile_subsample launch), see there
    int want_draws;          // 1: k_tile_scan stores every foreground pixel's U(0,1) draw beside its list entry -- when
                             //    subsampling is possible at all (max_num below the largest foreground_num the mask can have)
                             //    AND has its own pass (k_tile_subsample, !fuse_sub: subsampling is likely).  With fused
                             //    subsampling (unlikely by the host's rule) the few consumers evaluate the draw of a listed
                             //    pixel on demand (list_draw): the scan of a batch that never subsamples -- the benchmark --
                             //    then neither runs the generator nor writes 4 bytes per foreground pixel (-1.3 % per call)
    // fused argmax (decode_keypoint): when seg != nullptr the mask value is argmax_c seg[b,c,y,x]
    const float *seg;
    long long *mask_out;     // [B,H,W] int64 or nullptr
    int64_t gb, gc, gh, gw;  // element strides of seg
    int C;
};

constexpr uint32_t kTileNzMask = 0xfffu;   // tiles[] word: foreground pixels (0..2048) | weight sum (<= 2048*255) << 12

template <int ES>
__device__ __forceinline__ uint64_t load_elem(const void *base, int64_t off)
{
    if (ES == 1) return ((const uint8_t *)base)[off];
    if (ES == 2) return ((const uint16_t *)base)[off];
    if (ES == 4) return ((const uint32_t *)base)[off];
    return ((const uint64_t *)base)[off];
}

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

// weight of pixel p of image b: 0 = background; v3: low byte (P:125-126 sums the bytes), estimate: 1 (P:207-208).
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
    return (float)(rng_u32(a.seed, 0u, (uint32_t)(a.b0 + b), (uint32_t)p) >> 8) * 0x1p-24f;
}

// Draw of entry e of tile i's list: stored by the scan (draws != nullptr) or evaluated on demand from the (image, pixel)
// key -- the same number either way.
__device__ __forceinline__ float list_draw(const MaskArgs &a, int b, int i, int e, const unsigned short *__restrict__ lists /*of image b*/,
                                           const float *__restrict__ draws /*of image b, or nullptr*/)
{
    if (draws) return draws[(size_t)i * kTile + e];
    return selection_draw(a, b, i * kTile + (int)lists[(size_t)i * kTile + e]);
}

// Exclusive scan of the 32 (step, wave) segment counts of a tile by wave 0; seg[32] = the tile's total.
// Call with all threads; contains the barriers.
__device__ __forceinline__ void scan_segments(int *seg)
{
    __syncthreads();
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const int c = lane < kTileSteps * 4 ? seg[lane] : 0;
        const int inc = wave_incl_scan(c);
        if (lane < kTileSteps * 4) seg[lane] = inc - c;
        if (lane == kTileSteps * 4 - 1) seg[kTileSteps * 4] = inc;
    }
    __syncthreads();
}

// Pass 1 -- the ONLY pass that reads the mask.  Persistent over the B*T tiles (grid = the resident blocks, tile
// g = blockIdx.x + i * gridDim.x -> image g / T, tile g % T): a contiguous mask is read one tile AHEAD -- the eight loads
// of a thread's next tile are in flight while the ballots, the segment scan and the list stores of the current one run --
// so the kernel streams instead of paying a block launch and a cold load latency per 16 KB (round 2: 9600 short-lived
// blocks at B = 64, 4.0 TB/s on rotating batches).  A strided mask or the fused argmax take the same loop without the
// read-ahead.
template <int ES> struct RawElem;
template <> struct RawElem<1> { typedef uint8_t type; };
template <> struct RawElem<2> { typedef uint16_t type; };
template <> struct RawElem<4> { typedef uint32_t type; };
template <> struct RawElem<8> { typedef uint64_t type; };

template <int ES, bool AHEAD, int MODE /*MaskArgs.mode, compile-time here*/>
__global__ __launch_bounds__(kBlock) void k_tile_scan(MaskArgs a, uint32_t *__restrict__ tiles,
                                                      unsigned short *__restrict__ tile_list,
                                                      float *__restrict__ tile_draw, int total_tiles)
{
    typedef typename RawElem<ES>::type raw_t;
    __shared__ int seg2[2][kTileSteps * 4 + 1];                    // double-buffered by the parity of the iteration: the
    __shared__ int red2[2][4];                                     // next tile's counts are written while stragglers still read
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    // AHEAD (host: a.contig && !a.seg): the read-ahead instantiation.  The kernel issues about as many instructions per tile
    // as the memory system needs cycles to deliver it (round 3: ~400 per wave and tile against 16 KB per block), so the
    // tile walk avoids what it can: (image, tile) advance by a constant step instead of two divisions per tile, and a tile
    // that lies inside the image -- all but possibly the last one -- is loaded without per-element bounds checks.
    raw_t cur[AHEAD ? kTileSteps : 1] = {};
    int g = blockIdx.x;
    int b = g / a.T, t = g - b * a.T;
    const int step_b = (int)gridDim.x / a.T, step_t = (int)gridDim.x - step_b * a.T;
    [[maybe_unused]] auto load_tile = [&](raw_t (&dst)[AHEAD ? kTileSteps : 1], int bi, int ti) {
        const raw_t *src = (const raw_t *)a.mask + (int64_t)bi * a.sb + (int64_t)ti * kTile + threadIdx.x;
        if ((ti + 1) * kTile <= a.HW) {
#pragma unroll
            for (int s = 0; s < kTileSteps; ++s) dst[AHEAD ? s : 0] = src[s * kBlock];
        } else {
#pragma unroll
            for (int s = 0; s < kTileSteps; ++s)
                dst[AHEAD ? s : 0] = ti * kTile + s * kBlock + (int)threadIdx.x < a.HW ? src[s * kBlock] : (raw_t)0;
        }
    };
    if constexpr (AHEAD) if (g < total_tiles) load_tile(cur, b, t);
    for (int it = 0; g < total_tiles; ++it) {
        int *seg = seg2[it & 1], *red = red2[it & 1];
        raw_t nxt[AHEAD ? kTileSteps : 1] = {};
        const int gn = g + gridDim.x;
        int bn = b + step_b, tn = t + step_t;
        if (tn >= a.T) { tn -= a.T; ++bn; }
        if constexpr (AHEAD) if (gn < total_tiles) load_tile(nxt, bn, tn);
        unsigned long long m[kTileSteps];
        int pc[kTileSteps];
        int sum = 0;
#pragma unroll
        for (int s = 0; s < kTileSteps; ++s) {
            const int p = t * kTile + s * kBlock + threadIdx.x;
            int w = 0;
            if constexpr (AHEAD) {
                const uint64_t v = (uint64_t)cur[s];               // 0 beyond the image
                w = MODE == 0 ? (int)(v & 0xFF) : (v == 1 ? 1 : 0);
            } else if (p < a.HW) {
                w = mask_weight<ES>(a, b, p);
            }
            m[s] = __ballot(w != 0);
            pc[s] = __popcll(m[s]);                                // (scalar)
            sum += w;
        }
        // the weight sum only differs from the pixel count for byte masks with values above 1 (P:126 sums the BYTES)
        if (MODE == 0) sum = wave_total(sum);
        else sum = pc[0] + pc[1] + pc[2] + pc[3] + pc[4] + pc[5] + pc[6] + pc[7];
        if (lane == 0) {
#pragma unroll
            for (int s = 0; s < kTileSteps; ++s) seg[s * 4 + wave] = pc[s];
            red[wave] = sum;
        }
        scan_segments(seg);
        if (threadIdx.x == 0)
            tiles[b * a.T + t] = (uint32_t)seg[kTileSteps * 4] | ((uint32_t)(red[0] + red[1] + red[2] + red[3]) << 12);
        unsigned short *list = tile_list + ((size_t)b * a.T + t) * kTile;
        float *draw = tile_draw + ((size_t)b * a.T + t) * kTile;
        if (seg[kTileSteps * 4] != 0) {                            // (block-uniform: four tiles in five hold no foreground)
#pragma unroll
            for (int s = 0; s < kTileSteps; ++s)
                if ((m[s] >> lane) & 1ull) {
                    const int r = seg[s * 4 + wave] + __popcll(m[s] & ((1ull << lane) - 1ull));
                    list[r] = (unsigned short)(s * kBlock + threadIdx.x);
                    if (a.want_draws) draw[r] = selection_draw(a, b, t * kTile + s * kBlock + threadIdx.x);
                }
        }
        if constexpr (AHEAD) {
#pragma unroll
            for (int s = 0; s < kTileSteps; ++s) cur[s] = nxt[s];
        }
        g = gn; b = bn; t = tn;
    }
}

// ---------------------------------------------------------------------------------------------
// The mask scan of decode_keypoint for a TWO-CLASS seg (PVNet's seg_dim; resnet18.py:69 `torch.argmax(output['seg'], 1)`):
// the layout the real caller passes is two contiguous float32 planes per image (channel slices of the network's output
// tensor, resnet18.py:93).  One tile per block; a thread owns 2 x 4 CONSECUTIVE pixels and reads each plane with two
// 16-byte non-temporal loads (all four in flight at once: the kernel is a read-once stream), decides idx = 1 iff
// v1 > v0 or (v1 is NaN and v0 is not) -- torch.argmax's first-maximum / NaN rule for two classes -- and ranks its
// foreground pixels with a wave prefix sum of the per-thread counts (DPP) instead of eight ballots: 8 segments per tile
// instead of 32.  For two classes the weight of a foreground pixel is 1 in both modes (idx & 0xFF = 1; idx == 1).
// WRITE_MASK: also store the int64 mask here (two 16-byte stores per 4 pixels).  The host defers that store to
// k_mask_from_lists on a side stream whenever the tile lists stay complete (no k_tile_subsample): 8 B per pixel written
// behind a 8 B per pixel read would double the traffic of this kernel (70 us instead of 27 at B = 64).
// Needs: gw == 1, gh == W, H*W % 4 == 0, the planes 16-byte aligned (host: seg2_ok).
// ---------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef long long i64x2 __attribute__((ext_vector_type(2)));

template <bool WRITE_MASK>
__global__ __launch_bounds__(kBlock) void k_tile_scan_seg2(MaskArgs a, uint32_t *__restrict__ tiles,
                                                           unsigned short *__restrict__ tile_list,
                                                           float *__restrict__ tile_draw)
{
    __shared__ int seg[8];                                         // (half, wave) counts
    const int g = blockIdx.x;
    const int b = g / a.T, t = g - b * a.T;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const float *p0 = a.seg + (int64_t)b * a.gb + (int64_t)t * kTile + 4 * threadIdx.x;
    const float *p1 = p0 + a.gc;
    constexpr int kHalf = kTile / 2;                               // 1024 pixels = 256 threads x 4
    f32x4 v0[2], v1[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const bool in = t * kTile + s * kHalf + 4 * (int)threadIdx.x < a.HW;   // HW % 4 == 0: a vector is inside or outside
        v0[s] = in ? __builtin_nontemporal_load((const f32x4 *)(p0 + s * kHalf)) : f32x4{0.f, 0.f, 0.f, 0.f};
        v1[s] = in ? __builtin_nontemporal_load((const f32x4 *)(p1 + s * kHalf)) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    unsigned m4[2];
    int excl[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        unsigned m = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x0 = v0[s][j], x1 = v1[s][j];
            m |= ((x1 > x0) || (x1 != x1 && x0 == x0)) ? (1u << j) : 0u;
        }
        m4[s] = m;
        const int c = __popc(m);
        const int inc = wave_incl_scan(c);
        excl[s] = inc - c;
        if (lane == 63) seg[s * 4 + wave] = inc;
    }
    __syncthreads();
    int cnt[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) cnt[i] = seg[i];
    int nz = 0, base[2] = {0, 0};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (i < wave) base[0] += cnt[i];
        if (i < 4 + wave) base[1] += cnt[i];
        nz += cnt[i];
    }
    if (threadIdx.x == 0) tiles[g] = (uint32_t)nz | ((uint32_t)nz << 12);
    if (WRITE_MASK && a.mask_out) {
        long long *mo = a.mask_out + (int64_t)b * a.HW + (int64_t)t * kTile + 4 * threadIdx.x;
#pragma unroll
        for (int s = 0; s < 2; ++s)
            if (t * kTile + s * kHalf + 4 * (int)threadIdx.x < a.HW) {
                __builtin_nontemporal_store(i64x2{(long long)(m4[s] & 1u), (long long)((m4[s] >> 1) & 1u)}, (i64x2 *)(mo + s * kHalf));
                __builtin_nontemporal_store(i64x2{(long long)((m4[s] >> 2) & 1u), (long long)((m4[s] >> 3) & 1u)}, (i64x2 *)(mo + s * kHalf + 2));
            }
    }
    if (nz == 0) return;                                           // (block-uniform: four tiles in five hold no foreground)
    unsigned short *list = tile_list + (size_t)g * kTile;
    float *draw = tile_draw + (size_t)g * kTile;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        int r = base[s] + excl[s];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if ((m4[s] >> j) & 1u) {
                const int off = s * kHalf + 4 * (int)threadIdx.x + j;
                list[r] = (unsigned short)off;
                if (a.want_draws) draw[r] = selection_draw(a, b, t * kTile + off);
                ++r;
            }
    }
}

// The int64 mask of decode_keypoint from the tile lists (two-class seg: mask = 1 exactly on the listed pixels), when
// k_tile_scan_seg2 did not write it: launched on the library's side stream behind the compaction, it streams its 8 B per
// pixel beside the VALU-bound count pass instead of in front of it.  Persistent (the host launches two blocks per
// CU: the count kernel's blocks leave room for exactly that, and a queue of 10 000 short-lived blocks would compete with
// the other kernels' dispatch), a tile per trip; a store instruction covers 4 KB contiguous.
__global__ __launch_bounds__(kBlock) void k_mask_from_lists(const uint32_t *__restrict__ tiles, const unsigned short *__restrict__ tile_list,
                                                            long long *__restrict__ mask_out, int T, int HW, int total_tiles)
{
    __shared__ unsigned bm[kTile / 32];
    uint32_t word = (int)blockIdx.x < total_tiles ? tiles[blockIdx.x] : 0u;
    for (int g = blockIdx.x; g < total_tiles; g += gridDim.x) {
        const int b = g / T, t = g - b * T;
        const int nz = (int)(word & kTileNzMask);
        if (g + (int)gridDim.x < total_tiles) word = tiles[g + gridDim.x];   // the next trip's word, requested a trip ahead
        if (nz) {                                                  // (block-uniform)
            __syncthreads();                                       // the previous tile's bitmap is consumed
            if (threadIdx.x < kTile / 32) bm[threadIdx.x] = 0u;
            __syncthreads();
            const unsigned short *list = tile_list + (size_t)g * kTile;
            for (int e = threadIdx.x; e < nz; e += kBlock) {
                const unsigned off = list[e];
                atomicOr(&bm[off >> 5], 1u << (off & 31u));
            }
            __syncthreads();
        }
        long long *dst = mask_out + (int64_t)b * HW + (int64_t)t * kTile;
#pragma unroll
        for (int j = 0; j < kTile / (2 * kBlock); ++j) {
            const int px = (j * kBlock + (int)threadIdx.x) * 2;    // HW % 4 == 0: a pair is inside or outside
            if (t * kTile + px >= HW) break;
            const unsigned w = nz ? bm[px >> 5] >> (px & 31) : 0u;
#ifdef PVV_MASK_PLAIN_STORE                                            // (tuning builds: A/B of the store policy)
            *(i64x2 *)(dst + px) = i64x2{(long long)(w & 1u), (long long)((w >> 1) & 1u)};
#else
            __builtin_nontemporal_store(i64x2{(long long)(w & 1u), (long long)((w >> 1) & 1u)}, (i64x2 *)(dst + px));
#endif
        }
    }
}

// foreground_num of P:126 / P:208 (sum of the weights) and the number of foreground pixels of image b.
struct ImageTotals { long long fg; int total; int before; };

// One pass over the image's tile table and ONE block reduction for all three sums: foreground_num, the rows of the
// tiles before tile t, and the image's row count.  redl: 4 long long, red: 8 int.
__device__ __forceinline__ ImageTotals image_totals(const uint32_t *__restrict__ tiles, int b, int T, int t,
                                                    long long *redl, int *red)
{
    long long fgs = 0;
    int before = 0, total = 0;
    for (int i = threadIdx.x; i < T; i += kBlock) {
        const uint32_t w = tiles[b * T + i];
        fgs += w >> 12;
        const int c = (int)(w & kTileNzMask);
        total += c;
        if (i < t) before += c;
    }
    fgs = wave_total(fgs);
    before = wave_total(before);
    total = wave_total(total);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane_id() == 0) { redl[wave] = fgs; red[wave] = before; red[4 + wave] = total; }
    __syncthreads();
    ImageTotals r;
    r.fg = redl[0] + redl[1] + redl[2] + redl[3];
    r.before = red[0] + red[1] + red[2] + red[3];
    r.total = red[4] + red[5] + red[6] + red[7];
    return r;
}

// Filter one tile's list by the subsample draw (keep iff U < prob), order kept: survivors land in out[] (LDS or
// global, may alias nothing), their number is returned to every thread.  seg: kTileSteps*4+1 ints of LDS.
__device__ __forceinline__ int filter_tile_list(const MaskArgs &a, int b, int t, int nz, float prob,
                                                const unsigned short *__restrict__ img_lists, const float *__restrict__ img_draws /*or nullptr*/,
                                                unsigned short *out, int *seg)
{
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const unsigned short *list = img_lists + (size_t)t * kTile;
    unsigned long long m[kTileSteps];
    unsigned short off[kTileSteps];
#pragma unroll
    for (int s = 0; s < kTileSteps; ++s) {
        const int e = s * kBlock + threadIdx.x;
        bool f = false;
        off[s] = 0;
        if (e < nz) {
            off[s] = list[e];
            f = (img_draws ? img_draws[(size_t)t * kTile + e] : selection_draw(a, b, t * kTile + (int)off[s])) < prob;
        }
        m[s] = __ballot(f);
        if (lane == 0) seg[s * 4 + wave] = __popcll(m[s]);
    }
    scan_segments(seg);
#pragma unroll
    for (int s = 0; s < kTileSteps; ++s)
        if ((m[s] >> lane) & 1ull) out[seg[s * 4 + wave] + __popcll(m[s] & ((1ull << lane) - 1ull))] = off[s];
    return seg[kTileSteps * 4];
}

// P:135-138 / P:219-223 as its own pass (large images, or max_num so small that nearly every image is subsampled):
// when foreground_num > max_num every foreground pixel survives with probability max_num/foreground_num (binary32
// quotient).  Rewrites the tile's list in place and its count; tiles without foreground and images that are not
// subsampled exit at once.
__global__ __launch_bounds__(kBlock) void k_tile_subsample(MaskArgs a, uint32_t *__restrict__ tiles,
                                                           unsigned short *__restrict__ tile_list,
                                                           const float *__restrict__ tile_draw)
{
    __shared__ long long redl[4];
    __shared__ int red[8];
    __shared__ int seg[kTileSteps * 4 + 1];
    __shared__ unsigned short keep[kTile];
    const int t = blockIdx.x, b = blockIdx.y;
    const uint32_t w = tiles[b * a.T + t];
    const int nz = (int)(w & kTileNzMask);
    if (nz == 0) return;
    const ImageTotals tot = image_totals(tiles, b, a.T, t, redl, red);
    if (tot.fg <= (long long)a.max_num) return;
    const float prob = (float)a.max_num / (float)tot.fg;
    unsigned short *list = tile_list + ((size_t)b * a.T + t) * kTile;
    const int n = filter_tile_list(a, b, t, nz, prob, tile_list + (size_t)b * a.T * kTile, tile_draw + (size_t)b * a.T * kTile, keep, seg);
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += kBlock) list[i] = keep[i];
    // In place: the other blocks of the image may still be reading the table, but they only use the weight sums
    // (foreground_num), which this update keeps; the pixel COUNT is read by later kernels only.
    if (threadIdx.x == 0) tiles[b * a.T + t] = (uint32_t)n | (w & ~kTileNzMask);
}

struct VertexArgs {
    const float *vertex;
    int64_t sb, sh, sw, sk, sc;
    int K;
    int vec2;  // sc == 1 and every other stride even: (x,y) is one aligned 8-byte load
};

__device__ __forceinline__ float2 load_vertex(const VertexArgs &v, int b, int y, int x, int vi)
{
    const float *src = v.vertex + (int64_t)b * v.sb + (int64_t)y * v.sh + (int64_t)x * v.sw + (int64_t)vi * v.sk;
    if (v.vec2) return *(const float2 *)src;
    return make_float2(src[0], src[v.sc]);
}

// ---------------------------------------------------------------------------------------------
// Stage 2: hypotheses (replaces random_ P:145/P:235 + generate_hypothesis K:11-86); zeroes the inlier counters.
// ---------------------------------------------------------------------------------------------
struct HypArgs {
    const int32_t *idxs;     // [B,hn_first,K,2] or null
    const int32_t *idxs2;    // [B,hn-hn_first,K,2] or null (the estimate's rounds of a fused un_pnp call)
    int hn, hn_first;        // hypotheses [0,hn_first) draw from idxs / RNG stream `stream`, the rest from idxs2 / `stream2`
    uint32_t stream, stream2;
    float2 *hyps;            // [B,K,hn]
    int *counts;             // [B,K,hn]
    int32_t *draws_out;      // [B,K,hn,2] or null: the pixel (y*W+x) each index pair resolved to (tests)
    int blocks;              // hypothesis blocks per image
    int *surv;               // [B, kSurvCap] scratch: the survivors of a heavily subsampled image (see k_compact_hyp)
    int *lead;               // [B,K,8] or null: leader counts of the staged count pass (count_prune.hpp); [4..7] zeroed here
    int *miss;               // [B,K,hn] or null: the staged pass's shared miss counters (count_filter_runs.hpp), zeroed here
};

constexpr int kHypRejectTries = 1 << 12;
// Fused subsampling with a SMALL survival probability (e.g. a 0/255 byte mask: foreground_num sums the byte values,
// P:126, so 6144 pixels of 255 are subsampled to ~117): rejection sampling would need ~1/prob tries per index, so below
// kSurvMinProb every hypothesis block lists the image's survivors (few by construction: < total/64 <= 5120 for the
// <= 160-tile images that fuse) in row-major order and draws from that list -- exactly randint(0, tn) over the subsampled
// list.  Every hypothesis block of the image writes the SAME list to the image's scratch row and reads back what it wrote.
constexpr float kSurvMinProb = 1.f / 64.f;
constexpr int kSurvCap = 8192;

// Row t of the image's (not yet written) compacted list -> pixel: search the inclusive tile prefix, then one read of
// the tile's list.
__device__ __forceinline__ int select_pixel(const int *prefix, int T, const unsigned short *__restrict__ lists /*of image b*/, int t,
                                            size_t *entry = nullptr /*index of the list entry within the image's lists*/)
{
    int lo = 0, hi = T - 1;               // smallest i with prefix[i] > t
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (prefix[mid] > t) hi = mid; else lo = mid + 1;
    }
    const int r = t - (lo ? prefix[lo - 1] : 0);
    const size_t e = (size_t)lo * kTile + r;
    if (entry) *entry = e;
    return lo * kTile + (int)lists[e];
}

// Ordered scatter + hypotheses, one launch, grid (h.blocks + T, B):
//  * blocks x < h.blocks (hypotheses): 256 hypotheses each, straight from the tile lists and the vertex field.  First in
//    the grid: their dependency chain (table, prefix, search, list entry, two gathers) is the longer one, so they
//    should not also be dispatched last (-0.7 us at B = 64, -1.2 us at B = 8);
//  * the other T blocks (compaction): pixel list of tile x - h.blocks -> rows of the image's compacted arrays:
//    coords[b][r] = (x,y) (P:140-141) and dirs[b][vi][r] = vertex[b,y,x,vi,:] (P:142-143, planar per keypoint so that
//    the count kernel's loads are unit-stride).
// Dynamic LDS: T ints (the tile prefix of the hypothesis blocks).
__global__ __launch_bounds__(kBlock) void k_compact_hyp(MaskArgs a, VertexArgs v, HypArgs h,
                                                        const uint32_t *__restrict__ tiles,
                                                        const unsigned short *__restrict__ tile_list,
                                                        const float *__restrict__ tile_draw,
                                                        int *__restrict__ tn_out, float2 *__restrict__ coords,
                                                        float2 *__restrict__ dirs)
{
    extern __shared__ int s_prefix[];
    __shared__ long long redl[4];
    __shared__ int red[8];
    __shared__ int seg[kTileSteps * 4 + 1];
    __shared__ unsigned short list[kTile];
    const int b = blockIdx.y;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const unsigned short *img_lists = tile_list + (size_t)b * a.T * kTile;
    const float *img_draws = a.want_draws ? tile_draw + (size_t)b * a.T * kTile : nullptr;   // nullptr: on demand (list_draw)

    if ((int)blockIdx.x < h.blocks) {
        // ------------------------------------------------------------------ hypothesis block
        const int j = blockIdx.x;
        // inclusive prefix of the tile counts in LDS (+ foreground_num), 256 tiles per round
        long long fgs = 0;
        int carry = 0;
        for (int base = 0; base < a.T; base += kBlock) {
            const int i = base + threadIdx.x;
            const uint32_t w = i < a.T ? tiles[b * a.T + i] : 0u;
            fgs += w >> 12;
            const int inc = wave_incl_scan((int)(w & kTileNzMask));
            __syncthreads();
            if (lane == 63) red[wave] = inc;
            __syncthreads();
            int off = carry;
            for (int w2 = 0; w2 < wave; ++w2) off += red[w2];
            if (i < a.T) s_prefix[i] = inc + off;
            carry += red[0] + red[1] + red[2] + red[3];
        }
        fgs = wave_total(fgs);
        __syncthreads();
        if (lane == 0) redl[wave] = fgs;
        __syncthreads();
        const long long fg = redl[0] + redl[1] + redl[2] + redl[3];
        const int total = carry;                                  // rows before any truncation at cap
        int tn = total < a.cap ? total : a.cap;
        if (fg < (long long)a.min_num) tn = 0;                    // P:129-132 / P:211-216
        // subsampling fused into this launch (a.fuse_sub): the lists still hold EVERY foreground pixel; a pixel
        // survives iff its draw < prob.  The index pairs then come from rejection sampling -- uniform over the
        // survivors, which is what randint(0, tn) over the subsampled list is -- so nothing has to wait for the
        // compaction (the host never fuses when index pairs are injected: those address the subsampled order).
        const bool sub = a.fuse_sub && fg > (long long)a.max_num;
        const float prob = sub ? (float)a.max_num / (float)fg : 2.f;
        int nsurv = -1;                                           // >= 0: survivors listed in h.surv[b]
        if (sub && prob < kSurvMinProb) {
            int *sv = h.surv + (size_t)b * kSurvCap;
            nsurv = 0;
            for (int i = 0; i < a.T; ++i) {                       // block-uniform walk over the tiles, 256 entries a round
                const int ni = s_prefix[i] - (i ? s_prefix[i - 1] : 0);
                for (int e0 = 0; e0 < ni; e0 += kBlock) {
                    const int e = e0 + threadIdx.x;
                    const bool keep = e < ni && list_draw(a, b, i, e, img_lists, img_draws) < prob;
                    const unsigned long long m = __ballot(keep);
                    __syncthreads();
                    if (lane == 0) red[wave] = __popcll(m);
                    __syncthreads();
                    int off = nsurv;
                    for (int w2 = 0; w2 < wave; ++w2) off += red[w2];
                    off += __popcll(m & ((1ull << lane) - 1ull));
                    if (keep && off < kSurvCap) sv[off] = i * kTile + (int)img_lists[(size_t)i * kTile + e];
                    nsurv += red[0] + red[1] + red[2] + red[3];
                }
            }
            __threadfence_block();
            __syncthreads();
            // More survivors than the list holds: impossible for the device RNG (its binomial is >= 40 sigma below kSurvCap
            // for the <= 160-tile images that fuse) but an injected selection tensor may keep any number.  Then
            // at least kSurvCap / (160 * 2048) = 1/40 of the listed pixels survive and rejection sampling (below) finds one
            // in a few tries -- uniform over ALL survivors, nothing truncated.
            if (nsurv > kSurvCap) nsurv = -1;
            else if (nsurv < tn) tn = nsurv;                      // what the compaction blocks will report (cap applies there)
        }

        const int gid = j * kBlock + threadIdx.x;                 // hypothesis (vi, hi) of image b
        if (gid >= v.K * h.hn) return;
        const int vi = gid / h.hn, hi = gid - vi * h.hn;
        const size_t o = ((size_t)b * v.K + vi) * h.hn + hi;
        h.counts[o] = 0;
        if (h.miss) h.miss[o] = 0;
        if (h.lead && hi == 0) {                                  // all four, whatever hn is (hi < min(4, hn) left words unzeroed for hn < 4)
            int *lw = h.lead + ((size_t)b * v.K + vi) * 8 + 4;
            lw[0] = 0; lw[1] = 0; lw[2] = 0; lw[3] = 0;
        }
        if (tn <= 0) {
            h.hyps[o] = make_float2(0.f, 0.f);
            if (h.draws_out) { h.draws_out[2 * o] = -1; h.draws_out[2 * o + 1] = -1; }
            return;
        }
        const bool first = hi < h.hn_first;
        const int32_t *src = first ? h.idxs : h.idxs2;
        int p0, p1;
        if (src) {
            const int32_t *ip = first ? h.idxs + (((size_t)b * h.hn_first + hi) * v.K + vi) * 2
                                      : h.idxs2 + (((size_t)b * (h.hn - h.hn_first) + (hi - h.hn_first)) * v.K + vi) * 2;
            int t0 = ip[0], t1 = ip[1];
            // the reference reads out of bounds here; clamp instead of faulting
            t0 = t0 < 0 ? 0 : (t0 >= tn ? tn - 1 : t0);
            t1 = t1 < 0 ? 0 : (t1 >= tn ? tn - 1 : t1);
            p0 = select_pixel(s_prefix, a.T, img_lists, t0);
            p1 = select_pixel(s_prefix, a.T, img_lists, t1);
        } else {
            const uint32_t stream = first ? h.stream : h.stream2;
            const uint32_t c = (uint32_t)((first ? hi : hi - h.hn_first) * v.K + vi) * 2u;
            const uint32_t img = (uint32_t)(a.b0 + b);
            if (!sub) {
                p0 = select_pixel(s_prefix, a.T, img_lists, (int)(rng_u32(a.seed, stream, img, c) % (uint32_t)tn));
                p1 = select_pixel(s_prefix, a.T, img_lists, (int)(rng_u32(a.seed, stream, img, c + 1u) % (uint32_t)tn));
            } else if (nsurv >= 0) {
                const int *sv = h.surv + (size_t)b * kSurvCap;
                p0 = sv[rng_u32(a.seed, stream, img, c) % (uint32_t)tn];
                p1 = sv[rng_u32(a.seed, stream, img, c + 1u) % (uint32_t)tn];
            } else {
                p0 = p1 = -1;
                for (int tr = 0; tr < kHypRejectTries && (p0 < 0 || p1 < 0); ++tr) {
                    // try tr of draw c: key (stream + 16 + 4 tr, image, c).  The try index goes into the STREAM word, where it
                    // cannot wrap (`c + (tr << 24)` wrapped at tr = 256, so tries 256.. replayed tries 0..255 and a
                    // draw that had failed 256 times failed 4096 times); the streams in use are 1 and 3, so 17 + 4 tr and
                    // 19 + 4 tr never meet, and 0 (the subsample draws) is never touched.
                    size_t e;
                    const uint32_t stry = stream + 16u + 4u * (uint32_t)tr;
                    if (p0 < 0) {
                        const int p = select_pixel(s_prefix, a.T, img_lists, (int)(rng_u32(a.seed, stry, img, c) % (uint32_t)total), &e);
                        if ((img_draws ? img_draws[e] : selection_draw(a, b, p)) < prob) p0 = p;
                    }
                    if (p1 < 0) {
                        const int p = select_pixel(s_prefix, a.T, img_lists, (int)(rng_u32(a.seed, stry, img, c + 1u) % (uint32_t)total), &e);
                        if ((img_draws ? img_draws[e] : selection_draw(a, b, p)) < prob) p1 = p;
                    }
                }
                if (p0 < 0 || p1 < 0) {                            // no survivor found (prob ~ 0): degenerate pair
                    h.hyps[o] = make_float2(0.f, 0.f);
                    if (h.draws_out) { h.draws_out[2 * o] = p0; h.draws_out[2 * o + 1] = p1; }
                    return;
                }
            }
        }
        const int y0 = p0 / a.W, x0 = p0 - y0 * a.W, y1 = p1 / a.W, x1 = p1 - y1 * a.W;
        const float2 d0 = load_vertex(v, b, y0, x0, vi), d1 = load_vertex(v, b, y1, x1, vi);
        h.hyps[o] = hypothesis_exact(d0.x, d0.y, (float)x0, (float)y0, d1.x, d1.y, (float)x1, (float)y1);
        if (h.draws_out) { h.draws_out[2 * o] = p0; h.draws_out[2 * o + 1] = p1; }
        return;
    }

    // ---------------------------------------------------------------------- compaction block of tile t
    const int t = blockIdx.x - h.blocks;
    const int nz = (int)(tiles[b * a.T + t] & kTileNzMask);
    // background-only tile: nothing to scatter (tile 0 reports tn; with fused subsampling the last tile does)
    if (t != 0 && !(a.fuse_sub && t == a.T - 1) && nz == 0) return;
    const unsigned short *my_list = img_lists + (size_t)t * kTile;
    // this tile's list, requested before the reductions below so that the two latencies overlap
    unsigned short mine[kTileSteps];
#pragma unroll
    for (int s = 0; s < kTileSteps; ++s) {
        const int e = s * kBlock + threadIdx.x;
        mine[s] = e < nz ? my_list[e] : (unsigned short)0;
    }
    const ImageTotals tot = image_totals(tiles, b, a.T, t, redl, red);
    if (tot.fg < (long long)a.min_num) {  // P:129-132 / P:211-216: image skipped
        if (t == 0 && threadIdx.x == 0) {
            tn_out[b] = 0;
            if (a.tn_user) a.tn_user[b] = 0;
            if (a.status) a.status[b] = PVV_STATUS_SKIPPED;
        }
        return;
    }
    int before = tot.before, tile_n = nz;
    // P:135-138 / P:219-223 fused (a.fuse_sub; otherwise k_tile_subsample ran first and the lists are final): the draws
    // are keyed by (image, pixel), so this block redoes them for the pixels of the tiles before it -- it needs their
    // survivor count for its row offset -- and filters its own list.  Only for images with foreground_num > max_num,
    // and it saves a launch on every call; the last tile's block reports tn.
    const bool sub = a.fuse_sub && tot.fg > (long long)a.max_num;
    if (sub) {
        const float prob = (float)a.max_num / (float)tot.fg;
        int cnt = 0;
        if (img_draws || a.T > kBlock) {                                            // (never with the host's rule: fused = on-demand draws, <= 160 tiles)
            for (int i = 0; i < t; ++i) {
                const int ni = (int)(tiles[b * a.T + i] & kTileNzMask);             // block-uniform
                for (int e = threadIdx.x; e < ni; e += kBlock)
                    cnt += list_draw(a, b, i, e, img_lists, img_draws) < prob ? 1 : 0;
            }
        } else {
            // The lists of the tiles before t as ONE sequence of 8-entry groups (16 bytes), dealt round-robin to the threads, four
            // loads per thread in flight.  (Until round 4 this was a loop over the tiles with a block-uniform load of the tile's
            // count and then its entries -- two dependent round trips per tile, 47.6 us for the LAST tile of a dense 480x640 image,
            // tn = 30 000: profiles/r04_configs.json cfg2_dense_tn30000_B1.)  s_prefix (the dynamic LDS of the hypothesis blocks):
            // inclusive prefix of the tiles' group counts; the tile of a thread's next group is found by walking on from its last.
            const int ti = threadIdx.x;
            const int nzi = ti < t ? (int)(tiles[b * a.T + ti] & kTileNzMask) : 0;
            const int inc = wave_incl_scan((nzi + 7) >> 3);
            __syncthreads();
            if (lane == 63) red[wave] = inc;
            __syncthreads();
            int goff = 0;
            for (int w2 = 0; w2 < wave; ++w2) goff += red[w2];
            if (ti < a.T) s_prefix[ti] = inc + goff;
            list[ti] = (unsigned short)nzi;                                         // (list[] is free until filter_tile_list: the tiles' counts)
            __syncthreads();
            const int groups = t ? s_prefix[t - 1] : 0;
            constexpr int kAhead = 4;
            int tile = 0;
            for (int g0 = threadIdx.x; g0 < groups; g0 += kAhead * kBlock) {
                uint4 ent[kAhead];
                int left[kAhead], pix0[kAhead];
#pragma unroll
                for (int u = 0; u < kAhead; ++u) {
                    const int g = g0 + u * kBlock;
                    left[u] = 0;
                    if (g < groups) {
                        while (s_prefix[tile] <= g) ++tile;
                        const int e0 = (g - (tile ? s_prefix[tile - 1] : 0)) * 8;
                        ent[u] = *(const uint4 *)(img_lists + (size_t)tile * kTile + e0);
                        left[u] = (int)list[tile] - e0;                     // valid entries of the group (>= 1; more than 8: all)
                        pix0[u] = tile * kTile;
                    }
                }
#pragma unroll
                for (int u = 0; u < kAhead; ++u) {
                    const uint32_t wds[4] = {ent[u].x, ent[u].y, ent[u].z, ent[u].w};
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (q < left[u]) {
                            const int off = (int)((wds[q >> 1] >> (16 * (q & 1))) & 0xffffu);
                            cnt += selection_draw(a, b, pix0[u] + off) < prob ? 1 : 0;
                        }
                }
            }
        }
        before = block_sum(cnt, red);
        __syncthreads();
        tile_n = filter_tile_list(a, b, t, nz, prob, img_lists, img_draws, list, seg);
        __syncthreads();
        if (t == a.T - 1 && threadIdx.x == 0) {                                    // the last tile knows the subsampled total
            const int all = before + tile_n;
            tn_out[b] = all < a.cap ? all : a.cap;
            if (a.tn_user) a.tn_user[b] = tn_out[b];
            if (a.status) a.status[b] = PVV_STATUS_SUBSAMPLED | (all > a.cap ? PVV_STATUS_TRUNCATED : 0);
        }
    } else {
        if (t == 0 && threadIdx.x == 0) {
            tn_out[b] = tot.total < a.cap ? tot.total : a.cap;
            if (a.tn_user) a.tn_user[b] = tn_out[b];
            // not fused: k_tile_subsample has already rewritten the lists of an image with foreground_num > max_num
            if (a.status) a.status[b] = (tot.fg > (long long)a.max_num ? PVV_STATUS_SUBSAMPLED : 0) |
                                        (tot.total > a.cap ? PVV_STATUS_TRUNCATED : 0);
        }
#pragma unroll
        for (int s = 0; s < kTileSteps; ++s) {
            const int e = s * kBlock + threadIdx.x;
            if (e < nz) list[e] = mine[s];
        }
        __syncthreads();
    }
    const int room = a.cap - before;                                     // rows left in the image's list
    const int n = tile_n < room ? tile_n : (room > 0 ? room : 0);
    // Row li of the tile: its pixel once (ONE division by W), then coords and the K gathers of 8 bytes -- all K in flight,
    // consecutive threads on consecutive rows of every planar array.  (Rounds 1-2 flattened (keypoint, row) into one index
    // and paid two integer divisions -- ~35 instructions each, there is no hardware divide -- per GATHER: ~600 instructions
    // per thread and tile in a block whose whole life is a few microseconds.)
    constexpr int kKeys = 12;                                            // gathers in flight per trip of the keypoint loop (K = 9: one trip)
    for (int li = threadIdx.x; li < n; li += kBlock) {
        const int p = t * kTile + list[li];
        const int y = p / a.W, x = p - y * a.W;
        coords[(size_t)b * a.cap + before + li] = make_float2((float)x, (float)y);
        float2 *drow = dirs + (size_t)b * v.K * a.cap + before + li;
        for (int v0 = 0; v0 < v.K; v0 += kKeys) {
            float2 d[kKeys];
#pragma unroll
            for (int u = 0; u < kKeys; ++u)
                if (v0 + u < v.K) d[u] = load_vertex(v, b, y, x, v0 + u);
#pragma unroll
            for (int u = 0; u < kKeys; ++u)
                if (v0 + u < v.K) drow[(size_t)(v0 + u) * a.cap] = d[u];
        }
    }
}
