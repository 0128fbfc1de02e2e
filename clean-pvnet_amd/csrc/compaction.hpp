// compaction.hpp -- stage 1: mask scan, subsample, ordered compaction.
// Part of the single translation unit pvnet_vote.hip (included inside its anonymous namespace); see that file
// for the numerical contract and the reference citations (K = ransac_voting_kernel.cu, P = ransac_voting_gpu.py).
#pragma once

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
    int b0;                  // first_image: RNG key offset of image 0
    int *tn_user;            // the caller's tn[B] (or nullptr): written beside the workspace copy, no D2D copy later
    int fuse_sub;            // 1: k_compact applies the subsampling itself (no k_tile_subsample launch), see there
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
    return (float)(rng_u32(a.seed, 0u, (uint32_t)(a.b0 + b), (uint32_t)p) >> 8) * 0x1p-24f;
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
    __shared__ int red[8];
    __shared__ int seg[kTileSteps * 4 + 1];
    __shared__ unsigned short list[kTile];
    const int t = blockIdx.x, b = blockIdx.y;
    const int lane = lane_id(), wave = threadIdx.x >> 6;

    // background-only tile: nothing to scatter (tile 0 reports tn; with fused subsampling the last tile does)
    if (t != 0 && !(a.fuse_sub && t == a.T - 1) && tile_nz[b * a.T + t] == 0) return;

    // this tile's foreground map, requested before the reductions below so that the two latencies overlap
    const unsigned long long *wb = bits + ((size_t)b * a.T + t) * (kTileSteps * 4);
    unsigned long long word[kTileSteps];
#pragma unroll
    for (int s = 0; s < kTileSteps; ++s) word[s] = wb[s * 4 + wave];   // wave-uniform

    // one pass over the image's tile table and ONE block reduction for all three sums: foreground_num (P:126 / P:208),
    // the rows before this tile, and the image's row count
    long long fgs = 0;
    int before = 0, total = 0;
    for (int i = threadIdx.x; i < a.T; i += kBlock) {
        fgs += tile_sum[b * a.T + i];
        const int c = tile_nz[b * a.T + i];
        total += c;
        if (i < t) before += c;
    }
    fgs = wave_sum(fgs);
    before = wave_sum(before);
    total = wave_sum(total);
    if (lane == 0) { redl[wave] = fgs; red[wave] = before; red[4 + wave] = total; }
    __syncthreads();
    const long long fg = redl[0] + redl[1] + redl[2] + redl[3];
    before = red[0] + red[1] + red[2] + red[3];
    total = red[4] + red[5] + red[6] + red[7];
    if (fg < (long long)a.min_num) {  // P:129-132 / P:211-216: image skipped
        if (t == 0 && threadIdx.x == 0) {
            tn_out[b] = 0;
            if (a.tn_user) a.tn_user[b] = 0;
        }
        return;
    }
    // P:135-138 / P:219-223 fused (images of <= kFuseSubTiles tiles; larger ones go through k_tile_subsample first):
    // when foreground_num > max_num every foreground pixel survives with probability max_num/foreground_num.  The
    // draws are keyed by (image, pixel), so this block redoes them for its own tile AND for the tiles before it -- it
    // needs their survivor counts for its row offset.  Rare and bounded (<= kFuseSubTiles tiles), and it saves a launch
    // on every call; the last tile's block reports tn.
    const bool sub = a.fuse_sub && fg > (long long)a.max_num;
    if (sub) {
        const float prob = (float)a.max_num / (float)fg;
        int cnt = 0;
        for (int i = 0; i < t; ++i) {
            if (tile_nz[b * a.T + i] == 0) continue;                       // block-uniform
            const unsigned long long *wi = bits + ((size_t)b * a.T + i) * (kTileSteps * 4);
#pragma unroll
            for (int s = 0; s < kTileSteps; ++s) {
                const bool f = (wi[s * 4 + wave] >> lane) & 1ull;
                if (f) cnt += selection_draw(a, b, i * kTile + s * kBlock + threadIdx.x) < prob ? 1 : 0;
            }
        }
        __syncthreads();                                                   // red[] was read above
        before = block_sum(cnt, red);
#pragma unroll
        for (int s = 0; s < kTileSteps; ++s) {
            bool f = (word[s] >> lane) & 1ull;
            if (f) f = selection_draw(a, b, t * kTile + s * kBlock + threadIdx.x) < prob;
            word[s] = __ballot(f);
        }
    } else if (t == 0 && threadIdx.x == 0) {
        tn_out[b] = total < a.cap ? total : a.cap;
        if (a.tn_user) a.tn_user[b] = tn_out[b];
    }

#pragma unroll
    for (int s = 0; s < kTileSteps; ++s)
        if (lane == 0) seg[s * 4 + wave] = __popcll(word[s]);
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
        if (threadIdx.x == kTileSteps * 4 - 1) seg[kTileSteps * 4] = inc;   // foreground pixels of the tile (after subsampling)
    }
    __syncthreads();

    // foreground pixels of the tile -> LDS list (in rank order), so that the K-fold gather below is spread over
    // all 256 threads instead of looping inside the few lanes that own a foreground pixel
#pragma unroll
    for (int s = 0; s < kTileSteps; ++s) {
        const unsigned long long m = word[s];
        if (!((m >> lane) & 1ull)) continue;
        const int lr = seg[s * 4 + wave] + __popcll(m & ((1ull << lane) - 1ull));   // rank within the tile
        list[lr] = (unsigned short)(s * kBlock + threadIdx.x);
    }
    __syncthreads();
    const int tile_n = seg[kTileSteps * 4];
    if (sub && t == a.T - 1 && threadIdx.x == 0) {                       // the last tile knows the subsampled total
        const int all = before + tile_n;
        tn_out[b] = all < a.cap ? all : a.cap;
        if (a.tn_user) a.tn_user[b] = tn_out[b];
    }
    const int room = a.cap - before;                                     // rows left in the image's list
    const int n = tile_n < room ? tile_n : (room > 0 ? room : 0);
    for (int i = threadIdx.x; i < n; i += kBlock) {
        const int p = t * kTile + list[i];
        const int y = p / a.W;
        coords[(size_t)b * a.cap + before + i] = make_float2((float)(p - y * a.W), (float)y);
    }
    // n*K gathers of 8 bytes each; eight per thread in flight (all loads of a trip before its stores), otherwise every
    // trip of the loop pays a full memory latency: 9 trips at K = 9 and ~250 foreground pixels per tile
    constexpr int kGather = 8;
    const int total_g = n * v.K;
    for (int i0 = threadIdx.x; i0 < total_g; i0 += kGather * kBlock) {
        float2 d[kGather];
        size_t row[kGather];
        int xs[kGather], ys[kGather];
#pragma unroll
        for (int u = 0; u < kGather; ++u) {
            const int i = i0 + u * kBlock;
            d[u] = make_float2(0.f, 0.f);
            row[u] = 0;
            xs[u] = ys[u] = 0;
            if (i < total_g) {
                const int vi = i / n, li = i - vi * n;                   // consecutive threads -> consecutive rows
                const int p = t * kTile + list[li];
                const int y = p / a.W;
                const int x = p - y * a.W;
                const float *src = v.vertex + (int64_t)b * v.sb + (int64_t)y * v.sh + (int64_t)x * v.sw + (int64_t)vi * v.sk;
                if (v.vec2) {
                    d[u] = *(const float2 *)src;
                } else {
                    d[u].x = src[0];
                    d[u].y = src[v.sc];
                }
                row[u] = ((size_t)b * v.K + vi) * a.cap + before + li;
                xs[u] = x; ys[u] = y;
            }
        }
#pragma unroll
        for (int u = 0; u < kGather; ++u) {
            if (i0 + u * kBlock < total_g) {
                dirs[row[u]] = d[u];
                if (v.kappa != 0.0) recs[row[u]] = make_record((float)xs[u], (float)ys[u], d[u].x, d[u].y, v.kappa);
            }
        }
    }
}
