// DROPPED EXPERIMENT (round 2), kept for the record -- not compiled into the product.  One fused launch for the
// reference's default call (hn = 128, max_num = 100): bit-exact (the GPU parity suite passed with it), but no faster than
// the general pipeline: 28.1 vs 27.7 us of GPU time at B = 1, 73 vs 66 us at B = 64 (tools/ab in the git history of this
// round).  Lesson recorded in DESIGN.md: a dependent phase (barrier + memory round trip) costs the same 2.5-5 us inside a
// kernel as it does as its own launch; fusing launches does not shorten the dependency chain.

// tiny.hpp -- the whole back end of ransac_voting_layer_v3 in ONE launch when few pixels vote.
// Part of the single translation unit pvnet_vote.hip (included inside its anonymous namespace); see that file
// for the numerical contract and the reference citations (K = ransac_voting_kernel.cu, P = ransac_voting_gpu.py).
#pragma once

// ---------------------------------------------------------------------------------------------
// The reference's DEFAULT call (resnet18.py:75: 128 hypotheses, max_num = 100) subsamples every image down to ~100
// pixels; the work per image is then 128 x 9 x 100 evaluations -- nothing -- and the general pipeline (subsample,
// compaction + hypotheses, count, refit) is five launches of latency.  When the rows reserved per image fit LDS
// (cap <= kTinyCap) one 1024-thread block per (keypoint, image) does everything after the mask scan:
//   subsample (P:135-138, from the draws k_tile_scan stored) and ordered compaction of the image's pixel lists into
//   LDS (coords, P:140-141) -> this keypoint's directions (P:142-143) -> hypotheses (K:22-48) -> inlier counts
//   (K:100-125 through the sqrt/divide-free test with its guard band, exact sequence inside the band: same counts)
//   -> first-maximum winner (P:160) -> least-squares refit (P:176-191) -> partial normal equations for k_finalize_v3.
// The K blocks of an image redo the (cheap: a compare per listed pixel) subsample and compaction; nothing is written
// to the global compacted arrays.  Everything here is latency: 16 waves per block and branch-free batches of loads.
//
// The fast test (CPU model: tests/test_band_model.py::test_packed_valu_band): d = fl(h - c) as the exact
// path, nh = n/|n| and B = kappa perp(nh) binary64 quotients rounded once, a = fma(dx, nhx, dy nhy),
// b' = fma(dx, Bx, dy By), t = a - |b'|; decided from the sign of t when |t| - beta a > eps_abs, otherwise by the exact
// binary32 sequence.  A pixel the exact test can never accept (norm1 < 1e-6 or non-finite) is stored with cx = +inf
// (t = -inf, never in the band); a hypothesis that is non-finite or beyond 1e15 px takes the exact sequence throughout.
// ---------------------------------------------------------------------------------------------
constexpr int kTinyCap = 1024;        // rows per image held in LDS
constexpr int kTinySegs = 32;         // 256-entry segments of the pixel lists per batch (one keep bit each)
constexpr int kTinyMaxTiles = 2048;   // tile table in LDS (images up to 4 Mpixel; larger ones take the general path)

struct FastConsts {
    float beta;     // relative half-width of the guard band (in units of a)
    float eps_abs;  // absolute floor of the band, px
    float thresh;
    int use_fast;   // 0: thresh outside [0.5, 0.99995] -- every evaluation exact
    double kappa;
};

struct __attribute__((aligned(16))) TinyPix {      // 32 bytes of LDS per row
    float cx, cy, nhx, nhy, Bx, By, nx, ny;
};

// NT threads per block: 1024 when the batch is small (every phase is a latency chain, 16 waves shorten each), fewer
// when B*K blocks would not be resident at once (the host picks).  Dynamic LDS: tiny_lds_bytes(cap, T).
__host__ __device__ inline size_t tiny_lds_bytes(int cap, int T)
{
    return (size_t)cap * (sizeof(TinyPix) + sizeof(int)) + (size_t)T * (sizeof(int) + 2 * sizeof(unsigned short)) + 64;
}

template <int NT>
__global__ __launch_bounds__(NT) void k_tiny_vote(MaskArgs a, VertexArgs v, HypArgs h,
                                                          const uint32_t *__restrict__ tiles,
                                                          const unsigned short *__restrict__ tile_list,
                                                          const float *__restrict__ tile_draw,
                                                          int *__restrict__ tn_out, double *__restrict__ sums /*[B,K,1,5]*/,
                                                          int *__restrict__ win_counts /*[B,K] or null*/, FastConsts fc,
                                                          long long *__restrict__ dbg /*instrumented builds only*/)
{
#ifdef PVV_STAMPS
#define PVV_TSTAMP(i) do { if (dbg && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) dbg[(i)] = wall_clock64(); \
                           if (h.blocks == -(i) - 1) return; } while (0)     /* bisection: stop after phase i */
#else
#define PVV_TSTAMP(i) do { } while (0)
#endif
    constexpr int kTinyBlock = NT, kTinyWaves = NT / 64, QN = NT / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    TinyPix *s_pix = (TinyPix *)s_dyn;                              // [cap]
    int *s_rowp = (int *)(s_pix + a.cap);                           // [cap] pixel index y*W + x of every row
    int *s_sp = s_rowp + a.cap;                                     // [T] inclusive prefix of the 256-entry segment counts of
    unsigned short *s_ft = (unsigned short *)(s_sp + a.T);          // [T] the tiles with foreground, in order,
    unsigned short *s_fn = s_ft + a.T;                              // [T] and their list lengths - 1
    __shared__ int s_bbase[kTinySegs], s_bcnt[kTinySegs];   // the batch's segments: first list entry (index into the image's lists), entries
    __shared__ int s_rc[kTinySegs * 4 + 1];          // survivors per (segment, wave of the segment) of a batch
    __shared__ int s_cnt[kTinyBlock];
    __shared__ long long redl[kTinyWaves];
    __shared__ int red[2 * kTinyWaves];
    __shared__ int s_best[kTinyWaves], s_besti[kTinyWaves];
    __shared__ float2 s_besth[kTinyWaves];
    __shared__ double red5[5 * kTinyWaves];
    PVV_TSTAMP(0);
    const int vi = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bk = b * v.K + vi;
    const unsigned short *img_lists = tile_list + (size_t)b * a.T * kTile;
    const float *img_draws = tile_draw + (size_t)b * a.T * kTile;
    double *part = sums + (size_t)bk * 5;

    // ---- one pass over the tile table: foreground_num (P:126), the tiles that hold foreground and the inclusive
    //      prefix of their segment counts (a segment = up to 256 consecutive entries of one tile's list)
    long long fgs = 0;
    int nf = 0, nseg = 0;
    for (int base = 0; base < a.T; base += kTinyBlock) {
        const int i = base + threadIdx.x;
        const uint32_t w = i < a.T ? tiles[b * a.T + i] : 0u;
        fgs += w >> 12;
        const int nz = (int)(w & kTileNzMask);
        int inc = (nz + 255) >> 8, finc = nz > 0 ? 1 : 0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int n = __shfl_up(inc, o, 64), fn = __shfl_up(finc, o, 64);
            if (lane >= o) { inc += n; finc += fn; }
        }
        __syncthreads();
        if (lane == 63) { red[wave] = inc; red[kTinyWaves + wave] = finc; }
        __syncthreads();
        int off = nseg, foff = nf, tot = 0, ftot = 0;
#pragma unroll
        for (int w2 = 0; w2 < kTinyWaves; ++w2) {
            if (w2 < wave) { off += red[w2]; foff += red[kTinyWaves + w2]; }
            tot += red[w2]; ftot += red[kTinyWaves + w2];
        }
        if (nz > 0) {
            s_ft[foff + finc - 1] = (unsigned short)i;
            s_fn[foff + finc - 1] = (unsigned short)(nz - 1);
            s_sp[foff + finc - 1] = off + inc;
            // the first batch's segment table straight from here (later batches -- more than 32 segments, i.e. more
            // than ~8000 listed pixels -- search the prefix)
            const int seg_lo = off + inc - ((nz + 255) >> 8);
            for (int q = 0; q < (nz + 255) >> 8 && seg_lo + q < kTinySegs; ++q) {
                s_bbase[seg_lo + q] = i * kTile + q * 256;
                s_bcnt[seg_lo + q] = min(256, nz - q * 256);
            }
        }
        nseg += tot;
        nf += ftot;
    }
    fgs = wave_sum(fgs);
    __syncthreads();
    if (lane == 0) redl[wave] = fgs;
    __syncthreads();
    long long fg = 0;
#pragma unroll
    for (int w2 = 0; w2 < kTinyWaves; ++w2) fg += redl[w2];
    const bool skipped = fg < (long long)a.min_num;                 // P:129-132
    const bool sub = fg > (long long)a.max_num;                     // P:135-137
    const float prob = sub ? (float)a.max_num / (float)fg : 2.f;
    PVV_TSTAMP(1);

    // ---- subsample + ordered compaction into LDS + this keypoint's direction of every row, a batch of kTinySegs
    //      segments at a time: thread (q, l) = (threadIdx / 256, threadIdx % 256) looks at entry l of the segments
    //      q, q + 4, ... of the batch.  Phase 1: every draw in flight at once (no dependent addressing: the batch's
    //      segment table is built first), one keep bit per segment in a register, survivors counted per (segment, wave);
    //      one scan turns the counts into row offsets; phase 2: ranks from ballots of the kept bits, and only the
    //      survivors -- ~1.6 % of the entries in the reference's default call -- read their list entry and gather the
    //      direction.  Rows >= cap are dropped, as the general path does.
    const int q4 = threadIdx.x >> 8, l256 = threadIdx.x & 255, w4 = (threadIdx.x >> 6) & 3;   // q4 < QN
    int rows = 0;                                                   // survivors of the batches before
    const int S_total = skipped ? 0 : nseg;
    for (int s0 = 0; s0 < S_total && rows < a.cap; s0 += kTinySegs) {
        const int nb = min(kTinySegs, S_total - s0);
        if (s0 > 0 && threadIdx.x < nb) {                           // the batch's segment table
            const int sg = s0 + threadIdx.x;
            int lo = 0, hi = nf - 1;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (s_sp[mid] > sg) hi = mid; else lo = mid + 1;
            }
            const int q = sg - (lo ? s_sp[lo - 1] : 0);
            s_bbase[threadIdx.x] = (int)s_ft[lo] * kTile + q * 256;
            s_bcnt[threadIdx.x] = min(256, (int)s_fn[lo] + 1 - q * 256);
        }
        if (s0 > 0) __syncthreads();
        PVV_TSTAMP(8);
        unsigned kept = 0u;                                         // bit u: entry l256 of segment QN u + q4 survives
        {
            float dv[kTinySegs / QN];
            bool valid[kTinySegs / QN];
#pragma unroll
            for (int u = 0; u < kTinySegs / QN; ++u) {              // branch-free: an invalid slot reads entry 0
                const int sgi = QN * u + q4;
                valid[u] = sgi < nb && l256 < s_bcnt[sgi];
                dv[u] = img_draws[valid[u] && sub ? s_bbase[sgi] + l256 : 0];
            }
            PVV_TSTAMP(9);
#pragma unroll
            for (int u = 0; u < kTinySegs / QN; ++u) {
                const bool keep = valid[u] && (!sub || dv[u] < prob);
                kept |= keep ? 1u << u : 0u;
                const unsigned long long m = __ballot(keep);
                if (lane == 0) s_rc[(QN * u + q4) * 4 + w4] = __popcll(m);
            }
            PVV_TSTAMP(10);
        }
        __syncthreads();
        PVV_TSTAMP(2);
        if (threadIdx.x < 64) {                                     // exclusive scan of the 128 counts, two per lane
            const int i0 = 2 * lane, i1 = 2 * lane + 1;
            const int c0 = s_rc[i0], c1 = s_rc[i1];
            int inc = c0 + c1;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int n = __shfl_up(inc, o, 64);
                if (lane >= o) inc += n;
            }
            s_rc[i0] = inc - c0 - c1;
            s_rc[i1] = inc - c1;
            if (lane == 63) s_rc[kTinySegs * 4] = inc;
        }
        __syncthreads();
        PVV_TSTAMP(3);
        // ranks from ballots of the kept bits (no memory); a survivor parks its list position in its row's slot ...
#pragma unroll
        for (int u = 0; u < kTinySegs / QN; ++u) {
            const bool keep = (kept >> u) & 1u;
            const unsigned long long m = __ballot(keep);
            const int sgi = QN * u + q4;
            const int r = rows + s_rc[sgi * 4 + w4] + __popcll(m & ((1ull << lane) - 1ull));
            if (keep && r < a.cap) s_rowp[r] = s_bbase[sgi] + l256;
        }
        const int rows_end = min(a.cap, rows + s_rc[kTinySegs * 4]);
        __syncthreads();
        // ... and then every new row is completed by its own thread, all at once: list entry -> pixel -> this keypoint's
        // direction -> the record of the fast test (two dependent round trips in total, not per segment)
        for (int r = rows + threadIdx.x; r < rows_end; r += kTinyBlock) {
            const int le = s_rowp[r];
            const int p = (le & ~(kTile - 1)) + (int)img_lists[le];
            const int y = p / a.W, x = p - y * a.W;
            const float2 d = load_vertex(v, b, y, x, vi);
            TinyPix px;
            px.nx = d.x; px.ny = d.y;
            const float norm1 = sqrtf(d.x * d.x + d.y * d.y);       // the exact path's own norm1 (K:116)
            if (!lt_1e6(norm1) && norm1 < INFINITY && norm1 == norm1) {
                const double N1 = sqrt((double)d.x * (double)d.x + (double)d.y * (double)d.y);
                const double ux = (double)d.x / N1, uy = (double)d.y / N1;
                px.cx = (float)x; px.cy = (float)y; px.nhx = (float)ux; px.nhy = (float)uy;
                px.Bx = (float)(-fc.kappa * uy); px.By = (float)(fc.kappa * ux);
            } else {
                px.cx = INFINITY; px.cy = 0.f; px.nhx = 1.f; px.nhy = 0.f; px.Bx = 1.f; px.By = 0.f;
            }
            s_pix[r] = px;
            s_rowp[r] = p;                                       // true coordinates for everything exact
        }
        rows += s_rc[kTinySegs * 4];
        __syncthreads();
    }
    PVV_TSTAMP(4);
    const int tn = rows < a.cap ? rows : a.cap;
    if (vi == 0 && threadIdx.x == 0) {
        tn_out[b] = tn;
        if (a.tn_user) a.tn_user[b] = tn;
    }
    if (tn <= 0) {
        if (threadIdx.x < 5) part[threadIdx.x] = 0.0;
        if (threadIdx.x == 0 && win_counts) win_counts[bk] = 0;
        for (int hi = threadIdx.x; hi < h.hn; hi += kTinyBlock) {
            h.counts[(size_t)bk * h.hn + hi] = 0;
            h.hyps[(size_t)bk * h.hn + hi] = make_float2(0.f, 0.f);
            if (h.draws_out) { h.draws_out[2 * ((size_t)bk * h.hn + hi)] = -1; h.draws_out[2 * ((size_t)bk * h.hn + hi) + 1] = -1; }
        }
        return;
    }
    s_cnt[threadIdx.x] = 0;
    __syncthreads();

    // ---- hypotheses and their inlier counts, kTinyBlock threads = HB hypotheses x S pixel slices per batch
    int HB = kTinyBlock;
    while (HB > 64 && (HB >> 1) >= h.hn) HB >>= 1;               // hn = 128 -> 128 hypotheses x 8 slices
    const int S = kTinyBlock / HB;
    const int hl = threadIdx.x % HB, slice = threadIdx.x / HB;
    int best = -1, besti = 0x7fffffff;
    float2 besth = make_float2(0.f, 0.f);
    for (int hbase = 0; hbase < h.hn; hbase += HB) {
        const int hi = hbase + hl;
        float2 hyp = make_float2(0.f, 0.f);
        if (hi < h.hn) {
            int t0, t1;
            if (h.idxs) {
                const int32_t *ip = h.idxs + (((size_t)b * h.hn + hi) * v.K + vi) * 2;
                t0 = ip[0]; t1 = ip[1];
                t0 = t0 < 0 ? 0 : (t0 >= tn ? tn - 1 : t0);
                t1 = t1 < 0 ? 0 : (t1 >= tn ? tn - 1 : t1);
            } else {
                const uint32_t c = (uint32_t)(hi * v.K + vi) * 2u;
                t0 = (int)(rng_u32(a.seed, h.stream, (uint32_t)(a.b0 + b), c) % (uint32_t)tn);
                t1 = (int)(rng_u32(a.seed, h.stream, (uint32_t)(a.b0 + b), c + 1u) % (uint32_t)tn);
            }
            const int p0 = s_rowp[t0], p1 = s_rowp[t1];
            const int y0 = p0 / a.W, x0 = p0 - y0 * a.W, y1 = p1 / a.W, x1 = p1 - y1 * a.W;
            hyp = hypothesis_exact(s_pix[t0].nx, s_pix[t0].ny, (float)x0, (float)y0, s_pix[t1].nx, s_pix[t1].ny, (float)x1, (float)y1);
            if (slice == 0) {
                const size_t o = (size_t)bk * h.hn + hi;
                h.hyps[o] = hyp;
                if (h.draws_out) { h.draws_out[2 * o] = p0; h.draws_out[2 * o + 1] = p1; }
            }
            // K:100-125 for this slice of the pixels; the LDS reads are broadcasts (every lane of a slice the same row)
            const bool all_exact = !fc.use_fast || !(fabsf(hyp.x) < 1e15f && fabsf(hyp.y) < 1e15f);
            int inl = 0;
            for (int ti0 = slice; ti0 < tn; ti0 += 4 * S) {          // four independent evaluations per trip
                unsigned redo = all_exact ? 0xfu : 0u;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int ti = ti0 + u * S;
                    if (ti < tn) {
                        const TinyPix px = s_pix[ti];
                        const float dx = hyp.x - px.cx, dy = hyp.y - px.cy;
                        const float a1 = __builtin_fmaf(dx, px.nhx, dy * px.nhy);
                        const float b1 = __builtin_fmaf(dx, px.Bx, dy * px.By);
                        const float t = a1 - fabsf(b1);
                        inl += t > 0.f ? 1 : 0;
                        redo |= !(__builtin_fmaf(-fc.beta, a1, fabsf(t)) > fc.eps_abs) ? 1u << u : 0u;
                        redo |= t > 0.f ? 16u << u : 0u;           // what the fast test said, for the correction below
                    }
                }
                if (__builtin_expect((redo & 0xfu) != 0u, 0)) {
                    // inside the guard band (or a far hypothesis): the exact sequence decides (one copy of it, rolled)
#pragma unroll 1
                    for (int u = 0; u < 4; ++u) {
                        const int ti = ti0 + u * S;
                        if (!((redo >> u) & 1u) || ti >= tn) continue;
                        const int p = s_rowp[ti];
                        const int y = p / a.W;
                        const int exact = vote_exact((float)(p - y * a.W), (float)y, hyp.x, hyp.y, s_pix[ti].nx, s_pix[ti].ny, fc.thresh) ? 1 : 0;
                        inl += exact - (int)((redo >> (4 + u)) & 1u);
                    }
                }
            }
            if (inl) atomicAdd(&s_cnt[hl], inl);
        }
        __syncthreads();
        // running first-maximum (P:160): lower index wins ties; batches come in increasing index order
        if (slice == 0 && hi < h.hn) {
            const int c = s_cnt[hl];
            s_cnt[hl] = 0;                                       // for the next batch (barrier below)
            h.counts[(size_t)bk * h.hn + hi] = c;
            if (c > best) { best = c; besti = hi; besth = hyp; }
        }
        __syncthreads();
    }
    PVV_TSTAMP(5);
    // argmax over the threads (slice 0 holds candidates; the others carry best = -1)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int oc = __shfl_xor(best, o, 64), oi = __shfl_xor(besti, o, 64);
        const float ox = __shfl_xor(besth.x, o, 64), oy = __shfl_xor(besth.y, o, 64);
        if (oc > best || (oc == best && oi < besti)) { best = oc; besti = oi; besth = make_float2(ox, oy); }
    }
    if (lane == 0) { s_best[wave] = best; s_besti[wave] = besti; s_besth[wave] = besth; }
    __syncthreads();
    best = s_best[0]; besti = s_besti[0]; besth = s_besth[0];
#pragma unroll
    for (int w = 1; w < kTinyWaves; ++w)
        if (s_best[w] > best || (s_best[w] == best && s_besti[w] < besti)) { best = s_best[w]; besti = s_besti[w]; besth = s_besth[w]; }
    // P:162-167: all_win_ratio (0) < count/tn  <=>  count > 0; otherwise the winner stays (0,0)
    const float2 win = best > 0 ? besth : make_float2(0.f, 0.f);
    PVV_TSTAMP(6);

    // ---- P:176-191: re-vote the winner (exact sequence) and accumulate the normal equations in binary64
    double xx = 0, xy = 0, yy = 0, bx = 0, by = 0;
    for (int ti = threadIdx.x; ti < tn; ti += kTinyBlock) {
        const int p = s_rowp[ti];
        const int y = p / a.W;
        const float cx = (float)(p - y * a.W), cy = (float)y, dxv = s_pix[ti].nx, dyv = s_pix[ti].ny;
        if (!vote_exact(cx, cy, win.x, win.y, dxv, dyv, fc.thresh)) continue;
        const double nx = (double)dyv, ny = -(double)dxv;
        const double bb = nx * (double)cx + ny * (double)cy;
        xx += nx * nx; xy += nx * ny; yy += ny * ny;
        bx += nx * bb; by += ny * bb;
    }
    double r5[5] = {xx, xy, yy, bx, by};
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int i = 0; i < 5; ++i) r5[i] += __shfl_xor(r5[i], o, 64);
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 5; ++i) red5[wave * 5 + i] = r5[i];
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        double s = 0;
#pragma unroll
        for (int w = 0; w < kTinyWaves; ++w) s += red5[w * 5 + threadIdx.x];
        part[threadIdx.x] = s;
    }
    if (threadIdx.x == 0 && win_counts) win_counts[bk] = best;
    PVV_TSTAMP(7);
}
