// count_bf16.hpp -- stage 3: split-bf16 matrix-core prefilter inlier-count kernel (default).
// Part of the single translation unit pvnet_vote.hip (included inside its anonymous namespace); see that file
// for the numerical contract and the reference citations (K = ransac_voting_kernel.cu, P = ransac_voting_gpu.py).
#pragma once

// ---------------------------------------------------------------------------------------------
// Stage 3, matrix-core prefilter form (k_count_bf16).
//
// The sqrt/divide-free VALU kernel of round 1 (k_count_fast, removed) was bound by VALU issue (7.35 VALU instructions per evaluation, 84 % VALU busy); 4 of the 6.5
// useful ones are the two dot products a = d.nh and b' = kappa d x nh.  Those are bilinear in (hx,hy,1) and the
// pixel's (nh, -c.nh) / (B, -c.B): a rank-3 form.  The f32 MFMA shares the fp32 VALU datapath on gfx950
// (round 1's mfma_valu_overlap microbenchmark, profiles/r01_microbench.txt: no overlap), but the bf16 matrix core is a separate
// pipe that does overlap (tools/microbench/bf16_mfma_overlap.hip; how much: tools/microbench/count_pipe3.hip, round 6).  So every fp32 operand is split EXACTLY into three bf16
// pieces x = x0 + x1 + x2 (+ <= 2^-27 |x|), the six leading piece products of hx*nhx and of hy*nhy plus the three
// pieces of the constant are the 15 terms of a K=16 dot product, and ONE v_mfma_f32_32x32x16_bf16 delivers a
// (rows 0-15) and b' (rows 16-31) for 16 pixels x 32 hypotheses.  The VALU keeps t = a - |b'|, the sign-bit
// queue and the guard-band test: 21 instructions per 512 evaluations instead of 56 x 4.
//
// The MFMA result is only a PREFILTER: the decision is taken from it when |t| - beta*a > eps, otherwise that
// evaluation is redone with the exact binary32 sequence (K:100-125).  The hot loop tests a whole 16x32 tile against
// a per-hypothesis upper bound of beta*a + eps (one v_min3 chain); tiles that fail mark their in-band evaluations
// while a and t are in registers, and only those are re-decided after the 8 tiles of the hypothesis tile.
// Bound, with u = 2^-24, d = fl(h-c) as the exact path sees it, o = the block's integer origin, c' = c-o (exact),
// h' = fl(h-o), C1 = max |c'|_1 of the block:
//     piece residuals and dropped piece products          <= 0.5 u S,   S = |hx' nhx| + |hy' nhy| + |c'.nh|
//     bf16 MFMA accumulation (products exact in f32; 15 f32 roundings in any order)  <= 15 u S
//     (measured on MI355X: 3.9 u S including the split, bf16_mfma_overlap.hip)
//     fl(h-o), the exact path's fl(h-c), f32 unit normal (4u: v_rsq_f32, see the prologue), f32 c'.nh   
//  => |a_mfma - a_true| <= u (31 |d| + 35 C1);  for b' the operand B = kappa perp(nh) carries 6u instead of 4u:
//     kappa u (33 |d| + 35 C1);
//     beta = 1.25 (33 (1+kappa) + 8/(1-T^2)) u / T,     eps = 1.25 (1+kappa) 35 u C1 + eps_abs.
// Inlier counts stay bit-exact (tests/test_gpu_parity.py, every parity test runs through this kernel by default).
//
// Layout (MI355X_MICROARCH / verified in the microbenchmark): A operand lane l = row l%32, k = 8*(l/32)..+7;
// B operand lane l = column l%32, same k; D register r of lane l = row 4*(l/32) + r%4 + 8*(r/4), column l%32.
// Rows = (form, pixel) of a 16-pixel tile, columns = 32 hypotheses: every lane owns ONE hypothesis and 8 of the
// 16 pixels; lanes l and l^32 share a hypothesis and are merged in LDS.
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

#define kDdAlive __uint_as_float(0x2b8cbcceu)   // 1.0000002e-12f: the smallest dd with (double)sqrtf(dd) >= 1e-6
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

// B operand of hypothesis i of a group: lane l of tile ht holds column l%32, k = 8*(l/32)..+7 of
//   (qx0,qy0,qx1,qx2,qx0,qy1,qy2,qy0 || qx0,qy0,qx1,qy1, 1,1,1,0),   q = pieces of h' = fl(h - o);
// zeroes the hypothesis' LDS counter (flushed by the same thread later).  Returns 1 for a hypothesis that is non-finite
// or astronomically far (the whole group then takes the exact loop).
__device__ __forceinline__ int stage_hypothesis(bf16x8 *sB, int *sCnt, int i, float2 hp, float2 org, int cnt_init = 0)
{
    __bf16 qx[3], qy[3];
    split3(hp.x - org.x, qx);
    split3(hp.y - org.y, qy);
    const __bf16 one = (__bf16)1.f, zero = (__bf16)0.f;
    const bf16x8 lo8 = {qx[0], qy[0], qx[1], qx[2], qx[0], qy[1], qy[2], qy[0]};
    const bf16x8 hi8 = {qx[0], qy[0], qx[1], qy[1], one, one, one, zero};
    sB[(i >> 5) * 64 + (i & 31)] = lo8;
    sB[(i >> 5) * 64 + 32 + (i & 31)] = hi8;
    sCnt[i] = cnt_init;
    return !(fabsf(hp.x) < 1e15f && fabsf(hp.y) < 1e15f);
}

// phase timestamps of a few blocks, instrumented builds only (tools/build_variant.sh -DPVV_TUNING -DPVV_STAMPS +
// tools/phase_stamps.py); they cost registers, so timing sweeps use -DPVV_TUNING alone
#ifdef PVV_STAMPS
#define PVV_STAMP(i) do { if (dbg && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 97 || blockIdx.x == 401)) \
    dbg[((blockIdx.x == 0 ? 0 : (blockIdx.x == 97 ? 1 : 2)) * 16) + (i)] = wall_clock64(); } while (0)
// per-block census behind the three timelines, 16 words per block at dbg[64 + 16 * block]:
//   [0] entry, [1] exit (wall_clock64: the 100 MHz real-time counter), [2] hardware id (HW_ID | XCC_ID << 32), [3] items processed,
//   [4..11] SHADER cycles (s_memtime) thread 0 spent per phase -- kCensusPhases below --, [12] real-time ticks of the matrix-core
//   loop phase, [13] matrix-core tiles wave 0 multiplied.  Shader cycles / real time = the clock the chip ran at INSIDE the phase
//   (round 6, VERDICT r5 #1: the roofline's 2.4 GHz was never checked inside the kernels).
enum { kCsTable = 0, kCsItemPrologue, kCsAOperands, kCsNextGroup, kCsLoop, kCsLoopWait, kCsFlush, kCsOther, kCensusPhases };
#define PVV_CENSUS_IN() long long cs_last = 0; __shared__ long long s_cs[16]; \
    do { if (threadIdx.x == 0) { for (int i_ = 0; i_ < 16; ++i_) s_cs[i_] = 0; s_cs[0] = wall_clock64(); \
    s_cs[2] = (long long)__builtin_amdgcn_s_getreg(63492) | ((long long)__builtin_amdgcn_s_getreg(63508) << 32); \
    cs_last = (long long)__builtin_readcyclecounter(); } } while (0)
#define PVV_CS(i) do { if (threadIdx.x == 0) { const long long t_ = (long long)__builtin_readcyclecounter(); s_cs[4 + (i)] += t_ - cs_last; cs_last = t_; } } while (0)
#define PVV_CS_RT_IN() long long cs_rt = 0; do { if (threadIdx.x == 0) cs_rt = wall_clock64(); } while (0)
#define PVV_CS_RT_OUT(tiles) do { if (threadIdx.x == 0) { s_cs[12] += wall_clock64() - cs_rt; s_cs[13] += (tiles); } } while (0)
#define PVV_CENSUS_OUT(n) do { if (dbg && threadIdx.x == 0) { s_cs[1] = wall_clock64(); s_cs[3] = (n); \
    for (int i_ = 0; i_ < 16; ++i_) dbg[64 + 16 * (size_t)blockIdx.x + i_] = s_cs[i_]; } } while (0)
#else
#define PVV_STAMP(i) do { } while (0)
#define PVV_CENSUS_IN() do { } while (0)
#define PVV_CS(i) do { } while (0)
#define PVV_CS_RT_IN() do { } while (0)
#define PVV_CS_RT_OUT(tiles) do { } while (0)
#define PVV_CENSUS_OUT(n) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------
// Staged counting (ransac_voting_layer_v3 only): the layer needs the ARG-MAX of the counts and the winner's count
// (P:160-167), not the counts.  Two launches instead of one:
//   k_count_bf16<kCountFirst>   every hypothesis over the 512-pixel chunks whose index has a residue (mod M) in `mask`
//                               -- a quarter of an image's chunks, spread over the object;
//   k_lead (count_prune.hpp)    two leaders per (image, keypoint): their partial counts plus their SURE inliers among all
//                               the other pixels = lower bounds of two full counts; L* = the larger;
//   k_count_bf16<kCountFilter>  the other chunks, but only for hypotheses h with  partial(h) + R >= L*  (R = pixels not
//                               counted by the first launch): every other hypothesis has full(h) <= partial(h) + R < L*
//                               <= max and can neither win nor tie.  The filter is evaluated on the fly while a block
//                               stages its hypotheses (count, bound and four leader numbers per (image, keypoint): no
//                               survivor lists in memory); it reads counters that other blocks of the same launch are
//                               adding to, which is benign: a dropped hypothesis is never counted, so its counter never
//                               moves, and a kept one only grows -- the predicate is the same for every reader.
// Kept hypotheses end with their exact full count, dropped ones with a partial count below the maximum: winner, first-index
// tie rule and winner count are those of the full pass, bit for bit.  Images of fewer than n_min chunks are counted
// completely by the first launch (nothing to gain from a bound taken after half of their pixels).
// ---------------------------------------------------------------------------------------------
enum { kCountFull = 0, kCountFirst = 1, kCountFilter = 2 };

// The chunk schedule, compile-time (every index computation below is then shifts and constant multiplies; as run-time
// parameters they were a dozen integer divisions per work item): the first launch counts the 512-pixel chunks c with
// (c mod 8) in {1, 5} -- a quarter of the chunks, spread over the object (rows of the compacted list = raster order) --,
// the second launch the other residues; images of fewer than kStageMinChunks chunks are counted completely by the first.
#ifndef PVV_STAGE_M                    // (tools/build_variant.sh -DPVV_STAGE_M=.. -DPVV_STAGE_FIRST=..: schedule sweeps)
#define PVV_STAGE_M 8
#define PVV_STAGE_FIRST 0x22u
#endif
constexpr int kStageM = PVV_STAGE_M;
constexpr uint32_t kStageFirst = PVV_STAGE_FIRST;
// Round 4: with the second launch eliminating as it goes (count_filter_runs.hpp) a SMALLER first stage pays once its runs are
// long: an eighth of the chunks ({1} of 8) instead of a quarter is -4 % per call at config 3 / B = 128 and -9 % on config 5 /
// B = 16, +3 % at B = 64 and below (one-process A/B, profiles/DESIGN_rounds_1-4.md 4.7).  The schedule is a template parameter of the three kernels
// of the pass; the host picks it from the problem's size (launch_count_bf16).
constexpr uint32_t kStageFirstEighth = 0x02u;
constexpr uint32_t stage_rest_of(uint32_t first) { return ((1u << kStageM) - 1u) & ~first; }
constexpr int kStageMinChunks = 8;

struct StageArgs {
    const int *lead;      // kCountFilter: [B,K,8] leaders' partial counts [0..3] (-1: none) and their SURE inliers among the
                          // pixels the first launch did not count [4..7] (k_lead)
    int *any_staged;      // one word: kCountFirst writes whether ANY image is staged; k_lead and kCountFilter leave at once
                          // when none is (a batch of small masks then pays two empty launches, not two table builds)
    int *miss;            // k_count_filter_runs: [B,K,hn] misses proven so far among the pixels the first launch did not count
                          // (zeroed by k_compact_hyp; count_filter_runs.hpp)
    int sub_tenth;        // 1: the caller is estimate_voting_distribution_with_mean, which weighs every hypothesis whose ratio is
                          // within 0.1 of the best (P:262-264): the elimination bound is lowered accordingly (stage_bound)
    const float2 *mean;   // k_count_filter_runs, the estimate: [B,K] the keypoints the covariance is taken about, or nullptr.  Not
                          // nullptr: a run walks its chunks NEAREST (in y) to the keypoint first -- see the chunk order there
    int hstride;          // row length of hyps / counts / miss when the hn hypotheses counted are a column range of longer rows (the
                          // fused un_pnp call counts its 512 + 4096 hypotheses as two passes over one compaction); 0: hn
#ifdef PVV_STAMPS
    long long *dbg;       // instrumented builds: the phase census of k_count_filter_runs (count_filter_runs.hpp)
#endif
};

// The count a hypothesis must still be able to reach to matter.  ransac_voting_layer_v3 keeps the arg-max: L* itself.  The
// estimate zeroes every ratio below  max ratio - 0.1  in binary32 (k_covariance: thr = fl(fl(mx)/fl(tn)) - 0.1f, r = fl(c / tn),
// r < thr -> 0): a hypothesis whose full count is below  L* - 0.1 tn - margin  has weight zero with its full count and with any
// partial count (r is monotone in c), so it may be dropped.  margin: 0.1f = 0.1000000015 and three binary32 roundings of
// ratios <= 1 are < 4e-7 in ratio units = tn * 4e-7 counts; ceil(tn / 10) + 2 + tn / 2^20 covers both for every tn.
__device__ __forceinline__ int stage_bound(int lstar, int tn, int sub_tenth)
{
    if (sub_tenth && lstar >= 0) lstar -= (tn + 9) / 10 + 2 + (tn >> 20);
    return lstar;
}

// chunks with a residue in `mask` among the first n chunks, and the j-th of them (residues present in a last, partial
// period are a prefix of the sorted residues, so the j-th chunk does not depend on n)
template <uint32_t MASK>
__device__ __forceinline__ int stage_chunks(int n)
{
    return (n / kStageM) * __builtin_popcount(MASK) + __popc(MASK & ((1u << (n % kStageM)) - 1u));
}
template <uint32_t MASK>
__device__ __forceinline__ int stage_chunk_at(int j)
{
    constexpr int m = __builtin_popcount(MASK);
    const int per = j / m;
    uint32_t t = MASK;
    for (int k = j - per * m; k > 0; --k) t &= t - 1u;
    return per * kStageM + __builtin_ctz(t);
}
// pixels of an image of tn pixels (nch chunks of PC) that lie in chunks with a residue in MASK
template <uint32_t MASK>
__device__ __forceinline__ int stage_pixels(int tn, int nch, int PC)
{
    return stage_chunks<MASK>(nch) * PC - (((MASK >> ((nch - 1) % kStageM)) & 1u) ? nch * PC - tn : 0);
}

// 5 blocks (= 5 waves per SIMD) per CU: <= 96 VGPRs and 30 KB of LDS per block; measured -4.4 % against 4
template <int MODE, uint32_t FIRST = kStageFirst>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_count_bf16(
    const float2 *__restrict__ coords /*[B,cap]*/, const float2 *__restrict__ dirs /*[B,K,cap]*/,
    const float2 *__restrict__ hyps /*[B,K,hn]*/, int *__restrict__ counts /*[B,K,hn]*/,
    const int *__restrict__ tn_arr, int B, int K, int hn, int cap, float thresh, Bf16Consts fc, int target_items,
    long long *__restrict__ dbg /*tuning builds: phase timestamps; nullptr otherwise*/, StageArgs sa)
{
    PVV_STAMP(0);
    PVV_CENSUS_IN();
    int n_items_done = 0;
    __shared__ int chunk_end[kMaxBatchLds];         // inclusive prefix of the 512-pixel chunks per image
    __shared__ int s_htpi, s_gpi, s_chunks;
    __shared__ bf16x8 sB[kBfMaxHt * 64];            // B operands of the current hypothesis group (16 KB)
    __shared__ float4 sP[4 * kBfPixPerWave];        // per pixel: (nhx, nhy, c'x, c'y); nhx = NaN: can never vote  (8 KB)
    __shared__ int sCnt[kBfMaxHt * 32];
    __shared__ float sRed[4];
    __shared__ unsigned long long s_small[kMaxBatchLds / 64];   // staged: images counted completely by the first launch
    __shared__ int s_keep[8];                                   // kCountFilter: kept hypotheses per (pass, wave) of a group
    constexpr bool STAGED = MODE != kCountFull;
    constexpr bool FILTER = MODE == kCountFilter;
    constexpr uint32_t REST = stage_rest_of(FIRST);
    const int lane = lane_id(), wave = wave_id();
    constexpr int PC = 4 * kBfPixPerWave;
    const int nt = (hn + 31) >> 5;                  // 32-hypothesis tiles per keypoint
    const int hs = sa.hstride > 0 ? sa.hstride : hn; // row length of hyps / counts
    if constexpr (FILTER) if (*sa.any_staged == 0) return;

    // Work item = (image, keypoint, 512-pixel chunk, a run of hypothesis groups).  A group is up to 16 tiles (512
    // hypotheses, what fits the LDS staging); an item walks as many groups as possible (the pixel operands are
    // built once per item), fewer -- and smaller groups -- when the batch is too small to fill the chip.
    // ONE pass over tn[] by wave 0 gives the chunk prefix of the images (the item table: every block derives the same
    // one, no host sync) and, from the total, the item size.
    if (wave == 0) {
        int carry = 0;
        [[maybe_unused]] bool any_staged = false;
        for (int b0 = 0; b0 < B; b0 += 64) {
            const int b = b0 + lane;
            int inc = b < B ? (tn_arr[b] + PC - 1) / PC : 0;
            if constexpr (STAGED) {                                // this launch's chunks of the image
                const bool small = inc < kStageMinChunks;
                const unsigned long long sm = __ballot(small);
                any_staged |= sm != ~0ull;
                if (lane == 0) s_small[b0 >> 6] = sm;
                inc = small ? (MODE == kCountFirst ? inc : 0) : stage_chunks<MODE == kCountFirst ? FIRST : REST>(inc);
            }
            inc = wave_incl_scan(inc) + carry;
            if (b < B) chunk_end[b] = inc;
            carry = __builtin_amdgcn_readlane(inc, 63);
        }
        const long long chunks = (long long)carry * K;
        int htpi = min(nt, kBfMaxHt);
        int nhg = (nt + htpi - 1) / htpi;
        int gpi = nhg;                                              // groups per item
        // Long items (several 512-hypothesis groups behind one pixel-operand build) are split until there are THREE generations'
        // worth of them when the grid has that many blocks (B > 8: 15 or 48 per CU) -- 1296 eight-group items on 1280 block slots
        // (the 4096-hypothesis estimate at B = 12) leave 16 items a second round to themselves.  One-process A/B, count kernel,
        // 480x640, K = 9 (profiles/r04_experiments.txt (19)): 4096 hypotheses -19 % at B = 12, -10 % at 16, -7 % at 24, -4 % at 32,
        // +-0 from 48 on; 2048 hypotheses -7.5 % at B = 16; config 5's staged first launch -3 % at B = 16.  Not below B = 9 (the
        // one-generation grid walks its items round-robin: twice as many half-length items cost +9 % on config 5 at B = 2, and the
        // estimate at B = 8) and not for two groups (1024 hypotheses at B = 16: +1.6 %).
#ifndef PVV_GPI_FACTOR
#define PVV_GPI_FACTOR 3
#endif
        const long long gpi_target = (B > 8 && nhg >= 4) ? (long long)PVV_GPI_FACTOR * target_items : (long long)target_items;
        while (gpi > 1 && chunks * ((nhg + gpi - 1) / gpi) < gpi_target) gpi = (gpi + 1) >> 1;
        // groups below 4 tiles only for very small problems (< 128 (chunk, keypoint) pairs: every item they add is another
        // CU put to work); otherwise the prologue of an item is worth more than the 16 matrix-core tiles of a 2-tile group
        const int htpi_min = chunks < 128 ? 2 : 4;
        if (gpi == 1 && !FILTER)                                    // (the filter compacts groups of 512 hypotheses: no smaller ones)
            while (htpi > htpi_min && chunks * ((nt + htpi - 1) / htpi) < target_items) htpi = (htpi + 1) >> 1;
        if (lane == 0) { s_htpi = htpi; s_gpi = gpi; s_chunks = carry; }
        if constexpr (MODE == kCountFirst) if (blockIdx.x == 0 && lane == 0) *sa.any_staged = any_staged ? 1 : 0;
    }
    __syncthreads();
    PVV_STAMP(1);
    const int htpi = __builtin_amdgcn_readfirstlane(s_htpi);
    const int gpi = __builtin_amdgcn_readfirstlane(s_gpi);
    const int nhg = (nt + htpi - 1) / htpi;          // hypothesis groups per keypoint
    const int nruns = (nhg + gpi - 1) / gpi;         // runs of groups = items per (chunk, keypoint)
    const int per_chunk = K * nruns;
    const int total = __builtin_amdgcn_readfirstlane(s_chunks) * per_chunk;
    const int col = lane & 31, kslice = lane >> 5;
    PVV_STAMP(2);
    PVV_CS(kCsTable);

    // The host sizes the grid without knowing tn.  With few items (up to 7/4 of the one-generation grid = target_items)
    // only the first target_items blocks of a 15-per-CU grid take part -- one generation, a few blocks with two items,
    // measured faster than spreading them over three generations (B = 16: 54.9 vs 57.9 us) -- and the others leave here
    // (they pass through the slots the working blocks free: no measurable tail).  With many items every block works
    // (config 5 at B = 16: 1.43 vs 1.70 ms), and so does the 48-per-CU grid of the long-hypothesis configurations
    // (11 000 leaving blocks would be a tail: +10 % on config 4 at B = 16).
    const int nblk = (total <= target_items + (target_items >> 1) + (target_items >> 2) && (int)gridDim.x > target_items &&
                      (int)gridDim.x <= 3 * target_items) ? target_items : (int)gridDim.x;
    if ((int)blockIdx.x >= nblk) return;
    // item -> (chunk over the whole batch, rest) without a division per item: both advance by a constant step
    const int step_q = nblk / per_chunk, step_r = nblk - step_q * per_chunk;
    int gchunk = (int)blockIdx.x / per_chunk, rem = (int)blockIdx.x - gchunk * per_chunk;
    for (int item = blockIdx.x; item < total;
         item += nblk, gchunk += step_q + (rem + step_r >= per_chunk ? 1 : 0), rem += step_r - (rem + step_r >= per_chunk ? per_chunk : 0)) {
        int local;
        const int b = locate_item(chunk_end, B, gchunk, &local);   // image, and the chunk's index within it (this stage's)
        int chunk = local;
        if constexpr (STAGED)
            if (!((s_small[b >> 6] >> (b & 63)) & 1ull)) chunk = stage_chunk_at<MODE == kCountFirst ? FIRST : REST>(local);
        const int vi = nruns == 1 ? rem : rem / nruns;          // (one run per (chunk, keypoint) unless the batch is tiny)
        const int run = rem - vi * nruns;
        const int bk = b * K + vi;
        const float2 *hyp_k = hyps + (size_t)bk * hs;
        const float2 *crd = coords + (size_t)b * cap;
        const float2 *dir_k = dirs + (size_t)bk * cap;
        const int pb = chunk * PC;                              // first pixel of the block's chunk (< tn)
        const int g0 = run * gpi, g1 = min(nhg, g0 + gpi);

        __syncthreads();                                        // previous item's LDS fully consumed
        // The thread index, made opaque once per item: everything the prologue derives from it (addresses, row / slot
        // indices) is then recomputed per item instead of being hoisted out of the item loop and kept alive -- spilled --
        // across the matrix-core loop.  At the 96-register cap this removes every scratch access from the full and the
        // first-stage kernels (16 B -> 0) and a few from the filter kernel (52 B -> 44 B); -0.6 % per call at B = 64, -1 % at B = 1.
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        PVV_STAMP(3);
        PVV_CS(kCsOther);
        ++n_items_done;
        // ---- every global load of the item is issued here, before anything waits: tn, the chunk's origin, two pixels
        //      per thread (rows beyond tn are read -- the arrays reserve cap rows -- and masked below) and the
        //      hypotheses of the first group.  One memory round trip instead of three.
        const int tn_v = tn_arr[b];
        int lead_p = -1, lead_r = 0;                            // kCountFilter: a leader's partial count and its sure inliers in the rest
        if constexpr (FILTER) {
            lead_p = sa.lead[(size_t)bk * 8 + (lane & 3)];
            lead_r = sa.lead[(size_t)bk * 8 + 4 + (lane & 3)];
        }
        const float2 org = crd[pb];                             // integer origin: the chunk's first pixel
        float2 pc[2], pd[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int p = min(pb + tid + q * kBlock, cap - 1);
            pc[q] = crd[p];
            pd[q] = dir_k[p];
        }
        const int nht0 = min(nt, (g0 + 1) * htpi) - g0 * htpi;  // tiles of the first group
        float2 hp0[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = tid + q * kBlock;
            const int h = (g0 * htpi + (i >> 5)) * 32 + (i & 31);
            hp0[q] = (i < nht0 * 32 && h < hn) ? hyp_k[h] : make_float2(0.f, 0.f);
        }
        int cnt0[2] = {0, 0};                                   // kCountFilter: the first launch's counts of those hypotheses
        if constexpr (FILTER) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int i = tid + q * kBlock;
                const int h = g0 * htpi * 32 + i;
                if (i < nht0 * 32 && h < hn) cnt0[q] = counts[(size_t)bk * hs + h];
            }
        }
        const int tn = __builtin_amdgcn_readfirstlane(tn_v);
        // kCountFilter: R = pixels of the image the first launch did not count; L* = the larger of the leaders' lower bounds (k_lead)
        int R_rem = 0, lstar = 0, ns_g = 0;
        if constexpr (FILTER) {
            R_rem = stage_pixels<REST>(tn, (tn + PC - 1) / PC, PC);
            int full = lead_p >= 0 ? lead_p + lead_r : -1;
            full = max(full, PVV_DPP(full, full, 0xB1, 0xf, false));   // quad_perm [1,0,3,2]
            full = max(full, PVV_DPP(full, full, 0x4E, 0xf, false));   // quad_perm [2,3,0,1]: lanes 0-3 hold the four leaders' maximum
            lstar = stage_bound(__builtin_amdgcn_readfirstlane(full), tn, sa.sub_tenth);
        }
        // kCountFilter: stage only the hypotheses of group g that can still reach L*, densely from slot 0 (order = pass,
        // wave, lane: irrelevant, the counter word carries the hypothesis' index within the group in its high half); pads
        // the last tile with (0,0) hypotheses.  Contains one barrier; returns the far flag of stage_hypothesis.
        auto stage_kept = [&](int g, const int (&cnt)[2], const float2 (&hp)[2]) -> int {
            bool keep[2];
            unsigned long long m[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int i = tid + q * kBlock;
                keep[q] = i < htpi * 32 && g * htpi * 32 + i < hn && cnt[q] + R_rem >= lstar;
                m[q] = __ballot(keep[q]);
                if (lane == 0) s_keep[q * 4 + wave] = __popcll(m[q]);
            }
            __syncthreads();
            int tot = 0, base[2] = {0, 0};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int v = s_keep[j];
                tot += v;
                if (j < wave) base[0] += v;
                if (j < 4 + wave) base[1] += v;
            }
            int f = 0;
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (keep[q])
                    f |= stage_hypothesis(sB, sCnt, base[q] + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m[q] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m[q], 0u)), hp[q], org,
                                          (int)(tid + q * kBlock) << 16);
            const int pad = ((tot + 31) & ~31) - tot;
            if (tid < pad) stage_hypothesis(sB, sCnt, tot + tid, make_float2(0.f, 0.f), org, 0);
            ns_g = tot;
            return f;
        };

        // ---- per pixel (two per thread): the f32 unit normal and the translated coordinates (16 bytes of LDS; the
        //      kappa-scaled perpendicular and the constants -(c-o).nh, -(c-o).B are formed where they are used).  A pixel
        //      the exact test can never accept (K:121 norm1 < 1e-6, a non-finite norm1) or beyond tn is marked by
        //      nhx = NaN and becomes nh = B = 0, constant -1e30 in the A operand: a = -1e30, b' = 0, t < 0.
        float c1 = 0.f;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int pl = tid + q * kBlock, p = pb + pl;
            float4 rec = make_float4(__builtin_nanf(""), 0.f, 0.f, 0.f);
            if (p < tn) {
                const float2 c = pc[q], d = pd[q];
                const float cx = c.x - org.x, cy = c.y - org.y;  // exact (integers)
                c1 = fmaxf(c1, fabsf(cx) + fabsf(cy));
                // unit normal by v_rsq_f32 (one transcendental and two multiplies instead of a correctly rounded square root
                // and two divisions: ~60 VALU instructions less per wave and item, -1.6 % per call).  Its components are
                // within 4u of the true ones (dd 2u -> 1u, v_rsq_f32 <= 0.8633 ulp = 1.73u measured over ALL normal inputs,
                // the product 1u; 2.72u measured over 4e9 directions: tools/microbench/rsq_accuracy.hip), which is what
                // the guard bands assume (bf16_consts).  The exact test's reject  (double)sqrtf(dd) < 1e-6  (K:121) is a
                // threshold on dd itself, sqrtf being monotone: kDdAlive is the smallest binary32 it accepts
                // (tests/test_band_model.py recomputes it).
                const float dd = d.x * d.x + d.y * d.y;
                if (dd >= kDdAlive && dd < INFINITY) {
                    const float rinv = __builtin_amdgcn_rsqf(dd);
                    rec = make_float4(d.x * rinv, d.y * rinv, cx, cy);
                }
            }
            sP[pl] = rec;
        }
        c1 = wave_max(c1);
        if (lane == 0) sRed[wave] = c1;
        // ---- B operands of the first group (see stage_group below for the layout)
        int far = 0;
        if constexpr (FILTER) {
            far = stage_kept(g0, cnt0, hp0);
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int i = tid + q * kBlock;
                if (i < nht0 * 32) far |= stage_hypothesis(sB, sCnt, i, hp0[q], org);
            }
        }
        far = __syncthreads_or(far);
        PVV_STAMP(4);
        PVV_CS(kCsItemPrologue);
        const float C1 = fmaxf(fmaxf(sRed[0], sRed[1]), fmaxf(sRed[2], sRed[3]));
        const float eps = fc.eps0 + fc.eps_c * C1;
        const float epsw = __builtin_fmaf(fc.beta * 1.02f, C1, eps);   // band half-width at |h'| = 0

        // 16-pixel tiles of the chunk go round-robin to the 4 waves (tile 4*j + wave is this wave's j-th): a partial
        // chunk keeps all four waves busy and nobody multiplies tiles that hold no pixel
        const int ntile_c = (min(tn - pb, PC) + 15) >> 4;       // tiles of the chunk with at least one pixel (1..32)
        const int ntile_w = (ntile_c - wave + 3) >> 2;          // this wave's share (0..8)

        // ---- A operands: lane l = row l%32 (form = row/16, pixel = row%16), k = 8*(l/32)..+7 of
        //      (vx0,vy0,vx0,vx0,vx1,vy0,vy0,vy1 || vx2,vy2,vx1,vy1, cv0,cv1,cv2, 0); built once per item.  The order of
        //      the 15 terms is free; this one puts (qx0,qy0) first in BOTH k halves of the B operand (band width below).
        //      Lanes l and l+32 need the two k halves of the SAME row, so they share the work: the lower lane splits the
        //      row for tiles 0-3, the upper one for tiles 4-7, each forms both halves, and one v_permlane32_swap per
        //      register hands the partner its half.
        bf16x8 A[8];
        if (ntile_w > 0) {
            const int form = (lane >> 4) & 1, prow = lane & 15;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int j = kslice * 4 + i;
                const float4 rec = sP[(j * 4 + wave) * 16 + prow];
                // row of the a-form: (nh, -c'.nh); of the b'-form: (B, -c'.B) with B = kappa perp(nh)
                const bool dead = rec.x != rec.x;
                float rx = form ? -fc.kappa * rec.y : rec.x;
                float ry = form ? fc.kappa * rec.x : rec.y;
                float rz = -(rec.z * rx + rec.w * ry);
                if (dead) { rx = 0.f; ry = 0.f; rz = form ? 0.f : -1e30f; }
                __bf16 vx[3], vy[3], cv[3];
                split3(rx, vx);
                split3(ry, vy);
                split3(rz, cv);
                const __bf16 zero = (__bf16)0.f;
                const bf16x8 lo8 = {vx[0], vy[0], vx[0], vx[0], vx[1], vy[0], vy[0], vy[1]};
                const bf16x8 hi8 = {vx[2], vy[2], vx[1], vy[1], cv[0], cv[1], cv[2], zero};
                uint4 x = __builtin_bit_cast(uint4, lo8), y = __builtin_bit_cast(uint4, hi8);
                // swap(x, y): lanes 32-63 of x <-> lanes 0-31 of y.  Afterwards x = this lane's k half of tile i (own lo8
                // below 32, the partner's hi8 above), y = this lane's k half of tile 4 + i
                {
                    const auto r0 = __builtin_amdgcn_permlane32_swap(x.x, y.x, false, false);
                    const auto r1 = __builtin_amdgcn_permlane32_swap(x.y, y.y, false, false);
                    const auto r2 = __builtin_amdgcn_permlane32_swap(x.z, y.z, false, false);
                    const auto r3 = __builtin_amdgcn_permlane32_swap(x.w, y.w, false, false);
                    x = make_uint4(r0[0], r1[0], r2[0], r3[0]);
                    y = make_uint4(r0[1], r1[1], r2[1], r3[1]);
                }
                A[i] = __builtin_bit_cast(bf16x8, x);
                A[4 + i] = __builtin_bit_cast(bf16x8, y);
            }
        }
        PVV_STAMP(5);
        PVV_CS(kCsAOperands);
        const int ebase = kslice * 4;                            // this lane's pixels: ebase + e%4 + 8*(e/4)
        const float16v zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

        for (int g = g0; g < g1; ++g) {
            const int ht0 = g * htpi;
            if (g > g0) {
                // ---- B operands of the next group (the first group's were staged with the pixels)
                far = 0;
                if constexpr (FILTER) {
                    int cnt[2] = {0, 0};
                    float2 hp[2] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f)};
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int i = threadIdx.x + q * kBlock;
                        const int h = ht0 * 32 + i;
                        if (i < htpi * 32 && h < hn) { cnt[q] = counts[(size_t)bk * hs + h]; hp[q] = hyp_k[h]; }
                    }
                    far = stage_kept(g, cnt, hp);
                } else {
                    const int nhtg = min(nt, ht0 + htpi) - ht0;
                    for (int i = threadIdx.x; i < nhtg * 32; i += kBlock) {
                        const int h = (ht0 + (i >> 5)) * 32 + (i & 31);
                        far |= stage_hypothesis(sB, sCnt, i, h < hn ? hyp_k[h] : make_float2(0.f, 0.f), org);
                    }
                }
                far = __syncthreads_or(far);
            }
            // tiles of this group: its kept hypotheses (filter) or all of them; slot s of the staging holds hypothesis
            // ht0*32 + s, or -- filter -- ht0*32 + (sCnt[s] >> 16)
            const int nht = FILTER ? (ns_g + 31) >> 5 : min(nt, ht0 + htpi) - ht0;
            const int nslot = FILTER ? ns_g : min(hn - ht0 * 32, nht * 32);
            PVV_STAMP(6);
            PVV_CS(kCsNextGroup);
            PVV_CS_RT_IN();

            if (__builtin_expect(far, 0)) {
                // some hypothesis of the group is non-finite / astronomically far: exact loop (K:100-125)
                for (int ht = 0; ht < nht; ++ht) {
                    const int slot = ht * 32 + col;
                    if (slot >= nslot) continue;
                    const float2 hp = hyp_k[ht0 * 32 + (FILTER ? sCnt[slot] >> 16 : slot)];
                    int inl = 0;
                    for (int p = pb + wave * 2 + kslice; p < min(tn, pb + PC); p += 8) {
                        const float2 c = crd[p], d = dir_k[p];
                        inl += vote_exact(c.x, c.y, hp.x, hp.y, d.x, d.y, thresh) ? 1 : 0;
                    }
                    if (inl) atomicAdd(&sCnt[ht * 32 + col], inl);
                }
            } else if (ntile_w > 0) {
                for (int ht = 0; ht < nht; ++ht) {
                    const bf16x8 Bop = sB[ht * 64 + lane];
                    unsigned flagged = 0u;                            // wave-uniform: tiles with an evaluation in the band
                    // conservative band half-width of this lane's hypothesis for the whole item: a = d.nh <= |d|_2 <=
                    // |h'|_1 + |c'|_1 <= (|qx0| + |qy0|) (1 + 2^-8) + C1 -- the leading bf16 pieces are the first two elements
                    // of either half of the B operand; the 2 % slack covers that and the roundings of a
                    const unsigned q01 = __builtin_bit_cast(uint4, Bop).x;
                    const float wband = __builtin_fmaf(fc.beta * 1.02f,
                                                       fabsf(__uint_as_float(q01 << 16)) + fabsf(__uint_as_float(q01 & 0xffff0000u)),
                                                       epsw);
                    unsigned qs[2] = {0u, 0u};                        // sign-bit queues, one per 4 tiles (newest evaluation = bit 0)
                    unsigned mb[2] = {0u, 0u};                        // in-band evaluations of flagged tiles: bit 8*(j%4) + e
                    // ntile_w made opaque per hypothesis tile: the seven "j < ntile_w" below are then scalar compares in
                    // place instead of seven SGPR pairs computed per item and kept alive across the loop (26 instead of 41
                    // spilled scalars in the full kernel, 53 instead of 64 in the filter kernel; -0.1 ... -1.4 % per call)
                    int ntw = ntile_w;
                    asm volatile("" : "+s"(ntw));
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if (j < ntw) {
                            const float16v acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[j], Bop, zero16, 0, 0, 0);
                            // conservative band test per tile: min |t|  vs  beta * (bound on a) + eps
                            float tmin = INFINITY;
                            float t[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                t[e] = acc[e] - fabsf(acc[8 + e]);
                                qs[j >> 2] = __builtin_amdgcn_alignbit(qs[j >> 2], __float_as_uint(t[e]), 31);
                                tmin = fminf(tmin, fabsf(t[e]));
                            }
                            if (__builtin_expect(__ballot(tmin <= wband) != 0, 0)) {
                                // rare (a few % of the tiles): mark the evaluations that really are inside the band,
                                // |t| - beta a <= eps, while a and t are still in registers; they are re-decided below
                                // (sign bit of eps - z is set exactly when z > eps: the same bit queue as for t, 3 VALU per
                                // evaluation; evaluation e of the tile ends up in bit 7 - e, set = NOT in the band)
                                unsigned out_of_band = 0u;
#pragma unroll
                                for (int e = 0; e < 8; ++e)
                                    out_of_band = __builtin_amdgcn_alignbit(
                                        out_of_band, __float_as_uint(eps - __builtin_fmaf(-fc.beta, acc[e], fabsf(t[e]))), 31);
                                mb[j >> 2] |= (~out_of_band & 0xffu) << (8 * (j & 3));
                                flagged |= 1u << j;
                            }
                        }
                    }
                    int inl = 8 * ntile_w - __popc(qs[0]) - __popc(qs[1]);   // sign bit set = not an inlier
                    // flagged tiles (a third of the iterations has one) mostly hold no evaluation that really is in the band;
                    // only then the un-translated hypothesis is fetched (not ahead of the tiles either: that prefetch costs
                    // more issue slots in every iteration than the stall does in a sixth of them, measured +1.6 %)
                    if (__builtin_expect(flagged != 0u, 0) && __any((mb[0] | mb[1]) != 0u)) {
                        const int slot = ht * 32 + col;
                        const float2 hp = slot < nslot ? hyp_k[ht0 * 32 + (FILTER ? sCnt[slot] >> 16 : slot)] : make_float2(0.f, 0.f);
                        do {
                            // re-decide the marked evaluations of tile j exactly (K:100-125)
                            const int j = __builtin_ctz(flagged);
                            flagged &= flagged - 1;
                            const unsigned m = mb[j >> 2] >> (8 * (j & 3));
                            if (!__any((m & 0xffu) != 0u)) continue;
                            // position of evaluation (j, e) in its sign queue: tiles pushed after it, 8 bits each, then 7 - e
                            const int after = min(ntile_w - (j & 4), 4) - 1 - (j & 3);
                            const unsigned sgn = qs[j >> 2] >> (8 * after);
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const bool marked = (m >> (7 - e)) & 1u;
                                if (!__any(marked)) continue;
                                const int prow = (j * 4 + wave) * 16 + ebase + (e & 3) + 8 * (e >> 2);
                                const int p = pb + prow;
                                const int fast = ((sgn >> (7 - e)) & 1u) ? 0 : 1;
                                // second level: the sqrt/divide-free test of round 1 (k_count_fast) on d = fl(h - c) (the exact path's own
                                // d) with the f32 unit normal from LDS; its band (beta2, eps0) is ~10x narrower than the MFMA's
                                const float4 rec = sP[prow];
                                const float dx = hp.x - (rec.z + org.x), dy = hp.y - (rec.w + org.y);
                                const float a2 = __builtin_fmaf(dx, rec.x, dy * rec.y);
                                const float b2 = __builtin_fmaf(dx, -fc.kappa * rec.y, dy * (fc.kappa * rec.x));
                                const float t2 = a2 - fabsf(b2);
                                int decided = t2 > 0.f ? 1 : 0;
                                const bool unsure = marked && (!(__builtin_fmaf(-fc.beta2, a2, fabsf(t2)) > fc.eps0) || rec.x != rec.x);
                                if (__any(unsure)) {
                                    int exact = 0;
                                    if (unsure && p < tn) {
                                        const float2 c = crd[p], d = dir_k[p];
                                        exact = vote_exact(c.x, c.y, hp.x, hp.y, d.x, d.y, thresh) ? 1 : 0;
                                    }
                                    if (unsure) decided = exact;
                                }
                                if (p >= tn) decided = 0;
                                if (marked) inl += decided - fast;
                            }
                        } while (flagged != 0u);
                    }
                    if (inl) atomicAdd(&sCnt[ht * 32 + col], inl);    // LDS: 2 lanes x 4 waves per hypothesis
                }
            }
            PVV_STAMP(7);
            PVV_CS(kCsLoop);
            PVV_CS_RT_OUT(nht * ntile_w);
            __syncthreads();
            PVV_CS(kCsLoopWait);
            for (int i = threadIdx.x; i < nslot; i += kBlock) {
                const int v = sCnt[i];
                const int c = FILTER ? v & 0xffff : v;               // (filter: index within the group << 16 | count <= 512)
                if (c != 0) atomicAdd(&counts[(size_t)bk * hs + ht0 * 32 + (FILTER ? v >> 16 : i)], c);
            }
            PVV_STAMP(8);
            PVV_CS(kCsFlush);
        }
    }
    PVV_STAMP(9);
    PVV_CS(kCsOther);
    PVV_CENSUS_OUT(n_items_done);
    (void)n_items_done;
}
