// count_filter_runs.hpp -- stage 3, second launch of the staged count (ransac_voting_layer_v3; the estimate when forced), round 4:
// work items that OWN A RUN of an (image, keypoint)'s remaining chunks, with progressive elimination inside the run.
// Part of the single translation unit pvnet_vote.hip (included inside its anonymous namespace, after count_bf16.hpp);
// see that file for the numerical contract and the matrix-core prefilter, count_prune.hpp for the bound L*.
#pragma once

// ---------------------------------------------------------------------------------------------
// Round 3's second launch (k_count_bf16<kCountFilter>) took one 512-pixel chunk per work item: every item loaded the 512
// counters and hypotheses of its (image, keypoint), evaluated the keep predicate  partial(h) + R >= L*, compacted and
// staged the survivors, built its pixel operands, multiplied three hypothesis tiles and flushed -- ~570 VALU instructions
// of prologue around ~600 in the matrix-core loop (profiles/r03_summary.json: as many VALU instructions as the first
// launch for a fifth of its matrix-core work per item).  Here an item is (image, keypoint, a RUN of R consecutive remaining
// chunks):
//   * the predicate, the compaction of the survivors and the flush happen ONCE per run; per chunk the survivors' B
//     operands are re-staged against the chunk's origin from their indices (one cached load + the split, for the <= 2
//     slots a thread owns) -- no counters, no ballots, no prefix;
//   * COOPERATIVE PROGRESSIVE ELIMINATION: after every chunk the block adds, for each survivor h, the chunk's MISSES --
//     pixels exactly decided not to be inliers of h -- to a per-hypothesis counter miss[b,k,h] shared by all the runs of the
//     (image, keypoint), and learns from the returned value what all of them have proven so far.  With R = the pixels the
//     first launch did not count,  full(h) = partial(h) + R - (all misses of h among them) <= partial(h) + R - miss_seen(h),
//     so h is dropped from the rest of the run as soon as  partial(h) + R - miss_seen(h) < L*.  The same argument as for the
//     first elimination (count_bf16.hpp): a dropped hypothesis has full(h) < L* <= max, whatever partial count it is left with
//     is below the maximum, and every hypothesis whose full count equals the maximum is never dropped -- winner, first-index
//     tie rule and winner count are those of the full pass, bit for bit.  miss[] only grows, and each (pixel, h) pair is
//     evaluated by exactly one block at most once, so ANY value a block reads, however stale, is a lower bound of the true
//     misses: the scheme needs no ordering, fence or barrier between blocks, and its result does not depend on the order in
//     which blocks run (only how early hypotheses are dropped does).  Survivors are re-compacted when that frees a
//     32-hypothesis tile; what a dropped hypothesis counted in this run is discarded (its counter keeps a partial count).
//   * R is chosen on the device so that the launch still has about `target_items` items (one generation of blocks):
//     runs of 3 at config 3 / B = 64 (9 remaining chunks per image), of 9 at config 5 / B = 16 (44 of them).
// Hypothesis counts above 512 are handled in PASSES: the survivors of as many consecutive 512-hypothesis groups (at most 63) as
// fit the 512 staging slots are collected, the run is walked for them, and the next pass takes the next groups (with the 15 %
// survival of a clean field 2048 hypotheses are one pass).  For the estimate (StageArgs.sub_tenth: three quarters survive) an
// item takes ONE group, and the item count that R balances is chunks x groups.
// sCnt[slot] = (hypothesis index relative to the pass's first group) << 16 | inliers counted in this run (< 2^16: a run is
// at most 64 chunks); sPrev[slot] = that count at the slot's last elimination step (the next step publishes the difference).
// ---------------------------------------------------------------------------------------------
constexpr int kRunMaxChunks = 64;
constexpr int kRunSlots = kBfMaxHt * 32;          // 512 staging slots

// Phase census of this kernel, instrumented builds only (tools/build_variant.sh <name> -DPVV_TUNING -DPVV_STAMPS +
// tools/census_filter.py): thread 0 of every block adds the shader cycles (s_memtime) between consecutive marks to one
// of 10 phase accumulators in LDS and writes them, with its item / chunk / survivor / tile counts and its wall-clock
// entry and exit, to dbg[16 * block ...] on exit (StageArgs.dbg, a member that exists only in such builds).
#ifdef PVV_STAMPS
#define PVV_FS_DECL() __shared__ long long s_fs[16]; long long fs_last = 0; \
    if (threadIdx.x == 0) { for (int i_ = 0; i_ < 16; ++i_) s_fs[i_] = 0; s_fs[14] = wall_clock64(); fs_last = (long long)__builtin_readcyclecounter(); }
#define PVV_FS(i) do { if (threadIdx.x == 0) { const long long t_ = (long long)__builtin_readcyclecounter(); s_fs[i] += t_ - fs_last; fs_last = t_; } } while (0)
#define PVV_FS_ADD(i, n) do { if (threadIdx.x == 0) s_fs[i] += (n); } while (0)
#define PVV_FS_OUT() do { if (sa.dbg && threadIdx.x == 0) { s_fs[15] = wall_clock64(); \
    for (int i_ = 0; i_ < 16; ++i_) sa.dbg[(size_t)blockIdx.x * 16 + i_] = s_fs[i_]; } } while (0)
#else
#define PVV_FS_DECL() do { } while (0)
#define PVV_FS(i) do { } while (0)
#define PVV_FS_ADD(i, n) do { } while (0)
#define PVV_FS_OUT() do { } while (0)
#endif

// B operand of slot i (see stage_hypothesis): the counters are NOT touched.
__device__ __forceinline__ int stage_hypothesis_b(bf16x8 *sB, int i, float2 hp, float2 org)
{
    __bf16 qx[3], qy[3];
    split3(hp.x - org.x, qx);
    split3(hp.y - org.y, qy);
    const __bf16 one = (__bf16)1.f, zero = (__bf16)0.f;
    const bf16x8 lo8 = {qx[0], qy[0], qx[1], qx[2], qx[0], qy[1], qy[2], qy[0]};
    const bf16x8 hi8 = {qx[0], qy[0], qx[1], qy[1], one, one, one, zero};
    sB[(i >> 5) * 64 + (i & 31)] = lo8;
    sB[(i >> 5) * 64 + 32 + (i & 31)] = hi8;
    return !(fabsf(hp.x) < 1e15f && fabsf(hp.y) < 1e15f);
}

template <uint32_t FIRST>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_count_filter_runs(
    const float2 *__restrict__ coords /*[B,cap]*/, const float2 *__restrict__ dirs /*[B,K,cap]*/,
    const float2 *__restrict__ hyps /*[B,K,hn]*/, int *__restrict__ counts /*[B,K,hn]*/,
    const int *__restrict__ tn_arr, int B, int K, int hn, int cap, float thresh, Bf16Consts fc, int target_items, int run_r, StageArgs sa)
{
    // inclusive prefix of the runs per image: DYNAMIC shared memory, B ints (the host passes sizeof(int) * B).  Round 5: as a
    // static kMaxBatchLds array it made the block's LDS 32 064 B -- 26 of the CU's 1280-byte allocation granules, so that only
    // FOUR blocks fitted the 160 KB where the grid (5 per CU) and the register budget (96 VGPRs) were sized for five: a fifth of
    // the blocks waited ~26 us for a slot (tools/census_filter.py: 1024 of 1280 blocks alive).  27 968 B + 4 B per image is 23
    // granules at B = 64 and stays within 25 up to B = 1008.
    extern __shared__ int run_end[];
    __shared__ int s_R, s_runs, s_gpi;
    __shared__ bf16x8 sB[kBfMaxHt * 64];            // B operands of the staged survivors (16 KB)
    __shared__ float4 sP[4 * kBfPixPerWave];        // per pixel: (nhx, nhy, c'x, c'y); nhx = NaN: can never vote  (8 KB)
    __shared__ int sCnt[kRunSlots];
    __shared__ unsigned short sPrev[kRunSlots];     // a slot's inliers of this run at the last elimination step
    __shared__ float sRed[4];
    __shared__ int s_keep[8];
    __shared__ unsigned short s_order[64];          // the estimate: rank by distance to the keypoint -> index of the remaining chunk
    const int lane = lane_id(), wave = wave_id();
    constexpr int PC = 4 * kBfPixPerWave;
    constexpr int GH = kRunSlots;                   // hypotheses per group
    constexpr uint32_t REST = stage_rest_of(FIRST);  // the chunks the first launch left
    const int nhg = (hn + GH - 1) / GH;             // hypothesis groups per keypoint
    const int hs = sa.hstride > 0 ? sa.hstride : hn; // row length of hyps / counts / miss (StageArgs)
    if (*sa.any_staged == 0) return;
    PVV_FS_DECL();

    // ---- the item table: remaining chunks per image (0 for the images the first launch counted completely), the run
    //      length from their total, runs per image as an inclusive prefix.  Wave 0; every block derives the same table.
    if (wave == 0) {
        int total = 0;
        for (int b0 = 0; b0 < B; b0 += 64) {
            const int b = b0 + lane;
            int n = 0;
            if (b < B) {
                const int nch = (tn_arr[b] + PC - 1) / PC;
                n = nch < kStageMinChunks ? 0 : stage_chunks<REST>(nch);
                run_end[b] = n;
            }
            total += wave_total(n);
        }
        // The run length: as many chunks per run as still leave about target_items items (the size of the grid: one
        // generation of blocks), (total / R) * K >= target_items.  Measured on MI355X (one-process A/B of whole calls, config 3,
        // forced run lengths; profiles/DESIGN_rounds_1-4.md 4.7): B = 64 -> R = 3 wins (1728 items; 2: +4 %, 5: +2 %, 9: +5 %), B = 32 -> 2 or 3,
        // B = 128 -> 5, B = 16 -> 1, config 5 at B = 16 -> 9; the rule "at most ONE item per block" (longer runs: every item starts
        // at once, more elimination) lost 1-3 % at B = 16 ... 64.  run_r > 0 (tuning builds) forces the length.
        // Hypothesis groups per item: all of them when few hypotheses survive (ransac_voting_layer_v3: 15 % of 2048 are one pass), ONE
        // when most do (the estimate, sub_tenth: ~3/4 of 4096 survive the first stage -- an item that walked six passes over its
        // run was as long as six v3 items, and 1728 of those on 1280 block slots left a third of the chip idle: the staged estimate
        // measured 4-11 % SLOWER than the full pass until its items were cut by groups).  The work units that the run length
        // balances are then chunks x group ranges.
        const int gpi = sa.sub_tenth ? 1 : nhg;
        const int ngr = (nhg + gpi - 1) / gpi;
        int R = run_r;
        if (R <= 0) {
            const long long r = target_items > 0 ? ((long long)total * K * ngr) / target_items : 1;
            R = (int)(r < 1 ? 1 : r);
            // runs of equal length: nruns = ceil(n / R) runs of n / nruns chunks -- for the typical image of the batch, take
            // the length those runs really have (9 chunks, R = 4 -> 3 runs of 3)
        }
        R = R < 1 ? 1 : (R > kRunMaxChunks ? kRunMaxChunks : R);
        int carry = 0;
        for (int b0 = 0; b0 < B; b0 += 64) {
            const int b = b0 + lane;
            const int n = b < B ? run_end[b] : 0;
            int inc = (n + R - 1) / R;
            inc = wave_incl_scan(inc) + carry;
            if (b < B) run_end[b] = inc;
            carry = __builtin_amdgcn_readlane(inc, 63);
        }
        if (lane == 0) { s_R = R; s_runs = carry; s_gpi = gpi; }
    }
    __syncthreads();
    PVV_FS(0);                                                   // phase 0: entry -> item table built
    const int R = __builtin_amdgcn_readfirstlane(s_R);
    const int gpi = __builtin_amdgcn_readfirstlane(s_gpi), ngr = (nhg + gpi - 1) / gpi;   // groups per item, group ranges per keypoint
    const int per_run = K * ngr;
    const int total = __builtin_amdgcn_readfirstlane(s_runs) * per_run;
    const int col = lane & 31, kslice = lane >> 5;
    const int ebase = kslice * 4;                                // this lane's pixels of a tile: ebase + e%4 + 8*(e/4)
    const float16v zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    for (int item = blockIdx.x; item < total; item += gridDim.x) {
        const int grun = item / per_run, rem_i = item - grun * per_run;
        const int vi = ngr == 1 ? rem_i : rem_i / ngr, gr = rem_i - vi * ngr;
        const int g_begin = gr * gpi, g_end = min(nhg, g_begin + gpi);
        int r_img;
        const int b = locate_item(run_end, B, grun, &r_img);     // image, and the run's index within it
        const int bk = b * K + vi;
        const float2 *hyp_k = hyps + (size_t)bk * hs;
        const float2 *crd = coords + (size_t)b * cap;
        const float2 *dir_k = dirs + (size_t)bk * cap;
        int *cnt_k = counts + (size_t)bk * hs;
        int *miss_k = sa.miss + (size_t)bk * hs;
        const int tn = __builtin_amdgcn_readfirstlane(tn_arr[b]);
        const int nch = (tn + PC - 1) / PC;
        const int nrest = stage_chunks<REST>(nch);         // remaining chunks of the image (>= 1: the image is staged)
        const int nruns = (nrest + R - 1) / R;
        const int j0 = (int)((long long)r_img * nrest / nruns), j1 = (int)((long long)(r_img + 1) * nrest / nruns);
        const int R_rem = stage_pixels<REST>(tn, nch, PC);  // pixels of the image the first launch did not count
        // L* = the larger of the leaders' lower bounds (k_lead)
        int lstar;
        {
            const int lead_p = sa.lead[(size_t)bk * 8 + (lane & 3)], lead_r = sa.lead[(size_t)bk * 8 + 4 + (lane & 3)];
            int full = lead_p >= 0 ? lead_p + lead_r : -1;
            full = max(full, PVV_DPP(full, full, 0xB1, 0xf, false));   // quad_perm [1,0,3,2]
            full = max(full, PVV_DPP(full, full, 0x4E, 0xf, false));   // quad_perm [2,3,0,1]
            lstar = stage_bound(__builtin_amdgcn_readfirstlane(full), tn, sa.sub_tenth);
        }

        // ---- the ORDER of the chunks (round 5, the estimate: sa.mean).  Which pixels a hypothesis misses is not uniform: one that
        //      lies d px off the keypoint misses the pixels within ~7 d of it (the angle its offset subtends there exceeds the
        //      threshold's 8 degrees) and hits nearly all others.  The estimate drops a hypothesis after ~0.1 tn proven misses, and a
        //      typical one (ratio 0.85) has only 0.15 tn of them, ALL near the keypoint: in row-major order they trickle in over
        //      70 % of the pixels, nearest-first they are there after a few chunks.  Chunks are bands of ~6 image rows: ranked by
        //      |y of the chunk's middle pixel - y of the keypoint| (ties: by index); a run takes the ranks r_img, r_img + nruns, ...
        //      so that every run of the (image, keypoint) starts near.  Any order counts the same pairs: exactness is untouched.
        const bool prox = sa.mean != nullptr && nrest <= 64;
        if (prox && wave == 0) {
            float key = INFINITY;
            if (lane < nrest) {
                const int pm = min(stage_chunk_at<REST>(lane) * PC + PC / 2, tn - 1);
                key = fabsf(crd[pm].y - sa.mean[bk].y);
                if (!(key == key)) key = INFINITY;               // (a NaN keypoint: natural order)
            }
            int rank = 0;
            for (int o = 0; o < nrest; ++o) {
                const float ko = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(key), o));
                rank += (ko < key || (ko == key && o < lane)) ? 1 : 0;
            }
            if (lane < nrest) s_order[rank] = (unsigned short)lane;
        }
        const int nmine = prox ? (nrest - r_img + nruns - 1) / nruns : j1 - j0;   // chunks of this run
        PVV_FS(1);                                               // phase 1: item decode, tn, leaders -> L*
        PVV_FS_ADD(10, 1);
        for (int g = g_begin; g < g_end;) {
            // ================= a pass: the survivors of groups gp0 .. g-1 (as many consecutive groups as fit the slots)
            const int gp0 = g;
            int ns = 0;
            __syncthreads();                                     // the previous pass / item is done with sCnt, sPrev, s_keep
            for (; g < g_end && g - gp0 < 63; ++g) {             // (a slot's hypothesis index relative to the pass: 15 bits)
                bool keep[2];
                int slack[2];
                unsigned long long m[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int i = (int)threadIdx.x + q * kBlock, h = g * GH + i;
                    const int c = h < hn ? cnt_k[h] : 0;
                    const int m0 = h < hn ? miss_k[h] : 0;       // what other runs have already proven (>= 0, only grows)
                    slack[q] = c + R_rem - lstar - m0;
                    keep[q] = h < hn && slack[q] >= 0;
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
                const bool fits = ns + tot <= kRunSlots;          // (block-uniform; a group alone always fits)
                if (fits) {
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        if (keep[q]) {
                            const int slot = ns + base[q] + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m[q] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m[q], 0u));
                            sCnt[slot] = ((g - gp0) * GH + (int)threadIdx.x + q * kBlock) << 16;
                            sPrev[slot] = 0;
                        }
                    ns += tot;
                }
                __syncthreads();                                 // s_keep is free again; the slots are visible
                if (!fits) break;                                // group g opens the next pass
            }
            PVV_FS(2);                                           // phase 2: counters + miss read, keep predicate, survivor compaction
            if (ns == 0) continue;                               // nobody of these groups can still reach L*

            for (int it = 0; it < nmine && ns > 0; ++it) {
                const int j = prox ? __builtin_amdgcn_readfirstlane((int)s_order[r_img + it * nruns]) : j0 + it;
                const int chunk = stage_chunk_at<REST>(j);
                const int pb = chunk * PC;                       // first pixel of the chunk (< tn)
                int tid = threadIdx.x;
                asm volatile("" : "+v"(tid));                    // (see k_count_bf16: keeps the prologue's indices out of the loop's registers)
                // ---- every global load of the chunk first: the origin, two pixels per thread (rows beyond tn are read -- the
                //      arrays reserve cap rows -- and masked below), the survivors' hypotheses (cached: every chunk re-reads them).
                //      (Requesting the run's FIRST chunk together with the counters -- one memory round trip, like round 3's
                //      items, the survivors staged while they are compacted -- was built and measured: +1.5 ... 3 % per call at
                //      every batch size, 28 B of scratch; so was requesting the NEXT chunk's pixels behind the matrix-core loop, beside
                //      the elimination step's atomics (twice, before and after the counters became shared): +0.5 ... 1 %, 24 B of scratch.)
                const float2 org = crd[pb];
                float2 pc[2], pd[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int p = min(pb + tid + q * kBlock, cap - 1);
                    pc[q] = crd[p];
                    pd[q] = dir_k[p];
                }
                const int ns_pad = (ns + 31) & ~31;
                float2 hp[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int i = tid + q * kBlock;
                    hp[q] = i < ns ? hyp_k[gp0 * GH + (int)((unsigned)sCnt[i] >> 16)] : make_float2(0.f, 0.f);
                }
                // ---- per pixel: the f32 unit normal (v_rsq_f32, see k_count_bf16) and the translated coordinates
                float c1 = 0.f;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int pl = tid + q * kBlock, p = pb + pl;
                    float4 rec = make_float4(__builtin_nanf(""), 0.f, 0.f, 0.f);
                    if (p < tn) {
                        const float2 c = pc[q], d = pd[q];
                        const float cx = c.x - org.x, cy = c.y - org.y;  // exact (integers)
                        c1 = fmaxf(c1, fabsf(cx) + fabsf(cy));
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
                int far = 0;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int i = tid + q * kBlock;
                    if (i < ns_pad) far |= stage_hypothesis_b(sB, i, hp[q], org);
                }
                far = __syncthreads_or(far);
                PVV_FS(3);                                       // phase 3: the chunk's loads, pixel records, B staging, barrier
                PVV_FS_ADD(11, 1);
                PVV_FS_ADD(12, ns);
                const float C1 = fmaxf(fmaxf(sRed[0], sRed[1]), fmaxf(sRed[2], sRed[3]));
                const float eps = fc.eps0 + fc.eps_c * C1;
                const float epsw = __builtin_fmaf(fc.beta * 1.02f, C1, eps);   // band half-width at |h'| = 0
                const int npx = min(tn - pb, PC);
                const int ntile_c = (npx + 15) >> 4;              // tiles of the chunk with at least one pixel (1..32)
                const int ntile_w = (ntile_c - wave + 3) >> 2;    // this wave's share (0..8)
                // ---- A operands (k_count_bf16: lanes l and l+32 share the split of a row through v_permlane32_swap)
                bf16x8 A[8];
                if (ntile_w > 0) {
                    const int form = (lane >> 4) & 1, prow = lane & 15;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int jt = kslice * 4 + i;
                        const float4 rec = sP[(jt * 4 + wave) * 16 + prow];
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
                const int nht = ns_pad >> 5, nslot = ns;
                const float2 *hyp_p = hyp_k + gp0 * GH;           // slot s holds hypothesis hyp_p[sCnt[s] >> 16]
                PVV_FS(4);                                       // phase 4: A operands
                PVV_FS_ADD(13, nht * ntile_w);

                if (__builtin_expect(far, 0)) {
                    // some survivor is non-finite / astronomically far: exact loop (K:100-125)
                    for (int ht = 0; ht < nht; ++ht) {
                        const int slot = ht * 32 + col;
                        if (slot >= nslot) continue;
                        const float2 hq = hyp_p[(unsigned)sCnt[slot] >> 16];
                        int inl = 0;
                        for (int p = pb + wave * 2 + kslice; p < min(tn, pb + PC); p += 8) {
                            const float2 c = crd[p], d = dir_k[p];
                            inl += vote_exact(c.x, c.y, hq.x, hq.y, d.x, d.y, thresh) ? 1 : 0;
                        }
                        if (inl) atomicAdd(&sCnt[slot], inl);
                    }
                } else if (ntile_w > 0) {
                    // ---- the matrix-core loop of k_count_bf16 (see there for every step), slots instead of hypotheses
                    for (int ht = 0; ht < nht; ++ht) {
                        const bf16x8 Bop = sB[ht * 64 + lane];
                        unsigned flagged = 0u;
                        const unsigned q01 = __builtin_bit_cast(uint4, Bop).x;
                        const float wband = __builtin_fmaf(fc.beta * 1.02f,
                                                           fabsf(__uint_as_float(q01 << 16)) + fabsf(__uint_as_float(q01 & 0xffff0000u)),
                                                           epsw);
                        unsigned qs[2] = {0u, 0u};
                        unsigned mb[2] = {0u, 0u};
                        int ntw = ntile_w;
                        asm volatile("" : "+s"(ntw));
#pragma unroll
                        for (int jt = 0; jt < 8; ++jt) {
                            if (jt < ntw) {
                                const float16v acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[jt], Bop, zero16, 0, 0, 0);
                                float tmin = INFINITY;
                                float t[8];
#pragma unroll
                                for (int e = 0; e < 8; ++e) {
                                    t[e] = acc[e] - fabsf(acc[8 + e]);
                                    qs[jt >> 2] = __builtin_amdgcn_alignbit(qs[jt >> 2], __float_as_uint(t[e]), 31);
                                    tmin = fminf(tmin, fabsf(t[e]));
                                }
                                if (__builtin_expect(__ballot(tmin <= wband) != 0, 0)) {
                                    unsigned out_of_band = 0u;
#pragma unroll
                                    for (int e = 0; e < 8; ++e)
                                        out_of_band = __builtin_amdgcn_alignbit(
                                            out_of_band, __float_as_uint(eps - __builtin_fmaf(-fc.beta, acc[e], fabsf(t[e]))), 31);
                                    mb[jt >> 2] |= (~out_of_band & 0xffu) << (8 * (jt & 3));
                                    flagged |= 1u << jt;
                                }
                            }
                        }
                        int inl = 8 * ntile_w - __popc(qs[0]) - __popc(qs[1]);   // sign bit set = not an inlier
                        if (__builtin_expect(flagged != 0u, 0) && __any((mb[0] | mb[1]) != 0u)) {
                            const int slot = ht * 32 + col;
                            const float2 hq = slot < nslot ? hyp_p[(unsigned)sCnt[slot] >> 16] : make_float2(0.f, 0.f);
                            do {
                                const int jt = __builtin_ctz(flagged);
                                flagged &= flagged - 1;
                                const unsigned m = mb[jt >> 2] >> (8 * (jt & 3));
                                if (!__any((m & 0xffu) != 0u)) continue;
                                const int after = min(ntile_w - (jt & 4), 4) - 1 - (jt & 3);
                                const unsigned sgn = qs[jt >> 2] >> (8 * after);
#pragma unroll
                                for (int e = 0; e < 8; ++e) {
                                    const bool marked = (m >> (7 - e)) & 1u;
                                    if (!__any(marked)) continue;
                                    const int prow = (jt * 4 + wave) * 16 + ebase + (e & 3) + 8 * (e >> 2);
                                    const int p = pb + prow;
                                    const int fast = ((sgn >> (7 - e)) & 1u) ? 0 : 1;
                                    const float4 rec = sP[prow];
                                    const float dx = hq.x - (rec.z + org.x), dy = hq.y - (rec.w + org.y);
                                    const float a2 = __builtin_fmaf(dx, rec.x, dy * rec.y);
                                    const float b2 = __builtin_fmaf(dx, -fc.kappa * rec.y, dy * (fc.kappa * rec.x));
                                    const float t2 = a2 - fabsf(b2);
                                    int decided = t2 > 0.f ? 1 : 0;
                                    const bool unsure = marked && (!(__builtin_fmaf(-fc.beta2, a2, fabsf(t2)) > fc.eps0) || rec.x != rec.x);
                                    if (__any(unsure)) {
                                        int exact = 0;
                                        if (unsure && p < tn) {
                                            const float2 c = crd[p], d = dir_k[p];
                                            exact = vote_exact(c.x, c.y, hq.x, hq.y, d.x, d.y, thresh) ? 1 : 0;
                                        }
                                        if (unsure) decided = exact;
                                    }
                                    if (p >= tn) decided = 0;
                                    if (marked) inl += decided - fast;
                                }
                            } while (flagged != 0u);
                        }
                        if (inl) atomicAdd(&sCnt[ht * 32 + col], inl);    // LDS: 2 lanes x 4 waves per slot
                    }
                }
                PVV_FS(5);                                       // phase 5: the matrix-core loop (wave 0's own)
                __syncthreads();                                 // the chunk's counts are complete; sB / sP are free
                PVV_FS(6);                                       // phase 6: waiting for the other waves' loops
                // ---- COOPERATIVE progressive elimination.  Every survivor's misses of this chunk -- pixels that are exactly
                //      decided NOT to be its inliers -- go to miss[h], shared by all the runs of the (image, keypoint); the
                //      value the atomic returns counts what every run has proven so far.  full(h) = partial(h) + R - (all its
                //      misses among the R remaining pixels) <= partial(h) + R - miss_seen(h): h is dropped as soon as that is
                //      below L*.  miss[] only grows and every (pixel, h) is evaluated by exactly one block, so ANY value read
                //      -- however stale -- is a valid lower bound of the misses: no ordering between blocks is needed.
                // (running this step only every third chunk when the slots are full -- the estimate's passes: 512 returning atomics per
                //  chunk -- was measured: worse, the hypotheses it would have dropped are multiplied for two more chunks)
                {
                    const bool last = it + 1 >= nmine;
                    bool keep[2];
                    int w[2];
                    unsigned long long m[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int i = (int)threadIdx.x + q * kBlock;
                        keep[q] = false;
                        w[q] = 0;
                        if (i < ns) {
                            w[q] = sCnt[i];
                            const int c = w[q] & 0xffff, h = gp0 * GH + (int)((unsigned)w[q] >> 16);
                            const int dmiss = npx - (c - (int)sPrev[i]);
                            sPrev[i] = (unsigned short)c;
                            if (last) {
                                if (dmiss) atomicAdd(&miss_k[h], dmiss);               // for the other runs; nothing left to drop here
                            } else {
                                const int seen = atomicAdd(&miss_k[h], dmiss) + dmiss;
                                keep[q] = cnt_k[h] + R_rem - lstar - seen >= 0;        // (cnt_k[h] >= partial(h): other runs may have flushed)
                            }
                        }
                        m[q] = __ballot(keep[q]);
                        if (lane == 0) s_keep[q * 4 + wave] = __popcll(m[q]);
                    }
                    if (!last) {
                        __syncthreads();                         // (also: every thread has read its slots)
                        int tot = 0, base[2] = {0, 0};
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) {
                            const int v = s_keep[jj];
                            tot += v;
                            if (jj < wave) base[0] += v;
                            if (jj < 4 + wave) base[1] += v;
                        }
                        if (((tot + 31) >> 5) < ((ns + 31) >> 5)) {  // (block-uniform) a tile less: move the survivors down
#pragma unroll
                            for (int q = 0; q < 2; ++q)
                                if (keep[q]) {
                                    const int slot = base[q] + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m[q] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m[q], 0u));
                                    sCnt[slot] = w[q];
                                    sPrev[slot] = (unsigned short)(w[q] & 0xffff);
                                }
                            ns = tot;
                        }
                        __syncthreads();
                    }
                }
                PVV_FS(7);                                       // phase 7: elimination step (miss atomics, re-compaction, barriers)
            }
            // ---- flush the run's counts of the survivors
            __syncthreads();
            for (int i = threadIdx.x; i < ns; i += kBlock) {
                const int v = sCnt[i];
                if (v & 0xffff) atomicAdd(&cnt_k[gp0 * GH + (int)((unsigned)v >> 16)], v & 0xffff);
            }
            PVV_FS(8);                                           // phase 8: flush
        }
    }
    PVV_FS(9);
    PVV_FS_OUT();
}
