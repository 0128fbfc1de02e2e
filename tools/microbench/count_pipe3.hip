// Round 6: the per-tile loop of the count kernel again (count_pipe2.hip), this time with BOTH device counters
// read inside the kernel -- s_memtime (__builtin_readcyclecounter: shader cycles) and s_memrealtime (wall_clock64: 100 MHz) --
// so that "cycles" and "ns" are separate measurements and the clock the chip runs at INSIDE the loop is their ratio, not an
// assumption; with a 60 ms pre-warm of the same kernel in front of every timed group (the clock ramps for tens of ms after idle),
// a census of where the waves ran (HW_ID: waves per SIMD really resident), and the three knock-outs the verdict names.
//
// UNITS.  "cyc/tile/wave"  = shader cycles (s_memtime) one wave spends per matrix-core tile of its loop (one tile = one
//                            v_mfma_f32_32x32x16_bf16 + the VALU instructions of the mode), averaged over all waves' thread-0 stamps;
//         "ns/tile/wave"   = the same interval on the 100 MHz counter;
//         "GHz"            = their ratio = the effective shader clock inside the loop;
//         "ns/tile/SIMD"   = kernel time by HIP events / tiles per SIMD (W waves x ITERS x 8): what the SIMD really delivers, wall clock;
//         "cyc/tile/SIMD"  = ns/tile/SIMD x GHz: shader cycles the SIMD spends per tile -- the number to hold against the guide's
//                            32 cycles per 32x32x16 bf16 MFMA and 2 (full rate) / 4 (half rate) per wave64 VALU instruction.
//         (count_pipe2 divided a wave's cycles by 5, which is the SIMD's figure only if all five waves are resident for the whole
//         kernel, and took its clock from wave cycles / kernel time; both are replaced by the direct figures above.)
//
// Modes: 0 shipped (8 v_sub |abs|, 8 v_alignbit, 4 v_min3 |abs|, v_cmp + ballot) . 5 MFMA alone . 8 VALU of mode 0 alone (no MFMA)
//        6 knock-out: v_sub in its 32-bit VOP2 encoding on a pre-abs'd operand (no |abs| modifier, so no 64-bit VOP3 word)
//        7 knock-out: two independent sign queues per tile (a 4-deep dependent v_alignbit chain instead of 8-deep)
//        9 mode 0 without queue and band test (8 v_sub only)      1 / 2 / 3 / 4 as in count_pipe2 (clamp counting with / without its
//        detection; no sign queue; no band test)
// The third knock-out (MFMA results in AGPRs) is the same source compiled WITHOUT -mllvm -amdgpu-mfma-vgpr-form.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form count_pipe3.hip -o build/mb/cp3
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize count_pipe3.hip -o build/mb/cp3_agpr
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
typedef float float16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define ITERS 2048

template <int MODE, int W>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W, W))) void tile_kernel(float *out, float s, long long *st)
{
    bf16x8 A[8], Bop;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) A[j][i] = (__bf16)(float)((int)((threadIdx.x * 7 + i * 3 + j) % 13) - 6);
#pragma unroll
    for (int i = 0; i < 8; ++i) Bop[i] = (__bf16)(float)((int)((threadIdx.x + i) % 5) - 2);
    const float16v zero16 = {0};
    const float Wb = s * 2e-3f;
    int total = 0;
    unsigned flagged_any = 0;
    float16v fake = {0};
    if (MODE == 8) {
#pragma unroll
        for (int e = 0; e < 16; ++e) fake[e] = s * (float)((int)((threadIdx.x + e * 5) % 11) - 5);
    }
    const long long r0 = wall_clock64();
    const long long t0 = (long long)__builtin_readcyclecounter();
    if (MODE == 10) {
        // mode 10: the shipped VALU work, software-pipelined inside the wave -- the MFMA of tile j + 1 is issued BEFORE the 21 VALU
        // instructions that consume tile j (two accumulator sets: +16 VGPRs), so that a wave never waits on its own MFMA
        float16v cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], Bop, zero16, 0, 0, 0);
        for (int it = 0; it < ITERS; ++it) {
            unsigned flagged = 0;
            unsigned qs[2] = {0u, 0u};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float16v nxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[(j + 1) & 7], Bop, zero16, 0, 0, 0);
                asm volatile("" ::: "memory");
                float tmin = INFINITY;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float t = cur[e] - fabsf(cur[8 + e]);
                    qs[j >> 2] = __builtin_amdgcn_alignbit(qs[j >> 2], __float_as_uint(t), 31);
                    tmin = fminf(tmin, fabsf(t));
                }
                if (__ballot(tmin <= Wb) != 0) flagged |= 1u << j;
                cur = nxt;
            }
            total += 64 - __popc(qs[0]) - __popc(qs[1]);
            flagged_any |= flagged;
            asm volatile("" : "+v"(Bop));
        }
        total += (int)cur[0];
    } else
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
        unsigned flagged = 0;
        unsigned qs[2] = {0u, 0u}, qb[2] = {0u, 0u};
        unsigned sink = 0u;
        float cntf[2] = {0.f, 0.f}, sqf[2] = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 8) asm volatile("" : "+v"(fake));       // (opaque: the VALU work is redone per tile, nothing is multiplied)
            const float16v acc = MODE == 8 ? fake : __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[j], Bop, zero16, 0, 0, 0);
            if (MODE == 5) {
                sink |= __float_as_uint(acc[0]);              // one v_or per MFMA keeps it alive; consumed in order (no 8-deep hoist)
                asm volatile("" : "+v"(sink));
                continue;
            }
            if (MODE == 1 || MODE == 2) {
                // clamp counting (count_pipe2 modes 1 / 2): s = clamp(S t + 1/2) in {0, 1} outside the band (scale and 1/2 ride in the
                // MFMA operands), count = sum s, band detection sum s^2 != sum s: 24 FULL-rate instructions instead of 8 + 13 half-rate
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float sv = __builtin_amdgcn_fmed3f(acc[e] - fabsf(acc[8 + e]), 0.f, 1.f);   // folds into v_sub ... clamp
                    cntf[e & 1] += sv;
                    if (MODE == 1) sqf[e & 1] = __builtin_fmaf(sv, sv, sqf[e & 1]);
                }
                if (MODE == 1) { if (__ballot(cntf[0] + cntf[1] != sqf[0] + sqf[1]) != 0) flagged |= 1u << j; }
                continue;
            }
            float tmin = INFINITY;
            float t[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                t[e] = MODE == 6 ? acc[e] - acc[8 + e] : acc[e] - fabsf(acc[8 + e]);
                if (MODE == 7) {
                    if (e & 1) qb[j >> 2] = __builtin_amdgcn_alignbit(qb[j >> 2], __float_as_uint(t[e]), 31);
                    else qs[j >> 2] = __builtin_amdgcn_alignbit(qs[j >> 2], __float_as_uint(t[e]), 31);
                } else if (MODE != 3 && MODE != 9) {
                    qs[j >> 2] = __builtin_amdgcn_alignbit(qs[j >> 2], __float_as_uint(t[e]), 31);
                }
                if (MODE != 4 && MODE != 9) tmin = fminf(tmin, fabsf(t[e]));
            }
            if (MODE == 3 || MODE == 9) {
#pragma unroll
                for (int e = 0; e < 8; ++e) sink |= __float_as_uint(t[e]) >> 31;   // (keeps t alive; count_pipe2's convention)
            }
            if (MODE != 4 && MODE != 9) { if (__ballot(tmin <= Wb) != 0) flagged |= 1u << j; }
        }
        total += 64 - __popc(qs[0]) - __popc(qs[1]) - __popc(qb[0]) - __popc(qb[1]) + (int)sink + (int)(cntf[0] + cntf[1] + sqf[0] + sqf[1]);
        flagged_any |= flagged;
        asm volatile("" : "+v"(Bop));
    }
    const long long t1 = (long long)__builtin_readcyclecounter();
    const long long r1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) {
        long long *o = st + 8 * (size_t)(blockIdx.x * 4 + (threadIdx.x >> 6));
        o[0] = t1 - t0; o[1] = r1 - r0; o[2] = r0; o[3] = r1;
        o[4] = (long long)__builtin_amdgcn_s_getreg(63492) | ((long long)__builtin_amdgcn_s_getreg(63508) << 32);   // HW_ID | XCC_ID << 32
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)total + (float)flagged_any;
}

template <typename K> static void run(K kern, const char *label, int W, float *d, long long *dst)
{
    const int blocks = 256 * W;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    // 60 ms of the same kernel first: the clock a timed launch sees is that of a busy chip
    (void)hipEventRecord(e0);
    float warm = 0.f; int nwarm = 0;
    while (warm < 60.f) {
        for (int r = 0; r < 4; ++r, ++nwarm) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, dst);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&warm, e0, e1);
    }
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, dst);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const int nw = blocks * 4;
    static long long h[8 * 4 * 256 * 8];
    (void)hipMemcpy(h, dst, sizeof(long long) * 8 * nw, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0; long long first = h[2], last = h[3], last_in = h[2];
    std::map<long long, int> per_simd;
    for (int i = 0; i < nw; ++i) {
        cyc += (double)h[8 * i]; rt += (double)h[8 * i + 1];
        if (h[8 * i + 2] < first) first = h[8 * i + 2];
        if (h[8 * i + 2] > last_in) last_in = h[8 * i + 2];
        if (h[8 * i + 3] > last) last = h[8 * i + 3];
        const long long hw = h[8 * i + 4];
        // SIMD identity: xcc, se, sh, cu, simd (HW_ID bits: wave 3:0, simd 5:4, cu 11:8, sh 12, se 15:13)
        per_simd[((hw >> 32) & 0xf) << 16 | (hw & 0xff30)]++;
    }
    int wmin = 1 << 30, wmax = 0;
    for (auto &kv : per_simd) { if (kv.second < wmin) wmin = kv.second; if (kv.second > wmax) wmax = kv.second; }
    const double tiles = ITERS * 8.0;
    const double ghz = cyc / (rt * 10.0);
    const double ns_simd = ms * 1e6 / (W * tiles);
    printf("%-58s W=%d  %7.3f ms | per wave: %6.1f cyc/tile %6.1f ns/tile -> %.3f GHz | per SIMD: %6.2f ns/tile = %6.1f cyc/tile | "
           "SIMDs %zu, waves/SIMD %d..%d, last wave in +%.1f us, span %.3f ms\n", label, W, ms, cyc / nw / tiles, rt * 10.0 / nw / tiles, ghz,
           ns_simd, ns_simd * ghz, per_simd.size(), wmin, wmax, (last_in - first) / 100.0, (last - first) / 1e5);
    fflush(stdout);
}

int main()
{
    float *d; (void)hipMalloc(&d, (size_t)256 * 256 * 8 * sizeof(float));
    long long *dst; (void)hipMalloc(&dst, sizeof(long long) * 8 * 4 * 256 * 8);
    run(tile_kernel<5, 1>, "5 MFMA alone, ONE wave per SIMD (calibration: 32 cyc)", 1, d, dst);
    run(tile_kernel<5, 5>, "5 MFMA alone", 5, d, dst);
    run(tile_kernel<8, 5>, "8 VALU of the shipped loop alone (21, no MFMA)", 5, d, dst);
    run(tile_kernel<0, 5>, "0 shipped (8 sub|abs|, 8 alignbit, 4 min3, cmp)", 5, d, dst);
    run(tile_kernel<10, 5>, "10 shipped VALU, MFMA of the NEXT tile issued first (2 acc sets)", 5, d, dst);
    run(tile_kernel<10, 4>, "10 the same at FOUR waves per SIMD (its 112-VGPR budget)", 4, d, dst);
    run(tile_kernel<6, 5>, "6 knock-out: v_sub as VOP2, no |abs| modifier", 5, d, dst);
    run(tile_kernel<7, 5>, "7 knock-out: two independent sign queues", 5, d, dst);
    run(tile_kernel<3, 5>, "3 without the sign queue (8 sub, 8 or, 4 min3, cmp)", 5, d, dst);
    run(tile_kernel<4, 5>, "4 without the band test (8 sub, 8 alignbit)", 5, d, dst);
    run(tile_kernel<9, 5>, "9 8 sub only (+8 shift-or to keep them alive)", 5, d, dst);
    run(tile_kernel<1, 5>, "1 clamp counting (8 sub-clamp, 8 add, 8 fma, cmp: 2 chains)", 5, d, dst);
    run(tile_kernel<2, 5>, "2 clamp counting without detection (8 sub-clamp, 8 add)", 5, d, dst);
    run(tile_kernel<0, 4>, "0 shipped at FOUR waves per SIMD", 4, d, dst);
    run(tile_kernel<0, 6>, "0 shipped at SIX waves per SIMD", 6, d, dst);
    run(tile_kernel<0, 8>, "0 shipped at EIGHT waves per SIMD", 8, d, dst);
    run(tile_kernel<8, 8>, "8 VALU alone at EIGHT waves per SIMD", 8, d, dst);
    run(tile_kernel<1, 8>, "1 clamp counting at EIGHT waves per SIMD", 8, d, dst);
    run(tile_kernel<0, 5>, "0 shipped, again", 5, d, dst);
    run(tile_kernel<5, 5>, "5 MFMA alone, again", 5, d, dst);
    return 0;
}
