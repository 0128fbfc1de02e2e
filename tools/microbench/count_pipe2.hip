// Round 5: what the VALU side of the count kernel's per-tile loop costs, form by form, at the kernel's own occupancy
// (5 waves per SIMD, 96-register budget), beside the bf16 matrix-core instruction it consumes.  Relative numbers in one run:
// NOTE (round 6) on this file's UNITS: its "cycles per tile and SIMD" are ONE WAVE's s_memtime cycles per tile divided by 5 -- the SIMD's
// figure only if all five waves ran their loops over the whole kernel (they do not: waves of one SIMD finish up to a millisecond apart) -- and
// its "clock" is wave cycles / kernel time, low for the same reason (1.45-1.67 "GHz" where the chip ran at 2.3; also why "MFMA alone" read
// 23.3 where the matrix pipe needs 32-34 cycles).  count_pipe3.hip reads both device counters (s_memtime, s_memrealtime) and the kernel time and
// prints per-wave and per-SIMD figures separately: use that.  Kept because profiles/r05_count_pipe2.txt quotes this one.
//   0  as shipped: 8 v_sub |abs|, 8 v_alignbit (sign queue), 4 v_min3 |abs| (band test), v_cmp + ballot
//   1  clamp counting: 8 v_sub |abs| clamp (s = clamp(S t + 1/2): the scale S and the 1/2 ride in the MFMA operands), 8 v_add
//      (count), 8 v_fma (sum of squares: equal to the count iff no s is fractional, i.e. no evaluation in the band), v_cmp + ballot
//   2  form 1 without the detection (8 sub-clamp, 8 add): its floor
//   3  form 0 without the sign queue (8 sub, 4 min3, cmp)
//   4  form 0 without the band test (8 sub, 8 alignbit)
//   5  MFMA alone (results consumed by one v_or chain so that it is not dead)
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form count_pipe2.hip -o /tmp/cp2 && /tmp/cp2
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
typedef float float16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define ITERS 2048

__device__ __forceinline__ float sub_abs_clamp(float a, float b)
{
    float r;
    asm("v_sub_f32_e64 %0, %1, |%2| clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void tile_kernel(float *out, float s, long long *cyc)
{
    bf16x8 A[8], Bop;
    for (int j = 0; j < 8; ++j)
        for (int i = 0; i < 8; ++i) A[j][i] = (__bf16)(float)((threadIdx.x * 7 + i * 3 + j) % 13 - 6);
    for (int i = 0; i < 8; ++i) Bop[i] = (__bf16)(float)((threadIdx.x + i) % 5 - 2);
    const float16v zero16 = {0};
    const float W = s * 2e-3f;
    int total = 0;
    float totalf = 0.f;
    unsigned flagged_any = 0;
    const long long t0 = (long long)__builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
        unsigned flagged = 0;
        unsigned qs[2] = {0u, 0u};
        float cnt = 0.f, sq = 0.f;
        unsigned sink = 0u;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float16v acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[j], Bop, zero16, 0, 0, 0);
            if (MODE == 0 || MODE == 3 || MODE == 4) {
                float tmin = INFINITY;
                float t[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    t[e] = acc[e] - fabsf(acc[8 + e]);
                    if (MODE != 3) qs[j >> 2] = __builtin_amdgcn_alignbit(qs[j >> 2], __float_as_uint(t[e]), 31);
                    if (MODE != 4) tmin = fminf(tmin, fabsf(t[e]));
                }
                if (MODE == 3) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) sink |= __float_as_uint(t[e]) >> 31;   // (keeps t alive at 1 op; not part of the form)
                }
                if (MODE != 4) { if (__ballot(tmin <= W) != 0) flagged |= 1u << j; }
            } else if (MODE == 1 || MODE == 2) {
                const float before_c = cnt, before_q = sq;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float sv = sub_abs_clamp(acc[e], acc[8 + e]);
                    cnt += sv;
                    if (MODE == 1) sq = __builtin_fmaf(sv, sv, sq);
                }
                if (MODE == 1) { if (__ballot(cnt != sq) != 0) flagged |= 1u << j; }
                (void)before_c; (void)before_q;
            } else {
#pragma unroll
                for (int e = 0; e < 16; e += 4) sink |= __float_as_uint(acc[e]);
            }
        }
        total += 64 - __popc(qs[0]) - __popc(qs[1]) + (int)sink;
        totalf += cnt + sq;
        flagged_any |= flagged;
        asm volatile("" : "+v"(Bop));
    }
    const long long t1 = (long long)__builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = (float)total + totalf + (float)flagged_any;
}

template <typename K> static void run(K kern, const char *label, float *d, long long *dc, int blocks)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, dc);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, dc);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    static long long h[4096];
    (void)hipMemcpy(h, dc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double sum = 0; for (int i = 0; i < blocks; ++i) sum += (double)h[i];
    const double cyc_block = sum / blocks;                       // shader cycles one block (its wave 0) spent in the loop
    // 5 waves per SIMD walk ITERS * 8 tiles each: cycles per tile and SIMD = block cycles / (ITERS * 8) / 5 ... if all 5 run concurrently
    printf("%-64s %8.3f ms   %7.1f shader cycles per tile and wave   %6.1f per tile and SIMD (5 waves)   clock %.2f GHz\n", label, ms,
           cyc_block / (ITERS * 8.0), cyc_block / (ITERS * 8.0) / 5.0, cyc_block / (ms * 1e6));
}

int main()
{
    const int blocks = 256 * 5;
    float *d; (void)hipMalloc(&d, (size_t)256 * blocks * sizeof(float));
    long long *dc; (void)hipMalloc(&dc, sizeof(long long) * blocks);
    run(tile_kernel<0>, "0 shipped (8 sub, 8 alignbit, 4 min3, cmp)", d, dc, blocks);
    run(tile_kernel<1>, "1 clamp counting (8 sub-clamp, 8 add, 8 fma, cmp)", d, dc, blocks);
    run(tile_kernel<2>, "2 form 1 without detection (8 sub-clamp, 8 add)", d, dc, blocks);
    run(tile_kernel<3>, "3 form 0 without the sign queue (8 sub, 8 or, 4 min3, cmp)", d, dc, blocks);
    run(tile_kernel<4>, "4 form 0 without the band test (8 sub, 8 alignbit)", d, dc, blocks);
    run(tile_kernel<5>, "5 MFMA alone", d, dc, blocks);
    run(tile_kernel<0>, "0 shipped, again", d, dc, blocks);
    run(tile_kernel<1>, "1 clamp counting, again", d, dc, blocks);
    return 0;
}
