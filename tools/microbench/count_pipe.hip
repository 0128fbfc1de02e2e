// Candidates for the next step of k_count_bf16's hot loop, measured in isolation on gfx950:
//  (1) issue rate of the conversion / bit ops a matrix-core sign counter would need
//  (2) semantics of v_cvt_pkrtz_f16_f32 on huge / tiny / non-finite inputs (saturation to +-65504 under RTZ)
//  (3) the per-tile loop of the count kernel in four forms, 4 waves per SIMD:
//        0: as shipped   (8 sub, 8 alignbit, 4 min3, 4 max3, fma, cmp per bf16 MFMA)
//        1: sign counting moved to a second MFMA (4 cvt_pkrtz + v_mfma_f32_32x32x16_f16 instead of 8 alignbit)
//        2: same with v_mfma_f32_16x16x32_f16 (4 accumulator registers)
//        3: form 2 without the max-a chain (band half-width from a per-lane constant)
//        4: form 0 without the max-a chain
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form count_pipe.hip -o /tmp/cp && /tmp/cp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __fp16 half2v __attribute__((ext_vector_type(2)));
#define ITERS 4096

// ---------------- (1) issue rates: 8 independent chains of one instruction
#define RATE_KERNEL(NAME, ASM)                                                                               \
    __global__ __launch_bounds__(256) void NAME(float *out, float s)                                          \
    {                                                                                                         \
        float v0 = threadIdx.x + s, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6,   \
              v7 = v0 + 7, a = s * 3.f, b = s * 5.f;                                                            \
        for (int i = 0; i < ITERS; ++i) {                                                                     \
            asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                               \
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7)      \
                         : "v"(a), "v"(b));                                                                   \
        }                                                                                                     \
        out[blockIdx.x * 256 + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;                           \
    }
#define A_PKRTZ(i) "v_cvt_pkrtz_f16_f32 %" #i ", %" #i ", %8\n"
#define A_PKBF(i) "v_cvt_pk_bf16_f32 %" #i ", %" #i ", %8\n"
#define A_MAX3(i) "v_max3_f32 %" #i ", %" #i ", %8, %9\n"
#define A_MIN3A(i) "v_min3_f32 %" #i ", |%" #i "|, |%8|, |%9|\n"
#define A_OR(i) "v_or_b32 %" #i ", %" #i ", %8\n"
#define A_AND_OR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define A_SUBABS(i) "v_sub_f32 %" #i ", %" #i ", |%8|\n"
#define A_ALIGN(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 31\n"
#define A_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define A_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define A_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define A_SQRT(i) "v_sqrt_f32 %" #i ", %" #i "\n"
#define A_PKMINH(i) "v_pk_min_f16 %" #i ", %" #i ", %8\n"
#define A_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 16, 1\n"
#define A_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
RATE_KERNEL(r_pkrtz, A_PKRTZ)
RATE_KERNEL(r_pkbf, A_PKBF)
RATE_KERNEL(r_max3, A_MAX3)
RATE_KERNEL(r_min3a, A_MIN3A)
RATE_KERNEL(r_or, A_OR)
RATE_KERNEL(r_and_or, A_AND_OR)
RATE_KERNEL(r_subabs, A_SUBABS)
RATE_KERNEL(r_align, A_ALIGN)
RATE_KERNEL(r_perm, A_PERM)
RATE_KERNEL(r_mullo, A_MULLO)
RATE_KERNEL(r_rcp, A_RCP)
RATE_KERNEL(r_sqrt, A_SQRT)
RATE_KERNEL(r_pkminh, A_PKMINH)
RATE_KERNEL(r_bfe, A_BFE)
RATE_KERNEL(r_add3, A_ADD3)

template <typename K> static float time_kernel(K kern, float *d, int blocks)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1.0001f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1.0001f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 5;
}

// ---------------- (2) semantics of the saturating conversion
__global__ void sem_kernel(const float *in, uint32_t *out, int n)
{
    int i = threadIdx.x;
    if (i < n) {
        half2v h = __builtin_amdgcn_cvt_pkrtz(in[i], -in[i]);
        out[i] = __builtin_bit_cast(uint32_t, h);
    }
}

// one row of ones through the f16 matrix core: does D[0][n] = sum of the 16 (32x32x16) values of column n, exactly?
__global__ void sum_kernel(const float *t /*[64][8]*/, float *d32 /*[64]*/, float *d16 /*[64][2]*/)
{
    const int l = threadIdx.x;
    half8 b, a32, a16;
    for (int e = 0; e < 8; e += 2) {
        half2v h = __builtin_amdgcn_cvt_pkrtz(t[l * 8 + e], t[l * 8 + e + 1]);
        b[e] = (_Float16)h[0]; b[e + 1] = (_Float16)h[1];
    }
    for (int e = 0; e < 8; ++e) {
        a32[e] = (l % 32) == 0 ? (_Float16)1.f : (_Float16)0.f;                       // row 0 of a 32x32x16 A: all ones
        // 16x16x32: A lane l = row l%16, k = 8*(l/16)..+7; row 0 takes k blocks 0 and 2, row 1 blocks 1 and 3
        const int m = l % 16, kb = l / 16;
        a16[e] = ((m == 0 && (kb == 0 || kb == 2)) || (m == 1 && (kb == 1 || kb == 3))) ? (_Float16)1.f : (_Float16)0.f;
    }
    float16v c32 = {0};
    float4v c16 = {0};
    c32 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a32, b, c32, 0, 0, 0);
    c16 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16, b, c16, 0, 0, 0);
    d32[l] = c32[0];                    // lanes 0-31: row 0, column l
    d16[l * 2] = c16[0];                // lanes 0-15: row 0 (hyp l), row 1 (hyp l+16)
    d16[l * 2 + 1] = c16[1];
}

// ---------------- (3) the per-tile loop
template <int MODE>
__global__ __launch_bounds__(256) void tile_kernel(float *out, float s)
{
    const int lane = threadIdx.x & 63;
    bf16x8 A[8], Bop;
    for (int j = 0; j < 8; ++j)
        for (int i = 0; i < 8; ++i) A[j][i] = (__bf16)(float)((threadIdx.x * 7 + i * 3 + j) % 13 - 6);
    for (int i = 0; i < 8; ++i) Bop[i] = (__bf16)(float)((threadIdx.x + i) % 5 - 2);
    half8 ones32, ones16;
    for (int e = 0; e < 8; ++e) {
        ones32[e] = (lane % 32) == 0 ? (_Float16)1.f : (_Float16)0.f;
        const int m = lane % 16, kb = lane / 16;
        ones16[e] = ((m == 0 && (kb == 0 || kb == 2)) || (m == 1 && (kb == 1 || kb == 3))) ? (_Float16)1.f : (_Float16)0.f;
    }
    const float16v zero16 = {0};
    const float beta = s * 1e-5f, eps = s * 1e-3f, W = s * 2e-3f;
    int total = 0;
    unsigned flagged_any = 0;
    for (int it = 0; it < ITERS / 8; ++it) {
        unsigned flagged = 0;
        int inl = 0;
        float16v c32 = {0};
        float4v c16 = {0};
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            unsigned q = 0;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = half * 4 + jj;
                const float16v acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[j], Bop, zero16, 0, 0, 0);
                float tmin = INFINITY, amax = 0.f;
                float t[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    t[e] = acc[e] - fabsf(acc[8 + e]);
                    if (MODE == 0 || MODE == 4) q = __builtin_amdgcn_alignbit(q, __float_as_uint(t[e]), 31);
                    tmin = fminf(tmin, fabsf(t[e]));
                    if (MODE <= 2) amax = fmaxf(amax, acc[e]);
                }
                const float w = (MODE <= 2) ? __builtin_fmaf(beta, amax, eps) : W;
                const bool f = __ballot(tmin <= w) != 0;
                flagged |= f ? (1u << j) : 0u;
                if (MODE == 1 || MODE == 2 || MODE == 3) {
                    half8 pk;
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        half2v h = __builtin_amdgcn_cvt_pkrtz(t[e], t[e + 1]);
                        pk[e] = (_Float16)h[0]; pk[e + 1] = (_Float16)h[1];
                    }
                    if (!f) {
                        if (MODE == 1) c32 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones32, pk, c32, 0, 0, 0);
                        else c16 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones16, pk, c16, 0, 0, 0);
                    }
                }
            }
            if (MODE == 0 || MODE == 4) inl += 32 - __popc(q);
        }
        if (MODE == 1) inl = (int)__builtin_rintf(c32[0] * (1.f / 65504.f));
        if (MODE == 2 || MODE == 3) inl = (int)__builtin_rintf(c16[0] * (1.f / 65504.f)) + 1000 * (int)__builtin_rintf(c16[1] * (1.f / 65504.f));
        total += inl;
        flagged_any |= flagged;
        asm volatile("" : "+v"(Bop));         // keep the loop body from being hoisted
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)total + (float)flagged_any;
}

int main()
{
    float *d; (void)hipMalloc(&d, 256 * 4096 * sizeof(float));
    const int blocks = 256 * 4;     // 16 waves per CU = 4 per SIMD
    const double per_simd = (double)ITERS * 8 * 4;   // instructions of the measured kind per SIMD (8 per iteration, 4 waves)
#define RATE(NAME, LABEL) { float ms = time_kernel(NAME, d, blocks); printf("%-28s %.2f cycles/instr/SIMD\n", LABEL, ms * 1e-3 * 2.4e9 / per_simd); }
    RATE(r_subabs, "v_sub_f32 a,|b|")
    RATE(r_align, "v_alignbit_b32")
    RATE(r_pkrtz, "v_cvt_pkrtz_f16_f32")
    RATE(r_pkbf, "v_cvt_pk_bf16_f32")
    RATE(r_max3, "v_max3_f32")
    RATE(r_min3a, "v_min3_f32 |abs|")
    RATE(r_or, "v_or_b32")
    RATE(r_and_or, "v_and_or_b32")
    RATE(r_perm, "v_perm_b32")
    RATE(r_pkminh, "v_pk_min_f16")
    RATE(r_bfe, "v_bfe_u32")
    RATE(r_add3, "v_add3_u32")
    RATE(r_mullo, "v_mul_lo_u32")
    RATE(r_rcp, "v_rcp_f32")
    RATE(r_sqrt, "v_sqrt_f32")

    // (2)
    const float vals[] = {0.f, 1.f, 65504.f, 65520.f, 70000.f, 1e30f, 3e38f, INFINITY, NAN, 1e-30f, 6e-8f, 3e-8f, 65503.9f, 32768.f};
    const int n = sizeof(vals) / sizeof(vals[0]);
    float *din; uint32_t *dout;
    (void)hipMalloc(&din, sizeof(vals)); (void)hipMalloc(&dout, n * 4);
    (void)hipMemcpy(din, vals, sizeof(vals), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(sem_kernel, dim3(1), dim3(64), 0, 0, din, dout, n);
    std::vector<uint32_t> ho(n);
    (void)hipMemcpy(ho.data(), dout, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("cvt_pkrtz(%g, %g) = 0x%04x 0x%04x\n", vals[i], -vals[i], ho[i] & 0xffff, ho[i] >> 16);

    std::vector<float> t(64 * 8);
    for (int l = 0; l < 64; ++l) for (int e = 0; e < 8; ++e) t[l * 8 + e] = (((l * 8 + e) * 2654435761u >> 7) & 1) ? 1e20f : -3e12f;
    float *dt, *d32, *d16;
    (void)hipMalloc(&dt, 64 * 8 * 4); (void)hipMalloc(&d32, 64 * 4); (void)hipMalloc(&d16, 128 * 4);
    (void)hipMemcpy(dt, t.data(), 64 * 8 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(64), 0, 0, dt, d32, d16);
    std::vector<float> h32(64), h16(128);
    (void)hipMemcpy(h32.data(), d32, 64 * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(h16.data(), d16, 128 * 4, hipMemcpyDeviceToHost);
    int bad32 = 0, bad16 = 0;
    for (int nn = 0; nn < 32; ++nn) {
        int pos = 0;
        for (int e = 0; e < 8; ++e) { pos += t[nn * 8 + e] > 0; pos += t[(nn + 32) * 8 + e] > 0; }
        const float want = 65504.f * (2 * pos - 16);
        if (h32[nn] != want) ++bad32;
        const float got16 = nn < 16 ? h16[nn * 2] : h16[(nn - 16) * 2 + 1];
        if (got16 != want) ++bad16;
    }
    printf("sign sum through the f16 matrix core: 32x32x16 mismatches %d/32, 16x16x32 mismatches %d/32\n", bad32, bad16);

    // (3)
    const double tiles_per_simd = (double)ITERS * 4;
#define TILE(M, LABEL) { float ms = time_kernel(tile_kernel<M>, d, blocks); printf("tile loop %d %-58s %.1f cycles/tile/SIMD\n", M, LABEL, ms * 1e-3 * 2.4e9 / tiles_per_simd); }
    TILE(0, "(shipped: alignbit + min3 + max3)")
    TILE(1, "(pkrtz + 32x32x16 f16 MFMA counter)")
    TILE(2, "(pkrtz + 16x16x32 f16 MFMA counter)")
    TILE(3, "(form 2, per-lane band constant instead of max a)")
    TILE(4, "(form 0, per-lane band constant instead of max a)")
    return 0;
}
