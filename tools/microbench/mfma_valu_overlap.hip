// Do f32 MFMAs (v_mfma_f32_16x16x4_f32) overlap with VALU work on gfx950?  Three kernels over the same
// iteration count: MFMA only, VALU only (v_fma_f32), both interleaved in one wave.  If T(both) ~ max -> separate
// pipes; if T(both) ~ sum -> shared datapath.   hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o /tmp/ov && /tmp/ov
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float4v __attribute__((ext_vector_type(4)));
#define ITERS 2048

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float s)
{
    float4v acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
    float a = threadIdx.x * 0.001f, b = s;
    float v0 = a, v1 = a + 1, v2 = a + 2, v3 = a + 3, v4 = a + 4, v5 = a + 5, v6 = a + 6, v7 = a + 7;
    for (int i = 0; i < ITERS; ++i) {
        if (MODE == 0 || MODE == 2) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n v_mfma_f32_16x16x4_f32 %1, %4, %5, %1\n"
                         "v_mfma_f32_16x16x4_f32 %2, %4, %5, %2\n v_mfma_f32_16x16x4_f32 %3, %4, %5, %3\n"
                         : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(a), "v"(b));
        }
        if (MODE == 1 || MODE == 2) {   // 32 VALU ops ~ 4 MFMA x 32 cycles at 4 cycles each
#pragma unroll
            for (int r = 0; r < 4; ++r)
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(a), "v"(b));
        }
        if (MODE == 3) {  // bf16 MFMA for comparison of the concurrency claim
            typedef short short8 __attribute__((ext_vector_type(8)));
            short8 x = {1, 2, 3, 4, 5, 6, 7, 8};
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %4, %0\n v_mfma_f32_16x16x32_bf16 %1, %4, %4, %1\n"
                         "v_mfma_f32_16x16x32_bf16 %2, %4, %4, %2\n v_mfma_f32_16x16x32_bf16 %3, %4, %4, %3\n"
                         : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(x));
        }
    }
    asm volatile("s_nop 15\n s_nop 15");
    out[blockIdx.x * 256 + threadIdx.x] = acc0[0] + acc1[1] + acc2[2] + acc3[3] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}

template <int MODE> float run(float *d, int wpc)
{
    int blocks = 256 * wpc / 4;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 10;
}

int main()
{
    float *d; (void)hipMalloc(&d, 256 * 4096 * sizeof(float));
    for (int wpc : {4, 8, 16}) {
        float m = run<0>(d, wpc), v = run<1>(d, wpc), both = run<2>(d, wpc), bf = run<3>(d, wpc);
        double per = (double)ITERS * 4 * (wpc / 4.0);   // MFMAs per SIMD
        printf("waves/CU=%2d  f32-MFMA only %.3f ms (%.1f cyc/MFMA/SIMD)   VALU only %.3f ms   both %.3f ms   sum %.3f max %.3f   [bf16 16x16x32 only %.3f ms]\n",
               wpc, m, m * 1e-3 * 2.4e9 / per, v, both, m + v, m > v ? m : v, bf);
    }
    return 0;
}
