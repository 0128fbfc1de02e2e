// VALU issue-rate microbenchmark for gfx950: how many wave64 instructions per cycle per SIMD do
// v_fma_f32 / v_pk_fma_f32 / v_cmp+v_addc / SGPR-operand forms sustain?  Sizing input for the
// inlier-count kernel.  Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o /tmp/valu && /tmp/valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITERS 4096
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float s0, float s1)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    float2v q = {s0, s1};
    unsigned c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    for (int i = 0; i < ITERS; ++i) {
        if (MODE == 0) {  // 8 independent v_fma_f32 (VGPR operands)
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s0), "v"(s1));
        } else if (MODE == 1) {  // 8 independent v_pk_fma_f32
            asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                         "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(q));
        } else if (MODE == 2) {  // 4 x (v_cmp_gt_f32 vcc + v_addc_co_u32)
            asm volatile("v_cmp_gt_f32 vcc, %4, %5\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n v_cmp_gt_f32 vcc, %5, %4\n v_addc_co_u32 %1, vcc, 0, %1, vcc\n"
                         "v_cmp_gt_f32 vcc, %4, %6\n v_addc_co_u32 %2, vcc, 0, %2, vcc\n v_cmp_gt_f32 vcc, %6, %5\n v_addc_co_u32 %3, vcc, 0, %3, vcc\n"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2) : "vcc");
        } else if (MODE == 3) {  // 8 v_fma_f32 with one SGPR operand
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s0), "v"(s1));
        } else if (MODE == 4) {  // 8 v_mul_f32 e32 (VOP2) with SGPR src0
            asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                         "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s0));
        } else if (MODE == 5) {  // v_cmp to SGPR pair + s_bcnt1 + s_add (lanes = pixels style counting)
            unsigned long long m; unsigned t;
            asm volatile("v_cmp_gt_f32 %0, %2, %3\n s_bcnt1_i32_b64 %1, %0\n" : "=s"(m), "=s"(t) : "v"(a0), "v"(a1) : "scc");
            c0 += t;
            asm volatile("v_cmp_gt_f32 %0, %2, %3\n s_bcnt1_i32_b64 %1, %0\n" : "=s"(m), "=s"(t) : "v"(a1), "v"(a2) : "scc");
            c1 += t;
            asm volatile("v_cmp_gt_f32 %0, %2, %3\n s_bcnt1_i32_b64 %1, %0\n" : "=s"(m), "=s"(t) : "v"(a2), "v"(a3) : "scc");
            c2 += t;
            asm volatile("v_cmp_gt_f32 %0, %2, %3\n s_bcnt1_i32_b64 %1, %0\n" : "=s"(m), "=s"(t) : "v"(a3), "v"(a0) : "scc");
            c3 += t;
        } else if (MODE == 6) {  // 8 v_pk_mul_f32
            asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                         "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(q));
        } else if (MODE == 7) {  // 8 v_sub_f32 (VOP2, VGPR)
            asm volatile("v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n"
                         "v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s0));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + c0 + c1 + c2 + c3;
}

template <int MODE>
void run(const char *name, int instr_per_iter, int wpc, float *d)
{
    int blocks = 256 * wpc / 4;  // wpc waves per CU (4 waves per block)
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double winstr = (double)blocks * 4 * ITERS * instr_per_iter;      // wave-instructions
    double per_simd_per_s = winstr / (ms * 1e-3) / 1024.0;
    printf("%-34s waves/CU=%2d  %.3f ms  %.1f G wave-instr/s  -> %.2f cycles/instr/SIMD @2.4GHz\n", name, wpc, ms,
           winstr / (ms * 1e-3) / 1e9, 2.4e9 / per_simd_per_s);
}

int main()
{
    float *d; hipMalloc(&d, 256 * 2048 * 4 * sizeof(float));
    for (int wpc : {4, 8, 16, 32}) {
        run<0>("v_fma_f32 vgpr", 8, wpc, d);
        run<3>("v_fma_f32 sgpr operand", 8, wpc, d);
        run<4>("v_mul_f32 e32 sgpr", 8, wpc, d);
        run<7>("v_sub_f32 e32", 8, wpc, d);
        run<1>("v_pk_fma_f32", 8, wpc, d);
        run<6>("v_pk_mul_f32", 8, wpc, d);
        run<2>("v_cmp+v_addc (per instr)", 8, wpc, d);
        run<5>("v_cmp->sgpr + s_bcnt1 + s_add (per v_cmp)", 4, wpc, d);
    }
    return 0;
}
