// Second VALU microbenchmark (gfx950): counting idioms and VOP3 modifiers for the inlier-count kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 4096
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float s0, float s1)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    float2v q = {s0, s1};
    unsigned c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 1, c5 = 2, c6 = 3, c7 = 4;
    for (int i = 0; i < ITERS; ++i) {
        if (MODE == 0) {  // cmp -> distinct sgpr pairs, addc from them (no VCC serialisation)
            asm volatile("v_cmp_gt_f32 s[20:21], %4, %5\n v_cmp_gt_f32 s[22:23], %5, %6\n v_cmp_gt_f32 s[24:25], %6, %7\n v_cmp_gt_f32 s[26:27], %7, %4\n"
                         "v_addc_co_u32 %0, s[28:29], 0, %0, s[20:21]\n v_addc_co_u32 %1, s[28:29], 0, %1, s[22:23]\n"
                         "v_addc_co_u32 %2, s[28:29], 0, %2, s[24:25]\n v_addc_co_u32 %3, s[28:29], 0, %3, s[26:27]\n"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3)
                         : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29");
        } else if (MODE == 1) {  // v_alignbit_b32 acc, acc, t, 31
            asm volatile("v_alignbit_b32 %0, %0, %8, 31\n v_alignbit_b32 %1, %1, %9, 31\n v_alignbit_b32 %2, %2, %10, 31\n v_alignbit_b32 %3, %3, %11, 31\n"
                         "v_alignbit_b32 %4, %4, %8, 31\n v_alignbit_b32 %5, %5, %9, 31\n v_alignbit_b32 %6, %6, %10, 31\n v_alignbit_b32 %7, %7, %11, 31\n"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
        } else if (MODE == 2) {  // v_sub_f32 VOP3 with abs modifier
            asm volatile("v_sub_f32 %0, %0, |%8|\n v_sub_f32 %1, %1, |%8|\n v_sub_f32 %2, %2, |%8|\n v_sub_f32 %3, %3, |%8|\n"
                         "v_sub_f32 %4, %4, |%8|\n v_sub_f32 %5, %5, |%8|\n v_sub_f32 %6, %6, |%8|\n v_sub_f32 %7, %7, |%8|\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s0));
        } else if (MODE == 3) {  // v_min_f32
            asm volatile("v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n"
                         "v_min_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_min_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s0));
        } else if (MODE == 4) {  // v_pk_add_f32 with SGPR pair operand, op_sel broadcast, neg
            asm volatile("v_pk_add_f32 %0, %0, %4 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %1, %1, %4 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
                         "v_pk_add_f32 %2, %2, %4 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %3, %3, %4 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
                         "v_pk_add_f32 %0, %0, %4 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_add_f32 %1, %1, %4 op_sel:[0,1] op_sel_hi:[1,1]\n"
                         "v_pk_add_f32 %2, %2, %4 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_add_f32 %3, %3, %4 op_sel:[0,1] op_sel_hi:[1,1]\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "s"(q));
        } else if (MODE == 5) {  // v_pk_fma_f32 with SGPR pair operand (broadcast lo)
            asm volatile("v_pk_fma_f32 %0, %0, %4, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %1, %4, %1 op_sel_hi:[1,0,1]\n"
                         "v_pk_fma_f32 %2, %2, %4, %2 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %3, %4, %3 op_sel_hi:[1,0,1]\n"
                         "v_pk_fma_f32 %0, %0, %4, %1 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %1, %4, %2 op_sel_hi:[1,0,1]\n"
                         "v_pk_fma_f32 %2, %2, %4, %3 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %3, %4, %0 op_sel_hi:[1,0,1]\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "s"(q));
        } else if (MODE == 6) {  // v_fma_f32 with abs modifier + literal-free
            asm volatile("v_fma_f32 %0, %8, %9, |%0|\n v_fma_f32 %1, %8, %9, |%1|\n v_fma_f32 %2, %8, %9, |%2|\n v_fma_f32 %3, %8, %9, |%3|\n"
                         "v_fma_f32 %4, %8, %9, |%4|\n v_fma_f32 %5, %8, %9, |%5|\n v_fma_f32 %6, %8, %9, |%6|\n v_fma_f32 %7, %8, %9, |%7|\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s0), "v"(s1));
        } else if (MODE == 7) {  // v_min3_f32
            asm volatile("v_min3_f32 %0, %0, %8, %9\n v_min3_f32 %1, %1, %8, %9\n v_min3_f32 %2, %2, %8, %9\n v_min3_f32 %3, %3, %8, %9\n"
                         "v_min3_f32 %4, %4, %8, %9\n v_min3_f32 %5, %5, %8, %9\n v_min3_f32 %6, %6, %8, %9\n v_min3_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s0), "v"(s1));
        } else if (MODE == 8) {  // v_cmp_gt_f32 e64 with abs, alone (to sgpr pairs)
            asm volatile("v_cmp_gt_f32 s[20:21], %0, |%1|\n v_cmp_gt_f32 s[22:23], %1, |%2|\n v_cmp_gt_f32 s[24:25], %2, |%3|\n v_cmp_gt_f32 s[26:27], %3, |%0|\n"
                         "v_cmp_gt_f32 s[20:21], %1, |%0|\n v_cmp_gt_f32 s[22:23], %2, |%1|\n v_cmp_gt_f32 s[24:25], %3, |%2|\n v_cmp_gt_f32 s[26:27], %0, |%3|\n"
                         : : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        } else if (MODE == 9) {  // v_readlane_b32 x5 + 8 v_sub (pixel broadcast overhead model)
            int l = i & 63; float x0, x1, x2, x3, x4;
            asm volatile("v_readlane_b32 %0, %5, %6\n v_readlane_b32 %1, %5, %6\n v_readlane_b32 %2, %5, %6\n v_readlane_b32 %3, %5, %6\n v_readlane_b32 %4, %5, %6\n"
                         : "=s"(x0), "=s"(x1), "=s"(x2), "=s"(x3), "=s"(x4) : "v"(a0), "s"(l));
            asm volatile("v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %9\n v_sub_f32 %2, %2, %10\n v_sub_f32 %3, %3, %11\n"
                         "v_sub_f32 %4, %4, %12\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %9\n v_sub_f32 %7, %7, %10\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(x0), "s"(x1), "s"(x2), "s"(x3), "s"(x4));
        } else if (MODE == 10) {  // v_bcnt_u32_b32
            asm volatile("v_bcnt_u32_b32 %0, %4, %0\n v_bcnt_u32_b32 %1, %5, %1\n v_bcnt_u32_b32 %2, %6, %2\n v_bcnt_u32_b32 %3, %7, %3\n"
                         "v_bcnt_u32_b32 %0, %5, %0\n v_bcnt_u32_b32 %1, %6, %1\n v_bcnt_u32_b32 %2, %7, %2\n v_bcnt_u32_b32 %3, %4, %3\n"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(c4), "v"(c5), "v"(c6), "v"(c7));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}

template <int MODE>
void run(const char *name, int instr_per_iter, int wpc, float *d)
{
    int blocks = 256 * wpc / 4;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    double winstr = (double)blocks * 4 * ITERS * instr_per_iter;
    printf("%-44s waves/CU=%2d  %.3f ms  %7.1f G wave-instr/s  -> %.2f cyc/instr/SIMD @2.4GHz\n", name, wpc, ms,
           winstr / (ms * 1e-3) / 1e9, 2.4e9 / (winstr / (ms * 1e-3) / 1024.0));
}

int main()
{
    float *d; (void)hipMalloc(&d, 256 * 2048 * 4 * sizeof(float));
    for (int wpc : {16, 32}) {
        run<0>("v_cmp->s[..] + v_addc (per instr)", 8, wpc, d);
        run<8>("v_cmp_gt_f32 e64 |abs| -> sgpr", 8, wpc, d);
        run<1>("v_alignbit_b32", 8, wpc, d);
        run<10>("v_bcnt_u32_b32", 8, wpc, d);
        run<2>("v_sub_f32 e64 |abs|", 8, wpc, d);
        run<3>("v_min_f32", 8, wpc, d);
        run<7>("v_min3_f32", 8, wpc, d);
        run<6>("v_fma_f32 |abs|", 8, wpc, d);
        run<4>("v_pk_add_f32 sgpr op_sel", 8, wpc, d);
        run<5>("v_pk_fma_f32 sgpr op_sel", 8, wpc, d);
        run<9>("5 readlane + 8 v_sub sgpr (per 13)", 13, wpc, d);
    }
    return 0;
}
