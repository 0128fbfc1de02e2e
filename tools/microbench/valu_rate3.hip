// valu_rate3.hip -- how many cycles does a wave64 VALU instruction occupy its SIMD on gfx950?  (the
// round-1 figure "~4.2 cycles" against MI355X_MICROARCH.md's table "v_fma_f32 (wave64) 2 cyc (SIMD-32)".)
//
// Round 1's valu_rate.hip timed kernels with HIP events and converted to cycles with an ASSUMED 2.4 GHz.  This one counts
// TRUE shader-clock cycles: every wave reads s_memtime (clock64(): the shader core clock on gfx9) before and after its
// instruction stream; cycles per instruction and SIMD = max over waves of (t1 - t0) / (waves per SIMD x instructions per
// wave).  The same numbers can be cross-checked with counters (each instantiation is its own kernel symbol):
//   rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -- /tmp/valu3
// 8 or 16 INDEPENDENT dependency chains per wave (no instruction waits for its predecessor), 1 / 2 / 4 / 8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_rate3.hip -o /tmp/valu3 && /tmp/valu3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define ITERS 4096
typedef float float2v __attribute__((ext_vector_type(2)));

enum { FMA8, FMA16, ADD8, MUL8, PKFMA8, ALIGNBIT8, MIN3_8, SUBABS8, CMP8, LOOPMIX };

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, long long *cyc, float s0, float s1)
{
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x + i;
    float2v p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = float2v{a[i], a[i + 1]};
    unsigned q[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    const float2v sq = {s0, s1};
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < ITERS; ++it) {
        if (MODE == FMA8 || MODE == FMA16) {
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(s0), "v"(s1));
            if (MODE == FMA16)
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(s0), "v"(s1));
        } else if (MODE == ADD8) {
            asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(s0));
        } else if (MODE == MUL8) {
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(s0));
        } else if (MODE == PKFMA8) {
            asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                         "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n"
                         : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(sq));
        } else if (MODE == ALIGNBIT8) {
            asm volatile("v_alignbit_b32 %0, %0, %8, 31\n v_alignbit_b32 %1, %1, %8, 31\n v_alignbit_b32 %2, %2, %8, 31\n v_alignbit_b32 %3, %3, %8, 31\n"
                         "v_alignbit_b32 %4, %4, %8, 31\n v_alignbit_b32 %5, %5, %8, 31\n v_alignbit_b32 %6, %6, %8, 31\n v_alignbit_b32 %7, %7, %8, 31\n"
                         : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]) : "v"(s0));
        } else if (MODE == MIN3_8) {
            asm volatile("v_min3_f32 %0, %0, |%8|, |%9|\n v_min3_f32 %1, %1, |%8|, |%9|\n v_min3_f32 %2, %2, |%8|, |%9|\n v_min3_f32 %3, %3, |%8|, |%9|\n"
                         "v_min3_f32 %4, %4, |%8|, |%9|\n v_min3_f32 %5, %5, |%8|, |%9|\n v_min3_f32 %6, %6, |%8|, |%9|\n v_min3_f32 %7, %7, |%8|, |%9|\n"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(s0), "v"(s1));
        } else if (MODE == SUBABS8) {
            asm volatile("v_sub_f32 %0, %0, |%8|\n v_sub_f32 %1, %1, |%8|\n v_sub_f32 %2, %2, |%8|\n v_sub_f32 %3, %3, |%8|\n"
                         "v_sub_f32 %4, %4, |%8|\n v_sub_f32 %5, %5, |%8|\n v_sub_f32 %6, %6, |%8|\n v_sub_f32 %7, %7, |%8|\n"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(s0));
        } else if (MODE == CMP8) {
            asm volatile("v_cmp_gt_f32 vcc, %0, %8\n v_cmp_gt_f32 vcc, %1, %8\n v_cmp_gt_f32 vcc, %2, %8\n v_cmp_gt_f32 vcc, %3, %8\n"
                         "v_cmp_gt_f32 vcc, %4, %8\n v_cmp_gt_f32 vcc, %5, %8\n v_cmp_gt_f32 vcc, %6, %8\n v_cmp_gt_f32 vcc, %7, %8\n"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(s0) : "vcc");
        } else if (MODE == LOOPMIX) {
            // the steady-state mix of k_count_bf16 per matrix-core tile, scaled to 21: 8 v_sub |abs|, 8 v_alignbit, 4 v_min3, 1 v_cmp
            asm volatile("v_sub_f32 %0, %0, |%8|\n v_alignbit_b32 %4, %4, %0, 31\n v_sub_f32 %1, %1, |%8|\n v_alignbit_b32 %5, %5, %1, 31\n"
                         "v_sub_f32 %2, %2, |%8|\n v_alignbit_b32 %6, %6, %2, 31\n v_sub_f32 %3, %3, |%8|\n v_alignbit_b32 %7, %7, %3, 31\n"
                         "v_min3_f32 %9, %9, |%0|, |%1|\n v_min3_f32 %9, %9, |%2|, |%3|\n"
                         "v_sub_f32 %0, %0, |%8|\n v_alignbit_b32 %4, %4, %0, 31\n v_sub_f32 %1, %1, |%8|\n v_alignbit_b32 %5, %5, %1, 31\n"
                         "v_sub_f32 %2, %2, |%8|\n v_alignbit_b32 %6, %6, %2, 31\n v_sub_f32 %3, %3, |%8|\n v_alignbit_b32 %7, %7, %3, 31\n"
                         "v_min3_f32 %9, %9, |%0|, |%1|\n v_min3_f32 %9, %9, |%2|, |%3|\n v_cmp_gt_f32 vcc, %9, %8\n"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(s0), "+v"(a[8]) : : "vcc");
        }
    }
    const long long t1 = clock64();
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += a[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += p[i].x + p[i].y + (float)q[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
void run(const char *name, int instr_per_iter, int waves_per_simd, float *d, long long *dc, int cus)
{
    const int blocks = cus * waves_per_simd;        // 4 waves per block = 1 per SIMD: waves_per_simd blocks per CU
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, dc, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, dc, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> c(blocks * 4);
    hipMemcpy(c.data(), dc, sizeof(long long) * blocks * 4, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    const double n = (double)ITERS * instr_per_iter;
    const double med = (double)c[c.size() / 2], mx = (double)c.back();
    printf("%-22s waves/SIMD=%d  cycles per wave64 instruction and SIMD: %.2f (median wave) %.2f (slowest wave)   [%.3f ms by events -> %.2f GHz implied]\n",
           name, waves_per_simd, med / (waves_per_simd * n), mx / (waves_per_simd * n), ms, mx / (ms * 1e6));
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("%s, %d CUs, clockRate %.2f GHz; s_memtime = shader clock\n", prop.name, cus, prop.clockRate / 1e6);
    float *d; long long *dc;
    hipMalloc(&d, sizeof(float) * 256 * cus * 8);
    hipMalloc(&dc, sizeof(long long) * 4 * cus * 8);
    for (int w : {1, 2, 4, 8}) {
        run<FMA8>("v_fma_f32 x8 chains", 8, w, d, dc, cus);
        run<FMA16>("v_fma_f32 x16 chains", 16, w, d, dc, cus);
        run<MUL8>("v_mul_f32", 8, w, d, dc, cus);
        run<ADD8>("v_add_f32", 8, w, d, dc, cus);
        run<SUBABS8>("v_sub_f32 a,|b|", 8, w, d, dc, cus);
        run<PKFMA8>("v_pk_fma_f32", 8, w, d, dc, cus);
        run<ALIGNBIT8>("v_alignbit_b32", 8, w, d, dc, cus);
        run<MIN3_8>("v_min3_f32 |abs|", 8, w, d, dc, cus);
        run<CMP8>("v_cmp_gt_f32", 8, w, d, dc, cus);
        run<LOOPMIX>("count-loop mix (21)", 21, w, d, dc, cus);
    }
    return 0;
}
