// (1) Does v_mfma_f32_32x32x16_bf16 overlap with VALU work on gfx950?  (2) How accurate is a 3-way bf16 split dot
// product accumulated by the MFMA against the exactly rounded fp32 value?
// hipcc --offload-arch=gfx950 -O3 bf16_mfma_overlap.hip -o /tmp/bo && /tmp/bo
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
typedef float float16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define ITERS 2048

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float s)
{
    float16v acc = {0};
    bf16x8 x;
    for (int i = 0; i < 8; ++i) x[i] = (__bf16)(float)(threadIdx.x + i);
    float a = threadIdx.x * 0.001f, b = s;
    float v0 = a, v1 = a + 1, v2 = a + 2, v3 = a + 3, v4 = a + 4, v5 = a + 5, v6 = a + 6, v7 = a + 7;
    for (int i = 0; i < ITERS; ++i) {
        if (MODE == 0 || MODE == 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, acc, 0, 0, 0);
        if (MODE == 1 || MODE == 2) {   // 24 VALU ops (~ the per-MFMA VALU load of the count kernel)
#pragma unroll
            for (int r = 0; r < 3; ++r)
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(a), "v"(b));
        }
    }
    float r = 0; for (int i = 0; i < 16; ++i) r += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = r + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
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

// accuracy: D[i][j] = sum_k A[i][k] B[k][j], one MFMA, K = 16 terms chosen like the count kernel's split products
__global__ void acc_kernel(const __bf16 *A /*[32][16]*/, const __bf16 *B /*[16][32]*/, float *D /*[32][32]*/)
{
    int l = threadIdx.x;
    bf16x8 a, b;
    for (int t = 0; t < 8; ++t) {
        a[t] = A[(l % 32) * 16 + (l / 32) * 8 + t];      // lane l: row l%32, k = 8*(l/32)+t
        b[t] = B[((l / 32) * 8 + t) * 32 + (l % 32)];    // lane l: col l%32, k = 8*(l/32)+t
    }
    float16v acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        int row = (l / 32) * 4 + (r % 4) + 8 * (r / 4);
        D[row * 32 + (l % 32)] = acc[r];
    }
}

static void split3(float x, float *p)
{
    auto bf = [](float v) { __bf16 h = (__bf16)v; return (float)h; };
    p[0] = bf(x); p[1] = bf(x - p[0]); p[2] = bf(x - p[0] - p[1]);
}

int main()
{
    float *d; (void)hipMalloc(&d, 256 * 4096 * sizeof(float));
    for (int wpc : {4, 8, 16}) {
        float m = run<0>(d, wpc), v = run<1>(d, wpc), both = run<2>(d, wpc);
        double per = (double)ITERS * (wpc / 4.0);
        printf("waves/CU=%2d  bf16 32x32x16 MFMA only %.3f ms (%.1f cyc/MFMA/SIMD)  VALU(24) only %.3f ms  both %.3f ms  sum %.3f max %.3f\n",
               wpc, m, m * 1e-3 * 2.4e9 / per, v, both, m + v, m > v ? m : v);
    }
    // ---- accuracy of a = hx*nx + hy*ny + 1*cn with 3-way bf16 splits (6+6+3 = 15 terms)
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    std::vector<__bf16> A(32 * 16), B(16 * 32);
    std::vector<double> exact(32 * 32), scale(32 * 32);
    std::vector<float> hx(32), hy(32), nx(32), ny(32), cn(32);
    double worst = 0, worst_rel_u = 0;
    __bf16 *dA, *dB; float *dD;
    (void)hipMalloc(&dA, 32 * 16 * 2); (void)hipMalloc(&dB, 16 * 32 * 2); (void)hipMalloc(&dD, 32 * 32 * 4);
    for (int scenario = 0; scenario < 3; ++scenario) {
    worst = 0; worst_rel_u = 0;
    for (int trial = 0; trial < 400; ++trial) {
        // scenario 0: independent magnitudes; 1: heavy cancellation (hypothesis ~1 px from the pixel, both hundreds of
        // px from the origin: a = (h'-c').n is tiny against its terms); 2: terms spread over 8 orders of magnitude
        for (int i = 0; i < 32; ++i) { float t = U(rng) * 3.14159f; nx[i] = cosf(t); ny[i] = sinf(t); }
        std::vector<float> pcx(32), pcy(32);
        for (int i = 0; i < 32; ++i) { pcx[i] = floorf(U(rng) * 400); pcy[i] = floorf(U(rng) * 300); }
        for (int j = 0; j < 32; ++j) {
            if (scenario == 0) { hx[j] = U(rng) * 400; hy[j] = U(rng) * 300; }
            else if (scenario == 1) { hx[j] = pcx[j] + U(rng) * 1.5f; hy[j] = pcy[j] + U(rng) * 1.5f; }
            else { hx[j] = U(rng) * powf(10.f, U(rng) * 4); hy[j] = U(rng) * powf(10.f, U(rng) * 4); }
        }
        for (int i = 0; i < 32; ++i) cn[i] = -(pcx[i] * nx[i] + pcy[i] * ny[i]);
        static const int PI[6] = {0, 0, 0, 1, 1, 2}, PJ[6] = {0, 1, 2, 0, 1, 0};
        for (int i = 0; i < 32; ++i) {
            float px[3], py[3], pc[3]; split3(nx[i], px); split3(ny[i], py); split3(cn[i], pc);
            for (int t = 0; t < 6; ++t) { A[i * 16 + t] = (__bf16)px[PI[t]]; A[i * 16 + 6 + t] = (__bf16)py[PI[t]]; }
            for (int t = 0; t < 3; ++t) A[i * 16 + 12 + t] = (__bf16)pc[t];
            A[i * 16 + 15] = (__bf16)0.f;
        }
        for (int j = 0; j < 32; ++j) {
            float qx[3], qy[3]; split3(hx[j], qx); split3(hy[j], qy);
            for (int t = 0; t < 6; ++t) { B[t * 32 + j] = (__bf16)qx[PJ[t]]; B[(6 + t) * 32 + j] = (__bf16)qy[PJ[t]]; }
            for (int t = 0; t < 3; ++t) B[(12 + t) * 32 + j] = (__bf16)1.f;
            B[15 * 32 + j] = (__bf16)0.f;
        }
        (void)hipMemcpy(dA, A.data(), 32 * 16 * 2, hipMemcpyHostToDevice);
        (void)hipMemcpy(dB, B.data(), 16 * 32 * 2, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(acc_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        std::vector<float> D(32 * 32);
        (void)hipMemcpy(D.data(), dD, 32 * 32 * 4, hipMemcpyDeviceToHost);
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double ex = (double)hx[j] * nx[i] + (double)hy[j] * ny[i] + (double)cn[i];
            double sc = fabs((double)hx[j] * nx[i]) + fabs((double)hy[j] * ny[i]) + fabs((double)cn[i]);
            double err = fabs((double)D[i * 32 + j] - ex);
            if (err > worst) worst = err;
            if (sc > 0 && err / sc / 5.96e-8 > worst_rel_u) worst_rel_u = err / sc / 5.96e-8;
        }
    }
    printf("scenario %d: split-bf16 MFMA dot (15 terms): worst abs err %.3g, worst err / (sum|terms| * u) = %.2f\n", scenario, worst, worst_rel_u);
    }
    return 0;
}
