// sign_count.hip -- counting the negative ones among 8 floats per lane (the count kernel's sign-bit queue):
//   (a) 8 dependent v_alignbit_b32 (+ popcount once per 32 values)                               = 1 VALU per value
//   (b) 4 v_ashr_pk_i8_i32 (two values -> two bytes 0x00 / 0xff, low half then high half by op_sel) + 2 v_sad_u8
//       (sum of the four bytes into an accumulator = 255 per negative value)                       = 0.75 VALU per value
//   (c) the same with v_dot4_u32_u8 instead of v_sad_u8
// Checks that (b) and (c) count what (a) counts (semantics of the gfx950 instruction and of its op_sel destination half)
// and times the three as full-chip VALU-bound loops (1024 blocks x 256 threads, every SIMD several waves deep).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ unsigned pack4_signs(float t0, float t1, float t2, float t3)
{
    unsigned w;
    asm volatile("v_ashr_pk_i8_i32 %0, %1, %2, 31" : "=v"(w) : "v"(t0), "v"(t1));
    asm volatile("v_ashr_pk_i8_i32 %0, %1, %2, 31 op_sel:[0,0,0,1]" : "+v"(w) : "v"(t2), "v"(t3));
    return w;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_count(const float *__restrict__ in, int *__restrict__ out, int iters)
{
    float t[8];
    for (int e = 0; e < 8; ++e) t[e] = in[(blockIdx.x * 256 + threadIdx.x) * 8 + e];
    int total = 0;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            unsigned q = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int e = 0; e < 8; ++e) q = __builtin_amdgcn_alignbit(q, __float_as_uint(t[e]), 31);
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("v_xor_b32 %0, 0x80000000, %0" : "+v"(t[e]));   // flip: keeps the loop honest
            }
            total += __popc(q);
        } else {
            unsigned acc = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned w0 = pack4_signs(t[0], t[1], t[2], t[3]);
                const unsigned w1 = pack4_signs(t[4], t[5], t[6], t[7]);
                if (MODE == 1) {
                    asm volatile("v_sad_u8 %0, %1, 0, %0" : "+v"(acc) : "v"(w0));
                    asm volatile("v_sad_u8 %0, %1, 0, %0" : "+v"(acc) : "v"(w1));
                } else {
                    const unsigned ones = 0x01010101u;
                    asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(acc) : "v"(w0), "v"(ones));
                    asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(acc) : "v"(w1), "v"(ones));
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("v_xor_b32 %0, 0x80000000, %0" : "+v"(t[e]));
            }
            total += (acc + (acc >> 8) + 1) >> 8;          // acc = 255 n, n <= 256
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = total;
}

int main()
{
    const int blocks = 1024 * 4, n = blocks * 256;
    std::vector<float> h(n * 8);
    srand(7);
    for (auto &x : h) {
        const int r = rand() % 8;
        x = r == 0 ? -0.0f : r == 1 ? 0.0f : r == 2 ? -1e-38f : r == 3 ? 3e38f : r == 4 ? -3e38f : (rand() % 2000 - 1000) * 0.37f;
    }
    float *d_in;
    int *d_out[3];
    hipMalloc(&d_in, h.size() * 4);
    hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<int> r[3];
    const int iters = 2000;
    for (int m = 0; m < 3; ++m) {
        hipMalloc(&d_out[m], n * 4);
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        auto launch = [&](int it) {
            if (m == 0) hipLaunchKernelGGL(k_count<0>, dim3(blocks), dim3(256), 0, 0, d_in, d_out[m], it);
            if (m == 1) hipLaunchKernelGGL(k_count<1>, dim3(blocks), dim3(256), 0, 0, d_in, d_out[m], it);
            if (m == 2) hipLaunchKernelGGL(k_count<2>, dim3(blocks), dim3(256), 0, 0, d_in, d_out[m], it);
        };
        launch(iters); launch(iters);
        hipDeviceSynchronize();
        hipEventRecord(a);
        for (int i = 0; i < 5; ++i) launch(iters);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        ms /= 5;
        r[m].resize(n);
        launch(3);                                      // odd iteration count for the check: 3 x 4 rounds of flips
        hipMemcpy(r[m].data(), d_out[m], n * 4, hipMemcpyDeviceToHost);
        const double vals = (double)n * 32.0 * iters;
        printf("mode %d: %.3f ms  %.2f T values/s   (%s)\n", m, ms, vals / ms * 1e-9,
               m == 0 ? "8 x v_alignbit_b32" : m == 1 ? "4 x v_ashr_pk_i8_i32 + 2 x v_sad_u8" : "4 x v_ashr_pk_i8_i32 + 2 x v_dot4_u32_u8");
    }
    long bad1 = 0, bad2 = 0;
    for (int i = 0; i < n; ++i) { bad1 += r[1][i] != r[0][i]; bad2 += r[2][i] != r[0][i]; }
    printf("mismatches vs alignbit: sad %ld, dot4 %ld of %d lanes (lane 0: %d %d %d)\n", bad1, bad2, n, r[0][0], r[1][0], r[2][0]);
    return bad1 || bad2;
}
