// icache_cold.hip -- how much does straight-line code cost when a kernel is launched (instruction fetch)?
// One wave executes N independent-of-memory VALU instructions; timed per launch with HIP events, (a) the same kernel
// back to back, (b) alternating with another kernel of the same size (evicts / invalidates?), (c) 256 blocks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int N, int TAG>
__global__ void k_line(int *out, int x)
{
    int v = x + TAG;
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "v"(x));
    if (v == 0x7fffffff) out[0] = v;
}

template <int N>
__global__ void k_loop(int *out, int x, int iters)
{
    int v = x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < N; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "v"(x));
    }
    if (v == 0x7fffffff) out[0] = v;
}

template <typename F>
float time_us(F f, int reps)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f(); f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return 1e3f * ms / reps;
}

int main()
{
    int *d;
    hipMalloc(&d, 4);
    const int reps = 200;
#define RUN(N)                                                                                                        \
    {                                                                                                                 \
        float same1 = time_us([&] { hipLaunchKernelGGL((k_line<N, 0>), dim3(1), dim3(64), 0, 0, d, 1); }, reps);      \
        float alt1 = time_us([&] { hipLaunchKernelGGL((k_line<N, 0>), dim3(1), dim3(64), 0, 0, d, 1);                 \
                                   hipLaunchKernelGGL((k_line<N, 1>), dim3(1), dim3(64), 0, 0, d, 1); }, reps) / 2;   \
        float same256 = time_us([&] { hipLaunchKernelGGL((k_line<N, 0>), dim3(256), dim3(256), 0, 0, d, 1); }, reps); \
        float loop1 = time_us([&] { hipLaunchKernelGGL((k_loop<256>), dim3(1), dim3(64), 0, 0, d, 1, N / 256); }, reps); \
        printf("N=%6d straight-line: 1 wave %.2f us/launch, alternating kernels %.2f, 256x256 threads %.2f;  same work as a 256-instr loop: %.2f us\n", \
               N, same1, alt1, same256, loop1);                                                                       \
    }
    RUN(256) RUN(1024) RUN(4096) RUN(16384)
    return 0;
}
