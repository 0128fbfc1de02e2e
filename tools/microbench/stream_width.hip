// stream_width.hip -- read-once streaming of a 157 MB buffer (the int64 mask of a 64-image batch) by a persistent grid with
// different load shapes: does the width of the per-lane load (8 B as k_tile_scan issues them, 16 B as the bench's probe)
// or the number in flight limit what the box streams?  Three buffers rotate (cold caches).  GB/s per variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <typename V, int INFLIGHT, bool NT>
__global__ __launch_bounds__(256) void k_read(const V *__restrict__ src, size_t n, uint32_t *__restrict__ sink)
{
    uint32_t acc = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (INFLIGHT - 1) * stride < n; i += INFLIGHT * stride) {
        V v[INFLIGHT];
#pragma unroll
        for (int u = 0; u < INFLIGHT; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < INFLIGHT; ++u) {
            if constexpr (sizeof(V) == 16) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w; else acc += v[u].x ^ v[u].y;
        }
    }
    for (; i < n; i += stride) { V v = src[i]; acc += v.x; }
    if (acc == 0x12345678u) sink[0] = acc;
}

// tile-wise like k_tile_scan: a block owns 2048 consecutive 8-byte elements (16 KB), 8 loads of 8 B per thread, the next
// tile's loads in flight while the current one is consumed (a ballot + popcount per load stands in for the scan's work)
__global__ __launch_bounds__(256) void k_tiles(const unsigned long long *__restrict__ src, size_t ntiles, uint32_t *__restrict__ sink)
{
    unsigned long long cur[8], nxt[8];
    uint32_t acc = 0;
    size_t g = blockIdx.x;
    if (g < ntiles) for (int s = 0; s < 8; ++s) cur[s] = src[g * 2048 + s * 256 + threadIdx.x];
    for (; g < ntiles; g += gridDim.x) {
        const size_t gn = g + gridDim.x;
        if (gn < ntiles) for (int s = 0; s < 8; ++s) nxt[s] = src[gn * 2048 + s * 256 + threadIdx.x];
        for (int s = 0; s < 8; ++s) acc += __popcll(__ballot((cur[s] & 0xff) != 0));
        for (int s = 0; s < 8; ++s) cur[s] = nxt[s];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// the same with k_tile_scan's per-tile work added step by step: LEVEL 1 = + weight sum (wave reduction), 2 = + the segment
// scan (two barriers, wave 0 scans the 32 segment counts), 3 = + the ordered 16-bit list stores, 4 = + the packed tile word
template <int LEVEL>
__global__ __launch_bounds__(256) void k_scanlike(const unsigned long long *__restrict__ src, size_t ntiles, uint32_t *__restrict__ sink,
                                                  unsigned short *__restrict__ lists, uint32_t *__restrict__ tiles)
{
    __shared__ int seg2[2][33];
    __shared__ int red2[2][4];
    unsigned long long cur[8], nxt[8];
    uint32_t acc = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    size_t g = blockIdx.x;
    if (g < ntiles) for (int s = 0; s < 8; ++s) cur[s] = src[g * 2048 + s * 256 + threadIdx.x];
    for (int it = 0; g < ntiles; g += gridDim.x, ++it) {
        int *seg = seg2[it & 1], *red = red2[it & 1];
        const size_t gn = g + gridDim.x;
        if (gn < ntiles) for (int s = 0; s < 8; ++s) nxt[s] = src[gn * 2048 + s * 256 + threadIdx.x];
        unsigned long long m[8];
        int sum = 0;
        for (int s = 0; s < 8; ++s) {
            const int w = (int)(cur[s] & 0xff);
            m[s] = __ballot(w != 0 && ((threadIdx.x + s * 256 + g) % 50 == 0));     // ~2 % foreground
            if (lane == 0) seg[s * 4 + wave] = __popcll(m[s]);
            sum += w;
        }
        if (LEVEL >= 1) {
            for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
            if (lane == 0) red[wave] = sum;
        }
        if (LEVEL >= 2) {
            __syncthreads();
            if (threadIdx.x < 64) {
                int c = lane < 32 ? seg[lane] : 0, inc = c;
                for (int o = 1; o < 64; o <<= 1) { int n = __shfl_up(inc, o, 64); if (lane >= o) inc += n; }
                if (lane < 32) seg[lane] = inc - c;
                if (lane == 31) seg[32] = inc;
            }
            __syncthreads();
        }
        if (LEVEL >= 4 && threadIdx.x == 0) tiles[g] = (uint32_t)seg[32] | ((uint32_t)(red[0] + red[1] + red[2] + red[3]) << 12);
        if (LEVEL >= 3) {
            unsigned short *list = lists + g * 2048;
            for (int s = 0; s < 8; ++s)
                if ((m[s] >> lane) & 1ull) list[seg[s * 4 + wave] + __popcll(m[s] & ((1ull << lane) - 1ull))] = (unsigned short)(s * 256 + threadIdx.x);
        } else {
            for (int s = 0; s < 8; ++s) acc += __popcll(m[s]);
        }
        for (int s = 0; s < 8; ++s) cur[s] = nxt[s];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main()
{
    const size_t bytes = 64ull * 480 * 640 * 8;
    void *buf[3]; uint32_t *sink;
    for (auto &b : buf) { hipMalloc(&b, bytes); hipMemset(b, 1, bytes); }
    hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int cus = 256; hipDeviceProp_t p; if (hipGetDeviceProperties(&p, 0) == hipSuccess) cus = p.multiProcessorCount;
    auto time = [&](const char *name, auto launch) {
        for (int i = 0; i < 6; ++i) launch(buf[i % 3]);
        hipDeviceSynchronize();
        float best = 1e9f, sum = 0;
        for (int r = 0; r < 10; ++r) {
            hipEventRecord(e0);
            for (int i = 0; i < 6; ++i) launch(buf[i % 3]);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 6; sum += ms; if (ms < best) best = ms;
        }
        printf("%-44s %7.1f us  %7.1f GB/s (best %7.1f)\n", name, sum / 10 * 1e3, bytes / (sum / 10 * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e9);
    };
    unsigned short *lists; uint32_t *tiles;
    hipMalloc(&lists, bytes / 8 * 2); hipMalloc(&tiles, bytes / 16384 * 4);
    {
        const int grid = 8 * cus;
        printf("-- k_tile_scan's work added step by step, 8 blocks per CU\n");
        time("tile-wise + ballots (level 0)", [&](void *b) { hipLaunchKernelGGL((k_scanlike<0>), dim3(grid), dim3(256), 0, 0, (const unsigned long long *)b, bytes / 16384, sink, lists, tiles); });
        time("+ weight sum (1)", [&](void *b) { hipLaunchKernelGGL((k_scanlike<1>), dim3(grid), dim3(256), 0, 0, (const unsigned long long *)b, bytes / 16384, sink, lists, tiles); });
        time("+ segment scan, two barriers (2)", [&](void *b) { hipLaunchKernelGGL((k_scanlike<2>), dim3(grid), dim3(256), 0, 0, (const unsigned long long *)b, bytes / 16384, sink, lists, tiles); });
        time("+ ordered list stores (3)", [&](void *b) { hipLaunchKernelGGL((k_scanlike<3>), dim3(grid), dim3(256), 0, 0, (const unsigned long long *)b, bytes / 16384, sink, lists, tiles); });
        time("+ tile word (4)", [&](void *b) { hipLaunchKernelGGL((k_scanlike<4>), dim3(grid), dim3(256), 0, 0, (const unsigned long long *)b, bytes / 16384, sink, lists, tiles); });
        const int grid2 = 1920;
        time("level 4, 1920 blocks (5 tiles each, as the scan)", [&](void *b) { hipLaunchKernelGGL((k_scanlike<4>), dim3(grid2), dim3(256), 0, 0, (const unsigned long long *)b, bytes / 16384, sink, lists, tiles); });
    }
    for (int per_cu : {8}) {
        const int grid = per_cu * cus;
        printf("-- %d blocks per CU\n", per_cu);
        time("16 B/lane x 4 in flight, nontemporal", [&](void *b) { hipLaunchKernelGGL((k_read<u32x4, 4, true>), dim3(grid), dim3(256), 0, 0, (const u32x4 *)b, bytes / 16, sink); });
        time("16 B/lane x 4 in flight", [&](void *b) { hipLaunchKernelGGL((k_read<u32x4, 4, false>), dim3(grid), dim3(256), 0, 0, (const u32x4 *)b, bytes / 16, sink); });
        time("8 B/lane x 8 in flight", [&](void *b) { hipLaunchKernelGGL((k_read<u32x2, 8, false>), dim3(grid), dim3(256), 0, 0, (const u32x2 *)b, bytes / 8, sink); });
        time("8 B/lane x 16 in flight", [&](void *b) { hipLaunchKernelGGL((k_read<u32x2, 16, false>), dim3(grid), dim3(256), 0, 0, (const u32x2 *)b, bytes / 8, sink); });
        time("8 B/lane x 8, nontemporal", [&](void *b) { hipLaunchKernelGGL((k_read<u32x2, 8, true>), dim3(grid), dim3(256), 0, 0, (const u32x2 *)b, bytes / 8, sink); });
        time("tile-wise 16 KB, next tile ahead + ballots", [&](void *b) { hipLaunchKernelGGL(k_tiles, dim3(grid), dim3(256), 0, 0, (const unsigned long long *)b, bytes / 16384, sink); });
    }
    return 0;
}
