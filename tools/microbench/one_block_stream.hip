// one_block_stream.hip -- how fast can a FEW blocks stream one image's mask?  (VERDICT r3 #7 proposed, for B <= 4: "let ONE
// block per image walk its tiles with read-ahead and write the compacted rows itself".)  A 480x640 int64 mask is 2.4 MB =
// 150 tiles of 16 KB.  This benchmark reads that buffer with 1, 2, 4, 8, 16, 32, 150 blocks of 256 threads, every thread
// with its next tile's eight 8-byte loads in flight while it reduces the current one (the access shape of k_tile_scan's
// read-ahead instantiation), on a cold buffer (a different 2.4 MB slice of a 1 GB allocation per launch).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/one_block_stream.hip -o /tmp/obs && /tmp/obs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int kBlock = 256, kSteps = 8, kTile = kBlock * kSteps;

__global__ __launch_bounds__(kBlock) void walk(const uint64_t *__restrict__ mask, int tiles, unsigned *__restrict__ out)
{
    unsigned acc = 0;
    uint64_t cur[kSteps], nxt[kSteps];
    int t = blockIdx.x;
    if (t < tiles)
#pragma unroll
        for (int s = 0; s < kSteps; ++s) cur[s] = mask[(size_t)t * kTile + s * kBlock + threadIdx.x];
    for (; t < tiles; t += gridDim.x) {
        const int tn = t + gridDim.x;
        if (tn < tiles)
#pragma unroll
            for (int s = 0; s < kSteps; ++s) nxt[s] = mask[(size_t)tn * kTile + s * kBlock + threadIdx.x];
#pragma unroll
        for (int s = 0; s < kSteps; ++s) acc += __popcll(__ballot((cur[s] & 0xff) != 0));
#pragma unroll
        for (int s = 0; s < kSteps; ++s) cur[s] = nxt[s];
    }
    if (threadIdx.x == 0) atomicAdd(out, acc);
}

int main()
{
    const size_t total = (size_t)1 << 30;                      // 1 GB: every launch reads a slice nothing has touched recently
    const int tiles = 150;
    const size_t slice = (size_t)tiles * kTile * 8;
    uint64_t *buf; unsigned *out;
    hipMalloc(&buf, total); hipMalloc(&out, 4);
    hipMemset(buf, 1, total); hipMemset(out, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("one 480x640 int64 mask = %d tiles = %.2f MB, cold\n", tiles, slice / 1e6);
    size_t cursor = 0;
    for (int blocks : {1, 2, 4, 8, 16, 32, 64, 150}) {
        float best = 1e9f, sum = 0.f;
        const int reps = 12;
        for (int r = 0; r < reps; ++r) {
            cursor = (cursor + slice * 7) % (total - slice);
            cursor &= ~(size_t)4095;
            hipEventRecord(e0);
            hipLaunchKernelGGL(walk, dim3(blocks), dim3(kBlock), 0, 0, buf + cursor / 8, tiles, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (r >= 2) { sum += ms; if (ms < best) best = ms; }
        }
        printf("%3d blocks: %7.2f us mean, %7.2f us best  (%.1f GB/s)\n", blocks, 1e3 * sum / (reps - 2), 1e3 * best, slice / (best * 1e-3) / 1e9);
    }
    return 0;
}
