// count_exact.hpp -- stage 3: work-item table + the exact (sqrt/divide) inlier-count kernel.
// Part of the single translation unit pvnet_vote.hip (included inside its anonymous namespace); see that file
// for the numerical contract and the reference citations (K = ransac_voting_kernel.cu, P = ransac_voting_gpu.py).
#pragma once

// ---------------------------------------------------------------------------------------------
// Stage 3: inlier counting -- the hot kernel.  Replaces voting_for_hypothesis (K:88-167) +
// torch.sum(inlier, 2) (P:159 / P:243) without the [hn,vn,tn] byte scratch.
//
// Mapping (wave64): LANES ARE HYPOTHESES.  Each lane keeps R hypotheses of one keypoint and their
// R counters in VGPRs; the wave loads 64 compacted pixels with one coalesced load per array,
// then walks them one by one, broadcasting a pixel's (cx,cy,nx,ny,norm1) to SGPRs with
// v_readlane so that every evaluation is VGPR(hypothesis) x SGPR(pixel) arithmetic.  Counters
// are private per lane: no cross-lane reduction in the loop; one atomicAdd per (lane, r) per
// work item at the end (integer adds => order independent => bit-exact counts).
//
// Work item = (image b, keypoint vi, hypothesis tile of 64*R, pixel chunk of 4 waves x 64 px).
// The number of items depends on tn[b], which only the device knows, so the grid is persistent
// and every block derives the item list from tn[] itself (no host sync, no empty blocks).
// ---------------------------------------------------------------------------------------------
// ---- work-item table of the persistent count kernels ---------------------------------------------------------
// The number of pixel chunks of an image depends on tn[b], which only the device knows, so every block builds the
// same table itself: item_end[b] = inclusive prefix of (chunks of image b) * items_per_chunk.  No host sync, no
// empty blocks.  Returns the total number of items (valid in every thread after the barrier inside).
constexpr int kMaxBatchLds = 1024;  // images per launch (the table lives in LDS)

__device__ __forceinline__ int build_item_table(int *item_end, const int *__restrict__ tn_arr, int tn_fixed, int B,
                                                int pixels_per_chunk, int items_per_chunk)
{
    const int lane = lane_id();
    if (wave_id() == 0) {
        int carry = 0;
        for (int b0 = 0; b0 < B; b0 += 64) {
            const int b = b0 + lane;
            int n = 0;
            if (b < B) {
                const int tn = tn_arr ? tn_arr[b] : tn_fixed;
                n = ((tn + pixels_per_chunk - 1) / pixels_per_chunk) * items_per_chunk;
            }
            int inc = wave_incl_scan(n) + carry;
            if (b < B) item_end[b] = inc;
            carry = __builtin_amdgcn_readlane(inc, 63);
        }
    }
    __syncthreads();
    return item_end[B - 1];
}

// image of work item `item` (first b with item_end[b] > item) and the item's index within that image
__device__ __forceinline__ int locate_item(const int *item_end, int B, int item, int *local)
{
    int lo = 0, hi = B - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (item_end[mid] > item) hi = mid; else lo = mid + 1;
    }
    const int b = __builtin_amdgcn_readfirstlane(lo);
    *local = __builtin_amdgcn_readfirstlane(item - (b ? item_end[b - 1] : 0));
    return b;
}

struct CountArgs {
    const float2 *coords;  // pixel p of image b: coords[b*c_b + p]
    const float2 *dirs;    // dirs[b*d_b + vi*d_v + p*d_p]
    const float2 *hyps;    // hyps[b*h_b + vi*h_v + hi*h_h]
    int *counts;           // counts[b*h_b + vi*h_v + hi*h_h]
    const int *tn_arr;     // per image, or nullptr -> tn_fixed
    long long c_b, d_b, d_v, d_p, h_b, h_v, h_h;
    int tn_fixed;
    int B, K, hn;
    float thresh;
};

template <int R>
__global__ __launch_bounds__(kBlock) void k_count_inliers(CountArgs a)
{
    __shared__ int item_end[kMaxBatchLds];  // inclusive prefix of items per image
    const int lane = lane_id(), wave = wave_id();
    constexpr int HT = 64 * R;
    constexpr int PC = 4 * kPixPerWave;
    const int nht = (a.hn + HT - 1) / HT;
    const int per_chunk = a.K * nht;

    const int total = build_item_table(item_end, a.tn_arr, a.tn_fixed, a.B, PC, per_chunk);

    for (int item = blockIdx.x; item < total; item += gridDim.x) {
        int local;
        const int b = locate_item(item_end, a.B, item, &local);
        const int chunk = local / per_chunk;
        const int rem = local - chunk * per_chunk;
        const int vi = rem / nht;
        const int ht = rem - vi * nht;
        const int tn = __builtin_amdgcn_readfirstlane(a.tn_arr ? a.tn_arr[b] : a.tn_fixed);

        // this lane's R hypotheses (NaN => never an inlier => padding)
        float hx[R], hy[R];
        int cnt[R];
        const long long hbase = (long long)b * a.h_b + (long long)vi * a.h_v;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int h = ht * HT + r * 64 + lane;
            float2 hp = make_float2(NAN, NAN);
            if (h < a.hn) hp = a.hyps[hbase + (long long)h * a.h_h];
            hx[r] = hp.x;
            hy[r] = hp.y;
            cnt[r] = 0;
        }

        const int p0 = chunk * PC + wave * kPixPerWave;
        const int nvalid = min(kPixPerWave, tn - p0);  // wave-uniform
        if (nvalid > 0) {
            float cx = 0.f, cy = 0.f, nx = 0.f, ny = 0.f;
            if (lane < nvalid) {
                float2 c = a.coords[(long long)b * a.c_b + p0 + lane];
                float2 d = a.dirs[(long long)b * a.d_b + (long long)vi * a.d_v +
                                  (long long)(p0 + lane) * a.d_p];
                cx = c.x; cy = c.y; nx = d.x; ny = d.y;
            }
            float norm1 = sqrtf(nx * nx + ny * ny);
            if (lt_1e6(norm1)) norm1 = NAN;  // K:121 reject, folded into the quotient below

            for (int j = 0; j < nvalid; ++j) {
                const float scx = bcast(cx, j), scy = bcast(cy, j);
                const float snx = bcast(nx, j), sny = bcast(ny, j);
                const float sn1 = bcast(norm1, j);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    float dx = hx[r] - scx;
                    float dy = hy[r] - scy;
                    float norm2 = sqrtf(dx * dx + dy * dy);
                    float angle = (dx * snx + dy * sny) / (sn1 * norm2);
                    cnt[r] += (!lt_1e6(norm2) && angle > a.thresh) ? 1 : 0;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int h = ht * HT + r * 64 + lane;
            if (h < a.hn && cnt[r] != 0) atomicAdd(&a.counts[hbase + (long long)h * a.h_h], cnt[r]);
        }
    }
}
