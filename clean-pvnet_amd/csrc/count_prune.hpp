// count_prune.hpp -- stage 3, between two launches of k_count_bf16<true>: exact elimination of hypotheses that can no
// longer win (ransac_voting_layer_v3 only).
// Part of the single translation unit pvnet_vote.hip (included inside its anonymous namespace); see that file
// for the numerical contract and the reference citations (K = ransac_voting_kernel.cu, P = ransac_voting_gpu.py).
#pragma once

// ---------------------------------------------------------------------------------------------
// ransac_voting_layer_v3 keeps, per (image, keypoint), the FIRST hypothesis with the maximal inlier count and that count
// (torch.max over hn, P:160; ratio update P:162-167); the other hn - 1 counts never leave the layer.  So after the
// chunks of the stages run so far (a fraction f of the image's pixels, spread over the object) this kernel
//   1. takes a leader per wavefront (maximal PARTIAL count among the wave's share of the alive hypotheses, first index
//      among ties -- the four leaders include the overall partial leader) and counts it EXACTLY (K:100-125) over every
//      pixel not counted yet: four exactly known FULL counts, L* = the largest;
//   2. keeps hypothesis h alive iff  partial(h) + R >= L*,  R = the number of pixels not counted yet.  A dropped h has
//      full(h) <= partial(h) + R < L* <= max: it is neither the winner nor tied with it, and what stays in counts[] for
//      it (its partial count) is below the maximum, so the arg-max kernel (refit.hpp) is unaffected.  Every hypothesis
//      whose full count equals the maximum survives every stage and ends with its exact count: winner index (first
//      among ties), winner count and therefore the refit are those of the full pass, bit for bit;
//   3. writes the survivors -- indices and coordinates, dense, in index order -- for the next stage's launch.
// One block per (image, keypoint).  With a winner that explains nearly every pixel (LINEMOD-like: ratio ~0.99) a first
// stage over a quarter of the chunks leaves ~15 % of the hypotheses alive; with 30-50 % outlier pixels (config 4) the
// bound bites late and little is saved.
// ---------------------------------------------------------------------------------------------
struct PruneArgs {
    const int *tn_arr;
    const float2 *coords;    // [B,cap]
    const float2 *dirs;      // [B,K,cap]
    const float2 *hyps;      // [B,K,hn]
    const int *counts;       // [B,K,hn] partial counts (of the chunks in done_mask)
    const int *idx_in;       // [B,K,hn] alive before this prune, or nullptr = all hn
    const int *ns_in;        // [B,K] or nullptr
    float2 *hyp_out;         // [B,K,hn]
    int *idx_out;            // [B,K,hn]
    int *ns_out;             // [B,K]
    int K, hn, cap;
    float thresh;
    uint32_t done_mask;      // residues (mod M) of the 512-pixel chunks counted so far
    int M;
};

__global__ __launch_bounds__(kBlock) void k_prune(PruneArgs a)
{
    __shared__ int s_full[4], s_rem[4], s_red[4];
    const int vi = blockIdx.x, b = blockIdx.y, bk = b * a.K + vi;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    constexpr int PC = 4 * kBfPixPerWave;                          // the count kernel's chunk: 512 pixels
    const int tn = a.tn_arr[b];
    const int n_in = a.ns_in ? a.ns_in[bk] : a.hn;
    if (tn <= 0 || n_in <= 0) {                                    // image skipped (P:129-132): no stage has items for it
        if (threadIdx.x == 0) a.ns_out[bk] = 0;
        return;
    }
    const int *cp = a.counts + (size_t)bk * a.hn;
    const int *ip = a.idx_in ? a.idx_in + (size_t)bk * a.hn : nullptr;
    const float2 *hp = a.hyps + (size_t)bk * a.hn;

    // ---- 1. this wave's leader: maximal partial count, first index among ties
    int best = -1, besth = 0x7fffffff;
    for (int i = threadIdx.x; i < n_in; i += kBlock) {
        const int h = ip ? ip[i] : i;
        const int c = cp[h];
        if (c > best || (c == best && h < besth)) { best = c; besth = h; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int oc = __shfl_xor(best, o, 64), oh = __shfl_xor(besth, o, 64);
        if (oc > best || (oc == best && oh < besth)) { best = oc; besth = oh; }
    }
    // ---- its exact count over the pixels no stage has counted yet (K:100-125), four pixels per lane in flight
    const float2 *crd = a.coords + (size_t)b * a.cap;
    const float2 *dir = a.dirs + (size_t)bk * a.cap;
    const float2 lead = best >= 0 ? hp[besth] : make_float2(0.f, 0.f);
    int inl = 0, rem = 0;
    const int nch = (tn + PC - 1) / PC;
    for (int c = 0; c < nch; ++c) {
        if ((a.done_mask >> (c % a.M)) & 1u) continue;             // wave-uniform
        const int p0 = c * PC, pe = min(tn, p0 + PC);
        for (int p = p0 + lane; p < pe; p += 4 * 64) {
            float2 cc[4], dd[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = p + u * 64;
                cc[u] = q < pe ? crd[q] : make_float2(0.f, 0.f);
                dd[u] = q < pe ? dir[q] : make_float2(0.f, 0.f);   // zero direction: norm1 < 1e-6, never an inlier
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                inl += (best >= 0 && vote_exact(cc[u].x, cc[u].y, lead.x, lead.y, dd[u].x, dd[u].y, a.thresh)) ? 1 : 0;
                rem += p + u * 64 < pe ? 1 : 0;
            }
        }
    }
    inl = wave_sum(inl);
    rem = wave_sum(rem);
    if (lane == 0) { s_full[wave] = best >= 0 ? best + inl : -1; s_rem[wave] = rem; }
    __syncthreads();
    const int lstar = max(max(s_full[0], s_full[1]), max(s_full[2], s_full[3]));
    const int R = s_rem[0];                                        // every wave walked the same pixels

    // ---- 2./3. survivors, in index order
    const size_t row = (size_t)bk * a.hn;
    int base = 0;
    for (int i0 = 0; i0 < n_in; i0 += kBlock) {
        const int i = i0 + threadIdx.x;
        int h = 0;
        bool keep = false;
        if (i < n_in) {
            h = ip ? ip[i] : i;
            keep = cp[h] + R >= lstar;
        }
        const unsigned long long m = __ballot(keep);
        __syncthreads();
        if (lane == 0) s_red[wave] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += s_red[w];
        off += __popcll(m & ((1ull << lane) - 1ull));
        if (keep) {
            a.idx_out[row + off] = h;
            a.hyp_out[row + off] = hp[h];
        }
        base += s_red[0] + s_red[1] + s_red[2] + s_red[3];
    }
    if (threadIdx.x == 0) a.ns_out[bk] = base;
}
