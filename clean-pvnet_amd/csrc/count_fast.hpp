// count_fast.hpp -- stage 3: packed-VALU inlier-count kernel with guard band.
// Part of the single translation unit pvnet_vote.hip (included inside its anonymous namespace); see that file
// for the numerical contract and the reference citations (K = ransac_voting_kernel.cu, P = ransac_voting_gpu.py).
#pragma once

// ---------------------------------------------------------------------------------------------
// Stage 3, fast form.  Same mapping (lanes are hypotheses, R per lane) but:
//   * the pixel arrives as a 32-byte PixelRec through a SCALAR load (s_load_dwordx8): zero VALU
//     cycles, the operands of every evaluation are VGPR(hypothesis) x SGPR(pixel);
//   * hypotheses are processed in PAIRS with packed fp32 (v_pk_add/mul/fma_f32: two evaluations per
//     issue slot -- measured 4.2 cycles per packed wave-instruction vs 4.1 for a scalar v_fma_f32 on
//     gfx950, tools/microbench);
//   * no sqrt, no divide.  With d = h - c (the SAME rounded subtraction as the exact path),
//     nh = n/|n| and kappa = T/sqrt(1-T^2):
//         a  = d . nh            = |d| cos(theta)
//         b' = kappa * d x nh    = kappa |d| sin(theta)
//         cos(theta) > T  <=>  t := a - |b'| > 0
//   * the decision is taken from t only when it is OUTSIDE a guard band,  |t| - beta*a > eps_abs;
//     inside it (about 3e-6 of all evaluations) the pixel is re-evaluated with the exact binary32
//     sequence of K:100-125 and the counters are corrected.  Derivation of beta:
//     the exact path's computed cosine deviates from the true one by <= 8u (u = 2^-24), which is
//     |d| 8u/(1-T^2) in t; the fast path's t deviates by <= 3u(1+kappa)|d|; a ~ T|d| in the band.
//     eps_abs covers the exact path's norm2 < 1e-6 reject (K:121): |d| <= 1e-6 => |t| <= (1+kappa)|d|.
//   * hypotheses that are not finite or beyond 1e15 px (where the exact path's squares overflow and
//     the bounds above stop holding) send the whole work item down the exact loop.
// Inlier counts stay bit-exact against the oracle; tests/test_gpu_parity.py hammers the band.
// ---------------------------------------------------------------------------------------------
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float8v __attribute__((ext_vector_type(8)));

struct FastConsts {
    float beta;     // relative half-width of the guard band (in units of a)
    float eps_abs;  // absolute floor of the band, px
};

// (a, b') of one hypothesis pair against the pixel held in SGPRs.  cxy=(cx,cy), nh=(nhx,nhy), Bv=(Bx,By).
__device__ __forceinline__ void pk_project(float2v hx2, float2v hy2, float2v cxy, float2v nh, float2v Bv,
                                           float2v &a2, float2v &b2)
{
    float2v dx2, dy2, p2, q2;
    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dx2) : "v"(hx2), "s"(cxy));
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dy2) : "v"(hy2), "s"(cxy));
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(p2) : "v"(dy2), "s"(nh));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(a2) : "v"(dx2), "s"(nh), "v"(p2));
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(q2) : "v"(dy2), "s"(Bv));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(b2) : "v"(dx2), "s"(Bv), "v"(q2));
}

// t = a - |b|: one v_sub_f32 with an abs source modifier (f32 add/sub issue in 2.4 cycles on gfx950, every
// other VALU op in ~4.2).  Kept as asm so the SLP vectoriser cannot turn two of them into
// v_and + v_and + v_pk_add.
__device__ __forceinline__ float sub_abs(float a, float b)
{
    float t;
    asm("v_sub_f32 %0, %1, |%2|" : "=v"(t) : "v"(a), "v"(b));
    return t;
}

template <int R>
__global__ __launch_bounds__(kBlock) void k_count_fast(
    const float8v *__restrict__ recs /*[B,K,cap]*/, const float2 *__restrict__ hyps /*[B,K,hn]*/,
    int *__restrict__ counts /*[B,K,hn]*/, const int *__restrict__ tn_arr, int B, int K, int hn, int cap,
    float thresh, FastConsts fc, int max_pix_per_wave, int target_items)
{
    static_assert(R % 2 == 0, "hypotheses are processed in pairs");
    __shared__ int item_end[kMaxBatchLds];
    __shared__ int s_ppw;
    const int lane = lane_id(), wave = wave_id();
    constexpr int HT = 64 * R;
    const int nht = (hn + HT - 1) / HT;
    const int per_chunk = K * nht;

    // Pixels one wave walks per work item: as many as max_pix_per_wave (amortises the hypothesis loads and
    // the final atomics) but few enough that the batch still splits into >= target_items items -- a
    // single 480x640 image must spread over 256 CUs too.  Every block derives the same value from tn[].
    if (wave == 0) {
        long long px = 0;
        for (int b = lane; b < B; b += 64) px += tn_arr[b];
        px = wave_sum(px) * per_chunk;
        long long want = px / (4ll * target_items);
        int ppw = (int)(want < 16 ? 16 : (want > max_pix_per_wave ? max_pix_per_wave : want));
        if (lane == 0) s_ppw = ppw;
    }
    __syncthreads();
    const int pix_per_wave = __builtin_amdgcn_readfirstlane(s_ppw);
    const int PC = 4 * pix_per_wave;

    const int total = build_item_table(item_end, tn_arr, 0, B, PC, per_chunk);

    for (int item = blockIdx.x; item < total; item += gridDim.x) {
        int local;
        const int b = locate_item(item_end, B, item, &local);
        const int chunk = local / per_chunk;
        const int rem = local - chunk * per_chunk;
        const int vi = rem / nht;
        const int ht = rem - vi * nht;
        const int tn = __builtin_amdgcn_readfirstlane(tn_arr[b]);
        const int bk = b * K + vi;

        // this lane's R hypotheses; lanes past hn get (0,0) and never write their counters
        float2v hx2[R / 2], hy2[R / 2];
        int cnt[R];
        bool far = false;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int h = ht * HT + r * 64 + lane;
            float2 hp = make_float2(0.f, 0.f);
            if (h < hn) hp = hyps[(size_t)bk * hn + h];
            hx2[r / 2][r & 1] = hp.x;
            hy2[r / 2][r & 1] = hp.y;
            cnt[r] = 0;
            far |= !(fabsf(hp.x) < 1e15f && fabsf(hp.y) < 1e15f);
        }
        const int p0 = chunk * PC + wave * pix_per_wave;
        const int p1 = min(tn, p0 + pix_per_wave);
        const float8v *rp = recs + (size_t)bk * cap;

        if (__builtin_expect(__any(far), 0)) {
            // exact loop (K:100-125) for the whole work item
            for (int p = p0; p < p1; ++p) {
                const float8v rec = rp[p];
                const float cx = rec[0], cy = rec[1], nx = rec[6], ny = rec[7];
#pragma unroll
                for (int r = 0; r < R; ++r)
                    cnt[r] += vote_exact(cx, cy, hx2[r / 2][r & 1], hy2[r / 2][r & 1], nx, ny, thresh) ? 1 : 0;
            }
        } else if (p1 > p0) {
            // Counting by sign bit: t < 0 (not an inlier) shifts a 1 into a per-hypothesis bit queue
            // (one v_alignbit_b32 per evaluation); every 32 pixels the queue is popcounted.  t = +0 is
            // always inside the guard band, so "sign bit clear" == "fast path says inlier".
            int neg[R];
#pragma unroll
            for (int r = 0; r < R; ++r) neg[r] = 0;
            for (int pp = p0; pp < p1; pp += 32) {
                unsigned acc[R];
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] = 0u;
                const int pe = min(p1, pp + 32);
                for (int p = pp; p < pe; ++p) {
                    const float8v rec = rp[p];     // wave-uniform address -> scalar load
                    const float2v cxy = {rec[0], rec[1]}, nh = {rec[2], rec[3]}, Bv = {rec[4], rec[5]};
                    float zmin = INFINITY;
#pragma unroll
                    for (int q = 0; q < R / 2; ++q) {
                        float2v a2, b2;
                        pk_project(hx2[q], hy2[q], cxy, nh, Bv, a2, b2);
                        const float t0 = sub_abs(a2[0], b2[0]);
                        const float t1 = sub_abs(a2[1], b2[1]);
                        acc[2 * q] = __builtin_amdgcn_alignbit(acc[2 * q], __float_as_uint(t0), 31);
                        acc[2 * q + 1] = __builtin_amdgcn_alignbit(acc[2 * q + 1], __float_as_uint(t1), 31);
                        const float z0 = __builtin_fmaf(-fc.beta, a2[0], fabsf(t0));
                        const float z1 = __builtin_fmaf(-fc.beta, a2[1], fabsf(t1));
                        zmin = fminf(fminf(zmin, z0), z1);     // one v_min3_f32
                    }
                    if (__builtin_expect(__any(zmin <= fc.eps_abs), 0)) {
                        // some evaluation of this pixel sits inside the guard band: replace the fast
                        // decisions of the pixel by the exact ones
                        const float cx = rec[0], cy = rec[1], nx = rec[6], ny = rec[7];
#pragma unroll
                        for (int q = 0; q < R / 2; ++q) {
                            float2v a2, b2;
                            pk_project(hx2[q], hy2[q], cxy, nh, Bv, a2, b2);
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                const float t = sub_abs(a2[e], b2[e]);
                                const int fast = (__float_as_uint(t) >> 31) ? 0 : 1;
                                const int exact = vote_exact(cx, cy, hx2[q][e], hy2[q][e], nx, ny, thresh) ? 1 : 0;
                                cnt[2 * q + e] += exact - fast;
                            }
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < R; ++r) neg[r] += __popc(acc[r]);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) cnt[r] += (p1 - p0) - neg[r];
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int h = ht * HT + r * 64 + lane;
            if (h < hn && cnt[r] != 0) atomicAdd(&counts[(size_t)bk * hn + h], cnt[r]);
        }
    }
}
