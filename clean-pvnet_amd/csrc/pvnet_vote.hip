// pvnet_vote.hip -- RANSAC voting hot path of clean-pvnet as native HIP for gfx950 (MI355X).
//
// Written for CDNA4 only: 64-lane wavefronts are assumed everywhere (ballot masks are 64 bit,
// lane broadcasts use v_readlane), there is no CUDA path and no CPU fallback.
//
// Arithmetic contract (see oracle/vote_oracle.c): the inlier decision and the hypothesis
// intersection are IEEE binary32 with one rounding per source-level operation and NO fused
// multiply-add, so inlier counts are bit-exact against the oracle.  This file must be built
// with -ffp-contract=off; the pragma below enforces it even if the flag is forgotten.
//
// Reference behaviour restated (paths relative to /root/reference):
//   K = lib/csrc/ransac_voting/src/ransac_voting_kernel.cu
//   P = lib/csrc/ransac_voting/ransac_voting_gpu.py
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>

#include "pvnet_vote.h"

#pragma clang fp contract(off)

#define PVV_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

constexpr int kBlock = 256;          // 4 wavefronts
constexpr int kTileSteps = 8;
constexpr int kTile = kBlock * kTileSteps;  // pixels per compaction tile
// Images of up to this many 2048-pixel tiles (480x640 = 150) subsample inside k_compact_hyp: one launch less on every
// call (-3 % at B = 64, -5 % at B = 1 on MI355X).  When subsampling does trigger, every block redoes the draws of the
// tiles before it, a cost that grows with the square of the tile count: at 190 tiles (540x720, BASELINE config 5, whose
// 31 k foreground pixels ARE subsampled) it already outweighs the launch (+2 %), so larger images keep k_tile_subsample.
constexpr int kFuseSubTiles = 160;
constexpr int kMaxTiles = 16000;     // tile prefix of an image in (dynamic) LDS: 64 KB => images up to ~32 Mpixel
constexpr int kPixPerWave = 64;      // pixels one wave walks per work item of the exact count kernel

thread_local char g_err[512] = "";

int fail(int code, const char *msg)
{
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return PVV_OK;
}

#include "vote_common.hpp"
#include "compaction.hpp"
#include "count_exact.hpp"
#include "count_bf16.hpp"
#include "count_filter_runs.hpp"
#include "count_prune.hpp"
#include "refit.hpp"
#include "covariance.hpp"
#include "legacy_kernels.hpp"

// ---------------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------------
size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

struct Layout {
    int T;
    size_t tiles, tile_list, tile_draw, tn, surv, coords, dirs, hyps, counts, sums, total;
    size_t lead;                                     // staged counting (count_prune.hpp): [B,K,8] leader counts; 0 = not reserved
    size_t ratio;                                    // [B,K] winner count / tn (k_select_refit -> k_finalize_v3 -> stage hint)
    size_t miss;                                     // staged counting: [B,K,hn] shared miss counters (count_filter_runs.hpp); 0 = not reserved
};

bool use_bf16_count(const pvv_problem *p);
bool may_stage(const pvv_problem *p);

// Subsampling inside k_compact_hyp (no k_tile_subsample launch): only for images of <= kFuseSubTiles tiles and only when
// subsampling is unlikely -- max_num at least 1/16 of the image (30000 of 307200) -- and (make_front) only with the device RNG.
bool fuse_sub_shape(const pvv_problem *p, int T)
{
    return T <= kFuseSubTiles && (long long)p->max_num * 16 >= (long long)p->H * p->W;
}

// Does a call of this problem ever STORE per-pixel subsample draws (tile_draw: 4 B per pixel of the batch)?  Only the
// separate subsample pass reads them (k_tile_subsample; make_front: want_draws).  Whether that pass runs depends on pointer
// arguments the workspace query cannot see -- injected index pairs or draws switch the fused subsampling off -- unless the
// caller promises the device RNG (PVV_FLAG_DEVICE_RNG): then draws are stored only for images too large to fuse, and never
// when no mask of this shape can exceed max_num (the largest foreground_num: 255 per pixel, P:126).
bool may_store_draws(const pvv_problem *p, int T)
{
    // (the largest per-pixel weight: a byte mask's 255, or the fused argmax's class index seg_classes - 1 -- make_front's max_weight;
    // the workspace query cannot see which of the two the call will be, so the larger one decides: ADVICE r5)
    const long long max_weight = std::max(255ll, (long long)p->seg_classes - 1);
    if ((long long)p->max_num >= max_weight * p->H * p->W) return false;
    return !((p->flags & PVV_FLAG_DEVICE_RNG) && fuse_sub_shape(p, T));
}

// one set of leader words: [B,K,8] + the any_staged word
size_t lead_set_words(const pvv_problem *p) { return (size_t)p->B * p->K * 8 + 1; }

Layout make_layout(const pvv_problem *p)
{
    Layout L;
    const size_t HW = (size_t)p->H * p->W;
    L.T = (int)((HW + kTile - 1) / kTile);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes); return o; };
    L.tiles = take(sizeof(uint32_t) * (size_t)p->B * L.T);
    L.tile_list = take(sizeof(unsigned short) * (size_t)p->B * L.T * kTile);
    L.tile_draw = may_store_draws(p, L.T) ? take(sizeof(float) * (size_t)p->B * L.T * kTile) : 0;   // (0: not reserved)
    L.tn = take(sizeof(int) * (size_t)p->B);
    L.surv = take(sizeof(int) * (size_t)p->B * kSurvCap);
    L.coords = take(sizeof(float2) * (size_t)p->B * p->cap);
    L.dirs = take(sizeof(float2) * (size_t)p->B * p->K * p->cap);
    L.hyps = take(sizeof(float2) * (size_t)p->B * p->K * p->hn);
    L.counts = take(sizeof(int) * (size_t)p->B * p->K * p->hn);
    L.sums = take(sizeof(double) * (size_t)p->B * p->K * kRefitSplitMax * 5);
    L.ratio = take(sizeof(float) * (size_t)p->B * p->K);
    // (+ the any_staged word; TWO sets: the fused un_pnp call counts its v3 hypotheses and its estimate's as two staged passes)
    L.lead = may_stage(p) ? take(sizeof(int) * 2 * lead_set_words(p)) : 0;
    L.miss = may_stage(p) ? take(sizeof(int) * (size_t)p->B * p->K * p->hn) : 0;
    L.total = off;
    return L;
}

int validate(const pvv_problem *p)
{
    if (!p) return fail(PVV_E_ARG, "problem is NULL");
    if (p->B <= 0 || p->H <= 0 || p->W <= 0 || p->K <= 0 || p->hn <= 0)
        return fail(PVV_E_ARG, "B, H, W, K, hn must be positive");
    if (p->B > kMaxBatchLds) return fail(PVV_E_ARG, "B > 1024: split the batch");
    if (((long long)p->H * p->W + kTile - 1) / kTile > kMaxTiles) return fail(PVV_E_ARG, "H*W too large (more than 16000 tiles of 2048 pixels)");
    if ((long long)p->K * p->hn >= (1ll << 23)) return fail(PVV_E_ARG, "K*hn must be < 2^23");
    if (p->count_kernel < PVV_COUNT_AUTO || p->count_kernel > PVV_COUNT_STAGED_ESTIMATE) return fail(PVV_E_ARG, "unknown count_kernel");
    if (p->flags & ~PVV_FLAG_DEVICE_RNG) return fail(PVV_E_ARG, "unknown bits in flags");
    if (p->mask_elem_size != 1 && p->mask_elem_size != 2 && p->mask_elem_size != 4 &&
        p->mask_elem_size != 8)
        return fail(PVV_E_ARG, "mask_elem_size must be 1, 2, 4 or 8");
    if (p->cap <= 0 || (long long)p->cap > (long long)p->H * p->W)
        return fail(PVV_E_ARG, "cap must be in [1, H*W]");
    if ((long long)p->B * p->K * p->hn >= (1ll << 31) || (long long)p->B * p->K * p->cap >= (1ll << 40))
        return fail(PVV_E_ARG, "problem too large for one call");
    if (p->singular_policy < PVV_SINGULAR_REFERENCE || p->singular_policy > PVV_SINGULAR_IMAGE_ZERO)
        return fail(PVV_E_ARG, "unknown singular_policy");
    return PVV_OK;
}

int num_cus()
{
    // per device: one process may drive several GPUs (the bench and RCCL use one process per GPU, tests need not)
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) {
        hipDeviceProp_t prop;
        int n = 0;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        cus[dev] = n > 0 ? n : 256;
    }
    return cus[dev];
}

int launch_count(const CountArgs &a, hipStream_t st)
{
    const int grid = num_cus() * 8;
    if (a.hn <= 64)
        hipLaunchKernelGGL(k_count_inliers<1>, dim3(grid), dim3(kBlock), 0, st, a);
    else if (a.hn <= 128)
        hipLaunchKernelGGL(k_count_inliers<2>, dim3(grid), dim3(kBlock), 0, st, a);
    else if (a.hn <= 256)
        hipLaunchKernelGGL(k_count_inliers<4>, dim3(grid), dim3(kBlock), 0, st, a);
    else
        hipLaunchKernelGGL(k_count_inliers<8>, dim3(grid), dim3(kBlock), 0, st, a);
    return check_launch("k_count_inliers");
}

// The matrix-core prefilter needs 0 < T < 1 with a sane kappa and pixel coordinates well inside the range where the
// block extents of its guard band are exact; outside [0.5, 0.99995], for huge images, and when the caller asks for it
// (pvv_problem.count_kernel = PVV_COUNT_EXACT: the tests' cross-check) the exact kernel counts.
bool use_bf16_count(const pvv_problem *p)
{
    return p->count_kernel != PVV_COUNT_EXACT && p->inlier_thresh >= 0.5f && p->inlier_thresh <= 0.99995f &&
           p->H <= 16384 && p->W <= 16384;
}

// Staged counting with exact elimination (count_bf16.hpp / count_prune.hpp) is something only ransac_voting_layer_v3
// can use (it keeps the arg-max; the estimate weighs every hypothesis).  may_stage: the workspace reserves the leader
// words (a property of the problem alone, so that pvv_workspace_bytes needs no extra argument); a v3 call whose problem
// may stage takes the staged path.  PVV_COUNT_STAGED forces it wherever the matrix-core kernel is valid, PVV_COUNT_FULL
// forbids it, AUTO stages when the batch is large enough for the two extra launches (k_lead + the second count launch)
// to pay.  The host knows neither tn nor the winners' inlier ratios, so the rule is a proxy for the evaluations of a full
// pass, B*K*hn*H*W >= 2e10 -- measured on MI355X (tools/staged_ab.py): 480x640, K = 9, 512 hypotheses breaks even at
// B = 16 (2.3e10), +4 % at B = 32, +24 % at B = 64; 540x720, K = 17, 2048 hypotheses at B = 16 (2.2e11) +86 % -- and the
// DEVICE refines it per image: images of fewer than 8 chunks (tn <= 3584) are counted completely by the first launch, and
// when no image of the batch is staged the two later launches leave at their first instruction (config 4's sparse masks
// at B = 32: the call then costs ~5 us more than the full pass, the price of not knowing tn on the host).
// Round 4, later: that proxy is calibrated on LINEMOD-like frames (2 % foreground).  T-LESS votes on detector crops (128x128 / 256x256,
// a third of the pixels foreground, B = #detections): 16 crops of 256x256 are 1.7e9 evaluations -- the benchmark's 64
// frames are 1.8e9 -- behind a proxy of 4.8e9, and staging them is worth +34 % per call (profiles/r04_experiments.txt (17)).  So the
// decision now uses the evaluations themselves whenever the stage hint knows the last call's tn (work_equivalent: K*hn*sum(tn) scaled
// to the proxy's units, x 50 = 1 / 0.02), the proxy only before any call has reported; and may_stage -- what reserves the leader words
// and lets a call leave its hint -- asks whether the problem COULD reach the bound: B*K*hn*min(H*W, cap) evaluations at most.
constexpr double kStageMinWork = 2e10;
constexpr double kStageProxyFg = 0.02;      // tn / (H*W) of the frames the proxy and the threshold table were measured on
bool may_stage(const pvv_problem *p)
{
    if (!use_bf16_count(p) || p->count_kernel == PVV_COUNT_FULL) return false;
    if (p->count_kernel == PVV_COUNT_STAGED || p->count_kernel == PVV_COUNT_STAGED_ESTIMATE) return true;
    // cap bounds tn: with fewer than 8 chunks of rows reserved per image nothing can ever be staged (the reference's default
    // call, max_num = 100: 244 rows)
    const double rows = std::min((double)p->H * p->W, (double)p->cap);
    return p->hn >= 128 && p->cap >= kStageMinChunks * 4 * kBfPixPerWave &&
           (double)p->B * p->K * p->hn * rows >= kStageMinWork * kStageProxyFg;
}
double stage_proxy_work(const pvv_problem *p) { return (double)p->B * p->K * p->hn * p->H * p->W; }

// ---------------------------------------------------------------------------------------------
// The stage hint.  Whether staged counting pays depends on how clean the vector field is -- on the winners' inlier ratio
// rho, which the host cannot know before the call: measured on MI355X (tools/staged_ab.py --outlier .., profiles/DESIGN_rounds_1-4.md 4.6),
// 480x640, K = 9, 512 hypotheses at B = 64: +26 % at rho = 0.995, +5 % at 0.95, +-0 at 0.90, -4 % at 0.80 (a quarter of
// the pixels then eliminates nothing and the call pays for the two extra launches); at B = 16 the break-even is rho ~
// 0.97; 540x720, K = 17, 2048 hypotheses gains down to rho = 0.8 and below.  But consecutive calls see similar data (a
// video, a dataset, one network), and every call -- staged or not -- ends with the exact winner counts.  So
// k_finalize_v3 leaves each image's mean winner ratio in a small host-visible array (one per device, pinned, written by
// the GPU with plain stores), and AUTO stages a call only if the ratios the LAST completed calls left there reach a
// threshold that depends on the problem's size.  The hint lags by the calls still in flight and mixes calls when several
// streams interleave (a hint left by another problem SHAPE -- H, W, K, hn; the batch size may vary -- is ignored); it only ever selects between two exact paths.  No data yet (first call, or first
// call under stream capture, where no memory can be pinned): stage.  PVV_COUNT_STAGED / PVV_COUNT_FULL ignore it.
// ---------------------------------------------------------------------------------------------
struct StageHint {
    float *ratio = nullptr;   // 2 x kMaxBatchLds floats, hipHostMalloc: [b] winner ratio (< -1.5 = never written, -1 = skipped
                              // image), [kMaxBatchLds + b] the image's tn
    int n = 0;                // images of the last call that was given the buffer
    int shape[5] = {0, 0, 0, 0, 0};   // B, H, W, K, hn of that call: a hint left by another problem shape is ignored.
                                      // The BATCH SIZE is not part of the shape that must match: a detector hands over a different number
                                      // of crops with every frame; sums over the last call's images are scaled to this call's B
    bool tried = false;
};
StageHint g_hint[64];
std::mutex g_hint_mu;        // guards every field of g_hint[] (the GPU writes only the pinned ratio[] words)

// the device the STREAM belongs to (a caller may launch on a stream of a device that is not current)
int stream_device(hipStream_t st)
{
    int dev = -1;
    if (st) {
        hipDevice_t d;
        if (hipStreamGetDevice(st, &d) == hipSuccess) dev = (int)d; else (void)hipGetLastError();
    }
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return -1; }     // (no sticky error on GPU-less hosts)
    return dev >= 0 && dev < 64 ? dev : -1;
}

// the device's hint slot, with its pinned array allocated on first use (never while the stream is capturing).  Call with
// g_hint_mu held.
StageHint *stage_hint_locked(hipStream_t st, bool allocate)
{
    const int dev = stream_device(st);
    if (dev < 0) return nullptr;
    StageHint *g = &g_hint[dev];
    if (!g->tried && allocate) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (cs != hipStreamCaptureStatusNone) return nullptr;            // pinning memory is not allowed while capturing
        g->tried = true;
        void *q = nullptr;
        if (hipHostMalloc(&q, sizeof(float) * 2 * kMaxBatchLds, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); q = nullptr; }
        g->ratio = (float *)q;
        if (g->ratio) for (int i = 0; i < 2 * kMaxBatchLds; ++i) g->ratio[i] = -2.f;
    }
    return g->ratio ? g : nullptr;
}

bool hint_same_shape(const StageHint *g, const pvv_problem *p, bool any_hn = false)
{
    return g->shape[1] == p->H && g->shape[2] == p->W && g->shape[3] == p->K && (any_hn || g->shape[4] == p->hn);
}

// A v3 call is about to leave its winners' ratios in the device's slot: record whose they are and hand out the array.
float *stage_hint_claim(const pvv_problem *p, hipStream_t st)
{
    std::lock_guard<std::mutex> lock(g_hint_mu);
    StageHint *g = stage_hint_locked(st, true);
    if (!g) return nullptr;
    const int shape[5] = {p->B, p->H, p->W, p->K, p->hn};
    if (!hint_same_shape(g, p)) {
        // another problem shape: what the array holds says nothing about this one -- until this call reports, no data
        for (int i = 0; i < p->B; ++i) g->ratio[i] = -2.f;
    } else {
        // more images than the last call had: the new slots hold nothing of this shape until this call reports
        for (int i = g->n; i < p->B; ++i) g->ratio[i] = -2.f;
    }
    memcpy(g->shape, shape, sizeof(shape));
    g->n = p->B;
    return g->ratio;
}

// mean winner ratio of the images the last calls OF THIS SHAPE reported (< 0: no data) and the largest tn among them
// (any_hn: the hint of a v3 call on the same fields with ANOTHER hypothesis count counts too -- the estimate's question)
float stage_hint_mean(const pvv_problem *p, hipStream_t st, float *max_tn = nullptr, double *sum_tn = nullptr, bool any_hn = false)
{
    if (max_tn) *max_tn = -1.f;
    if (sum_tn) *sum_tn = -1.0;
    std::lock_guard<std::mutex> lock(g_hint_mu);
    StageHint *g = stage_hint_locked(st, false);
    if (!g || g->n <= 0) return -1.f;
    if (p && !hint_same_shape(g, p, any_hn)) return -1.f;
    double sum = 0, stn = 0;
    int cnt = 0;
    float mt = 0.f;
    const volatile float *r = g->ratio;
    for (int i = 0; i < g->n; ++i) {
        const float v = r[i];
        if (v < -1.5f) return -1.f;                                       // an image of the last call has not reported yet
        if (v >= 0.f) { sum += v; ++cnt; }
        mt = std::max(mt, (float)r[kMaxBatchLds + i]);
        stn += std::max(0.f, (float)r[kMaxBatchLds + i]);
    }
    if (max_tn) *max_tn = mt;
    if (sum_tn && cnt) *sum_tn = p ? stn * (double)p->B / (double)g->n : stn;   // (scaled to this call's batch size)
    return cnt ? (float)(sum / cnt) : -1.f;
}

// Chunks the SECOND launch of a staged pass would walk, summed over the images the last call of this shape reported (-1: no
// data): the host's preview of the run length k_count_filter_runs will pick (count_filter_runs.hpp).
long long stage_hint_rest_chunks(const pvv_problem *p, hipStream_t st, uint32_t rest_mask)
{
    std::lock_guard<std::mutex> lock(g_hint_mu);
    StageHint *g = stage_hint_locked(st, false);
    if (!g || g->n <= 0) return -1;
    if (!hint_same_shape(g, p)) return -1;
    const volatile float *r = g->ratio;
    long long total = 0;
    constexpr int PC = 4 * kBfPixPerWave;
    for (int i = 0; i < g->n; ++i) {
        if (r[i] < -1.5f) return -1;                                      // an image of the last call has not reported yet
        const int nch = ((int)r[kMaxBatchLds + i] + PC - 1) / PC;
        if (nch >= kStageMinChunks)
            total += (nch / kStageM) * __builtin_popcount(rest_mask) + __builtin_popcount(rest_mask & ((1u << (nch % kStageM)) - 1u));
    }
    return total * p->B / g->n;                                           // (scaled to this call's batch size)
}

// The break-even winner ratio, from one-process A/B measurements of whole calls on MI355X with the round-4 second launch
// (k_count_filter_runs; tools/staged_ab.py --outlier ..., profiles/r04_staged_ab_outliers.json, profiles/DESIGN_rounds_1-4.md 4.7): 480x640, K = 9,
// 512 hypotheses breaks even at rho = 0.990 for B = 16, 0.966 for 24, 0.957 for 32, 0.91 for 48, 0.765 for 64 and below 0.6
// for 128 (round 3's kernel: 0.976 / 0.941 / 0.906 at B = 16 / 32 / 64); 540x720, K = 17, 2048 hypotheses at B = 16 gains at
// every ratio measured (+10 % at 0.71).  Interpolated in x = log2(B*K*hn*H*W / 2.26e10); more hypotheses per keypoint
// lower it further.
float stage_hint_threshold(const pvv_problem *p, double work)
{
    static const double xs[6] = {0.0, 0.585, 1.0, 1.585, 2.0, 3.0}, ys[6] = {0.990, 0.966, 0.957, 0.910, 0.765, 0.50};
    const double x = std::log2(std::max(work, 1.0) / 2.26e10);
    double thr;
    if (x <= xs[0]) thr = ys[0] + (ys[1] - ys[0]) / (xs[1] - xs[0]) * (x - xs[0]);
    else if (x >= xs[5]) thr = ys[5];
    else {
        int i = 0;
        while (x > xs[i + 1]) ++i;
        thr = ys[i] + (ys[i + 1] - ys[i]) * (x - xs[i]) / (xs[i + 1] - xs[i]);
    }
    thr -= 0.1 * std::log2(std::max(p->hn, 512) / 512.0);
    return (float)std::min(0.995, std::max(0.5, thr));
}

// The evaluations of a full pass in the proxy's units (B*K*hn*H*W of a frame with 2 % foreground): K*hn*sum(tn) / 0.02 when the last
// call of this shape has reported its images' tn, the proxy itself otherwise.  *mean / *max_tn: the hint's (see stage_hint_mean).
double stage_work(const pvv_problem *p, hipStream_t st, float *mean = nullptr, float *max_tn = nullptr)
{
    double sum_tn = -1.0;
    float mt = -1.f;
    const float m = stage_hint_mean(p, st, &mt, &sum_tn);
    if (mean) *mean = m;
    if (max_tn) *max_tn = mt;
    if (m < 0.f || sum_tn < 0.0) return stage_proxy_work(p);
    return (double)p->K * p->hn * sum_tn / kStageProxyFg;
}

bool stage_hint_allows(const pvv_problem *p, hipStream_t st)
{
    if (p->count_kernel != PVV_COUNT_AUTO) return true;
    float max_tn = -1.f, m = -1.f;
    const double work = stage_work(p, st, &m, &max_tn);
    if (work < kStageMinWork) return false;                               // too few evaluations for the two extra launches to pay
    if (m < 0.f) return true;                                             // no data yet: by the proxy alone
    // (images of fewer than kStageMinChunks chunks are counted completely by the first launch: when the last calls held
    // no larger one -- config 4's sparse masks -- the two later launches would only be launched to leave again)
    return m >= stage_hint_threshold(p, work) && max_tn > (float)((kStageMinChunks - 1) * 4 * kBfPixPerWave);
}

Bf16Consts bf16_consts(float thresh)
{
    const double T = (double)thresh, s2 = 1.0 - T * T, kappa = T / std::sqrt(s2);
    const double u = 0x1p-24;
    Bf16Consts fc;
    // (round 3: the unit normals come from v_rsq_f32, components within 4u instead of the 3u of sqrt + divide: 32 -> 33,
    // 34 -> 35 in the first level, 6 -> 8 in the second)
    fc.beta = (float)(1.25 * (33.0 * (1.0 + kappa) + 8.0 / s2) * u / T);
    fc.eps_c = (float)(1.25 * (1.0 + kappa) * 35.0 * u);
    fc.eps0 = (float)(1.5e-6 * (1.0 + kappa));
    fc.kappa = (float)kappa;
    // second level: d exact-path's own, nh/B computed in f32 (<= 4u / 5u relative), one fma each => 8u(1+kappa)|d|
    fc.beta2 = (float)(1.25 * (8.0 * (1.0 + kappa) + 8.0 / s2) * u / T);
#ifdef PVV_TUNING
    // timing experiments only (tools/build_variant.sh -DPVV_TUNING); != 1 voids the exactness guarantee
    static const char *dbg = getenv("PVV_DEBUG_BAND_SCALE");
    if (dbg && *dbg) {
        const float k = (float)atof(dbg);
        fc.beta *= k; fc.eps_c *= k; fc.eps0 *= k;
    }
#endif
    return fc;
}

// tuning knobs exist only in experimental builds (tools/build_variant.sh -DPVV_TUNING): the shipped library reads no
// environment variable
int tuning_int(const char *name, int dflt)
{
#ifdef PVV_TUNING
    // read ONCE per knob (tools/variant_ab.py loads a copy of the library per setting and restores the environment after
    // its first launches)
    static const char *names[32];
    static int vals[32], n = 0;
    for (int i = 0; i < n; ++i) if (!strcmp(names[i], name)) return vals[i] == INT32_MIN ? dflt : vals[i];
    const char *e = getenv(name);
    if (n < 32) { names[n] = name; vals[n] = e && *e ? atoi(e) : INT32_MIN; ++n; }
    return e && *e ? atoi(e) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

long long *tuning_ptr(const char *name)
{
#ifdef PVV_TUNING
    const char *e = getenv(name);      // re-read on every call: the tuning scripts change it between calls
    return e && *e ? (long long *)strtoull(e, nullptr, 0) : nullptr;
#else
    (void)name;
    return nullptr;
#endif
}

// optional stage-boundary events (pvv_problem.ev_marks, a measurement aid): mark i is recorded on the call's stream
int mark(const pvv_problem *p, int i, hipStream_t st)
{
    if (!p->ev_marks || !p->ev_marks[i]) return PVV_OK;
    if (hipEventRecord((hipEvent_t)p->ev_marks[i], st) != hipSuccess) return fail(PVV_E_ARG, "ev_marks holds an invalid hipEvent_t");
    return PVV_OK;
}

struct StagedLaunch {
    const pvv_problem *p;
    char *ws;
    const Layout *L;
    hipStream_t st;
    Bf16Consts fc;
    long long *dbg;
    int per_cu_first, per_cu_filter, target_first, target_filter;
    int sub_tenth;       // 1: the estimate's bound (StageArgs.sub_tenth)
    int col0, hstride, lead_set;   // CountCols (0, 0, 0: whole rows of p->hn hypotheses)
    const float *mean;
};

// the three launches of a staged count pass for one chunk schedule (FIRST = the residues mod 8 the first launch counts)
template <uint32_t FIRST>
int launch_staged(const StagedLaunch &a)
{
    const pvv_problem *p = a.p;
    const Layout &L = *a.L;
    char *ws = a.ws;
    hipStream_t st = a.st;
    const Bf16Consts fc = a.fc;
    const float2 *coords = (const float2 *)(ws + L.coords), *dirs = (const float2 *)(ws + L.dirs);
    const float2 *hyps = (const float2 *)(ws + L.hyps) + a.col0;
    int *counts = (int *)(ws + L.counts) + a.col0;
    const int *tn = (const int *)(ws + L.tn);
    int *lead = (int *)(ws + L.lead) + (size_t)a.lead_set * lead_set_words(p);
    StageArgs sa;
    sa.lead = nullptr;
    sa.any_staged = lead + (size_t)p->B * p->K * 8;
    sa.miss = (int *)(ws + L.miss) + a.col0;
    sa.sub_tenth = a.sub_tenth;
    sa.hstride = a.hstride;
    sa.mean = (a.sub_tenth && tuning_int("PVV_PROX", 1) != 0) ? (const float2 *)a.mean : nullptr;
#ifdef PVV_STAMPS
    sa.dbg = tuning_ptr("PVV_DBG_PTR_FILTER");                    // phase census of the second launch (tools/census_filter.py)
#endif
    hipLaunchKernelGGL((k_count_bf16<kCountFirst, FIRST>), dim3(a.per_cu_first * num_cus()), dim3(kBlock), 0, st, coords, dirs, hyps, counts, tn,
                       p->B, p->K, p->hn, p->cap, p->inlier_thresh, fc, a.target_first, a.dbg, sa);
    if (int e = check_launch("k_count_bf16<first>")) return e;
    if (int e = mark(p, PVV_MARK_STAGE0, st)) return e;
    LeadArgs la;
    la.tn_arr = tn; la.coords = coords; la.dirs = dirs; la.hyps = hyps; la.counts = counts; la.lead = lead;
    la.K = p->K; la.hn = p->hn; la.cap = p->cap;
    la.hstride = a.hstride > 0 ? a.hstride : p->hn;
    la.kappa = fc.kappa; la.beta = 2.f * fc.beta2; la.eps = 2.f * fc.eps0;
    la.any_staged = sa.any_staged;
    // shares per (image, keypoint): the largest power of two <= 16 that keeps the grid within one generation of blocks
    // (8 per CU)
    la.nsplit = 16;
    while (la.nsplit > 1 && (long long)p->B * p->K * la.nsplit > 8ll * num_cus()) la.nsplit >>= 1;
    la.nsplit = tuning_int("PVV_LEAD_SPLIT", la.nsplit);
    hipLaunchKernelGGL(k_lead<FIRST>, dim3(p->K * la.nsplit, p->B), dim3(kBlock), 0, st, la);
    if (int e = check_launch("k_lead")) return e;
    if (int e = mark(p, PVV_MARK_PRUNE0, st)) return e;
    sa.lead = lead;
    // round 4: the second launch's items own a RUN of an (image, keypoint)'s remaining chunks and keep eliminating inside it
    // (count_filter_runs.hpp); one generation of blocks, the run length adapts the item count to it.  (Round 3's
    // one-chunk items stay reachable in tuning builds: PVV_FILTER_OLD=1.)
    // With runs of ONE chunk the new items only add their elimination step to round 3's (+2 % per call at config 3, B = 16 / 24):
    // when the images the last call of this shape reported (the stage hint: AUTO only) predict that, round 3's kernel runs.
    const long long rest = p->count_kernel == PVV_COUNT_AUTO ? stage_hint_rest_chunks(p, st, stage_rest_of(FIRST)) : -1;
    // (Only up to 512 hypotheses: with several hypothesis groups a run-owning item walks the survivors of all of them in ONE pass,
    // round 3's items one group each -- config 5 at B = 2, runs of one chunk: round 3's kernel +2.3 % per call.)
    const bool runs = rest < 0 || rest * p->K >= 2ll * a.target_filter || p->hn > 512;
    if (tuning_int("PVV_FILTER_OLD", runs ? 0 : 1) == 0) {
        hipLaunchKernelGGL(k_count_filter_runs<FIRST>, dim3(tuning_int("PVV_GRID_PER_CU_FILTER", 5) * num_cus()), dim3(kBlock), sizeof(int) * (size_t)p->B, st, coords, dirs,
                           hyps, counts, tn, p->B, p->K, p->hn, p->cap, p->inlier_thresh, fc, a.target_filter, tuning_int("PVV_RUN_R", 0), sa);
        return check_launch("k_count_filter_runs");
    }
    hipLaunchKernelGGL((k_count_bf16<kCountFilter, FIRST>), dim3(a.per_cu_filter * num_cus()), dim3(kBlock), 0, st, coords, dirs, hyps, counts, tn,
                       p->B, p->K, p->hn, p->cap, p->inlier_thresh, fc, a.target_filter, a.dbg, sa);
    return check_launch("k_count_bf16<filter>");
}

// The hypotheses a count pass covers when they are a column range of longer rows: the fused un_pnp call keeps rows of
// hn + hn_est hypotheses (one compaction, one hypothesis launch) and counts [0, hn) as ransac_voting_layer_v3's and [hn, hn + hn_est)
// as the estimate's, each full or in stages as it would be alone.  p->hn is then the number of columns counted, hstride the row.
struct CountCols {
    int col0 = 0;        // first column
    int hstride = 0;     // row length (0: p->hn, whole rows)
    int lead_set = 0;    // which of the workspace's two sets of leader words the pass uses
    const float *mean = nullptr;   // the estimate in stages: [B,K,2] keypoints (device) its second launch orders the chunks by, or nullptr
};

int launch_count_bf16(const pvv_problem *p, const Layout &L, char *ws, hipStream_t st, int staged /*0: full pass, 1: v3 in stages, 2: the estimate in stages*/,
                      const CountCols &cc = CountCols())
{
    // persistent blocks per CU (5 are resident): every block builds the item table once, so few blocks are better when
    // items are short (hn <= 512: one hypothesis group per item), more when they are long and uneven; several
    // generations of blocks also stagger the latency-bound prologues against the VALU-bound loops (exactly 5 per CU
    // runs them in lockstep: +11 %).  Measured on MI355X: 15 vs 24 per CU = -2 % at cfg3 (B = 64) and -20 % at B = 1;
    // 48 vs 24 = -1 % at cfg5 (2048 hypotheses).  A single atomic work queue instead of the static round-robin was
    // 12-150 % slower: device-scope atomics on one address serialise at ~20 ns each; an effective grid that gives every
    // block the same NUMBER of items was 6 % slower too -- its stride (32 images' worth of items) lines the near-empty
    // last chunks of all images up in the same blocks.
    const int per_cu_t = tuning_int("PVV_GRID_PER_CU", 0);
    // Item size: the kernel splits (chunk, keypoint) pairs into runs / groups of hypothesis tiles until there are at least
    // 5 items per CU (round-2 sweep, tools/sweep_count.py: against 2 per CU -12 % at B = 8, -8 % at B = 4, -20 % for the
    // 4096-hypothesis estimate at B <= 8, +-0 from B = 24 on).
    const int items_per_cu = tuning_int("PVV_ITEMS_PER_CU", 5);
    // Grid: up to ~2000 items one generation of blocks (the 5 resident ones per CU, a few of them take two items) -- a
    // block without an item still costs ~1 us (it has to read tn[] to find that out), and three generations of them kept
    // the kernel alive 2 us after the last working block at B = 1; 15 per CU lose 6 % at B = 16 and win 8 % at B = 24.
    // The host does not know tn: up to B = 8 it launches the one generation, beyond that 15 per CU (48 for >= 2048
    // hypotheses: long, uneven items) and the KERNEL falls back to one generation when it finds few items (count_bf16.hpp).
    // (Round 4: when the stage hint knows the tn of the last call of this shape, "few items" is decided on them instead of on B --
    // eight dense 256x256 crops are 3240 (chunk, keypoint) pairs, as many as 34 LINEMOD frames: 15 per CU -3.4 % per call there.)
    bool few = p->B <= 8;
    if (p->count_kernel == PVV_COUNT_AUTO) {
        double sum_tn = -1.0;
        if (stage_hint_mean(p, st, nullptr, &sum_tn) >= 0.f && sum_tn >= 0.0)
            few = sum_tn / (4 * kBfPixPerWave) * p->K * ((p->hn + 511) / 512) <= 2000.0;
    }
    // (not few at B <= 8 -- only the hint can say so --: 15 per CU whatever hn; config 5 at B = 2, staged: 48 per CU +2 %)
    const int per_cu = per_cu_t > 0 ? per_cu_t : (few ? 5 : ((p->hn < 2048 || p->B <= 8) ? 15 : 48));
    const float2 *coords = (const float2 *)(ws + L.coords), *dirs = (const float2 *)(ws + L.dirs);
    const float2 *hyps = (const float2 *)(ws + L.hyps) + cc.col0;
    int *counts = (int *)(ws + L.counts) + cc.col0;
    const int *tn = (const int *)(ws + L.tn);
    const Bf16Consts fc = bf16_consts(p->inlier_thresh);
    const int target = tuning_int("PVV_TARGET_ITEMS", items_per_cu * num_cus());
    const int target_first = tuning_int("PVV_TARGET_ITEMS_FIRST", target), target_filter = tuning_int("PVV_TARGET_ITEMS_FILTER", target);
    // the second launch's items are short (a few matrix-core tiles behind the same prologue): one generation of blocks
    // walks them as fast as three (-0.6 % per call at B = 64, -1.8 % at B = 32) and an EMPTY second launch -- a batch of
    // small masks -- costs a third (-1.5 % on config 4 at B = 32)
    const int per_cu_first = tuning_int("PVV_GRID_PER_CU_FIRST", per_cu);
    const int per_cu_filter = tuning_int("PVV_GRID_PER_CU_FILTER", p->hn < 2048 ? 5 : per_cu);
    long long *dbg = tuning_ptr("PVV_DBG_PTR");
    if (!staged) {
        StageArgs full{};
        full.hstride = cc.hstride;
        hipLaunchKernelGGL(k_count_bf16<kCountFull>, dim3(per_cu * num_cus()), dim3(kBlock), 0, st, coords, dirs, hyps, counts, tn,
                           p->B, p->K, p->hn, p->cap, p->inlier_thresh, fc, target, dbg, full);
        return check_launch("k_count_bf16");
    }
    // ransac_voting_layer_v3, staged: count a spread part of the chunks for every hypothesis, bound the winner's count from
    // below through two leaders (k_lead), then the rest only for the hypotheses that can still reach that bound.  The first
    // stage is a QUARTER of the chunks, or an EIGHTH when the problem is so large that the second launch's runs are long
    // (measured, one-process A/B of whole calls: -5 % at config 3 / B = 96, -4 % at B = 128, -9 % on config 5 / B = 16, -5 % at its
    // B = 8; +2.5 % at config 3 / B = 64, +4.8 % on config 5 / B = 4 with unequal keypoints).  "Long" is a property of the RUNS, not of
    // the hypothesis count: x = K * sum(tn) / 512 chunks per block slot of the second launch -- 5.4 at config 3 / B = 64, 8.1 at 96,
    // 6.2 / 3.1 on config 5 at B = 8 / 4 -- and the eighth pays from x = 5.8 on.  (Until round 5 the bound was on the work
    // K * hn * sum(tn), which told the same for 512 hypotheses and sent config 5's 2048 to the eighth from B = 2 on: the one case
    // of tools/auto_regret.py above 1.03.)
    StagedLaunch sl;
    sl.p = p; sl.ws = ws; sl.L = &L; sl.st = st; sl.fc = fc; sl.dbg = dbg;
    sl.per_cu_first = per_cu_first; sl.per_cu_filter = per_cu_filter; sl.target_first = target_first; sl.target_filter = target_filter;
    sl.sub_tenth = staged == 2 ? 1 : 0;
    sl.col0 = cc.col0; sl.hstride = cc.hstride; sl.lead_set = cc.lead_set; sl.mean = cc.mean;
    // (the ESTIMATE keeps the quarter at every size: its second launch walks the chunks nearest to the keypoint first, and an
    // eighth is 1.5 % slower at B = 64, 2-3 % at B = 6-8, +-1 % on config 5 -- profiles/r05_experiments.txt (15))
    const double sum_tn = (p->count_kernel == PVV_COUNT_AUTO ? stage_work(p, st) : stage_proxy_work(p)) * kStageProxyFg / ((double)p->K * p->hn);
    const double chunks_per_slot = (double)p->K * sum_tn / (4.0 * kBfPixPerWave) / (5.0 * num_cus());
    const bool eighth = tuning_int("PVV_STAGE_EIGHTH", (staged != 2 && chunks_per_slot >= 5.8) ? 1 : 0) != 0;
    return eighth ? launch_staged<kStageFirstEighth>(sl) : launch_staged<kStageFirst>(sl);
}

CountArgs planar_count_args(const pvv_problem *p, const Layout &L, char *ws)
{
    CountArgs a;
    a.coords = (const float2 *)(ws + L.coords);
    a.dirs = (const float2 *)(ws + L.dirs);
    a.hyps = (const float2 *)(ws + L.hyps);
    a.counts = (int *)(ws + L.counts);
    a.tn_arr = (const int *)(ws + L.tn);
    a.c_b = p->cap;
    a.d_b = (long long)p->K * p->cap; a.d_v = p->cap; a.d_p = 1;
    a.h_b = (long long)p->K * p->hn; a.h_v = p->hn; a.h_h = 1;
    a.tn_fixed = 0;
    a.B = p->B; a.K = p->K; a.hn = p->hn;
    a.thresh = p->inlier_thresh;
    return a;
}

// The ESTIMATE in stages: estimate_voting_distribution_with_mean weighs every hypothesis whose ratio lies within 0.1 of the best
// (P:262-264), so its count pass may drop what provably falls below that window (stage_bound, count_bf16.hpp) when nobody asked
// for the counts themselves.  Covariances and PnP weights are bit-identical to the full pass (tests/test_gpu_staged.py); a third of
// the hypothesis-tile work goes away.  Round 4 measured it exact and NOT faster (1.437 vs 1.402 ms at B = 64, 4096 hypotheses) and
// kept AUTO away from it -- with a second launch that, unnoticed, ran four blocks per CU instead of five (count_filter_runs.hpp:
// its LDS was one allocation granule too large).  With five (round 5, tools/estimate_ab.py, 480x640, K = 9, 4096 hypotheses,
// staged vs full): 1.2385 vs 1.3992 ms at B = 64 (+13 %; +12 % with 9.5 % outlier pixels), 0.657 vs 0.709 at B = 32, 0.369 vs 0.377
// at B = 16, 0.205 vs 0.218 at B = 8, 0.075 vs 0.053 at B = 1; 540x720, K = 17, 2048 hypotheses at B = 16: 1.438 vs 1.532; B = 24 / 48:
// +10 % / +13 %; 2048 hypotheses at B = 64 +8.5 %, 1024 at B = 64 +2 %, at B = 16 -10 %.  On fields with STRUCTURED errors (a third
// of the object voting for a wrong point plus keypoints of very different quality: synth wrong_region / kp_outlier) the gain shrinks
// to +3 % at B = 64 and turns into -6 % at B = 16, -2 % at B = 8 (profiles/r05_experiments.txt (8)).  So AUTO staged an estimate
// only from kEstStageMinWork = 2e11 evaluations-equivalent on (B*K*hn*H*W: 18 LINEMOD frames at 4096 hypotheses, 540x720 / K = 17 /
// 2048 at B = 16) and >= 1024 hypotheses; PVV_COUNT_STAGED_ESTIMATE forces it at every size (the tests' cross-check of the
// bound), PVV_COUNT_FULL forbids it, and PVV_COUNT_STAGED means v3 alone.
// With the chunks of a run walked NEAREST to the keypoint first (count_filter_runs.hpp: a hypothesis' misses sit around the
// keypoint) the staged pass is another 5-7 % faster and wins from 6 LINEMOD frames on: staged / full on clean fields 0.93 at
// B = 6, 0.94 at 8, 0.91 at 12, 0.85 at 24, 0.83-0.85 at 48-64 (1.10 at B = 4); 540x720 / K = 17 / 2048 hypotheses 0.80 at B = 4, 0.76 at
// 8; 9.5 % outlier pixels 0.95-0.96 at 8-16; 30 % outliers 1.00 / 0.99 / 0.88 at 8 / 16 / 64; structured errors 1.05 / 1.07 / 1.00 /
// 0.96 at 8 / 16 / 24 / 64 (profiles/r05_experiments.txt (15)).  What separates those fields is what the v3 call that precedes every
// estimate (resnet18.py:71-72) has just reported: the winners' ratios and the images' tn -- the stage hint of the shape (H, W, K),
// whatever its hn.  With a hint the bound is on the REAL work K * hn * sum(tn) / 0.02: 6e10 for ratios >= 0.95, 9e10 from 0.85 on,
// kEstStageMinWork below; without one the proxy against kEstStageMinWork, as before.
constexpr double kEstStageMinWork = 2e11;
bool est_stage_auto(const pvv_problem *p, hipStream_t st)
{
    if (p->hn < 1024) return false;
    double sum_tn = -1.0;
    float max_tn = -1.f;
    const float rho = stage_hint_mean(p, st, &max_tn, &sum_tn, /*any_hn=*/true);
    if (rho < 0.f || sum_tn < 0.0) return stage_proxy_work(p) >= kEstStageMinWork;
    // (as in stage_hint_allows: when the last calls held no image of kStageMinChunks chunks, the first launch counts everything and
    // k_lead + the second launch would only be launched to leave again)
    if (!(max_tn > (float)((kStageMinChunks - 1) * 4 * kBfPixPerWave))) return false;
    const double work = (double)p->K * p->hn * sum_tn / kStageProxyFg;
    return work >= (rho >= 0.95f ? 6e10 : (rho >= 0.85f ? 9e10 : kEstStageMinWork));
}
//
// kind: 0 = the pass must deliver every count (the estimate when its counts are an output, the fused un_pnp pass);
//       1 = ransac_voting_layer_v3 proper, which keeps the arg-max and may count in stages;
//       2 = the estimate without a counts output, which may count in stages against its own bound
// -> 0: full pass, 1: v3 in stages, 2: the estimate in stages.  Decided ONCE per call, before the front kernels are launched:
// k_compact_hyp zeroes the miss counters and leader words only for a call that will stage.
int decide_staged(const pvv_problem *p, const Layout &L, hipStream_t st, int kind)
{
    if (!may_stage(p) || L.lead == 0) return 0;
    if (kind == 1 && stage_hint_allows(p, st)) return 1;
    if (kind == 2 && (p->count_kernel == PVV_COUNT_STAGED_ESTIMATE || (p->count_kernel == PVV_COUNT_AUTO && est_stage_auto(p, st)))) return 2;
    return 0;
}

int launch_count_any(const pvv_problem *p, const Layout &L, char *ws, hipStream_t st, int staged, const CountCols &cc = CountCols())
{
    if (p->ev_count_begin && hipEventRecord((hipEvent_t)p->ev_count_begin, st) != hipSuccess)
        return fail(PVV_E_ARG, "ev_count_begin is not a valid hipEvent_t");
    if (!use_bf16_count(p) && (cc.col0 || cc.hstride)) return fail(PVV_E_ARG, "internal: a column range needs the bf16 count kernel");
    const int e = use_bf16_count(p) ? launch_count_bf16(p, L, ws, st, staged, cc) : launch_count(planar_count_args(p, L, ws), st);
    if (e) return e;
    if (p->ev_count_end && hipEventRecord((hipEvent_t)p->ev_count_end, st) != hipSuccess)
        return fail(PVV_E_ARG, "ev_count_end is not a valid hipEvent_t");
    return PVV_OK;
}

template <int ES, int MODE>
void launch_scan(const MaskArgs &m, const Layout &L, char *ws, int B, hipStream_t st)
{
    const long long total = (long long)L.T * B;
    uint32_t *tiles = (uint32_t *)(ws + L.tiles);
    unsigned short *lists = (unsigned short *)(ws + L.tile_list);
    float *draws = (float *)(ws + L.tile_draw);
    // the read-ahead needs a contiguous mask; a strided mask or the fused argmax keep one short-lived block per tile
    // (their loads are not issued ahead, and a persistent block would walk its tiles one load latency at a time)
    if (!(m.contig && !m.seg)) {
        hipLaunchKernelGGL((k_tile_scan<ES, false, MODE>), dim3((unsigned)total), dim3(kBlock), 0, st, m, tiles, lists, draws, (int)total);
        return;
    }
    // ONE tile per block.  Round 2 made this kernel persistent over the resident blocks (8 per CU, every block 4-5 tiles with
    // the next tile's loads ahead: 4.0 -> 4.4 TB/s on the cold int64 mask) when it issued ~400 instructions per wave and
    // tile; at ~300 (round 3) the dispatcher's overlap of short blocks wins again -- one-process A/B over 4 ... 80 blocks
    // per CU in the grid rule, monotone: -2.1 % per call at B = 64, -1.1 % at B = 16, -2.4 % at B = 128, +-0 at B = 1 and
    // on config 5 against the persistent grid.  (The kernel still walks gridDim-strided tiles with its read-ahead if it is
    // ever launched with fewer blocks: PVV_SCAN_PER_CU in tuning builds.)
    const int per_cu_t = tuning_int("PVV_SCAN_PER_CU", 0);
    long long per_block = 1;
    if (per_cu_t > 0) per_block = (total + (long long)per_cu_t * num_cus() - 1) / ((long long)per_cu_t * num_cus());
    const int grid = (int)((total + per_block - 1) / per_block);
    hipLaunchKernelGGL((k_tile_scan<ES, true, MODE>), dim3(grid), dim3(kBlock), 0, st, m, tiles, lists, draws, (int)total);
}

struct Front {
    MaskArgs m;
    VertexArgs v;
    HypArgs h;
    bool can_subsample;
    bool seg2;                // two-class seg in two contiguous planes: k_tile_scan_seg2 (16-byte loads)
    long long *mask_deferred; // != nullptr: the scan does not write the int64 mask; k_mask_from_lists does, on the side stream
};

// stream_first / stream_rest: RNG stream of the hypotheses [0, hn_first) / [hn_first, hn) -- 1 for
// ransac_voting_layer_v3, 3 for the estimate, so that the two layers never share draws under one seed
Front make_front(const pvv_problem *p, int mode, const void *d_mask, const float *d_vertex, const int32_t *d_idxs,
                 const float *d_selection, char *ws, const Layout &L, int32_t *d_tn, const float *d_seg,
                 int64_t *d_mask_out, const int32_t *d_idxs2, int hn_first, uint32_t stream_first, uint32_t stream_rest)
{
    Front f;
    MaskArgs &m = f.m;
    m.mask = d_mask;
    m.seg = d_seg;
    m.mask_out = (long long *)d_mask_out;
    m.gb = p->seg_stride[0]; m.gc = p->seg_stride[1]; m.gh = p->seg_stride[2]; m.gw = p->seg_stride[3];
    m.C = p->seg_classes;
    m.selection = d_selection;
    m.sb = p->mask_stride[0]; m.sh = p->mask_stride[1]; m.sw = p->mask_stride[2];
    m.es = p->mask_elem_size;
    m.contig = (m.sw == 1 && m.sh == p->W) ? 1 : 0;
    m.mode = mode;
    m.W = p->W; m.HW = p->H * p->W; m.T = L.T;
    m.min_num = p->min_num; m.max_num = p->max_num; m.cap = p->cap;
    m.seed = p->seed;
    m.b0 = p->first_image;
    m.tn_user = d_tn;
    m.status = p->d_status;
    // Subsampling inside k_compact_hyp (no k_tile_subsample launch) only for images of <= kFuseSubTiles tiles, only when
    // subsampling is unlikely -- max_num at least 1/16 of the image (30000 of 307200) -- and only with the device RNG:
    // injected index pairs address rows of the SUBSAMPLED list, which the hypothesis blocks can read off the tile lists
    // only after k_tile_subsample has rewritten them.
    m.fuse_sub = (fuse_sub_shape(p, L.T) && !d_idxs && !d_idxs2) ? 1 : 0;
    // largest possible foreground_num: the sum of byte values (P:126), of class indices (fused argmax) or of ones (P:208)
    const long long max_weight = mode == 1 ? 1 : (d_seg ? (p->seg_classes > 1 ? p->seg_classes - 1 : 1) : 255);
    f.can_subsample = (long long)p->max_num < max_weight * (long long)p->H * p->W;
    m.want_draws = (f.can_subsample && !m.fuse_sub) ? 1 : 0;     // fused subsampling evaluates its draws on demand (compaction.hpp)
    // decode_keypoint's real layout (resnet18.py:69,93): a two-class seg in two contiguous float32 planes
    f.seg2 = d_seg && p->seg_classes == 2 && m.gw == 1 && m.gh == p->W && (((long long)p->H * p->W) & 3) == 0 && (m.gb & 3) == 0 &&
             (m.gc & 3) == 0 && ((uintptr_t)d_seg & 15) == 0;
    // ... whose int64 mask (8 B per pixel, as much as the scan reads) can be written from the tile lists beside the rest of
    // the call instead of by the scan, when the lists stay complete (k_tile_subsample would rewrite them) and the batch is
    // large enough for the two cross-stream events to pay (measured: B = 64 480x640, scan 70.5 -> ~27 us)
    f.mask_deferred = nullptr;
    if (f.seg2 && d_mask_out && !(f.can_subsample && !m.fuse_sub) && (long long)p->B * p->H * p->W >= (1ll << 21) &&
        ((uintptr_t)d_mask_out & 15) == 0)
        f.mask_deferred = (long long *)d_mask_out;
    VertexArgs &v = f.v;
    v.vertex = d_vertex;
    v.sb = p->vertex_stride[0]; v.sh = p->vertex_stride[1]; v.sw = p->vertex_stride[2];
    v.sk = p->vertex_stride[3]; v.sc = p->vertex_stride[4];
    v.K = p->K;
    v.vec2 = (v.sc == 1 && !(v.sb & 1) && !(v.sh & 1) && !(v.sw & 1) && !(v.sk & 1) &&
              ((uintptr_t)d_vertex % 8 == 0)) ? 1 : 0;
    HypArgs &h = f.h;
    h.idxs = d_idxs; h.idxs2 = d_idxs2;
    h.hn = p->hn; h.hn_first = hn_first < 0 ? p->hn : hn_first;
    h.stream = stream_first; h.stream2 = stream_rest;
    h.hyps = (float2 *)(ws + L.hyps);
    h.counts = (int *)(ws + L.counts);
    h.draws_out = p->d_draws_out;
    h.blocks = (int)(((long long)p->K * p->hn + kBlock - 1) / kBlock);
    h.surv = (int *)(ws + L.surv);
    h.lead = L.lead ? (int *)(ws + L.lead) : nullptr;
    h.miss = L.miss ? (int *)(ws + L.miss) : nullptr;
    return f;
}

int run_scan(const pvv_problem *p, const Front &f, char *ws, const Layout &L, hipStream_t st, bool write_mask)
{
    if (f.seg2) {
        MaskArgs m = f.m;
        if (!write_mask) m.mask_out = nullptr;
        const unsigned total = (unsigned)((long long)L.T * p->B);
        uint32_t *tiles = (uint32_t *)(ws + L.tiles);
        unsigned short *lists = (unsigned short *)(ws + L.tile_list);
        float *draws = (float *)(ws + L.tile_draw);
        if (write_mask) hipLaunchKernelGGL(k_tile_scan_seg2<true>, dim3(total), dim3(kBlock), 0, st, m, tiles, lists, draws);
        else hipLaunchKernelGGL(k_tile_scan_seg2<false>, dim3(total), dim3(kBlock), 0, st, m, tiles, lists, draws);
        return check_launch("k_tile_scan_seg2");
    }
    // (the mask's interpretation -- low byte for v3, == 1 for the estimate -- is a template parameter of the scan)
    auto go = [&](auto es) {
        constexpr int ES = decltype(es)::value;
        if (f.m.mode == 0) launch_scan<ES, 0>(f.m, L, ws, p->B, st); else launch_scan<ES, 1>(f.m, L, ws, p->B, st);
    };
    switch (f.m.es) {
    case 1: go(std::integral_constant<int, 1>{}); break;
    case 2: go(std::integral_constant<int, 2>{}); break;
    case 4: go(std::integral_constant<int, 4>{}); break;
    default: go(std::integral_constant<int, 8>{}); break;
    }
    return check_launch("k_tile_scan");
}

// ---------------------------------------------------------------------------------------------
// The side stream: one per device, created on first use, for work of a call that nothing in the call waits for -- today
// the deferred int64 mask of decode_keypoint (k_mask_from_lists).  Fork: an event recorded on the caller's stream behind
// the scan, the side stream waits for it; join: an event recorded behind the side kernel, the caller's stream waits for
// it at the end of run_front -- by then the kernel has long finished, it ran beside the compaction and the count pass.
// hipStreamWaitEvent takes the event's state at the time of the call, so the events are reused (a ring of 8 pairs).
// Not used while the caller's stream is capturing (no stream or event is created under capture) or when the stream
// belongs to a device that is not current.
// ---------------------------------------------------------------------------------------------
// Both streams are on ONE device: the events between them need no system-scope fence (host / other devices do not look at
// what they order), only the agent-scope release every kernel ends with.
#ifdef PVV_SIDE_EVENT_SYSTEM_FENCE                                    // (tuning builds: A/B of the event flags)
constexpr unsigned kSideEventFlags = hipEventDisableTiming;
#else
constexpr unsigned kSideEventFlags = hipEventDisableTiming | hipEventDisableSystemFence;
#endif
struct SideStream {
    hipStream_t st = nullptr;
    hipEvent_t fork[8] = {}, join[8] = {};
    int next = 0;
    bool tried = false, ok = false;
};
SideStream g_side[64];
std::mutex g_side_mu;

// can work of a call on `st` be forked to the side stream?  (creates it on first use)
SideStream *side_get(hipStream_t st)
{
    int cur = -1;
    const int dev = stream_device(st);
    if (dev < 0 || hipGetDevice(&cur) != hipSuccess || cur != dev) return nullptr;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (cs != hipStreamCaptureStatusNone) return nullptr;
    std::lock_guard<std::mutex> lock(g_side_mu);
    SideStream *g = &g_side[dev];
    if (!g->tried) {
        g->tried = true;
        bool ok = hipStreamCreateWithFlags(&g->st, hipStreamNonBlocking) == hipSuccess;
        for (int i = 0; ok && i < 8; ++i)
            ok = hipEventCreateWithFlags(&g->fork[i], kSideEventFlags) == hipSuccess &&
                 hipEventCreateWithFlags(&g->join[i], kSideEventFlags) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        g->ok = ok;
    }
    return g->ok ? g : nullptr;
}

// Fork: `side` waits for everything enqueued on `st` so far.  Call with g_side_mu held (SideFork below): the ring slot, the
// fork record and the side stream's wait are one step, and so are the kernel launched on the side stream and its join record --
// two threads forking at once would otherwise interleave records and launches on the one side stream.
bool side_fork_locked(SideStream *g, hipStream_t st, hipStream_t *side, hipEvent_t *join)
{
    const int i = g->next;
    g->next = (i + 1) & 7;
    if (hipEventRecord(g->fork[i], st) != hipSuccess || hipStreamWaitEvent(g->st, g->fork[i], 0) != hipSuccess) { (void)hipGetLastError(); return false; }
    *side = g->st;
    *join = g->join[i];
    return true;
}

// Joins the side stream back into the caller's stream on EVERY way out of run_front once work has been forked (an
// error return between fork and join used to leave the side kernel unordered with `stream` while the caller already held
// an error code).  The join event of a ring slot may be re-recorded by a later fork before this wait is issued: both
// records are on the one in-order side stream, so waiting for the later one still covers this call's kernel.
struct SideJoin {
    hipStream_t st = nullptr;
    hipEvent_t join = nullptr;
    bool armed = false;
    int finish()                                                   // explicit join on the success path: its failure is an error
    {
        if (!armed) return PVV_OK;
        armed = false;
        return hipStreamWaitEvent(st, join, 0) == hipSuccess ? PVV_OK : fail(PVV_E_ARG, "side stream: hipStreamWaitEvent failed");
    }
    ~SideJoin()
    {
        if (armed && hipStreamWaitEvent(st, join, 0) != hipSuccess) (void)hipGetLastError();   // (the call is failing already)
    }
};

// What run_front launches as "the count pass" when it is not one pass over whole rows: the first of the fused un_pnp call's two
// passes (pvv_decode_keypoint_un_pnp); the caller launches the second one itself.
struct CountPlan {
    const pvv_problem *p;    // the problem of that pass: p->hn columns of rows of cols.hstride
    int staged;              // decide_staged() for it
    CountCols cols;
};

// mask scan + (subsample) + compaction and hypotheses + counting, shared by both layers.
int run_front(const pvv_problem *p, int mode, const void *d_mask, const float *d_vertex,
              const int32_t *d_idxs, const float *d_selection, char *ws, const Layout &L,
              hipStream_t st, int32_t *d_tn, const float *d_seg = nullptr, int64_t *d_mask_out = nullptr,
              const int32_t *d_idxs2 = nullptr, int hn_first = -1, uint32_t stream_first = 1u, uint32_t stream_rest = 3u,
              int stage_kind = 0 /*decide_staged's kind*/, const CountPlan *plan = nullptr,
              const float *d_mean = nullptr /*the estimate: the keypoints its staged pass orders the chunks by*/)
{
    if ((p->flags & PVV_FLAG_DEVICE_RNG) && (d_idxs || d_idxs2 || d_selection))
        return fail(PVV_E_ARG, "PVV_FLAG_DEVICE_RNG promises d_idxs = d_idxs_est = d_selection = NULL");
    Front f = make_front(p, mode, d_mask, d_vertex, d_idxs, d_selection, ws, L, d_tn, d_seg, d_mask_out, d_idxs2,
                         hn_first, stream_first, stream_rest);
    if (f.m.want_draws && L.tile_draw == 0) return fail(PVV_E_WORKSPACE, "workspace holds no draw storage for this call");   // (cannot happen: may_store_draws)
    // whether the count pass runs in stages is decided HERE, once: a call that will not stage neither zeroes nor reads the
    // miss counters and leader words (k_compact_hyp would otherwise clear B*K*hn words for nothing on every AUTO call)
    // (a plan: some pass of the call stages -- the caller decided -- so the words are zeroed for all columns)
    const int staged = plan ? 1 : decide_staged(p, L, st, stage_kind);
    if (!staged) { f.h.miss = nullptr; f.h.lead = nullptr; }
    if (int e = mark(p, PVV_MARK_BEGIN, st)) return e;
    SideJoin sj;
    // no side stream to be had (the caller's stream is being captured, ...): the scan writes the mask itself
    SideStream *sg = (f.mask_deferred && tuning_int("PVV_MASK_DEFER", 1) != 0) ? side_get(st) : nullptr;
    const bool defer_mask = sg != nullptr;
    if (int e = run_scan(p, f, ws, L, st, !defer_mask)) return e;
    if (!f.m.fuse_sub && f.can_subsample) {
        hipLaunchKernelGGL(k_tile_subsample, dim3(L.T, p->B), dim3(kBlock), 0, st, f.m, (uint32_t *)(ws + L.tiles),
                           (unsigned short *)(ws + L.tile_list), (const float *)(ws + L.tile_draw));
        if (int e = check_launch("k_tile_subsample")) return e;
    }
    if (int e = mark(p, PVV_MARK_SCAN, st)) return e;
    hipLaunchKernelGGL(k_compact_hyp, dim3(L.T + f.h.blocks, p->B), dim3(kBlock), sizeof(int) * (size_t)L.T, st, f.m, f.v,
                       f.h, (const uint32_t *)(ws + L.tiles), (const unsigned short *)(ws + L.tile_list),
                       (const float *)(ws + L.tile_draw), (int *)(ws + L.tn), (float2 *)(ws + L.coords),
                       (float2 *)(ws + L.dirs));
    if (int e = check_launch("k_compact_hyp")) return e;
    if (int e = mark(p, PVV_MARK_COMPACT, st)) return e;
    if (defer_mask) {
        // forked BEHIND the compaction (measured: forked behind the scan, the 157 MB of stores cost the latency-bound
        // compaction +14 us at B = 64 and the count pass +9; the count pass alone is 130 us of mostly VALU work)
        std::lock_guard<std::mutex> lock(g_side_mu);                // fork, launch and join record: one step (see side_fork_locked)
        hipStream_t side = nullptr;
        hipEvent_t join = nullptr;
        const bool forked = side_fork_locked(sg, st, &side, &join);
        const hipStream_t ms = forked ? side : st;                  // the fork failed: the same kernel, in line
        const long long total = (long long)L.T * p->B;
        const int grid = (int)std::min<long long>(total, (long long)tuning_int("PVV_MASK_GRID_PER_CU", 2) * num_cus());
        hipLaunchKernelGGL(k_mask_from_lists, dim3(grid), dim3(kBlock), 0, ms, (const uint32_t *)(ws + L.tiles),
                           (const unsigned short *)(ws + L.tile_list), f.mask_deferred, L.T, p->H * p->W, (int)total);
        const int le = check_launch("k_mask_from_lists");
        if (forked) {
            // whatever happened to the launch, the side stream now holds this call's fork: join it back on every exit
            if (hipEventRecord(join, side) == hipSuccess) { sj.st = st; sj.join = join; sj.armed = true; }
            else {
                (void)hipGetLastError();
                (void)hipStreamSynchronize(side);                   // no event to wait for: drain the side stream instead
                if (!le) return fail(PVV_E_ARG, "side stream: hipEventRecord failed");
            }
        }
        if (le) return le;
    }
    CountCols whole;
    whole.mean = d_mean;
    if (int e = plan ? launch_count_any(plan->p, L, ws, st, plan->staged, plan->cols) : launch_count_any(p, L, ws, st, staged, whole)) return e;
    if (int e = sj.finish()) return e;
    return mark(p, PVV_MARK_COUNT, st);
}

int check_ptrs(const pvv_problem *p, const void *mask, const void *vertex, void *ws, size_t ws_bytes,
               Layout *L)
{
    if (int e = validate(p)) return e;
    if (!mask || !vertex || !ws) return fail(PVV_E_ARG, "mask, vertex and workspace must be non-NULL");
    if ((uintptr_t)ws % 256 != 0) return fail(PVV_E_ARG, "workspace must be 256-byte aligned");
    *L = make_layout(p);
    if (ws_bytes < L->total) return fail(PVV_E_WORKSPACE, "workspace too small");
    return PVV_OK;
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
PVV_EXPORT int pvv_abi_version(void) { return PVV_ABI_VERSION; }
PVV_EXPORT const char *pvv_last_error(void) { return g_err; }

// ABI v8: give back what the library created behind the caller's back (see the header).  Everything is released even when
// a call fails; the first error is reported.
PVV_EXPORT int pvv_shutdown(void)
{
    int rc = PVV_OK;
    auto note = [&](hipError_t e, const char *what) {
        if (e != hipSuccess) {
            (void)hipGetLastError();
            if (rc == PVV_OK) { snprintf(g_err, sizeof(g_err), "pvv_shutdown: %s: %s", what, hipGetErrorString(e)); rc = (int)e; }
        }
    };
    int cur = -1;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    if (!have_cur) (void)hipGetLastError();
    {
        std::lock_guard<std::mutex> lock(g_side_mu);
        for (int dev = 0; dev < 64; ++dev) {
            SideStream *g = &g_side[dev];
            if (g->tried) {
                if (have_cur && hipSetDevice(dev) != hipSuccess) (void)hipGetLastError();
                if (g->st) { note(hipStreamSynchronize(g->st), "hipStreamSynchronize(side stream)"); note(hipStreamDestroy(g->st), "hipStreamDestroy"); }
                for (int i = 0; i < 8; ++i) {
                    if (g->fork[i]) note(hipEventDestroy(g->fork[i]), "hipEventDestroy");
                    if (g->join[i]) note(hipEventDestroy(g->join[i]), "hipEventDestroy");
                }
            }
            *g = SideStream();
        }
    }
    {
        std::lock_guard<std::mutex> lock(g_hint_mu);
        for (int dev = 0; dev < 64; ++dev) {
            StageHint *g = &g_hint[dev];
            if (g->ratio) note(hipHostFree(g->ratio), "hipHostFree(stage hint)");
            *g = StageHint();
        }
    }
    if (have_cur && hipSetDevice(cur) != hipSuccess) (void)hipGetLastError();
    return rc;
}

PVV_EXPORT int pvv_stage_hint_query(float *mean_ratio, float *threshold, const pvv_problem *p, void *stream)
{
    float m = stage_hint_mean(p, (hipStream_t)stream);
    if (threshold) *threshold = -1.f;
    if (p) {
        const double work = stage_work(p, (hipStream_t)stream, &m);
        // (a problem whose evaluations stay below the bound is never staged by AUTO: reported as a threshold no ratio reaches)
        if (threshold) *threshold = work >= kStageMinWork ? stage_hint_threshold(p, work) : 2.f;
    }
    if (mean_ratio) *mean_ratio = m;
    return m >= 0.f ? 1 : 0;
}

PVV_EXPORT int32_t pvv_default_cap(int32_t H, int32_t W, int32_t max_num)
{
    long long hw = (long long)H * W;
    if (max_num < 0) max_num = 0;
    long long cap = (long long)max_num + (long long)(8.0 * std::sqrt((double)max_num)) + 64;
    return (int32_t)(cap < hw ? cap : hw);
}

PVV_EXPORT size_t pvv_workspace_bytes(const pvv_problem *p)
{
    if (validate(p)) return 0;
    return make_layout(p).total;
}

// hstride: row length of hyps / counts in the workspace (= p->hn unless the row also holds the hypotheses of a fused
// estimate, pvv_decode_keypoint_un_pnp)
static int finish_v3(const pvv_problem *p, const Layout &L, char *ws, float *d_out, int32_t *d_win_counts,
                     hipStream_t st, int hstride = 0)
{
    if (hstride <= 0) hstride = p->hn;
    // blocks per (keypoint, image): the largest power of two that keeps the grid within ~6 blocks per CU (measured on
    // MI355X: 2 at B*K = 576, 8 for a single image)
    int nsplit = kRefitSplitMax;
    while (nsplit > 1 && (long long)p->B * p->K * nsplit > 6ll * num_cus()) nsplit >>= 1;
    nsplit = tuning_int("PVV_REFIT_SPLIT", nsplit);
    // the re-vote's prefilter is the count kernel's second-level test: used where that kernel is valid and the grid is
    // large enough to be VALU-bound (measured: -1.3 % per call at B = 64, +-0 at B = 8, +1.2 % at B = 1)
    const Bf16Consts fc = bf16_consts(p->inlier_thresh);
    const bool band = use_bf16_count(p) && (long long)p->B * p->K >= 128;
    auto launch_refit = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(p->K * nsplit, p->B), dim3(kBlock), 0, st,
                           (const int *)(ws + L.tn), (const float2 *)(ws + L.coords),
                           (const float2 *)(ws + L.dirs), (const float2 *)(ws + L.hyps),
                           (const int *)(ws + L.counts), (double *)(ws + L.sums), d_win_counts, p->K, p->hn, hstride,
                           p->cap, p->inlier_thresh, nsplit, (float *)(ws + L.ratio), fc);
    };
    if (band) launch_refit(k_select_refit<true>); else launch_refit(k_select_refit<false>);
    if (int e = check_launch("k_select_refit")) return e;
    if (int e = mark(p, PVV_MARK_SELECT, st)) return e;
    // the stage hint is only ever consulted by AUTO for problems that may stage: other calls do not write it (a store to
    // host-visible memory at the end of a 3 us kernel: the reference's own eval batch is B = 1)
    float *hint = (p->count_kernel == PVV_COUNT_AUTO && may_stage(p)) ? stage_hint_claim(p, st) : nullptr;
    hipLaunchKernelGGL(k_finalize_v3, dim3(p->B), dim3(64), 0, st, (const int *)(ws + L.tn),
                       (const double *)(ws + L.sums), (float2 *)d_out, p->K, p->singular_policy, nsplit,
                       (const float *)(ws + L.ratio), hint, kMaxBatchLds);
    if (int e = check_launch("k_finalize_v3")) return e;
    return mark(p, PVV_MARK_END, st);
}

PVV_EXPORT int pvv_ransac_voting_v3(const pvv_problem *p, const void *d_mask, const float *d_vertex,
                                    const int32_t *d_idxs, const float *d_selection,
                                    void *d_workspace, size_t workspace_bytes, float *d_out,
                                    int32_t *d_win_counts, int32_t *d_tn, void *stream)
{
    Layout L;
    if (int e = check_ptrs(p, d_mask, d_vertex, d_workspace, workspace_bytes, &L)) return e;
    if (!d_out) return fail(PVV_E_ARG, "d_out is NULL");
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)d_workspace;
    if (int e = run_front(p, 0, d_mask, d_vertex, d_idxs, d_selection, ws, L, st, d_tn, nullptr, nullptr, nullptr, -1, 1u, 3u, 1))
        return e;
    return finish_v3(p, L, ws, d_out, d_win_counts, st);
}

PVV_EXPORT int pvv_decode_keypoint_v3(const pvv_problem *p, const float *d_seg, const float *d_vertex,
                                      const int32_t *d_idxs, const float *d_selection, void *d_workspace,
                                      size_t workspace_bytes, int64_t *d_mask_out, float *d_out,
                                      int32_t *d_win_counts, int32_t *d_tn, void *stream)
{
    Layout L;
    if (p && p->mask_elem_size == 0) return fail(PVV_E_ARG, "set mask_elem_size = 8 (the int64 mask this call emits)");
    if (int e = check_ptrs(p, d_seg, d_vertex, d_workspace, workspace_bytes, &L)) return e;
    if (!d_out) return fail(PVV_E_ARG, "d_out is NULL");
    if (p->seg_classes < 1 || p->seg_classes > 256) return fail(PVV_E_ARG, "seg_classes must be in [1, 256]");
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)d_workspace;
    if (int e = run_front(p, 0, nullptr, d_vertex, d_idxs, d_selection, ws, L, st, d_tn, d_seg, d_mask_out, nullptr, -1, 1u, 3u, 1))
        return e;
    return finish_v3(p, L, ws, d_out, d_win_counts, st);
}

PVV_EXPORT int pvv_estimate_voting_distribution(const pvv_problem *p, const void *d_mask,
                                                const float *d_vertex, const int32_t *d_idxs,
                                                const float *d_selection, const float *d_mean,
                                                void *d_workspace, size_t workspace_bytes,
                                                float *d_cov, float *d_hyp, int32_t *d_counts,
                                                int32_t *d_tn, float *d_weights, void *stream)
{
    Layout L;
    if (int e = check_ptrs(p, d_mask, d_vertex, d_workspace, workspace_bytes, &L)) return e;
    if (!d_mean || !d_cov) return fail(PVV_E_ARG, "d_mean / d_cov is NULL");
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)d_workspace;
    // the counts are an output (d_counts): every one of them is needed; otherwise the pass may drop what cannot carry weight
    if (int e = run_front(p, 1, d_mask, d_vertex, d_idxs, d_selection, ws, L, st, d_tn, nullptr, nullptr, nullptr, -1, 3u, 3u, d_counts ? 0 : 2, nullptr, d_mean)) return e;
    hipLaunchKernelGGL(k_covariance, dim3(p->K, p->B), dim3(kBlock), 0, st,
                       (const int *)(ws + L.tn), (const float2 *)(ws + L.hyps),
                       (const int *)(ws + L.counts), (const float2 *)d_mean, d_cov, (float2 *)d_hyp,
                       d_counts, d_weights, p->K, p->hn, p->hn, 0);
    if (int e = check_launch("k_covariance")) return e;
    return mark(p, PVV_MARK_END, st);
}

PVV_EXPORT int pvv_estimate_counts_in_stages(const pvv_problem *p)
{
    if (int e = validate(p)) return e < 0 ? e : -e;
    if (!may_stage(p)) return 0;
    return (p->count_kernel == PVV_COUNT_STAGED_ESTIMATE || (p->count_kernel == PVV_COUNT_AUTO && est_stage_auto(p, nullptr))) ? 1 : 0;
}

// resnet18.py:65-72 with cfg.test.un_pnp as ONE pass (see the header): the row of every (image, keypoint) holds the hn
// hypotheses of ransac_voting_layer_v3 followed by the hn_est of estimate_voting_distribution_with_mean.
static pvv_problem un_pnp_problem(const pvv_problem *p, int32_t hn_est)
{
    pvv_problem q = *p;
    q.hn = p->hn + hn_est;
    return q;
}

PVV_EXPORT size_t pvv_workspace_bytes_un_pnp(const pvv_problem *p, int32_t hn_est)
{
    if (validate(p) || hn_est <= 0) return 0;
    const pvv_problem q = un_pnp_problem(p, hn_est);
    if (validate(&q)) return 0;
    return make_layout(&q).total;
}

PVV_EXPORT int pvv_decode_keypoint_un_pnp(const pvv_problem *p, int32_t hn_est, const float *d_seg,
                                          const float *d_vertex, const int32_t *d_idxs, const int32_t *d_idxs_est,
                                          const float *d_selection, void *d_workspace, size_t workspace_bytes,
                                          int64_t *d_mask_out, float *d_kpt, float *d_cov, float *d_weights,
                                          int32_t *d_win_counts, int32_t *d_tn, void *stream)
{
    if (int e = validate(p)) return e;
    if (hn_est <= 0) return fail(PVV_E_ARG, "hn_est must be positive");
    if (p->mask_elem_size == 0) return fail(PVV_E_ARG, "set mask_elem_size = 8 (the int64 mask this call emits)");
    // v3 takes `mask != 0` as foreground, the estimate `mask == 1` (P:125 vs P:207): one compaction serves both only
    // when the argmax cannot produce anything but 0 and 1
    if (p->seg_classes != 2) return fail(PVV_E_ARG, "the fused un_pnp pass needs seg_classes == 2 (PVNet's seg_dim)");
    const pvv_problem q = un_pnp_problem(p, hn_est);
    Layout L;
    if (int e = check_ptrs(&q, d_seg, d_vertex, d_workspace, workspace_bytes, &L)) return e;
    if (!d_kpt || !d_cov) return fail(PVV_E_ARG, "d_kpt / d_cov is NULL");
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)d_workspace;
    // Round 5: when the estimate alone would count in stages (decide_staged: large batches, AUTO; or forced) the rows are counted
    // as TWO passes over the one compaction -- columns [0, hn) as ransac_voting_layer_v3 would (full or in stages, the same rule
    // and the same stage hint as the layer's own call), then, behind the refit that delivers the mean, columns [hn, hn + hn_est)
    // against the estimate's bound.  Same results as the one full pass over all columns, bit for bit; what the two separate calls
    // cost more -- a second mask scan and a second compaction -- stays saved.
    pvv_problem pe = *p;
    pe.hn = hn_est;
    const int est_staged = decide_staged(&pe, L, st, 2);
    if (est_staged) {
        const size_t set_bytes = sizeof(int) * lead_set_words(p);
        // the second pass's leader words (k_compact_hyp zeroes the first set): cleared up front, off the kernels' chain
        if (hipMemsetAsync(ws + L.lead + set_bytes, 0, set_bytes, st) != hipSuccess) return fail(PVV_E_ARG, "hipMemsetAsync(leader words) failed");
        // the caller's event pair spans BOTH count passes (and the refit between them): begin is recorded by the first pass only,
        // end by the second only -- a consumer never sees the estimate's pass alone
        pvv_problem p1 = *p;
        p1.ev_count_end = nullptr;
        pe.ev_count_begin = nullptr;
        CountPlan plan;
        plan.p = &p1;
        plan.staged = decide_staged(p, L, st, 1);
        plan.cols.col0 = 0; plan.cols.hstride = q.hn; plan.cols.lead_set = 0;
        if (int e = run_front(&q, 0, nullptr, d_vertex, d_idxs, d_selection, ws, L, st, d_tn, d_seg, d_mask_out, d_idxs_est,
                              p->hn, 1u, 3u, 0, &plan))
            return e;
        if (int e = finish_v3(p, L, ws, d_kpt, d_win_counts, st, q.hn)) return e;
        CountCols ec;
        ec.col0 = p->hn; ec.hstride = q.hn; ec.lead_set = 1; ec.mean = d_kpt;
        if (int e = launch_count_any(&pe, L, ws, st, est_staged, ec)) return e;
    } else {
        if (int e = run_front(&q, 0, nullptr, d_vertex, d_idxs, d_selection, ws, L, st, d_tn, d_seg, d_mask_out, d_idxs_est,
                              p->hn))
            return e;
        if (int e = finish_v3(p, L, ws, d_kpt, d_win_counts, st, q.hn)) return e;      // mean = winner refit over the first hn
    }
    hipLaunchKernelGGL(k_covariance, dim3(p->K, p->B), dim3(kBlock), 0, st, (const int *)(ws + L.tn),
                       (const float2 *)(ws + L.hyps), (const int *)(ws + L.counts), (const float2 *)d_kpt, d_cov,
                       (float2 *)nullptr, (int *)nullptr, d_weights, p->K, (int)hn_est, q.hn, p->hn);
    if (int e = check_launch("k_covariance")) return e;
    return PVV_OK;
}

PVV_EXPORT int pvv_rerun_count_kernel(const pvv_problem *p, void *d_workspace, size_t workspace_bytes,
                                      int zero_counts, void *stream)
{
    // Re-runs the count pass on the state a previous call left in the workspace.  That call may have been the estimate or
    // the fused un_pnp pass, which need EVERY count: so the pass is re-run in stages only when the caller says so
    // explicitly (PVV_COUNT_STAGED -- the state must then be a v3 call's), never under AUTO.  A staged pass
    // compares PARTIAL counts and therefore needs cleared counters.
    if (p && p->count_kernel == PVV_COUNT_STAGED && !zero_counts)
        return fail(PVV_E_ARG, "a staged count pass needs zero_counts = 1");
    const int v3 = (p && p->count_kernel == PVV_COUNT_STAGED) ? 1 : 0;
    if (int e = validate(p)) return e;
    if (!d_workspace) return fail(PVV_E_ARG, "workspace is NULL");
    Layout L = make_layout(p);
    if (workspace_bytes < L.total) return fail(PVV_E_WORKSPACE, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)d_workspace;
    if (zero_counts) {
        hipError_t e = hipMemsetAsync(ws + L.counts, 0, sizeof(int) * (size_t)p->B * p->K * p->hn, st);
        if (e == hipSuccess && L.lead) e = hipMemsetAsync(ws + L.lead, 0, sizeof(int) * lead_set_words(p), st);
        if (e == hipSuccess && L.miss) e = hipMemsetAsync(ws + L.miss, 0, sizeof(int) * (size_t)p->B * p->K * p->hn, st);
        if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    }
    return launch_count_any(p, L, ws, st, decide_staged(p, L, st, v3));
}

// ---- streaming-read probe (bench aid; SURVEY 8(d): what a read-once stream reaches on this box) ------------------
namespace {
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(kBlock) void k_stream_read(const u32x4 *__restrict__ src, size_t n16, uint32_t *__restrict__ sink)
{
    // persistent grid, every lane four independent 16-byte loads per trip (64 B in flight per lane), non-temporal: the
    // data is read once
    uint32_t acc = 0;
    const size_t stride = (size_t)gridDim.x * kBlock;
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const u32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
        const u32x4 c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
        acc += (a.x ^ a.y ^ a.z ^ a.w) + (b.x ^ b.y ^ b.z ^ b.w) + (c.x ^ c.y ^ c.z ^ c.w) + (d.x ^ d.y ^ d.z ^ d.w);
    }
    for (; i < n16; i += stride) {
        const u32x4 a = __builtin_nontemporal_load(src + i);
        acc += a.x ^ a.y ^ a.z ^ a.w;
    }
    acc = wave_sum(acc);
    if (lane_id() == 0 && acc == 0x9e3779b9u) atomicAdd(sink, acc);   // practically never: the sum only has to be observable
}
}  // namespace

PVV_EXPORT int pvv_stream_read_probe(const void *d_buf, size_t bytes, uint32_t *d_sink, void *stream)
{
    if (!d_buf || !d_sink || bytes < 16 || (bytes & 15) || ((uintptr_t)d_buf & 15)) return fail(PVV_E_ARG, "probe: 16-byte aligned buffer and size");
    hipLaunchKernelGGL(k_stream_read, dim3(num_cus() * 8), dim3(kBlock), 0, (hipStream_t)stream, (const u32x4 *)d_buf, bytes / 16, d_sink);
    return check_launch("k_stream_read");
}

// ---- legacy module surface ---------------------------------------------------------------
static int check_legacy(const void *a, const void *b, const void *c, const void *d, int tn, int vn,
                        int hn)
{
    if (!a || !b || !c || !d) return fail(PVV_E_ARG, "NULL device pointer");
    if (tn < 0 || vn <= 0 || hn < 0) return fail(PVV_E_ARG, "tn, hn must be >= 0 and vn > 0");
    if ((long long)hn * vn >= (1ll << 29) || (long long)tn * vn >= (1ll << 29) ||
        (long long)hn * vn * (long long)(tn > 0 ? tn : 1) >= (1ll << 40))
        return fail(PVV_E_ARG, "problem too large");
    return PVV_OK;
}

PVV_EXPORT int pvv_generate_hypothesis(const float *d_direct, const float *d_coords,
                                       const int32_t *d_idxs, float *d_hypo_pts, int tn, int vn,
                                       int hn, void *stream)
{
    if (int e = check_legacy(d_direct, d_coords, d_idxs, d_hypo_pts, tn, vn, hn)) return e;
    if (hn == 0) return PVV_OK;
    hipLaunchKernelGGL(k_legacy_gen, dim3((hn * vn + kBlock - 1) / kBlock), dim3(kBlock), 0,
                       (hipStream_t)stream, d_direct, d_coords, d_idxs, d_hypo_pts, tn, vn, hn);
    return check_launch("k_legacy_gen");
}

PVV_EXPORT int pvv_generate_hypothesis_vanishing_point(const float *d_direct, const float *d_coords,
                                                       const int32_t *d_idxs, float *d_hypo_pts,
                                                       int tn, int vn, int hn, void *stream)
{
    if (int e = check_legacy(d_direct, d_coords, d_idxs, d_hypo_pts, tn, vn, hn)) return e;
    if (hn == 0) return PVV_OK;
    hipLaunchKernelGGL(k_legacy_gen_vp, dim3((hn * vn + kBlock - 1) / kBlock), dim3(kBlock), 0,
                       (hipStream_t)stream, d_direct, d_coords, d_idxs, d_hypo_pts, tn, vn, hn);
    return check_launch("k_legacy_gen_vp");
}

static void legacy_vote_shape(int tn, int vn, int hn, dim3 *grid, int *h_per_block)
{
    // enough blocks to fill 256 CUs, few enough that each thread amortises its pixel load
    int tiles = (tn + kBlock - 1) / kBlock;
    int hz = 1;
    while ((long long)tiles * vn * hz < 4096 && hz < hn) hz <<= 1;
    if (hz > hn) hz = hn;
    if (hz > 65535) hz = 65535;
    *h_per_block = (hn + hz - 1) / hz;
    *grid = dim3(tiles, vn, (hn + *h_per_block - 1) / *h_per_block);
}

PVV_EXPORT int pvv_voting_for_hypothesis(const float *d_direct, const float *d_coords,
                                         const float *d_hypo_pts, uint8_t *d_inliers, int tn, int vn,
                                         int hn, float inlier_thresh, void *stream)
{
    if (int e = check_legacy(d_direct, d_coords, d_hypo_pts, d_inliers, tn, vn, hn)) return e;
    if (vn > 65535) return fail(PVV_E_ARG, "vn > 65535");
    if (hn == 0 || tn == 0) return PVV_OK;
    dim3 grid; int hpb;
    legacy_vote_shape(tn, vn, hn, &grid, &hpb);
    hipLaunchKernelGGL(k_legacy_vote, grid, dim3(kBlock), 0, (hipStream_t)stream, d_direct, d_coords,
                       d_hypo_pts, d_inliers, tn, vn, hn, hpb, inlier_thresh);
    return check_launch("k_legacy_vote");
}

PVV_EXPORT int pvv_voting_for_hypothesis_vanishing_point(const float *d_direct, const float *d_coords,
                                                         const float *d_hypo_pts, uint8_t *d_inliers,
                                                         int tn, int vn, int hn, float inlier_thresh,
                                                         void *stream)
{
    if (int e = check_legacy(d_direct, d_coords, d_hypo_pts, d_inliers, tn, vn, hn)) return e;
    if (vn > 65535) return fail(PVV_E_ARG, "vn > 65535");
    if (hn == 0 || tn == 0) return PVV_OK;
    dim3 grid; int hpb;
    legacy_vote_shape(tn, vn, hn, &grid, &hpb);
    hipLaunchKernelGGL(k_legacy_vote_vp, grid, dim3(kBlock), 0, (hipStream_t)stream, d_direct,
                       d_coords, d_hypo_pts, d_inliers, tn, vn, hn, hpb, inlier_thresh);
    return check_launch("k_legacy_vote_vp");
}

PVV_EXPORT int pvv_count_inliers(const float *d_direct, const float *d_coords, const float *d_hypo_pts,
                                 int32_t *d_counts, int tn, int vn, int hn, float inlier_thresh,
                                 void *stream)
{
    if (int e = check_legacy(d_direct, d_coords, d_hypo_pts, d_counts, tn, vn, hn)) return e;
    if (hn == 0) return PVV_OK;
    hipStream_t st = (hipStream_t)stream;
    hipError_t me = hipMemsetAsync(d_counts, 0, sizeof(int) * (size_t)hn * vn, st);
    if (me != hipSuccess) return fail((int)me, hipGetErrorString(me));
    if (tn == 0) return PVV_OK;
    CountArgs a;
    a.coords = (const float2 *)d_coords;
    a.dirs = (const float2 *)d_direct;
    a.hyps = (const float2 *)d_hypo_pts;
    a.counts = d_counts;
    a.tn_arr = nullptr;
    a.c_b = 0;
    a.d_b = 0; a.d_v = 1; a.d_p = vn;      // direct[ti,vi]
    a.h_b = 0; a.h_v = 1; a.h_h = vn;      // hypo[hi,vi], counts[hi,vi]
    a.tn_fixed = tn;
    a.B = 1; a.K = vn; a.hn = hn;
    a.thresh = inlier_thresh;
    return launch_count(a, st);
}
