/*
 * pvnet_vote.h -- C ABI of libpvnet_vote.so: clean-pvnet's RANSAC voting hot
 * path as hand-written HIP kernels for gfx950 (MI355X).
 *
 * Every pointer named d_* is a DEVICE pointer (HBM) owned by the caller; the
 * library allocates no device memory, synchronises nothing and launches
 * everything on the `stream` it is given (a hipStream_t passed as void*;
 * NULL = the null stream).  What it does own -- per device, created on first
 * use: 8 KB of pinned host memory (the stage hint) and one side stream with
 * 16 events (the deferred mask of the fused decode) -- pvv_shutdown() gives
 * back.  All entry points return 0 on success and a non-zero code on
 * failure (PVV_E_*; > 0 values are hipError_t from a failed launch);
 * pvv_last_error() returns a thread-local human readable message.  There is
 * no CPU fallback anywhere in this library.
 *
 * Citations are into /root/reference (zju3dv/clean-pvnet):
 *   K = lib/csrc/ransac_voting/src/ransac_voting_kernel.cu
 *   C = lib/csrc/ransac_voting/src/ransac_voting.cpp
 *   P = lib/csrc/ransac_voting/ransac_voting_gpu.py
 */
#ifndef PVNET_VOTE_H_
#define PVNET_VOTE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PVV_OK 0
#define PVV_E_ARG (-1)       /* bad shape / size / NULL pointer              */
#define PVV_E_WORKSPACE (-2) /* workspace smaller than pvv_workspace_bytes() */

/* ABI version of this header; pvv_abi_version() must return the same. */
#define PVV_ABI_VERSION 8

int pvv_abi_version(void);
const char *pvv_last_error(void);

/* ABI v8.  Library lifecycle: releases everything the library created behind the caller's back -- per device the pinned
 * host array of the stage hint (see pvv_stage_hint_query) and the side stream with its events (ABI v7) -- and forgets
 * every hint, so that the next call starts like the first one of a fresh process (it re-creates what it needs).  The
 * side streams are synchronised first; the caller must have no pvv_* call in flight on another thread and should have
 * synchronised the streams it launched on (a kernel still running may write its hint into the array this frees).  Call
 * it before unloading the library or resetting the device, or between workloads to make timing independent of call
 * history.  Returns 0, or the first hipError_t met (everything is released regardless). */
int pvv_shutdown(void);

/* ------------------------------------------------------------------------
 * Legacy extension-module surface: the four functions the reference's
 * pybind module `ransac_voting` exports (C:102-107).  Layouts exactly as the
 * reference: direct [tn,vn,2] f32, coords [tn,2] f32 (x,y), idxs [hn,vn,2]
 * i32, hypo_pts [hn,vn,2|3] f32, inliers [hn,vn,tn] u8 -- all contiguous.
 * ---------------------------------------------------------------------- */

/* Replaces generate_hypothesis (C:20-31 -> K:51-86 -> kernel K:11-49).
 * d_hypo_pts is fully written: (0,0) for degenerate pairs, as at::zeros + the
 * kernel's early return give in the reference (K:42-43,75). */
int pvv_generate_hypothesis(const float *d_direct, const float *d_coords,
                            const int32_t *d_idxs, float *d_hypo_pts, int tn,
                            int vn, int hn, void *stream);

/* Replaces voting_for_hypothesis (C:41-55 -> K:129-167 -> kernel K:88-126).
 * In/out d_inliers: only ever written with 1; the caller pre-zeroes (P:155). */
int pvv_voting_for_hypothesis(const float *d_direct, const float *d_coords,
                              const float *d_hypo_pts, uint8_t *d_inliers,
                              int tn, int vn, int hn, float inlier_thresh,
                              void *stream);

/* Replaces generate_hypothesis_vanishing_point (C:64-75 -> K:231-266 -> K:170-229). */
int pvv_generate_hypothesis_vanishing_point(const float *d_direct,
                                            const float *d_coords,
                                            const int32_t *d_idxs,
                                            float *d_hypo_pts, int tn, int vn,
                                            int hn, void *stream);

/* Replaces voting_for_hypothesis_vanishing_point (C:85-99 -> K:313-351 -> K:268-310). */
int pvv_voting_for_hypothesis_vanishing_point(const float *d_direct,
                                              const float *d_coords,
                                              const float *d_hypo_pts,
                                              uint8_t *d_inliers, int tn,
                                              int vn, int hn,
                                              float inlier_thresh,
                                              void *stream);

/* Fused voting_for_hypothesis + torch.sum(inlier, 2) (P:155-159) on the
 * reference layouts: d_counts [hn,vn] i32, fully written.  No [hn,vn,tn]
 * scratch exists. */
int pvv_count_inliers(const float *d_direct, const float *d_coords,
                      const float *d_hypo_pts, int32_t *d_counts, int tn,
                      int vn, int hn, float inlier_thresh, void *stream);

/* ------------------------------------------------------------------------
 * Batched, sync-free voting layers (replace the per-image Python loops of
 * P:112-199 and P:202-274).  One call = the whole batch, no host read-back.
 * ---------------------------------------------------------------------- */

typedef struct pvv_problem {
    int32_t B, H, W, K;      /* images, rows, cols, keypoints (vn)            */
    int32_t hn;              /* hypotheses per keypoint evaluated in this call */
    int32_t mask_elem_size;  /* bytes per mask element: 1,2,4,8 (bool/intN)   */
    int32_t min_num;         /* P:129 / P:211                                  */
    int32_t max_num;         /* P:135 / P:219                                  */
    int32_t cap;             /* rows reserved per image for compacted pixels;
                                use pvv_default_cap()                          */
    int32_t singular_policy; /* PVV_SINGULAR_*; v3 only                        */
    float inlier_thresh;     /* P:112 / P:202                                  */
    int64_t mask_stride[3];  /* element strides of mask   [B,H,W]              */
    int64_t vertex_stride[5];/* element strides of vertex [B,H,W,K,2] (any view,
                                e.g. the planar permute of resnet18.py:66-68)  */
    uint64_t seed;           /* counter-based RNG key, used when d_idxs /
                                d_selection are NULL                           */
    /* pvv_decode_keypoint_v3 only (ignored elsewhere): the segmentation logits */
    int32_t seg_classes;     /* C of seg [B,C,H,W] (2 for PVNet, config.py:108-112) */
    int32_t first_image;     /* index of image 0 of this call in the caller's larger batch: the device RNG is keyed
                                by (seed, first_image + b), so a batch split over several calls draws the same
                                numbers as one call.  0 for a whole batch (was reserved_, keep 0 if unsure) */
    int64_t seg_stride[4];   /* element strides of seg [B,C,H,W] (a channel slice
                                of the network output, resnet18.py:93)          */
    /* ---- ABI v5 ---- */
    int32_t count_kernel;    /* PVV_COUNT_*: which inlier-count kernel runs; 0 (AUTO) unless cross-checking */
    int32_t flags;           /* PVV_FLAG_* bits (ABI v8; was reserved0: 0 keeps the v7 behaviour) */
    int32_t *d_draws_out;    /* optional DEVICE buffer [B,K,hn,2] i32 (NULL = off): the pixel (y*W + x) each
                                hypothesis' index pair resolved to, -1 where none (image skipped).  Lets a test
                                replay the device RNG's draws through the oracle; for the fused un_pnp call hn is
                                p->hn + hn_est */
    void *ev_count_begin;    /* optional hipEvent_t pair (NULL = off), recorded on `stream` immediately before and   */
    void *ev_count_end;      /* after the inlier-count launch of THIS call: the dominant kernel's duration as it runs
                                inside the pipeline (a measurement aid).  pvv_decode_keypoint_un_pnp counting its rows as
                                two passes records begin before the first and end after the second: the pair then spans
                                v3's count pass, the refit and the estimate's count pass; its PVV_MARK_END (ev_marks) is
                                recorded behind the refit, i.e. BEFORE the estimate's pass */
    /* ---- ABI v6 ---- */
    int32_t *d_status;       /* optional DEVICE buffer [B] i32 (NULL = off): PVV_STATUS_* bits per image, written with tn.
                                The one condition a caller cannot see otherwise is PVV_STATUS_TRUNCATED: the subsample of
                                P:135-138 came out longer than the `cap` rows reserved (beyond 8 sigma with
                                pvv_default_cap) and was cut */
    void **ev_marks;         /* optional HOST array of PVV_N_MARKS hipEvent_t (NULL = off; NULL entries are skipped):
                                recorded on `stream` at the stage boundaries of THIS call (PVV_MARK_*), so that every
                                kernel's duration can be read as it runs inside the pipeline (bench.py's per-kernel
                                rooflines).  A measurement aid: the records cost ~1 us each */
} pvv_problem;

/* pvv_problem.flags (ABI v8) */
#define PVV_FLAG_DEVICE_RNG 1    /* the caller promises d_idxs = d_idxs_est = d_selection = NULL for the call this problem
                                    describes (the device RNG draws everything).  Only then is it known BEFORE the call that
                                    no per-pixel subsample draw is ever stored (small images subsample inside the compaction
                                    kernel and evaluate draws on demand), and pvv_workspace_bytes() leaves the 4 B x H x W per
                                    image of draw storage out: -27 % at 480x640, B = 64.  A call that sets the flag and
                                    passes one of those pointers fails with PVV_E_ARG */

/* pvv_problem.d_status bits */
#define PVV_STATUS_SKIPPED 1     /* foreground_num < min_num: the image's keypoints are zeros (P:129-132 / P:211-216) */
#define PVV_STATUS_SUBSAMPLED 2  /* foreground_num > max_num: pixels were kept with probability max_num/foreground_num  */
#define PVV_STATUS_TRUNCATED 4   /* more rows than `cap`: the list was cut at cap rows (results are those of the cut list) */

/* pvv_problem.ev_marks indices: the event is recorded AFTER the named stage has been enqueued */
#define PVV_MARK_BEGIN 0    /* before the first kernel of the call                                    */
#define PVV_MARK_SCAN 1     /* k_tile_scan (+ k_tile_subsample)                                       */
#define PVV_MARK_COMPACT 2  /* k_compact_hyp                                                          */
#define PVV_MARK_COUNT 3    /* the whole inlier-count pass (both launches and k_lead when staged)     */
#define PVV_MARK_SELECT 4   /* k_select_refit (v3)                                                    */
#define PVV_MARK_END 5      /* k_finalize_v3 / k_covariance: the call is complete                     */
#define PVV_MARK_STAGE0 6   /* staged count only: the first k_count_bf16 launch                       */
#define PVV_MARK_PRUNE0 7   /* staged count only: k_lead                                                 */
#define PVV_N_MARKS 8

/* pvv_problem.count_kernel.  AUTO: the split-bf16 matrix-core prefilter with its guard band wherever it is valid
 * (0.5 <= inlier_thresh <= 0.99995, H and W <= 16384), the exact kernel elsewhere.  EXACT: the reference's own
 * arithmetic (sqrt, divide) for every evaluation -- identical counts, ~9x slower; what the tests cross-check AUTO
 * against.  (Round 1 selected this with an environment variable; the library now reads no environment.) */
#define PVV_COUNT_AUTO 0
#define PVV_COUNT_EXACT 1
/* ABI v7 (round 4) -- no signature changed, three behaviours did:
 *  - the staged count's second launch owns RUNS of an (image, keypoint)'s remaining chunks and eliminates cooperatively
 *    through per-hypothesis miss counters (count_filter_runs.hpp): the workspace reserves them, pvv_workspace_bytes() grew;
 *  - pvv_rerun_count_kernel re-runs the pass in stages only for an explicit PVV_COUNT_STAGED (see there);
 *  - pvv_decode_keypoint_v3 / pvv_decode_keypoint_un_pnp on a two-class seg in two contiguous planes may write d_mask_out
 *    from a second, library-owned HIP stream (one per device, created on first use) that is forked from and joined back
 *    into `stream` inside the call: for the caller everything stays ordered on `stream`.  Not used while `stream` is being
 *    captured into a graph.  (v8: joined back on EVERY exit path, errors included.  There is ONE side stream per device: calls
 *    from several caller streams or threads fork onto it one after the other -- fork, launch and join record are one critical
 *    section -- so the deferred masks of concurrent calls serialise among themselves, and a call's join may wait for a mask
 *    kernel another call queued before it; the voting kernels of different streams stay independent.) */

/* ABI v6.  ransac_voting_layer_v3 keeps only the arg-max of the counts (P:160-167), so pvv_ransac_voting_v3 /
 * pvv_decode_keypoint_v3 may count in STAGES: every hypothesis over a spread quarter of the pixels, then only the
 * hypotheses that can still reach a lower bound of a leader's full count over the rest (k_lead, count_prune.hpp).  Winner,
 * first-index
 * tie rule, winner count and refit are bit-identical to the full pass; what differs is that the counters of eliminated
 * hypotheses hold partial counts (they are not an output of v3).  AUTO stages when the batch is large enough for the
 * two extra launches to pay; FULL = the matrix-core kernel over everything, never staged; STAGED = staged wherever the
 * matrix-core kernel is valid (what the tests force at every size).  The estimate weighs the hypotheses within 0.1 of the best
 * ratio and, since ABI v8, counts in stages against that bound where it pays (AUTO: est_stage_auto -- the stage hint of the v3
 * call that precedes every estimate, from ~6 LINEMOD frames on; pvv_estimate_counts_in_stages tells; never when its counts are
 * an output).  The fused un_pnp call (pvv_decode_keypoint_un_pnp) follows the same rule: where the estimate alone would stage,
 * its rows are counted as TWO passes -- v3's columns, then the estimate's against its bound -- otherwise as one full pass. */
#define PVV_COUNT_FULL 2
#define PVV_COUNT_STAGED 3
/* ABI v8: PVV_COUNT_STAGED stages ransac_voting_layer_v3 only (its v6 meaning; under v7 it also staged an estimate whose
 * counts are not an output).  The estimate counted in stages against its own bound (every hypothesis whose ratio can still
 * come within 0.1 of the best, P:262-264) is exact and, since round 5, faster than the full pass from ~6 LINEMOD frames on
 * (staged / full 0.83-0.85 at B = 24-64, >= 1.0 below 6 frames: DESIGN.md 4.2), which is where AUTO takes it; this value FORCES it
 * at every size: what the tests use to cross-check the bound.  For v3 calls it behaves like PVV_COUNT_STAGED. */
#define PVV_COUNT_STAGED_ESTIMATE 4

/* b_inv (P:97-109) falls back to the identity for the WHOLE image when the
 * batched solve raises; REFERENCE reproduces that (x = ATb for every keypoint
 * of an image that has a singular keypoint), ZERO confines the damage to the
 * singular keypoint, which becomes (0,0). */
#define PVV_SINGULAR_REFERENCE 0
#define PVV_SINGULAR_ZERO 1
/* ransac_voting_layer (v1, P:86-91): torch.inverse raises for the whole image
 * -> every keypoint of that image becomes (0,0). */
#define PVV_SINGULAR_IMAGE_ZERO 2

/* Rows to reserve per image: H*W when max_num >= H*W, otherwise max_num plus
 * 8 sigma of the binomial subsample of P:135-138 (a longer list is truncated). */
int32_t pvv_default_cap(int32_t H, int32_t W, int32_t max_num);

/* Bytes of device scratch the layer calls below need for `p`; the scratch must be
 * 256-byte aligned (any hipMalloc / torch allocation is). */
size_t pvv_workspace_bytes(const pvv_problem *p);

/* ransac_voting_layer_v3 (P:112-199).
 *   d_mask      [B,H,W] integer/bool mask, foreground = low byte != 0 (P:125)
 *   d_vertex    [B,H,W,K,2] f32 through p->vertex_stride
 *   d_idxs      [B,hn,K,2] i32 injected index pairs of P:145, or NULL (device RNG)
 *   d_selection [B,H,W] f32 injected U(0,1) draws of P:136, or NULL (device RNG)
 *   d_out       [B,K,2] f32 keypoint means
 *   d_win_counts[B,K] i32 inlier count of each winner (optional, may be NULL)
 *   d_tn        [B] i32 foreground pixels used per image (optional)
 * The confidence loop of P:150-174 cannot change the result (idxs are drawn
 * once, P:145) and is not executed. */
int pvv_ransac_voting_v3(const pvv_problem *p, const void *d_mask,
                         const float *d_vertex, const int32_t *d_idxs,
                         const float *d_selection, void *d_workspace,
                         size_t workspace_bytes, float *d_out,
                         int32_t *d_win_counts, int32_t *d_tn, void *stream);

/* Resnet18.decode_keypoint (lib/networks/pvnet/resnet18.py:65-76) with the
 * argmax fused into the mask scan: mask = argmax(seg, 1) (first maximum; a NaN
 * logit wins, as torch.argmax) is computed while the foreground is counted, so
 * the int64 mask is written once and never read back.
 *   d_seg       [B,C,H,W] f32 class logits through p->seg_stride
 *   d_mask_out  [B,H,W] i64, contiguous: the `mask` entry of the output dict
 *               (resnet18.py:72,76); may be NULL when the caller does not need it
 * everything else as pvv_ransac_voting_v3 (foreground = class != 0). */
int pvv_decode_keypoint_v3(const pvv_problem *p, const float *d_seg,
                           const float *d_vertex, const int32_t *d_idxs,
                           const float *d_selection, void *d_workspace,
                           size_t workspace_bytes, int64_t *d_mask_out,
                           float *d_out, int32_t *d_win_counts, int32_t *d_tn,
                           void *stream);

/* estimate_voting_distribution_with_mean (P:202-274); p->hn is the TOTAL
 * number of hypotheses (round_num * round_hyp_num, P:231-249).
 *   d_mask      foreground = element == 1 (P:207)
 *   d_idxs      [B,hn,K,2] i32 (the rounds of P:235 concatenated) or NULL
 *   d_mean      [B,K,2] f32
 *   d_cov       [B,K,2,2] f32
 *   d_hyp       [B,K,hn,2] f32 all hypotheses (optional, may be NULL)
 *   d_counts    [B,K,hn] i32 their inlier counts (optional)
 *   d_weights   [B,K,3] f32 (wxx,wxy,wyy) of inv(sqrtm(cov)) (optional): the
 *               per-keypoint weights the evaluators hand to uncertainty_pnp
 *               (lib/evaluators/linemod/pvnet.py:118-130), zeros where
 *               cov[0][0] < 1e-6, any entry is NaN, or cov is not positive definite */
int pvv_estimate_voting_distribution(const pvv_problem *p, const void *d_mask,
                                     const float *d_vertex,
                                     const int32_t *d_idxs,
                                     const float *d_selection,
                                     const float *d_mean, void *d_workspace,
                                     size_t workspace_bytes, float *d_cov,
                                     float *d_hyp, int32_t *d_counts,
                                     int32_t *d_tn, float *d_weights,
                                     void *stream);

/* Resnet18.decode_keypoint with cfg.test.un_pnp (resnet18.py:65-72) as ONE pass:
 *     mask   = argmax(seg, 1)
 *     mean   = ransac_voting_layer_v3(mask, vertex, p->hn, inlier_thresh)        (resnet18.py:71)
 *     kpt, var = estimate_voting_distribution_with_mean(mask, vertex, mean)       (resnet18.py:72, P:202-274)
 * The two layers scan and compact the same mask, so here the mask is scanned once, the foreground compacted once,
 * and ONE hypothesis launch + ONE inlier-count launch cover the p->hn hypotheses of the v3 layer and the hn_est
 * (= ceil(min_hyp_num / round_hyp_num) * round_hyp_num, 4096 by default) of the estimate; the refit reads the
 * first p->hn counts of a row, the covariance the rest.  Results are bit-identical to pvv_decode_keypoint_v3
 * followed by pvv_estimate_voting_distribution on the same draws.
 * Where the estimate alone would count in stages (pvv_estimate_counts_in_stages(p with hn = hn_est): large batches under
 * PVV_COUNT_AUTO, or PVV_COUNT_STAGED_ESTIMATE) the rows are counted as TWO passes over the one compaction instead: the
 * columns [0, p->hn) exactly as pvv_ransac_voting_v3 would count them (in full or in stages, the same rule and stage hint),
 * then, behind the refit, the columns [p->hn, p->hn + hn_est) against the estimate's bound.  Same results, bit for bit.
 * Requires p->seg_classes == 2 (PVNet's seg_dim, config.py:108-112): v3 votes with `mask != 0`, the estimate with
 * `mask == 1` (P:125 vs P:207), which coincide only for a two-class argmax -- PVV_E_ARG otherwise (make the two
 * calls then).  min_num / max_num of p apply to both layers, as in the reference's call (defaults of both).
 *   d_idxs      [B,p->hn,K,2] i32 or NULL      d_idxs_est  [B,hn_est,K,2] i32 or NULL   (device RNG where NULL)
 *   d_kpt       [B,K,2] f32 = mean             d_cov       [B,K,2,2] f32
 *   d_weights   [B,K,3] f32 (wxx,wxy,wyy) of inv(sqrtm(cov)) or NULL (evaluators/linemod/pvnet.py:118-128)
 * workspace: pvv_workspace_bytes_un_pnp(p, hn_est). */
size_t pvv_workspace_bytes_un_pnp(const pvv_problem *p, int32_t hn_est);

/* ABI v8: 1 when pvv_estimate_voting_distribution would count this problem IN STAGES (d_counts == NULL; PVV_COUNT_STAGED_ESTIMATE,
 * or PVV_COUNT_AUTO: from ~2e11 evaluations-equivalent B*K*hn*H*W on -- 18 LINEMOD frames at 4096 hypotheses -- or, once a v3 call
 * on fields of the same H, W, K has reported its winners' ratios and tn on the current device (the stage hint; resnet18.py:71-72
 * runs v3 right before every estimate), from 6e10 of REAL work K*hn*sum(tn)/0.02 on clean fields (6 LINEMOD frames), 9e10 for
 * ratios >= 0.85), 0 when it counts in full, < 0 for an invalid problem.
 * pvv_decode_keypoint_un_pnp applies the same rule to its estimate columns (see there), so a host no longer has to choose
 * between the fused call and the two calls; the query stays for hosts that want to know which pass will run. */
int pvv_estimate_counts_in_stages(const pvv_problem *p);
int pvv_decode_keypoint_un_pnp(const pvv_problem *p, int32_t hn_est, const float *d_seg,
                               const float *d_vertex, const int32_t *d_idxs,
                               const int32_t *d_idxs_est, const float *d_selection,
                               void *d_workspace, size_t workspace_bytes,
                               int64_t *d_mask_out, float *d_kpt, float *d_cov,
                               float *d_weights, int32_t *d_win_counts, int32_t *d_tn,
                               void *stream);

/* Bench aid: SURVEY 8(d)'s "achievable number from a streaming-read microbenchmark on the box".  Reads `bytes` (a
 * multiple of 16) of d_buf once with 16-byte loads per lane from a persistent grid and writes a 4-byte checksum to
 * d_sink (so the loads cannot be elided); bracket it with events on `stream`.  Nothing of the voting path uses it. */
int pvv_stream_read_probe(const void *d_buf, size_t bytes, uint32_t *d_sink, void *stream);

/* The stage hint (ABI v6).  Whether staged counting pays depends on how clean the vector field is -- on the winners'
 * inlier ratio, which nobody knows before the call -- but consecutive calls see similar data, and every v3 call ends with
 * the exact winner counts.  The library therefore keeps, per device, the mean winner ratio (winner count / tn) of every
 * image of the last completed v3 calls in a small pinned array the GPU writes, and PVV_COUNT_AUTO stages a call only if
 * that mean reaches a threshold that depends on the problem's size (pvnet_vote.hip, stage_hint_threshold; DESIGN.md
 * 4.6-4.7).  The same array carries every image's tn, so the size that decides is the call's real work -- K * hn * sum(tn)
 * evaluations as the last call of this shape reported them (dense detector crops stage at a batch size where sparse full
 * frames do not; "shape" = H, W, K, hn: the batch size may change from call to call, the sums are scaled to it) -- and
 * B*K*hn*H*W, a proxy calibrated on frames with 2 % foreground, only while no call has reported.  The
 * hint lags by the calls in flight and only selects between two exact paths; with no data yet the proxy alone decides;
 * PVV_COUNT_STAGED / PVV_COUNT_FULL ignore it.  This query is for tests, benches and the curious: returns 1 and
 * the mean when data is there (0 and -1 otherwise), and the threshold of `p` (p may be NULL: -1; 2 = a problem with too
 * little work to be staged at all). */
int pvv_stage_hint_query(float *mean_ratio, float *threshold, const pvv_problem *p, void *stream);

/* Bench / profiling aid: re-runs ONLY the inlier-count kernel of the last
 * layer call recorded in `d_workspace` (same problem), so its duration can be
 * bracketed with HIP events on `stream`.  zero_counts != 0 first clears the
 * counters (a separate memset node) so the result stays valid; with 0 nothing
 * but the kernel is enqueued and the counters keep accumulating.  The pass is
 * re-run IN STAGES (both launches and k_lead) only when count_kernel is
 * PVV_COUNT_STAGED explicitly -- the workspace must then hold a v3 call's
 * state and zero_counts must be 1 (the elimination compares partial counts);
 * under AUTO the full kernel runs: the workspace may be the estimate's or the
 * fused un_pnp pass's, which need every count.  `p` must describe the call
 * that filled the workspace INCLUDING `cap` (the offsets depend on it). */
int pvv_rerun_count_kernel(const pvv_problem *p, void *d_workspace,
                           size_t workspace_bytes, int zero_counts,
                           void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PVNET_VOTE_H_ */
