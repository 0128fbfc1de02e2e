// ransac_voting_ext.cpp -- pybind11/torch shim over the C ABI of libpvnet_vote.so.
//
// Reproduces the extension-module surface of the reference
// (/root/reference/lib/csrc/ransac_voting/src/ransac_voting.cpp:102-107): the module is called
// `ransac_voting` and exports generate_hypothesis, voting_for_hypothesis,
// generate_hypothesis_vanishing_point and voting_for_hypothesis_vanishing_point with the
// reference's argument order and ownership rules (generate_* allocate and return a tensor,
// voting_* mutate `inliers` in place).  Differences, all deliberate:
//   * launches go to the CURRENT torch HIP stream, not the legacy default stream
//     (ransac_voting_kernel.cu:76,159 use <<<bdim,tdim>>>);
//   * device / dtype / contiguity / every dimension are checked with TORCH_CHECK
//     (-> Python RuntimeError) where the reference has CHECK_INPUT (ransac_voting.cpp:7-9) plus
//     bare assert()s (kernel.cu:61-65) and exit() on a launch error (cuda_common.h:19-26).
// This file contains no device code; it is compiled by the host compiler only.
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime_api.h>
#include <torch/extension.h>

#include <optional>
#include <tuple>
#include <vector>

#include "pvnet_vote.h"

namespace {

void *cur_stream(const at::Tensor &t)
{
    return (void *)c10::hip::getCurrentHIPStream(t.device().index()).stream();
}

void check_dev(const at::Tensor &t, const char *name, at::ScalarType st)
{
    TORCH_CHECK(t.is_cuda(), name, " must be a CUDA tensor");  // message as ransac_voting.cpp:7
    TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");  // ransac_voting.cpp:8
    TORCH_CHECK(t.scalar_type() == st, name, " must have dtype ", st, ", got ", t.scalar_type());
}

void same_device(const at::Tensor &a, const at::Tensor &b, const char *nb)
{
    TORCH_CHECK(a.device() == b.device(), nb, " is on ", b.device(), ", expected ", a.device());
}

void ok(int code, const char *what)
{
    TORCH_CHECK(code == 0, what, " failed (", code, "): ", pvv_last_error());
}

struct Dims { int tn, vn, hn; };

Dims check_gen(const at::Tensor &direct, const at::Tensor &coords, const at::Tensor &idxs)
{
    check_dev(direct, "direct", at::kFloat);
    check_dev(coords, "coords", at::kFloat);
    check_dev(idxs, "idxs", at::kInt);
    same_device(direct, coords, "coords");
    same_device(direct, idxs, "idxs");
    TORCH_CHECK(direct.dim() == 3 && direct.size(2) == 2, "direct must be [tn,vn,2]");
    const int64_t tn = direct.size(0), vn = direct.size(1);
    TORCH_CHECK(vn > 0, "direct must have vn > 0");
    TORCH_CHECK(coords.dim() == 2 && coords.size(0) == tn && coords.size(1) == 2, "coords must be [tn,2]");
    TORCH_CHECK(idxs.dim() == 3 && idxs.size(1) == vn && idxs.size(2) == 2, "idxs must be [hn,vn,2]");
    TORCH_CHECK(tn > 0 || idxs.size(0) == 0, "idxs index an empty pixel list");
    return {(int)tn, (int)vn, (int)idxs.size(0)};
}

Dims check_vote(const at::Tensor &direct, const at::Tensor &coords, const at::Tensor &hypo_pts,
                const at::Tensor &inliers, int hdim)
{
    check_dev(direct, "direct", at::kFloat);
    check_dev(coords, "coords", at::kFloat);
    check_dev(hypo_pts, "hypo_pts", at::kFloat);
    check_dev(inliers, "inliers", at::kByte);
    same_device(direct, coords, "coords");
    same_device(direct, hypo_pts, "hypo_pts");
    same_device(direct, inliers, "inliers");
    TORCH_CHECK(direct.dim() == 3 && direct.size(2) == 2, "direct must be [tn,vn,2]");
    const int64_t tn = direct.size(0), vn = direct.size(1);
    TORCH_CHECK(vn > 0, "direct must have vn > 0");
    TORCH_CHECK(coords.dim() == 2 && coords.size(0) == tn && coords.size(1) == 2, "coords must be [tn,2]");
    TORCH_CHECK(hypo_pts.dim() == 3 && hypo_pts.size(1) == vn && hypo_pts.size(2) == hdim,
                "hypo_pts must be [hn,vn,", hdim, "]");
    const int64_t hn = hypo_pts.size(0);
    TORCH_CHECK(inliers.dim() == 3 && inliers.size(0) == hn && inliers.size(1) == vn && inliers.size(2) == tn,
                "inliers must be [hn,vn,tn]");
    return {(int)tn, (int)vn, (int)hn};
}

// ---- reference module surface -----------------------------------------------------------

at::Tensor generate_hypothesis(at::Tensor direct, at::Tensor coords, at::Tensor idxs)
{
    const c10::DeviceGuard device_guard(direct.device());   // launch on the tensors' GPU, whatever the current device is
    Dims d = check_gen(direct, coords, idxs);
    auto hypo_pts = at::empty({d.hn, d.vn, 2}, direct.options());
    ok(pvv_generate_hypothesis(direct.data_ptr<float>(), coords.data_ptr<float>(), idxs.data_ptr<int32_t>(),
                               hypo_pts.data_ptr<float>(), d.tn, d.vn, d.hn, cur_stream(direct)),
       "generate_hypothesis");
    return hypo_pts;
}

void voting_for_hypothesis(at::Tensor direct, at::Tensor coords, at::Tensor hypo_pts, at::Tensor inliers,
                           float inlier_thresh)
{
    const c10::DeviceGuard device_guard(direct.device());   // launch on the tensors' GPU, whatever the current device is
    Dims d = check_vote(direct, coords, hypo_pts, inliers, 2);
    ok(pvv_voting_for_hypothesis(direct.data_ptr<float>(), coords.data_ptr<float>(), hypo_pts.data_ptr<float>(),
                                 inliers.data_ptr<uint8_t>(), d.tn, d.vn, d.hn, inlier_thresh,
                                 cur_stream(direct)),
       "voting_for_hypothesis");
}

at::Tensor generate_hypothesis_vanishing_point(at::Tensor direct, at::Tensor coords, at::Tensor idxs)
{
    const c10::DeviceGuard device_guard(direct.device());   // launch on the tensors' GPU, whatever the current device is
    Dims d = check_gen(direct, coords, idxs);
    auto hypo_pts = at::empty({d.hn, d.vn, 3}, direct.options());
    ok(pvv_generate_hypothesis_vanishing_point(direct.data_ptr<float>(), coords.data_ptr<float>(),
                                               idxs.data_ptr<int32_t>(), hypo_pts.data_ptr<float>(), d.tn,
                                               d.vn, d.hn, cur_stream(direct)),
       "generate_hypothesis_vanishing_point");
    return hypo_pts;
}

void voting_for_hypothesis_vanishing_point(at::Tensor direct, at::Tensor coords, at::Tensor hypo_pts,
                                           at::Tensor inliers, float inlier_thresh)
{
    const c10::DeviceGuard device_guard(direct.device());   // launch on the tensors' GPU, whatever the current device is
    Dims d = check_vote(direct, coords, hypo_pts, inliers, 3);
    ok(pvv_voting_for_hypothesis_vanishing_point(direct.data_ptr<float>(), coords.data_ptr<float>(),
                                                 hypo_pts.data_ptr<float>(), inliers.data_ptr<uint8_t>(),
                                                 d.tn, d.vn, d.hn, inlier_thresh, cur_stream(direct)),
       "voting_for_hypothesis_vanishing_point");
}

// ---- additions -----------------------------------------------------------------------------

// voting_for_hypothesis + sum over tn without the [hn,vn,tn] scratch -> [hn,vn] int32
at::Tensor count_inliers(at::Tensor direct, at::Tensor coords, at::Tensor hypo_pts, float inlier_thresh)
{
    const c10::DeviceGuard device_guard(direct.device());   // launch on the tensors' GPU, whatever the current device is
    check_dev(direct, "direct", at::kFloat);
    check_dev(coords, "coords", at::kFloat);
    check_dev(hypo_pts, "hypo_pts", at::kFloat);
    same_device(direct, coords, "coords");
    same_device(direct, hypo_pts, "hypo_pts");
    TORCH_CHECK(direct.dim() == 3 && direct.size(2) == 2, "direct must be [tn,vn,2]");
    const int64_t tn = direct.size(0), vn = direct.size(1);
    TORCH_CHECK(vn > 0, "direct must have vn > 0");
    TORCH_CHECK(coords.dim() == 2 && coords.size(0) == tn && coords.size(1) == 2, "coords must be [tn,2]");
    TORCH_CHECK(hypo_pts.dim() == 3 && hypo_pts.size(1) == vn && hypo_pts.size(2) == 2, "hypo_pts must be [hn,vn,2]");
    auto counts = at::empty({hypo_pts.size(0), vn}, direct.options().dtype(at::kInt));
    ok(pvv_count_inliers(direct.data_ptr<float>(), coords.data_ptr<float>(), hypo_pts.data_ptr<float>(),
                         counts.data_ptr<int32_t>(), (int)tn, (int)vn, (int)hypo_pts.size(0), inlier_thresh,
                         cur_stream(direct)),
       "count_inliers");
    return counts;
}

int mask_elem_size(const at::Tensor &mask)
{
    switch (mask.scalar_type()) {
    case at::kBool: case at::kByte: case at::kChar: return 1;
    case at::kShort: return 2;
    case at::kInt: return 4;
    case at::kLong: return 8;
    default: TORCH_CHECK(false, "mask must be a bool or integer tensor, got ", mask.scalar_type());
    }
}

pvv_problem make_problem(const at::Tensor &mask, const at::Tensor &vertex, int64_t hn, double thresh,
                         int64_t min_num, int64_t max_num, int64_t policy, int64_t seed)
{
    TORCH_CHECK(mask.is_cuda(), "mask must be a CUDA tensor");
    TORCH_CHECK(vertex.is_cuda(), "vertex must be a CUDA tensor");
    same_device(mask, vertex, "vertex");
    TORCH_CHECK(vertex.scalar_type() == at::kFloat, "vertex must be float32, got ", vertex.scalar_type());
    TORCH_CHECK(vertex.dim() == 5 && vertex.size(4) == 2, "vertex must be [b,h,w,vn,2]");
    TORCH_CHECK(mask.dim() == 3 && mask.size(0) == vertex.size(0) && mask.size(1) == vertex.size(1) &&
                    mask.size(2) == vertex.size(2),
                "mask must be [b,h,w] matching vertex");
    TORCH_CHECK(hn > 0 && hn < (1 << 24), "hypothesis count must be in [1, 2^24)");
    TORCH_CHECK(vertex.size(0) > 0 && vertex.size(3) > 0, "empty batch / no keypoints");
    pvv_problem p;
    memset(&p, 0, sizeof(p));
    p.B = (int32_t)vertex.size(0); p.H = (int32_t)vertex.size(1); p.W = (int32_t)vertex.size(2);
    p.K = (int32_t)vertex.size(3);
    p.hn = (int32_t)hn;
    p.mask_elem_size = mask_elem_size(mask);
    const int64_t big = std::numeric_limits<int32_t>::max();
    p.min_num = (int32_t)std::max<int64_t>(-big, std::min<int64_t>(big, min_num));
    p.max_num = (int32_t)std::max<int64_t>(0, std::min<int64_t>(big, max_num));
    p.cap = pvv_default_cap(p.H, p.W, p.max_num);
    p.singular_policy = (int32_t)policy;
    p.inlier_thresh = (float)thresh;
    for (int i = 0; i < 3; ++i) p.mask_stride[i] = mask.stride(i);
    for (int i = 0; i < 5; ++i) p.vertex_stride[i] = vertex.stride(i);
    p.seed = (uint64_t)seed;
    return p;
}

const int32_t *opt_idxs(const std::optional<at::Tensor> &idxs, const at::Tensor &vertex, const pvv_problem &p)
{
    if (!idxs.has_value()) return nullptr;
    const at::Tensor &t = *idxs;
    check_dev(t, "idxs", at::kInt);
    same_device(vertex, t, "idxs");
    TORCH_CHECK(t.dim() == 4 && t.size(0) == p.B && t.size(1) == p.hn && t.size(2) == p.K && t.size(3) == 2,
                "idxs must be [b,hn,vn,2] = [", p.B, ",", p.hn, ",", p.K, ",2]");
    return t.data_ptr<int32_t>();
}

// Injected U(0,1) draws may let ANY number of foreground pixels survive (all zeros keep every pixel), so the rows
// reserved per image are not bounded by max_num + 8 sigma then: reserve the whole image (call before make_workspace).
void cap_for_selection(const std::optional<at::Tensor> &sel, pvv_problem &p)
{
    if (sel.has_value()) p.cap = (int32_t)std::min<int64_t>((int64_t)p.H * p.W, std::numeric_limits<int32_t>::max());
}

const float *opt_selection(const std::optional<at::Tensor> &sel, const at::Tensor &vertex, const pvv_problem &p)
{
    if (!sel.has_value()) return nullptr;
    const at::Tensor &t = *sel;
    check_dev(t, "selection", at::kFloat);
    same_device(vertex, t, "selection");
    TORCH_CHECK(t.dim() == 3 && t.size(0) == p.B && t.size(1) == p.H && t.size(2) == p.W,
                "selection must be [b,h,w]");
    return t.data_ptr<float>();
}

// ABI v8: with nothing injected the device RNG draws everything -- the library then knows before the call that no per-pixel
// subsample draw is ever stored for images small enough to subsample inside the compaction kernel, and leaves that storage
// (4 B per pixel of the batch) out of the workspace.  Call before make_workspace.
void promise_device_rng(pvv_problem &p, const std::optional<at::Tensor> &idxs, const std::optional<at::Tensor> &idxs2,
                        const std::optional<at::Tensor> &selection)
{
    if (!idxs.has_value() && !idxs2.has_value() && !selection.has_value()) p.flags |= PVV_FLAG_DEVICE_RNG;
}

// `out` (optional): the caller's [b,vn,2] float32 result buffer -- e.g. this rank's rows of a persistent all_gather buffer
// (clean_pvnet_amd.dist.GatherBuffer), so that the exchange needs no copy -- instead of a fresh tensor
at::Tensor result_buffer(const std::optional<at::Tensor> &out, const pvv_problem &p, const at::Tensor &vertex)
{
    if (!out.has_value()) return at::empty({p.B, p.K, 2}, vertex.options());
    check_dev(*out, "out", at::kFloat);
    same_device(vertex, *out, "out");
    TORCH_CHECK(out->dim() == 3 && out->size(0) == p.B && out->size(1) == p.K && out->size(2) == 2, "out must be [b,vn,2] = [", p.B, ",", p.K, ",2]");
    return *out;
}

at::Tensor make_workspace(const pvv_problem &p, const at::Tensor &like)
{
    size_t n = pvv_workspace_bytes(&p);
    TORCH_CHECK(n > 0, "invalid voting problem: ", pvv_last_error());
    return at::empty({(int64_t)n}, like.options().dtype(at::kByte));
}

// ransac_voting_layer_v3 for the whole batch -> (kpt [b,vn,2], win_counts [b,vn], tn [b], workspace)
std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor> ransac_voting_v3(
    at::Tensor mask, at::Tensor vertex, int64_t round_hyp_num, double inlier_thresh, int64_t min_num,
    int64_t max_num, std::optional<at::Tensor> idxs, std::optional<at::Tensor> selection, int64_t seed,
    int64_t singular_policy, int64_t first_image, int64_t count_kernel, std::optional<at::Tensor> status,
    std::optional<int64_t> cap, std::optional<at::Tensor> out_buf)
{
    const c10::DeviceGuard device_guard(vertex.device());   // launch on the tensors' GPU, whatever the current device is
    pvv_problem p = make_problem(mask, vertex, round_hyp_num, inlier_thresh, min_num, max_num,
                                 singular_policy, seed);
    p.first_image = (int32_t)first_image;
    p.count_kernel = (int32_t)count_kernel;
    if (status.has_value()) {                               // per-image PVV_STATUS_* bits (e.g. list truncated at cap)
        check_dev(*status, "status", at::kInt);
        same_device(vertex, *status, "status");
        TORCH_CHECK(status->dim() == 1 && status->size(0) == p.B, "status must be [b]");
        p.d_status = status->data_ptr<int32_t>();
    }
    cap_for_selection(selection, p);
    if (cap.has_value()) {                                  // rows reserved per image (default: pvv_default_cap)
        TORCH_CHECK(*cap >= 1 && *cap <= (int64_t)p.H * p.W, "cap must be in [1, h*w]");
        p.cap = (int32_t)*cap;
    }
    const int32_t *ip = opt_idxs(idxs, vertex, p);
    const float *sp = opt_selection(selection, vertex, p);
    promise_device_rng(p, idxs, std::nullopt, selection);
    at::Tensor ws = make_workspace(p, vertex);
    auto out = result_buffer(out_buf, p, vertex);
    auto win = at::empty({p.B, p.K}, vertex.options().dtype(at::kInt));
    auto tn = at::empty({p.B}, vertex.options().dtype(at::kInt));
    ok(pvv_ransac_voting_v3(&p, mask.data_ptr(), vertex.data_ptr<float>(), ip, sp, ws.data_ptr(),
                            (size_t)ws.numel(), out.data_ptr<float>(), win.data_ptr<int32_t>(),
                            tn.data_ptr<int32_t>(), cur_stream(vertex)),
       "ransac_voting_v3");
    return {out, win, tn, ws};
}

// Resnet18.decode_keypoint's front half fused: mask = argmax(seg, 1) + ransac_voting_layer_v3(mask, vertex, ...)
// -> (kpt [b,vn,2], mask [b,h,w] int64, win_counts [b,vn], tn [b])
std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor> decode_keypoint_v3(
    at::Tensor seg, at::Tensor vertex, int64_t round_hyp_num, double inlier_thresh, int64_t min_num, int64_t max_num,
    std::optional<at::Tensor> idxs, std::optional<at::Tensor> selection, int64_t seed, int64_t singular_policy,
    int64_t first_image, int64_t count_kernel)
{
    const c10::DeviceGuard device_guard(vertex.device());   // launch on the tensors' GPU, whatever the current device is
    TORCH_CHECK(seg.is_cuda(), "seg must be a CUDA tensor");
    TORCH_CHECK(seg.scalar_type() == at::kFloat, "seg must be float32, got ", seg.scalar_type());
    TORCH_CHECK(seg.dim() == 4 && vertex.dim() == 5 && seg.size(0) == vertex.size(0) && seg.size(2) == vertex.size(1) &&
                    seg.size(3) == vertex.size(2),
                "seg must be [b,c,h,w] matching vertex [b,h,w,vn,2]");
    auto mask = at::empty({seg.size(0), seg.size(2), seg.size(3)}, seg.options().dtype(at::kLong));
    pvv_problem p = make_problem(mask, vertex, round_hyp_num, inlier_thresh, min_num, max_num, singular_policy, seed);
    p.first_image = (int32_t)first_image;
    p.count_kernel = (int32_t)count_kernel;
    p.seg_classes = (int32_t)seg.size(1);
    for (int i = 0; i < 4; ++i) p.seg_stride[i] = seg.stride(i);
    cap_for_selection(selection, p);
    const int32_t *ip = opt_idxs(idxs, vertex, p);
    const float *sp = opt_selection(selection, vertex, p);
    promise_device_rng(p, idxs, std::nullopt, selection);
    at::Tensor ws = make_workspace(p, vertex);
    auto out = at::empty({p.B, p.K, 2}, vertex.options());
    auto win = at::empty({p.B, p.K}, vertex.options().dtype(at::kInt));
    auto tn = at::empty({p.B}, vertex.options().dtype(at::kInt));
    ok(pvv_decode_keypoint_v3(&p, seg.data_ptr<float>(), vertex.data_ptr<float>(), ip, sp, ws.data_ptr(),
                              (size_t)ws.numel(), mask.data_ptr<int64_t>(), out.data_ptr<float>(),
                              win.data_ptr<int32_t>(), tn.data_ptr<int32_t>(), cur_stream(vertex)),
       "decode_keypoint_v3");
    return {out, mask, win, tn};
}

// Resnet18.decode_keypoint with cfg.test.un_pnp in one pass (pvv_decode_keypoint_un_pnp): one mask scan, one compaction,
// one hypothesis + count launch for the round_hyp_num hypotheses of v3 and the hyp_est of the estimate.
// -> (kpt [b,vn,2], mask [b,h,w] int64, cov [b,vn,2,2], weights [b,vn,3], win_counts [b,vn], tn [b])
std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor> decode_keypoint_un_pnp(
    at::Tensor seg, at::Tensor vertex, int64_t round_hyp_num, int64_t hyp_est, double inlier_thresh, int64_t min_num,
    int64_t max_num, std::optional<at::Tensor> idxs, std::optional<at::Tensor> idxs_est,
    std::optional<at::Tensor> selection, int64_t seed, int64_t singular_policy, int64_t first_image, int64_t count_kernel)
{
    const c10::DeviceGuard device_guard(vertex.device());
    TORCH_CHECK(seg.is_cuda(), "seg must be a CUDA tensor");
    TORCH_CHECK(seg.scalar_type() == at::kFloat, "seg must be float32, got ", seg.scalar_type());
    TORCH_CHECK(seg.dim() == 4 && vertex.dim() == 5 && seg.size(0) == vertex.size(0) && seg.size(2) == vertex.size(1) &&
                    seg.size(3) == vertex.size(2),
                "seg must be [b,c,h,w] matching vertex [b,h,w,vn,2]");
    TORCH_CHECK(hyp_est > 0 && hyp_est < (1 << 24), "hypothesis count of the estimate must be in [1, 2^24)");
    auto mask = at::empty({seg.size(0), seg.size(2), seg.size(3)}, seg.options().dtype(at::kLong));
    pvv_problem p = make_problem(mask, vertex, round_hyp_num, inlier_thresh, min_num, max_num, singular_policy, seed);
    p.first_image = (int32_t)first_image;
    p.count_kernel = (int32_t)count_kernel;
    p.seg_classes = (int32_t)seg.size(1);
    for (int i = 0; i < 4; ++i) p.seg_stride[i] = seg.stride(i);
    cap_for_selection(selection, p);
    const int32_t *ip = opt_idxs(idxs, vertex, p);
    pvv_problem pe = p;
    pe.hn = (int32_t)hyp_est;
    const int32_t *ie = opt_idxs(idxs_est, vertex, pe);
    const float *sp = opt_selection(selection, vertex, p);
    promise_device_rng(p, idxs, idxs_est, selection);
    const size_t n = pvv_workspace_bytes_un_pnp(&p, (int32_t)hyp_est);
    TORCH_CHECK(n > 0, "invalid voting problem: ", pvv_last_error());
    at::Tensor ws = at::empty({(int64_t)n}, vertex.options().dtype(at::kByte));
    auto kpt = at::empty({p.B, p.K, 2}, vertex.options());
    auto cov = at::empty({p.B, p.K, 2, 2}, vertex.options());
    auto weights = at::empty({p.B, p.K, 3}, vertex.options());
    auto win = at::empty({p.B, p.K}, vertex.options().dtype(at::kInt));
    auto tn = at::empty({p.B}, vertex.options().dtype(at::kInt));
    ok(pvv_decode_keypoint_un_pnp(&p, (int32_t)hyp_est, seg.data_ptr<float>(), vertex.data_ptr<float>(), ip, ie, sp,
                                  ws.data_ptr(), n, mask.data_ptr<int64_t>(), kpt.data_ptr<float>(), cov.data_ptr<float>(),
                                  weights.data_ptr<float>(), win.data_ptr<int32_t>(), tn.data_ptr<int32_t>(),
                                  cur_stream(vertex)),
       "decode_keypoint_un_pnp");
    return {kpt, mask, cov, weights, win, tn};
}

// estimate_voting_distribution_with_mean for the whole batch
// -> (cov [b,vn,2,2], hyp [b,vn,hn,2] | empty, counts [b,vn,hn] | empty, tn [b], weights [b,vn,3])
std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor> estimate_voting_distribution(
    at::Tensor mask, at::Tensor vertex, at::Tensor mean, int64_t hyp_total, double inlier_thresh,
    int64_t min_num, int64_t max_num, std::optional<at::Tensor> idxs, std::optional<at::Tensor> selection,
    int64_t seed, bool want_hyp, int64_t first_image, int64_t count_kernel)
{
    const c10::DeviceGuard device_guard(vertex.device());   // launch on the tensors' GPU, whatever the current device is
    pvv_problem p = make_problem(mask, vertex, hyp_total, inlier_thresh, min_num, max_num, 0, seed);
    p.first_image = (int32_t)first_image;
    p.count_kernel = (int32_t)count_kernel;
    cap_for_selection(selection, p);
    check_dev(mean, "mean", at::kFloat);
    same_device(vertex, mean, "mean");
    TORCH_CHECK(mean.dim() == 3 && mean.size(0) == p.B && mean.size(1) == p.K && mean.size(2) == 2,
                "mean must be [b,vn,2]");
    const int32_t *ip = opt_idxs(idxs, vertex, p);
    const float *sp = opt_selection(selection, vertex, p);
    promise_device_rng(p, idxs, std::nullopt, selection);
    at::Tensor ws = make_workspace(p, vertex);
    auto cov = at::empty({p.B, p.K, 2, 2}, vertex.options());
    auto tn = at::empty({p.B}, vertex.options().dtype(at::kInt));
    at::Tensor hyp, counts;
    if (want_hyp) {
        hyp = at::empty({p.B, p.K, p.hn, 2}, vertex.options());
        counts = at::empty({p.B, p.K, p.hn}, vertex.options().dtype(at::kInt));
    } else {
        hyp = at::empty({0}, vertex.options());
        counts = at::empty({0}, vertex.options().dtype(at::kInt));
    }
    auto weights = at::empty({p.B, p.K, 3}, vertex.options());
    ok(pvv_estimate_voting_distribution(&p, mask.data_ptr(), vertex.data_ptr<float>(), ip, sp,
                                        mean.data_ptr<float>(), ws.data_ptr(), (size_t)ws.numel(),
                                        cov.data_ptr<float>(), want_hyp ? hyp.data_ptr<float>() : nullptr,
                                        want_hyp ? counts.data_ptr<int32_t>() : nullptr,
                                        tn.data_ptr<int32_t>(), weights.data_ptr<float>(), cur_stream(vertex)),
       "estimate_voting_distribution");
    return {cov, hyp, counts, tn, weights};
}

// `reps` full ransac_voting_layer_v3 calls cycling over the given batches, each with a HIP event pair recorded around its
// inlier-count launch (pvv_problem.ev_count_begin / ev_count_end) -> the kernel's duration inside the pipeline, ms per call.
// One synchronisation at the end; a measurement aid (bench.py), not part of the voting path.
std::vector<double> count_kernel_ms_in_pipeline(std::vector<at::Tensor> masks, std::vector<at::Tensor> vertices,
                                                int64_t round_hyp_num, double inlier_thresh, int64_t min_num,
                                                int64_t max_num, int64_t seed, int64_t reps)
{
    TORCH_CHECK(!masks.empty() && masks.size() == vertices.size(), "need as many masks as vertex fields");
    const c10::DeviceGuard device_guard(vertices[0].device());
    std::vector<hipEvent_t> ev(2 * (size_t)reps);
    for (auto &e : ev) TORCH_CHECK(hipEventCreate(&e) == hipSuccess, "hipEventCreate failed");
    std::vector<at::Tensor> keep;
    for (int64_t r = 0; r < reps; ++r) {
        const at::Tensor &mask = masks[r % masks.size()], &vertex = vertices[r % masks.size()];
        pvv_problem p = make_problem(mask, vertex, round_hyp_num, inlier_thresh, min_num, max_num, PVV_SINGULAR_REFERENCE,
                                     seed + r);
        p.ev_count_begin = (void *)ev[2 * r];
        p.ev_count_end = (void *)ev[2 * r + 1];
        p.flags |= PVV_FLAG_DEVICE_RNG;
        at::Tensor ws = make_workspace(p, vertex);
        auto out = at::empty({p.B, p.K, 2}, vertex.options());
        ok(pvv_ransac_voting_v3(&p, mask.data_ptr(), vertex.data_ptr<float>(), nullptr, nullptr, ws.data_ptr(),
                                (size_t)ws.numel(), out.data_ptr<float>(), nullptr, nullptr, cur_stream(vertex)),
           "ransac_voting_v3");
        keep.push_back(ws);
        keep.push_back(out);
    }
    TORCH_CHECK(hipStreamSynchronize((hipStream_t)cur_stream(vertices[0])) == hipSuccess, "hipStreamSynchronize failed");
    std::vector<double> ms((size_t)reps);
    for (int64_t r = 0; r < reps; ++r) {
        float t = 0.f;
        TORCH_CHECK(hipEventElapsedTime(&t, ev[2 * r], ev[2 * r + 1]) == hipSuccess, "hipEventElapsedTime failed");
        ms[(size_t)r] = t;
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
    return ms;
}

// `reps` full ransac_voting_layer_v3 calls cycling over the given batches with a HIP event at every stage boundary
// (pvv_problem.ev_marks) -> per call the PVV_N_MARKS - 1 durations in ms: [scan, compact+hyp, count pass, select_refit,
// finalize, (staged only:) first count launch, first prune, 0].  One synchronisation at the end; a measurement aid.
std::vector<std::vector<double>> stage_ms_in_pipeline(std::vector<at::Tensor> masks, std::vector<at::Tensor> vertices,
                                                      int64_t round_hyp_num, double inlier_thresh, int64_t min_num,
                                                      int64_t max_num, int64_t seed, int64_t reps, int64_t count_kernel,
                                                      bool inner_marks, bool estimate, std::vector<at::Tensor> segs,
                                                      std::vector<at::Tensor> means)
{
    // means (optional, with estimate: one [b,vn,2] float32 tensor per vertex field): the keypoints the estimate is taken about --
    // its staged pass orders the chunks by them; empty: zeros (row-major order, the timing of rounds 2-4)
    TORCH_CHECK(means.empty() || (estimate && means.size() == vertices.size()), "means: one per vertex field, estimate only");
    // segs (optional, one [b,c,h,w] float32 tensor per vertex field): time pvv_decode_keypoint_v3 -- the class argmax fused
    // into the mask scan, the int64 mask written out -- instead of pvv_ransac_voting_v3; `masks` may then be empty
    const bool fused = !segs.empty();
    TORCH_CHECK(!vertices.empty() && (fused ? segs.size() == vertices.size() : masks.size() == vertices.size()),
                "need as many masks (or segs) as vertex fields");
    TORCH_CHECK(!(fused && estimate), "segs and estimate are exclusive");
    const c10::DeviceGuard device_guard(vertices[0].device());
    std::vector<hipEvent_t> ev((size_t)PVV_N_MARKS * (size_t)reps);
    for (auto &e : ev) TORCH_CHECK(hipEventCreate(&e) == hipSuccess, "hipEventCreate failed");
    std::vector<at::Tensor> keep;
    for (int64_t r = 0; r < reps; ++r) {
        const at::Tensor &vertex = vertices[r % vertices.size()];
        at::Tensor mask;
        if (fused) {
            const at::Tensor &seg = segs[r % segs.size()];
            TORCH_CHECK(seg.is_cuda() && seg.scalar_type() == at::kFloat && seg.dim() == 4 && seg.size(0) == vertex.size(0) &&
                            seg.size(2) == vertex.size(1) && seg.size(3) == vertex.size(2), "seg must be float32 [b,c,h,w] matching vertex");
            mask = at::empty({seg.size(0), seg.size(2), seg.size(3)}, seg.options().dtype(at::kLong));
        } else {
            mask = masks[r % masks.size()];
        }
        pvv_problem p = make_problem(mask, vertex, round_hyp_num, inlier_thresh, min_num, max_num, PVV_SINGULAR_REFERENCE,
                                     seed + r);
        p.count_kernel = (int32_t)count_kernel;
        if (fused) {
            const at::Tensor &seg = segs[r % segs.size()];
            p.seg_classes = (int32_t)seg.size(1);
            for (int i = 0; i < 4; ++i) p.seg_stride[i] = seg.stride(i);
        }
        std::vector<void *> marks(PVV_N_MARKS);
        for (int i = 0; i < PVV_N_MARKS; ++i) marks[(size_t)i] = (void *)ev[(size_t)PVV_N_MARKS * (size_t)r + (size_t)i];
        // inner_marks = false: no records INSIDE the count pass (each costs ~2 us): its duration is then what rocprofv3 sees
        if (!inner_marks) marks[PVV_MARK_STAGE0] = marks[PVV_MARK_PRUNE0] = nullptr;
        p.ev_marks = marks.data();
        p.flags |= PVV_FLAG_DEVICE_RNG;
        at::Tensor ws = make_workspace(p, vertex);
        auto out = at::empty({p.B, p.K, 2}, vertex.options());
        if (estimate) {      // estimate_voting_distribution_with_mean with round_hyp_num hypotheses in total (mean = zeros: timing only)
            if (means.empty()) out.zero_();
            else {
                const at::Tensor &mn = means[r % means.size()];
                check_dev(mn, "mean", at::kFloat);
                TORCH_CHECK(mn.dim() == 3 && mn.size(0) == p.B && mn.size(1) == p.K && mn.size(2) == 2, "mean must be [b,vn,2]");
                out.copy_(mn);
            }
            auto cov = at::empty({p.B, p.K, 2, 2}, vertex.options());
            ok(pvv_estimate_voting_distribution(&p, mask.data_ptr(), vertex.data_ptr<float>(), nullptr, nullptr, out.data_ptr<float>(),
                                                ws.data_ptr(), (size_t)ws.numel(), cov.data_ptr<float>(), nullptr, nullptr, nullptr,
                                                nullptr, cur_stream(vertex)),
               "estimate_voting_distribution");
            keep.push_back(cov);
        } else if (fused) {
            ok(pvv_decode_keypoint_v3(&p, segs[r % segs.size()].data_ptr<float>(), vertex.data_ptr<float>(), nullptr, nullptr, ws.data_ptr(),
                                      (size_t)ws.numel(), mask.data_ptr<int64_t>(), out.data_ptr<float>(), nullptr, nullptr,
                                      cur_stream(vertex)),
               "decode_keypoint_v3");
            keep.push_back(mask);
        } else {
            ok(pvv_ransac_voting_v3(&p, mask.data_ptr(), vertex.data_ptr<float>(), nullptr, nullptr, ws.data_ptr(),
                                    (size_t)ws.numel(), out.data_ptr<float>(), nullptr, nullptr, cur_stream(vertex)),
               "ransac_voting_v3");
        }
        keep.push_back(ws);
        keep.push_back(out);
    }
    TORCH_CHECK(hipStreamSynchronize((hipStream_t)cur_stream(vertices[0])) == hipSuccess, "hipStreamSynchronize failed");
    std::vector<std::vector<double>> ms((size_t)reps, std::vector<double>(PVV_N_MARKS - 1, 0.0));
    for (int64_t r = 0; r < reps; ++r) {
        hipEvent_t *e = &ev[(size_t)PVV_N_MARKS * (size_t)r];
        auto dt = [&](int a, int b) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, e[a], e[b]) == hipSuccess) return (double)t;
            (void)hipGetLastError();       // a mark that was not recorded (not staged): clear the sticky error, report -1
            return -1.0;
        };
        ms[(size_t)r][0] = dt(PVV_MARK_BEGIN, PVV_MARK_SCAN);
        ms[(size_t)r][1] = dt(PVV_MARK_SCAN, PVV_MARK_COMPACT);
        ms[(size_t)r][2] = dt(PVV_MARK_COMPACT, PVV_MARK_COUNT);
        ms[(size_t)r][3] = estimate ? dt(PVV_MARK_COUNT, PVV_MARK_END) : dt(PVV_MARK_COUNT, PVV_MARK_SELECT);   // estimate: k_covariance
        ms[(size_t)r][4] = estimate ? 0.0 : dt(PVV_MARK_SELECT, PVV_MARK_END);
        ms[(size_t)r][5] = dt(PVV_MARK_COMPACT, PVV_MARK_STAGE0);
        ms[(size_t)r][6] = dt(PVV_MARK_STAGE0, PVV_MARK_PRUNE0);
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
    return ms;
}

// SURVEY 8(d)'s streaming-read microbenchmark: one read-once pass over `buf` (pvv_stream_read_probe); the caller brackets
// it with events
void stream_read_probe(at::Tensor buf, at::Tensor sink)
{
    const c10::DeviceGuard device_guard(buf.device());
    TORCH_CHECK(buf.is_cuda() && buf.is_contiguous() && sink.is_cuda() && sink.scalar_type() == at::kInt && sink.numel() >= 1,
                "stream_read_probe: contiguous CUDA buffer and an int32 CUDA sink");
    const size_t bytes = ((size_t)buf.numel() * buf.element_size()) & ~(size_t)15;
    ok(pvv_stream_read_probe(buf.data_ptr(), bytes, (uint32_t *)sink.data_ptr<int32_t>(), cur_stream(buf)), "stream_read_probe");
}

// Re-run only the inlier-count kernel on the state a previous ransac_voting_v3 call left in `ws`
// (bench.py brackets this with HIP events to get the dominant kernel's duration).
void rerun_count_kernel(at::Tensor mask, at::Tensor vertex, int64_t hn, double inlier_thresh, int64_t min_num,
                        int64_t max_num, at::Tensor ws, bool zero_counts, int64_t count_kernel, std::optional<int64_t> cap, bool device_rng)
{
    const c10::DeviceGuard device_guard(vertex.device());   // launch on the tensors' GPU, whatever the current device is
    pvv_problem p = make_problem(mask, vertex, hn, inlier_thresh, min_num, max_num, 0, 0);
    p.count_kernel = (int32_t)count_kernel;
    // the workspace offsets depend on the flags the call that made `ws` carried: ransac_voting_v3 promises the device RNG
    // (PVV_FLAG_DEVICE_RNG: no draw storage) exactly when neither idxs nor selection was injected -- say so here
    if (device_rng) p.flags |= PVV_FLAG_DEVICE_RNG;
    // ... and on cap: a workspace made with ransac_voting_v3(..., cap=) needs the same value here
    if (cap.has_value()) {
        TORCH_CHECK(*cap >= 1 && *cap <= (int64_t)p.H * p.W, "cap must be in [1, H*W]");
        p.cap = (int32_t)*cap;
    }
    check_dev(ws, "workspace", at::kByte);
    // ransac_voting_v3 sizes its workspace to exactly pvv_workspace_bytes of ITS problem: a different total here means different
    // flags / cap / hn, i.e. every offset behind the tile lists would be shifted and the kernel would count on garbage.
    // (The staged pass's leader words and miss counters come LAST in the layout, so a workspace made under AUTO / STAGED serves a
    // re-run of the full pass: the two totals that differ only in that tail are both accepted.)
    pvv_problem p_auto = p;
    p_auto.count_kernel = PVV_COUNT_AUTO;
    TORCH_CHECK((size_t)ws.numel() == pvv_workspace_bytes(&p) || (size_t)ws.numel() == pvv_workspace_bytes(&p_auto),
                "rerun_count_kernel: the workspace holds ", ws.numel(), " bytes but this problem lays out ", pvv_workspace_bytes(&p),
                " (", pvv_workspace_bytes(&p_auto), " with the staged pass's tail): pass device_rng=False if the producing call injected idxs "
                "or selection, and the same hn / cap");
    ok(pvv_rerun_count_kernel(&p, ws.data_ptr(), (size_t)ws.numel(), zero_counts ? 1 : 0, cur_stream(vertex)),
       "rerun_count_kernel");
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    // the reference's four (ransac_voting.cpp:102-107), same names and docstrings' meaning
    m.def("generate_hypothesis", &generate_hypothesis, "generate hypothesis");
    m.def("voting_for_hypothesis", &voting_for_hypothesis, "voting for hypothesis");
    m.def("generate_hypothesis_vanishing_point", &generate_hypothesis_vanishing_point,
          "generate hypothesis vanishing point");
    m.def("voting_for_hypothesis_vanishing_point", &voting_for_hypothesis_vanishing_point,
          "voting for hypothesis vanishing point");
    // fused / batched additions
    m.def("count_inliers", &count_inliers, "fused vote + count -> [hn,vn] int32");
    // first_image: index of image 0 in the caller's larger batch (device RNG key), so that a batch split over several
    // calls draws the same numbers as one call
    namespace py = pybind11;
    m.def("ransac_voting_v3", &ransac_voting_v3, "batched ransac_voting_layer_v3", py::arg("mask"), py::arg("vertex"),
          py::arg("round_hyp_num"), py::arg("inlier_thresh"), py::arg("min_num"), py::arg("max_num"), py::arg("idxs"),
          py::arg("selection"), py::arg("seed"), py::arg("singular_policy"), py::arg("first_image") = 0,
          py::arg("count_kernel") = 0, py::arg("status") = py::none(), py::arg("cap") = py::none(), py::arg("out") = py::none());
    m.def("shutdown", []() { ok(pvv_shutdown(), "pvv_shutdown"); },
          "pvv_shutdown: release the library's per-device side streams, events and pinned stage-hint arrays; forget every hint");
    m.def("estimate_counts_in_stages", [](int64_t B, int64_t H, int64_t W, int64_t K, int64_t hn, int64_t max_num) {
              pvv_problem p;
              memset(&p, 0, sizeof(p));
              p.B = (int32_t)B; p.H = (int32_t)H; p.W = (int32_t)W; p.K = (int32_t)K; p.hn = (int32_t)hn;
              p.mask_elem_size = 8; p.min_num = 5; p.max_num = (int32_t)max_num;
              p.cap = pvv_default_cap(p.H, p.W, p.max_num); p.inlier_thresh = 0.99f;
              return pvv_estimate_counts_in_stages(&p) == 1;
          }, "pvv_estimate_counts_in_stages: would AUTO count an estimate of this size in stages? (host-only)", py::arg("B"), py::arg("H"),
          py::arg("W"), py::arg("K"), py::arg("hn"), py::arg("max_num") = 30000);
    m.def("workspace_bytes", [](int64_t B, int64_t H, int64_t W, int64_t K, int64_t hn, int64_t max_num, int64_t mask_elem_size,
                                int64_t count_kernel, bool device_rng) {
              pvv_problem p;
              memset(&p, 0, sizeof(p));
              p.B = (int32_t)B; p.H = (int32_t)H; p.W = (int32_t)W; p.K = (int32_t)K; p.hn = (int32_t)hn;
              p.mask_elem_size = (int32_t)mask_elem_size; p.min_num = 5; p.max_num = (int32_t)max_num;
              p.cap = pvv_default_cap(p.H, p.W, p.max_num); p.inlier_thresh = 0.99f; p.count_kernel = (int32_t)count_kernel;
              p.flags = device_rng ? PVV_FLAG_DEVICE_RNG : 0;
              return (int64_t)pvv_workspace_bytes(&p);
          }, "pvv_workspace_bytes for a contiguous problem (host-only: no GPU needed)", py::arg("B"), py::arg("H"), py::arg("W"),
          py::arg("K"), py::arg("hn"), py::arg("max_num") = 30000, py::arg("mask_elem_size") = 8, py::arg("count_kernel") = 0,
          py::arg("device_rng") = true);
    m.def("decode_keypoint_v3", &decode_keypoint_v3, "argmax(seg) fused with batched ransac_voting_layer_v3",
          py::arg("seg"), py::arg("vertex"), py::arg("round_hyp_num"), py::arg("inlier_thresh"), py::arg("min_num"),
          py::arg("max_num"), py::arg("idxs"), py::arg("selection"), py::arg("seed"), py::arg("singular_policy"),
          py::arg("first_image") = 0, py::arg("count_kernel") = 0);
    m.def("decode_keypoint_un_pnp", &decode_keypoint_un_pnp,
          "argmax(seg) + ransac_voting_layer_v3 + estimate_voting_distribution_with_mean in one pass (two-class seg)",
          py::arg("seg"), py::arg("vertex"), py::arg("round_hyp_num"), py::arg("hyp_est"), py::arg("inlier_thresh"),
          py::arg("min_num"), py::arg("max_num"), py::arg("idxs"), py::arg("idxs_est"), py::arg("selection"), py::arg("seed"),
          py::arg("singular_policy"), py::arg("first_image") = 0, py::arg("count_kernel") = 0);
    m.def("estimate_voting_distribution", &estimate_voting_distribution,
          "batched estimate_voting_distribution_with_mean", py::arg("mask"), py::arg("vertex"), py::arg("mean"),
          py::arg("hyp_total"), py::arg("inlier_thresh"), py::arg("min_num"), py::arg("max_num"), py::arg("idxs"),
          py::arg("selection"), py::arg("seed"), py::arg("want_hyp"), py::arg("first_image") = 0, py::arg("count_kernel") = 0);
    m.def("rerun_count_kernel", &rerun_count_kernel, "re-launch the inlier-count pass (profiling aid)", py::arg("mask"),
          py::arg("vertex"), py::arg("hn"), py::arg("inlier_thresh"), py::arg("min_num"), py::arg("max_num"), py::arg("ws"),
          py::arg("zero_counts"), py::arg("count_kernel") = 0, py::arg("cap") = py::none(), py::arg("device_rng") = true);
    m.def("stage_ms_in_pipeline", &stage_ms_in_pipeline,
          "per-stage durations inside full v3 calls, HIP events at the stage boundaries (profiling aid)", py::arg("masks"),
          py::arg("vertices"), py::arg("round_hyp_num"), py::arg("inlier_thresh"), py::arg("min_num"), py::arg("max_num"),
          py::arg("seed"), py::arg("reps"), py::arg("count_kernel") = 0, py::arg("inner_marks") = true, py::arg("estimate") = false,
          py::arg("segs") = std::vector<at::Tensor>(), py::arg("means") = std::vector<at::Tensor>());
    m.def("stage_hint", [](at::Tensor mask, at::Tensor vertex, int64_t hn) {
              pvv_problem p = make_problem(mask, vertex, hn, 0.99, 5, 30000, 0, 0);
              float mean = -1.f, thr = -1.f;
              const int valid = pvv_stage_hint_query(&mean, &thr, &p, cur_stream(vertex));
              return std::make_tuple(valid != 0, (double)mean, (double)thr);
          }, "pvv_stage_hint_query for this problem shape -> (data there, mean winner ratio of the last calls, AUTO's threshold)",
          py::arg("mask"), py::arg("vertex"), py::arg("hn"));
    m.def("stream_read_probe", &stream_read_probe, "one read-once streaming pass over a buffer (bench aid)");
    m.def("count_kernel_ms_in_pipeline", &count_kernel_ms_in_pipeline,
          "duration of the inlier-count kernel inside full v3 calls, HIP events around its launch (profiling aid)",
          py::arg("masks"), py::arg("vertices"), py::arg("round_hyp_num"), py::arg("inlier_thresh"), py::arg("min_num"),
          py::arg("max_num"), py::arg("seed"), py::arg("reps"));
    m.attr("abi_version") = pvv_abi_version();
    m.attr("SINGULAR_REFERENCE") = (int)PVV_SINGULAR_REFERENCE;
    m.attr("SINGULAR_ZERO") = (int)PVV_SINGULAR_ZERO;
    m.attr("SINGULAR_IMAGE_ZERO") = (int)PVV_SINGULAR_IMAGE_ZERO;
    m.attr("COUNT_AUTO") = (int)PVV_COUNT_AUTO;
    m.attr("COUNT_EXACT") = (int)PVV_COUNT_EXACT;
    m.attr("COUNT_FULL") = (int)PVV_COUNT_FULL;
    m.attr("COUNT_STAGED") = (int)PVV_COUNT_STAGED;
    m.attr("COUNT_STAGED_ESTIMATE") = (int)PVV_COUNT_STAGED_ESTIMATE;
    m.attr("STATUS_SKIPPED") = (int)PVV_STATUS_SKIPPED;
    m.attr("STATUS_SUBSAMPLED") = (int)PVV_STATUS_SUBSAMPLED;
    m.attr("STATUS_TRUNCATED") = (int)PVV_STATUS_TRUNCATED;
}
