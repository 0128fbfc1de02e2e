// oracle/_ref: the reference's OWN kernels as the pin of the oracle.  TEST INFRASTRUCTURE ONLY.
//
// This translation unit #includes /root/reference/lib/csrc/ransac_voting/src/ransac_voting_kernel.cu where it lies
// (nothing of it is copied into this repository) and compiles it with hipcc for gfx950 through the shim headers in
// oracle/ref_shim/ (the CUDA runtime names and the five at::Tensor members that file uses).  What runs is the
// reference's code, unmodified: its four __global__ kernels, launched by its own launchers with its own launch
// shapes (getGPULayout, cuda_common.h:35-55) and its own zero-initialised outputs (at::zeros, kernel.cu:75,255).
// Built with -ffp-contract=off, the arithmetic contract of SURVEY appendix A (one rounding per source-level
// operation); `make -C oracle _ref_fma` builds the same file with contraction allowed, to measure how much nvcc's
// default FMA contraction can change the result.
//
// The extern "C" wrappers below take raw device pointers and synchronise before returning.  Only tests/ load this
// library; it is built by __graft_entry__.build() when /root/reference is present and travels to the GPU box as a
// prebuilt file (oracle/_ref/ is git-ignored, not gpurun-ignored).
#include <cstdint>
#include <cstdio>

#include "ransac_voting_kernel.cu"   // -I /root/reference/lib/csrc/ransac_voting/src

namespace {
at::Tensor view(const void *p, int elem, std::initializer_list<int64_t> shape)
{
    at::Tensor t;
    t.ptr = const_cast<void *>(p);
    t.elem_size = elem;
    for (int64_t s : shape) t.sizes[t.ndim++] = s;
    return t;
}
int finish(at::Tensor produced, void *d_out)
{
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess && produced.ptr && d_out)
        e = hipMemcpy(d_out, produced.ptr, (size_t)produced.numel() * produced.elem_size, hipMemcpyDeviceToDevice);
    if (produced.owned) (void)hipFree(produced.ptr);
    return (int)e;
}
}  // namespace

extern "C" __attribute__((visibility("default"))) int ref_generate_hypothesis(
    const float *d_direct /*[tn,vn,2]*/, const float *d_coords /*[tn,2]*/, const int32_t *d_idxs /*[hn,vn,2]*/,
    float *d_hypo /*[hn,vn,2] out*/, int tn, int vn, int hn)
{
    at::Tensor out = generate_hypothesis_launcher(view(d_direct, 4, {tn, vn, 2}), view(d_coords, 4, {tn, 2}),
                                                  view(d_idxs, 4, {hn, vn, 2}));
    return finish(out, d_hypo);
}

extern "C" __attribute__((visibility("default"))) int ref_voting_for_hypothesis(
    const float *d_direct, const float *d_coords, const float *d_hypo /*[hn,vn,2]*/,
    unsigned char *d_inliers /*[hn,vn,tn], zero-filled by the caller like P:155*/, int tn, int vn, int hn, float thresh)
{
    voting_for_hypothesis_launcher(view(d_direct, 4, {tn, vn, 2}), view(d_coords, 4, {tn, 2}), view(d_hypo, 4, {hn, vn, 2}),
                                   view(d_inliers, 1, {hn, vn, tn}), thresh);
    return finish(at::Tensor(), nullptr);
}

extern "C" __attribute__((visibility("default"))) int ref_generate_hypothesis_vanishing_point(
    const float *d_direct, const float *d_coords, const int32_t *d_idxs, float *d_hypo /*[hn,vn,3] out*/, int tn, int vn,
    int hn)
{
    at::Tensor out = generate_hypothesis_vanishing_point_launcher(view(d_direct, 4, {tn, vn, 2}), view(d_coords, 4, {tn, 2}),
                                                                  view(d_idxs, 4, {hn, vn, 2}));
    return finish(out, d_hypo);
}

extern "C" __attribute__((visibility("default"))) int ref_voting_for_hypothesis_vanishing_point(
    const float *d_direct, const float *d_coords, const float *d_hypo /*[hn,vn,3]*/, unsigned char *d_inliers, int tn, int vn,
    int hn, float thresh)
{
    voting_for_hypothesis_vanishing_point_launcher(view(d_direct, 4, {tn, vn, 2}), view(d_coords, 4, {tn, 2}),
                                                   view(d_hypo, 4, {hn, vn, 3}), view(d_inliers, 1, {hn, vn, tn}), thresh);
    return finish(at::Tensor(), nullptr);
}
