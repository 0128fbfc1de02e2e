/* Shim (test infrastructure): the five members of at::Tensor the reference's launchers touch (ransac_voting_kernel.cu:
 * .size(i), .data<T>(), .type(), at::zeros({..}, type)), as a view of raw device memory.  The reference was written
 * against torch 1.1's ATen; today's ATen no longer accepts `at::zeros(sizes, tensor.type())`, and the oracle must not
 * depend on torch anyway. */
#ifndef PVV_REF_SHIM_ATEN_H_
#define PVV_REF_SHIM_ATEN_H_
#include <hip/hip_runtime.h>
#include <cassert>
#include <cstdint>
#include <initializer_list>

namespace at {
struct DeprecatedTypeProperties {
    int elem_size;
};
struct Tensor {
    void *ptr = nullptr;
    int64_t sizes[4] = {0, 0, 0, 0};
    int ndim = 0;
    int elem_size = 4;
    bool owned = false;      // allocated by at::zeros below; the caller of the launcher frees it
    int64_t size(int i) const { return sizes[i]; }
    template <typename T> T *data() const { return (T *)ptr; }
    DeprecatedTypeProperties type() const { return DeprecatedTypeProperties{elem_size}; }
    int64_t numel() const
    {
        int64_t n = 1;
        for (int i = 0; i < ndim; ++i) n *= sizes[i];
        return n;
    }
};
inline Tensor zeros(std::initializer_list<int> shape, DeprecatedTypeProperties t)
{
    Tensor r;
    r.elem_size = t.elem_size;
    for (int s : shape) r.sizes[r.ndim++] = s;
    const size_t bytes = (size_t)r.numel() * r.elem_size;
    if (hipMalloc(&r.ptr, bytes ? bytes : 4) != hipSuccess) return r;
    (void)hipMemset(r.ptr, 0, bytes);
    r.owned = true;
    return r;
}
}  // namespace at
#endif
