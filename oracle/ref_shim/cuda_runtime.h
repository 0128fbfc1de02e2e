/* Shim (test infrastructure, NOT product code): lets hipcc compile the reference's own
 * lib/csrc/ransac_voting/src/ransac_voting_kernel.cu, unmodified and where it lies under /root/reference, into
 * oracle/_ref/ -- the handful of CUDA runtime names that file, cuda_common.h and lib/csrc/nn/src/nearest_neighborhood.cu use, spelled with their HIP
 * equivalents.  See oracle/ref_build.hip and oracle/Makefile (target _ref). */
#ifndef PVV_REF_SHIM_CUDA_RUNTIME_H_
#define PVV_REF_SHIM_CUDA_RUNTIME_H_
#include <hip/hip_runtime.h>
typedef hipError_t cudaError_t;
#define cudaSuccess hipSuccess
#define cudaGetErrorString hipGetErrorString
#define cudaGetLastError hipGetLastError
/* lib/csrc/nn/src/nearest_neighborhood.cu:136-161 */
#define cudaMalloc hipMalloc
#define cudaFree hipFree
#define cudaMemcpy hipMemcpy
#define cudaMemcpyHostToDevice hipMemcpyHostToDevice
#define cudaMemcpyDeviceToHost hipMemcpyDeviceToHost
#endif
