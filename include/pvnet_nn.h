/*
 * pvnet_nn.h -- C ABI of libpvnet_nn.so: clean-pvnet's brute-force nearest-neighbour index search
 * (lib/csrc/nn/src/nearest_neighborhood.cu, used by the ADD-S metric, lib/evaluators/linemod/pvnet.py:74)
 * as a HIP kernel for gfx950.  SURVEY.md section 8(f) rank 4.
 *
 * findNearestPointIdxLauncher keeps the reference's exported name and signature
 * (nearest_neighborhood.cu:123-163, declared in src/ext.h) so its cffi binding (nn_utils.py:5-20) works
 * unchanged: HOST pointers in and out, the library allocates, copies and frees device memory per call like the
 * original.  pvv_nn_find_nearest is the same search on DEVICE pointers without any allocation or copy.
 */
#ifndef PVNET_NN_H_
#define PVNET_NN_H_

#ifdef __cplusplus
extern "C" {
#endif

/* For every query point the index of the nearest reference point (squared Euclidean distance in binary32,
 * first minimum wins: `dist < min_dist`, nearest_neighborhood.cu:75-79).  ref_pts [b,pn1,dim], que_pts
 * [b,pn2,dim], idxs [b,pn2] int32, dim in {2,3}; exclude_self != 0 skips p1i == p2i.  Host pointers.
 * On a HIP error the reference prints and exit()s (gpuErrchk, cuda_common.h:19-26); this one prints the error to
 * stderr and leaves idxs untouched. */
void findNearestPointIdxLauncher(float *ref_pts, float *que_pts, int *idxs, int b, int pn1, int pn2, int dim,
                                 int exclude_self);

/* Same search on device pointers; launches on `stream` (hipStream_t as void*), returns 0 or a hipError_t / -1. */
int pvv_nn_find_nearest(const float *d_ref_pts, const float *d_que_pts, int *d_idxs, int b, int pn1, int pn2,
                        int dim, int exclude_self, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PVNET_NN_H_ */
