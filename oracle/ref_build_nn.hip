// oracle/_ref, second library: the reference's OWN ADD-S nearest-neighbour search.  TEST INFRASTRUCTURE ONLY.
// #includes /root/reference/lib/csrc/nn/src/nearest_neighborhood.cu where it lies (nothing copied) and compiles it with
// hipcc for gfx950 through oracle/ref_shim/; the wrapper calls the reference's launcher (host pointers, its own
// malloc / copy / kernel / copy / free sequence, nearest_neighborhood.cu:123-163) under a name that cannot collide
// with the product's export of the same symbol.  See ref_build.hip for the rules.
#include "nearest_neighborhood.cu"   // -I /root/reference/lib/csrc/nn/src

extern "C" __attribute__((visibility("default"))) void ref_findNearestPointIdxLauncher(
    float *ref_pts, float *que_pts, int *idxs, int b, int pn1, int pn2, int dim, int exclude_self)
{
    findNearestPointIdxLauncher(ref_pts, que_pts, idxs, b, pn1, pn2, dim, exclude_self);
}
