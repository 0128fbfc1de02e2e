// rsq_accuracy.hip -- v_rsq_f32 on gfx950 against the correctly rounded 1/sqrt(x), EXHAUSTIVELY over every positive normal
// binary32 input: the largest error in units of the last place of the correctly rounded result, how many inputs are not
// correctly rounded, and the largest error of the unit-normal component  nx = fl(dx * rsq(fl(dx^2 + dy^2)))  against the
// true dx / |d| on a dense set of directions -- the number the guard bands of count_bf16.hpp / count_prune.hpp assume
// ("f32 unit normal").
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void k_rsq(unsigned long long *worst /*[0]: max |err| in 1/1024 ulp, [1]: inputs not correctly rounded*/)
{
    unsigned long long bad = 0, w = 0;
    const uint32_t stride = gridDim.x * 256;
    for (uint64_t bits = 0x00800000ull + blockIdx.x * 256 + threadIdx.x; bits < 0x7f800000ull; bits += stride) {
        const float x = __uint_as_float((uint32_t)bits);
        const float r = __builtin_amdgcn_rsqf(x);
        const double t = 1.0 / sqrt((double)x);                      // correctly rounded binary64 ops: error << 1 ulp of binary32
        const float c = (float)t;                                   // correctly rounded binary32 result (up to double rounding, ~2^-29 ulp)
        int e;
        frexp(t, &e);                                               // t = m 2^e, m in [0.5, 1): ulp of binary32 at t = 2^(e-24)
        const double ulp = ldexp(1.0, e - 24);
        const double err = fabs((double)r - t) / ulp;
        const unsigned long long q = (unsigned long long)(err * 1024.0);
        if (q > w) w = q;
        if (r != c) ++bad;
    }
    atomicMax(&worst[0], w);
    atomicAdd(&worst[1], bad);
}

// unit normal components through rsq, directions on a dense grid of (dx, dy) with magnitudes over 12 octaves
__global__ __launch_bounds__(256) void k_normal(unsigned long long *worst /*[2]: max relative error of nx, ny in 1/1024 u, u = 2^-24*/)
{
    unsigned long long w = 0;
    const uint32_t id = blockIdx.x * 256 + threadIdx.x;
    for (uint32_t k = 0; k < 4096; ++k) {
        // a pseudo-random direction and magnitude from (id, k)
        uint32_t s = id * 2654435761u + k * 40503u + 12345u;
        s ^= s >> 15; s *= 2246822519u; s ^= s >> 13; s *= 3266489917u; s ^= s >> 16;
        const float ang = (float)(s & 0xffffff) * (6.2831853f / 16777216.f);
        const float mag = ldexpf(1.f + (float)((s >> 8) & 0xffff) / 65536.f, (int)(s >> 28) - 8);
        const float dx = mag * cosf(ang), dy = mag * sinf(ang);
        const float dd = dx * dx + dy * dy;                        // (contract off: two roundings of the squares, one of the sum)
        const float r = __builtin_amdgcn_rsqf(dd);
        const float nx = dx * r, ny = dy * r;
        const double n = sqrt((double)dx * dx + (double)dy * dy);
        const double tx = dx / n, ty = dy / n;
        // error relative to |component| is unbounded near 0; the band argument needs it relative to 1 (|n| = 1) -- and, for the
        // dominant component, relative to itself.  Report relative to max(|tx|, |ty|) >= 0.707
        const double den = fmax(fabs(tx), fabs(ty));
        const double e = fmax(fabs(nx - tx), fabs(ny - ty)) / den / 5.9604644775390625e-08;
        const unsigned long long q = (unsigned long long)(e * 1024.0);
        if (q > w) w = q;
    }
    atomicMax(&worst[2], w);
}

int main()
{
    unsigned long long *d, h[3] = {0, 0, 0};
    hipMalloc(&d, sizeof(h));
    hipMemset(d, 0, sizeof(h));
    hipLaunchKernelGGL(k_rsq, dim3(256 * 16), dim3(256), 0, 0, d);
    hipLaunchKernelGGL(k_normal, dim3(256 * 16), dim3(256), 0, 0, d);
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("v_rsq_f32, all %u positive normal inputs: max error %.4f ulp of the correctly rounded result; %llu inputs (%.2f %%) not correctly rounded\n",
           0x7f800000u - 0x00800000u, h[0] / 1024.0, h[1], 100.0 * h[1] / (double)(0x7f800000u - 0x00800000u));
    printf("unit normal n = d * rsq(dx^2 + dy^2), 4.3e9 directions: max component error %.3f u (u = 2^-24), relative to the larger component\n", h[2] / 1024.0);
    return 0;
}
