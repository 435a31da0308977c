// recip_exact.hip -- is a short reciprocal sequence the correctly rounded 1.0f / x?  Every x with |x| in [2^-20, 2^20] (688 M bit
// patterns) through  A: r1 = fma(fma(-x, r0, 1), r0, r0) with r0 = v_rcp_f32(x);  B: A, then r2 = fma(fma(-x, r1, 1), r1, r1);
// C: div_by(div_prepare(x), 1.0f) (rt_device.h: the division's own sequence), each compared with the compiler's 1.0f / x.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off recip_exact.hip -o recip_exact ; run: ./recip_exact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(unsigned long long *bad, uint32_t *first_bad)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, n = gridDim.x * blockDim.x;
    unsigned long long ba = 0, bb = 0, bc = 0;
    for (uint32_t e = 107; e <= 147; ++e)                                  // biased exponents of 2^-20 .. 2^20
        for (uint32_t m = tid; m < (1u << 23); m += n)
            for (uint32_t s = 0; s < 2; ++s) {
                const uint32_t bits = (s << 31) | (e << 23) | m;
                if (e == 147 && m != 0) continue;                          // |x| <= 2^20
                const float x = __uint_as_float(bits);
                const float want = 1.0f / x;
                const float r0 = __builtin_amdgcn_rcpf(x);
                const float r1 = __builtin_fmaf(__builtin_fmaf(-x, r0, 1.0f), r0, r0);
                const float r2 = __builtin_fmaf(__builtin_fmaf(-x, r1, 1.0f), r1, r1);
                const float q0 = 1.0f * r1;
                const float q1 = __builtin_fmaf(__builtin_fmaf(-x, q0, 1.0f), r1, q0);
                const float q2 = __builtin_fmaf(__builtin_fmaf(-x, q1, 1.0f), r1, q1);
                if (__float_as_uint(r1) != __float_as_uint(want)) { if (!ba) atomicMin(&first_bad[0], bits); ++ba; }
                if (__float_as_uint(r2) != __float_as_uint(want)) { if (!bb) atomicMin(&first_bad[1], bits); ++bb; }
                if (__float_as_uint(q2) != __float_as_uint(want)) { if (!bc) atomicMin(&first_bad[2], bits); ++bc; }
            }
    atomicAdd(&bad[0], ba); atomicAdd(&bad[1], bb); atomicAdd(&bad[2], bc);
}
int main()
{
    unsigned long long *bad; uint32_t *fb;
    if (hipMalloc(&bad, 24) != hipSuccess || hipMalloc(&fb, 12) != hipSuccess) return 1;
    (void)hipMemset(bad, 0, 24); (void)hipMemset(fb, 0xff, 12);
    hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, bad, fb);
    unsigned long long h[3]; uint32_t f[3];
    if (hipMemcpy(h, bad, 24, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    (void)hipMemcpy(f, fb, 12, hipMemcpyDeviceToHost);
    printf("x with |x| in [2^-20, 2^20], against 1.0f / x:  A (rcp + 2 fma) differs for %llu (first 0x%08x)   B (rcp + 4 fma) %llu (0x%08x)   C (div_by, numerator 1) %llu (0x%08x)\n",
           h[0], f[0], h[1], f[1], h[2], f[2]);
    return 0;
}
