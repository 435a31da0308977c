// shader_clock.hip -- the shader clock a gfx950 part actually sustains under a dense fp32 VALU stream: every wave reads the
// shader-clock counter (s_memtime) and the constant 100 MHz counter (s_memrealtime) around the same loop of v_add_f32;
// cycles / microseconds = MHz.  It separates the two readings of "a plain v_add_f32 stream tops out at ~60 T lane-ops/s
// against a nominal 78.6 T" (profiles/r01_valu_slot_costs.txt): a lower clock under load, or issue slots lost at 2.4 GHz.
// Build: hipcc --offload-arch=gfx950 -O3 shader_clock.hip -o shader_clock ; run: ./shader_clock [waves per SIMD, default 8]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define N_ITER 200000
// the same loop over packed fp32 instructions (two floats per lane per instruction, operands in register pairs): does a SIMD issue
// them at the rate of the plain ones (then an f3 sub/mul pair costs half) or at half of it (then nothing is gained)?
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) kp(float *out, unsigned long long *t, float a)
{
    f2 x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
    const f2 aa = {a, a};
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int i = 0; i < N_ITER; ++i) {
#define P0(x) if (MODE == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(aa)); \
              else if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(aa)); \
              else asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(aa));
        P0(x0) P0(x1) P0(x2) P0(x3) P0(x4) P0(x5) P0(x6) P0(x7) P0(x0) P0(x1) P0(x2) P0(x3) P0(x4) P0(x5) P0(x6) P0(x7)
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        t[2 * w] = c1 - c0; t[2 * w + 1] = r1 - r0;
    }
    const f2 s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}
__global__ void __launch_bounds__(256) k(float *out, unsigned long long *t, float a)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int i = 0; i < N_ITER; ++i) {
#define S0(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(a));
        S0(x0) S0(x1) S0(x2) S0(x3) S0(x4) S0(x5) S0(x6) S0(x7) S0(x0) S0(x1) S0(x2) S0(x3) S0(x4) S0(x5) S0(x6) S0(x7)
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        t[2 * w] = c1 - c0; t[2 * w + 1] = r1 - r0;
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

int main(int argc, char **argv)
{
    const int per_simd = argc > 1 ? atoi(argv[1]) : 8;
    const int mode = argc > 2 ? atoi(argv[2]) : 0;          // 0 v_add_f32, 1 v_pk_add_f32, 2 v_pk_mul_f32, 3 v_pk_fma_f32
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, grid = cus * per_simd;     // 256 threads = 4 waves = one per SIMD; per_simd workgroups per CU
    float *out; unsigned long long *t;
    hipMalloc(&out, (size_t)grid * 256 * 4); hipMalloc(&t, (size_t)grid * 4 * 16);
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, t, 1.0f);
        else if (mode == 1) hipLaunchKernelGGL(kp<1>, dim3(grid), dim3(256), 0, 0, out, t, 1.0f);
        else if (mode == 2) hipLaunchKernelGGL(kp<2>, dim3(grid), dim3(256), 0, 0, out, t, 1.0f);
        else hipLaunchKernelGGL(kp<3>, dim3(grid), dim3(256), 0, 0, out, t, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h((size_t)grid * 8);
        hipMemcpy(h.data(), t, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> mhz;
        for (size_t w = 0; w < (size_t)grid * 4; ++w) if (h[2 * w + 1]) mhz.push_back((double)h[2 * w] / ((double)h[2 * w + 1] / 100.0));
        std::sort(mhz.begin(), mhz.end());
        const double lane_ops = (double)grid * 256 * 16.0 * N_ITER;
        printf("%d CUs, %d waves/SIMD: kernel %.2f ms, %.1f T lane-ops/s; s_memtime / s_memrealtime per wave: median %.0f MHz (min %.0f, max %.0f); "
               "cycles per wave64 instruction (mode %d) per SIMD at that clock: %.2f\n", cus, per_simd, ms, lane_ops / ms / 1e9, mhz[mhz.size() / 2], mhz.front(), mhz.back(), mode,
               mhz[mhz.size() / 2] * 1e6 * (ms * 1e-3) / (16.0 * N_ITER * per_simd));
    }
    return 0;
}
