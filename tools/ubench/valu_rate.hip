// valu_rate.hip -- what one SIMD of gfx950 sustains for plain (non-packed) fp32 VALU work as a function of
// resident waves: cycles per wave64 instruction for v_add_f32 / v_mul_f32 / v_fma_f32 / a mul+add pair /
// v_mad_u64_u32 / v_rcp / f64 fma.  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>

#define N_ITER 2048
template <int OP>
__global__ void k(float *out, unsigned long long *cyc, float a, float b)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < N_ITER; ++i) {
#define R8(S) S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7)
        if (OP == 0) {
#define S0(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(a));
            R8(S0) R8(S0)
        } else if (OP == 1) {
#define S1(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
            R8(S1) R8(S1)
        } else if (OP == 2) {
#define S2(x) asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2" : "+v"(x) : "v"(a), "v"(b));
            R8(S2)
        } else if (OP == 3) {
            unsigned long long p;
#define S3(x) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p) : "v"(__float_as_uint(x)), "v"(0xD2511F53u) : "vcc"); x = __uint_as_float((unsigned)(p >> 32));
            R8(S3) R8(S3)
        } else if (OP == 4) {
#define S4(x) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
            R8(S4) R8(S4)
        } else if (OP == 5) {
            static_assert(true, "");
            double d0 = x0, d1 = x1, d2 = x2, d3 = x3;
#define S5(d) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d) : "v"((double)a), "v"((double)b));
            S5(d0) S5(d1) S5(d2) S5(d3) S5(d0) S5(d1) S5(d2) S5(d3) S5(d0) S5(d1) S5(d2) S5(d3) S5(d0) S5(d1) S5(d2) S5(d3)
            x0 = d0; x1 = d1; x2 = d2; x3 = d3;
        } else if (OP == 6) {   // VALU + SALU interleaved 1:1
#define S6(x) asm volatile("v_add_f32 %0, %0, %1\n s_add_u32 s20, s20, 1" : "+v"(x) : "v"(a) : "s20");
            R8(S6) R8(S6)
        } else if (OP == 8) {   // 12 v_add_f32 + 4 v_fma_f64 interleaved: do the f64 ops ride beside the f32 stream?
            double d0 = x0, d1 = x1, d2 = x2, d3 = x3;
            S0(x4) S0(x5) S0(x6) S5(d0) S0(x7) S0(x4) S0(x5) S5(d1) S0(x6) S0(x7) S0(x4) S5(d2) S0(x5) S0(x6) S0(x7) S5(d3)
            x0 = d0; x1 = d1; x2 = d2; x3 = d3;
        } else if (OP == 9) {   // 12 v_add_f32 alone (the f32 part of case 8)
            S0(x4) S0(x5) S0(x6) S0(x7) S0(x4) S0(x5) S0(x6) S0(x7) S0(x4) S0(x5) S0(x6) S0(x7)
        } else if (OP == 10) {  // 12 v_add_f32 + 4 v_rcp_f32 interleaved
            S0(x4) S0(x5) S0(x6) S4(x0) S0(x7) S0(x4) S0(x5) S4(x1) S0(x6) S0(x7) S0(x4) S4(x2) S0(x5) S0(x6) S0(x7) S4(x3)
        } else if (OP == 11) {  // 12 v_add_f32 + 4 v_mad_u64_u32 interleaved
            unsigned long long p;
            S0(x4) S0(x5) S0(x6) S3(x0) S0(x7) S0(x4) S0(x5) S3(x1) S0(x6) S0(x7) S0(x4) S3(x2) S0(x5) S0(x6) S0(x7) S3(x3)
        } else if (OP >= 20 && OP < 40) {   // 12 v_add_f32 + 4 x one other instruction class, interleaved
            unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1), u2 = __float_as_uint(x2), u3 = __float_as_uint(x3);
#define MIX(A, B, C, D) S0(x4) S0(x5) S0(x6) A S0(x7) S0(x4) S0(x5) B S0(x6) S0(x7) S0(x4) C S0(x5) S0(x6) S0(x7) D
#define T20(u) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(u));
#define T21(u) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u) : "v"(__float_as_uint(a)));
#define T22(u) asm volatile("v_lshl_or_b32 %0, %0, 1, 1" : "+v"(u));
#define T23(u) asm volatile("v_cmp_eq_u32 vcc, %1, %0\n v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(u) : "v"(__float_as_uint(b)) : "vcc");
#define T24(u) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u) : "v"(__float_as_uint(a)) : "vcc");
#define T25(u) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(u), "v"(a) : "vcc");
#define T26(u) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(u) : "v"(__float_as_uint(a)), "v"(__float_as_uint(b)));
#define T27(u) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(u) : "v"(a));
#define T28(u) asm volatile("v_sqrt_f32 %0, %0" : "+v"(u));
#define T29(u) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(dd) : "v"(u)); u += (unsigned)(unsigned long long)__double_as_longlong(dd);
#define T30(u) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(u) : "v"(a), "v"(b));
#define T31(u) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(u) : "v"(a) : "vcc");
#define T32(u) asm volatile("v_mov_b32 %0, %1" : "=v"(u) : "v"(__float_as_uint(a)));
#define T33(u) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(u) : "v"(a), "v"(b));
#define T34(u) asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(((unsigned long long)u << 32) | u), "v"(k64) : "vcc");
#define T35(u) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_eq_f32 s[20:21], %0, %1" : : "v"(u), "v"(a) : "vcc", "s20", "s21");
            double dd; const unsigned long long k64 = ((unsigned long long)__float_as_uint(a) << 32) | __float_as_uint(b);
            if (OP == 20) { MIX(T20(u0), T20(u1), T20(u2), T20(u3)) }
            if (OP == 21) { MIX(T21(u0), T21(u1), T21(u2), T21(u3)) }
            if (OP == 22) { MIX(T22(u0), T22(u1), T22(u2), T22(u3)) }
            if (OP == 23) { MIX(T23(u0), T23(u1), T23(u2), T23(u3)) }
            if (OP == 24) { MIX(T24(u0), T24(u1), T24(u2), T24(u3)) }
            if (OP == 25) { MIX(T25(u0), T25(u1), T25(u2), T25(u3)) }
            if (OP == 26) { MIX(T26(u0), T26(u1), T26(u2), T26(u3)) }
            if (OP == 27) { MIX(T27(u0), T27(u1), T27(u2), T27(u3)) }
            if (OP == 28) { MIX(T28(u0), T28(u1), T28(u2), T28(u3)) }
            if (OP == 29) { MIX(T29(u0), T29(u1), T29(u2), T29(u3)) }
            if (OP == 30) { MIX(T30(u0), T30(u1), T30(u2), T30(u3)) }
            if (OP == 31) { MIX(T31(u0), T31(u1), T31(u2), T31(u3)) }
            if (OP == 32) { MIX(T32(u0), T32(u1), T32(u2), T32(u3)) }
            if (OP == 33) { MIX(T33(u0), T33(u1), T33(u2), T33(u3)) }
            if (OP == 34) { MIX(T34(u0), T34(u1), T34(u2), T34(u3)) }
            if (OP == 35) { MIX(T35(u0), T35(u1), T35(u2), T35(u3)) }
            x0 = __uint_as_float(u0); x1 = __uint_as_float(u1); x2 = __uint_as_float(u2); x3 = __uint_as_float(u3);
        } else if (OP == 7) {
#define S7(x) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p2##x) : "v"(pa));
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p2x0 = { x0, x1 }, p2x1 = { x1, x2 }, p2x2 = { x2, x3 }, p2x3 = { x3, x4 }, p2x4 = { x4, x5 }, p2x5 = { x5, x6 }, p2x6 = { x6, x7 }, p2x7 = { x7, x0 }, pa = { a, b };
            for (int j = 0; j < 1; ++j) { R8(S7) R8(S7) }
            x0 = p2x0.x + p2x0.y; x1 = p2x1.x; x2 = p2x2.y; x3 = p2x3.x; x4 = p2x4.y; x5 = p2x5.x; x6 = p2x6.x; x7 = p2x7.y;
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP> void run(const char *name, int per_iter)
{
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 8 * 256 * 4 * sizeof(float)); hipMalloc(&cyc, 8192 * 8);
    for (int wps : { 1, 2, 3, 4, 6, 8 }) {              // waves per SIMD: blocks of 256 threads (4 waves -> 1 per SIMD), wps blocks per CU
        int blocks = 256 * wps;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0001f, 0.5f);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0001f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks); hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= blocks;
        double instr_per_wave = (double)N_ITER * per_iter;
        // cycles per wave-instruction per SIMD = wave cycles / (instr per wave * waves per SIMD)
        printf("%-22s waves/SIMD %d: %7.2f cyc/instr per wave, %6.2f cyc/instr per SIMD, wall %.3f ms -> %.2f Tinstr-lanes/s\n", name, wps,
               avg / instr_per_wave, avg / instr_per_wave / wps, ms, instr_per_wave * 64.0 * blocks * 4 / (ms * 1e-3) / 1e12);
    }
}

int main(int argc, char **argv)
{
    if (argc > 1 && atoi(argv[1]) == 0) { run<0>("v_add_f32", 16); run<1>("v_fma_f32", 16); run<2>("v_mul+v_add (dep)", 16); run<3>("v_mad_u64_u32", 16); run<4>("v_rcp_f32", 16); }
    if (argc > 1 && atoi(argv[1]) == 5) run<5>("v_fma_f64", 16);
    if (argc > 1 && atoi(argv[1]) == 8) { run<9>("12 v_add_f32", 12); run<8>("12 v_add_f32 + 4 v_fma_f64", 16); run<10>("12 v_add_f32 + 4 v_rcp_f32", 16); run<11>("12 v_add_f32 + 4 v_mad_u64", 16); }
    if (argc > 1 && atoi(argv[1]) == 20) {
        run<9>("12 v_add_f32", 12); run<20>("+4 v_cvt_f32_u32", 16); run<21>("+4 v_xor_b32", 16); run<22>("+4 v_lshl_or_b32", 16); run<23>("+4 (v_cmp_eq_u32; v_addc)", 20);
        run<24>("+4 v_cndmask_b32", 16); run<25>("+4 v_cmp_lt_f32", 16); run<26>("+4 v_bitop3_b32", 16); run<27>("+4 v_mul_f32", 16); run<28>("+4 v_sqrt_f32", 16);
        run<29>("+4 v_cvt_f64_u32 (+add)", 16); run<30>("+4 v_div_fixup_f32", 16); run<31>("+4 v_div_scale_f32", 16); run<32>("+4 v_mov_b32", 16); run<33>("+4 v_max3_f32", 16);
    }
    if (argc > 1 && atoi(argv[1]) == 34) { run<9>("12 v_add_f32", 12); run<34>("+4 v_cmp_lt_u64", 16); run<35>("+4 (v_cmp_lt_f32; v_cmp_eq_f32)", 20); run<25>("+4 v_cmp_lt_f32", 16); }
    if (argc > 1 && atoi(argv[1]) == 6) run<6>("v_add_f32 + s_add", 32);
    if (argc > 1 && atoi(argv[1]) == 7) run<7>("v_pk_add_f32", 16);
    return 0;
}
