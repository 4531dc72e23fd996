// Micro-benchmark: issue rate of VALU instruction kinds on gfx950 (one wave per SIMD slot set, independent chains): cycles per instruction
// per wave for v_add_u32 (reference), v_mul_lo_u32, v_mul_u32_u24, v_mad_u64_u32, v_mad_u32_u24, v_cvt_f64_f32, v_add_f64, v_fma_f64, v_rcp_f32,
// v_readlane_b32.  Build: hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o tools/micro/valu_rate_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned *out, long long *cyc, int iters) {
    unsigned a = threadIdx.x + 1, b = threadIdx.x * 3 + 7, c = threadIdx.x ^ 5, d = threadIdx.x + 11;
    double fa = a, fb = b, fc = c, fd = d;
    float ga = a, gb = b, gc = c, gd = d;
    unsigned long long la = a, lb = b, lc = c, ld = d;
    const unsigned m = out[0] | 3u;   // unknown to the compiler
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) { REP8(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
        if (KIND == 1) { REP8(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
        if (KIND == 2) { REP8(asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
        if (KIND == 3) { REP8(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3" : "+v"(la), "+v"(lb), "+v"(lc), "+v"(ld) : "v"(m), "v"(a) : "vcc");) }
        if (KIND == 4) { REP8(asm volatile("v_mad_u32_u24 %0, %0, %4, %4\n v_mad_u32_u24 %1, %1, %4, %4\n v_mad_u32_u24 %2, %2, %4, %4\n v_mad_u32_u24 %3, %3, %4, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
        if (KIND == 5) { REP8(asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd) : "v"(ga), "v"(gb), "v"(gc), "v"(gd));) }
        if (KIND == 6) { REP8(asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %1, %1, %2\n v_add_f64 %2, %2, %3\n v_add_f64 %3, %3, %0" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));) }
        if (KIND == 7) { REP8(asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %2, %2, %3, %0\n v_fma_f64 %3, %3, %0, %1" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));) }
        if (KIND == 8) { REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(ga), "+v"(gb), "+v"(gc), "+v"(gd));) }
        if (KIND == 9) { unsigned s0, s1, s2, s3; REP8(asm volatile("v_readlane_b32 %0, %4, 3\n v_readlane_b32 %1, %5, 5\n v_readlane_b32 %2, %6, 7\n v_readlane_b32 %3, %7, 9" : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(a), "v"(b), "v"(c), "v"(d));) a += s0 + s1 + s2 + s3; }
        if (KIND == 10) { REP8(asm volatile("v_mul_f64 %0, %0, %1\n v_mul_f64 %1, %1, %2\n v_mul_f64 %2, %2, %3\n v_mul_f64 %3, %3, %0" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));) }
        if (KIND == 11) { REP8(asm volatile("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7" : "+v"(ga), "+v"(gb), "+v"(gc), "+v"(gd) : "v"(fa), "v"(fb), "v"(fc), "v"(fd));) }
        if (KIND == 12) { REP8(asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3" : "+v"(ga), "+v"(gb), "+v"(gc), "+v"(gd));) }
        if (KIND == 13) { REP8(asm volatile("v_mov_b32_dpp %0, %4 row_ror:4 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %5 row_ror:4 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %6 row_ror:4 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %7 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(ga), "v"(gb), "v"(gc), "v"(gd));) }
        if (KIND == 14) { REP8(asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %2\n v_pk_mul_f32 %2, %2, %3\n v_pk_mul_f32 %3, %3, %0" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));) }
        if (KIND == 15) { REP8(asm volatile("v_div_fmas_f32 %0, %0, %1, %2\n v_div_fmas_f32 %1, %1, %2, %3\n v_div_fmas_f32 %2, %2, %3, %0\n v_div_fmas_f32 %3, %3, %0, %1" : "+v"(ga), "+v"(gb), "+v"(gc), "+v"(gd) :: "vcc");) }
        if (KIND == 16) { REP8(asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %2\n v_div_scale_f32 %1, vcc, %1, %2, %3\n v_div_scale_f32 %2, vcc, %2, %3, %0\n v_div_scale_f32 %3, vcc, %3, %0, %1" : "+v"(ga), "+v"(gb), "+v"(gc), "+v"(gd) :: "vcc");) }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + (unsigned)(fa + fb + fc + fd) + (unsigned)(ga + gb + gc + gd) + (unsigned)(la + lb + lc + ld);
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int KIND> void run(const char *name, unsigned *out, long long *cyc, int wavesPerSimd) {
    const int iters = 2000;
    const int blocks = 256 * wavesPerSimd;   // 256 CUs x (4 waves per block = one per SIMD) x wavesPerSimd
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double ninstr = (double)iters * 32;   // per wave
    printf("%-16s waves/SIMD %d: %.2f shader-clock ticks per instruction per wave (x waves/SIMD = SIMD time per instruction); kernel %.3f ms -> %.2f ns per wave-instruction per SIMD\n",
           name, wavesPerSimd, (double)c / ninstr, ms, ms * 1e6 / (ninstr * wavesPerSimd));
}
int main() {
    unsigned *out; long long *cyc;
    hipMalloc(&out, 4 * 256 * 256 * 8); hipMemset(out, 0, 4 * 256 * 256 * 8); hipMalloc(&cyc, 64);
    for (int w : {1, 4}) {
        run<0>("v_add_u32", out, cyc, w); run<1>("v_mul_lo_u32", out, cyc, w); run<2>("v_mul_u32_u24", out, cyc, w); run<3>("v_mad_u64_u32", out, cyc, w);
        run<4>("v_mad_u32_u24", out, cyc, w); run<5>("v_cvt_f64_f32", out, cyc, w); run<6>("v_add_f64", out, cyc, w); run<7>("v_fma_f64", out, cyc, w);
        run<10>("v_mul_f64", out, cyc, w); run<11>("v_cvt_f32_f64", out, cyc, w); run<8>("v_rcp_f32", out, cyc, w); run<12>("v_sqrt_f32", out, cyc, w);
        run<9>("v_readlane_b32", out, cyc, w); run<13>("v_mov_b32_dpp", out, cyc, w); run<14>("v_pk_mul_f32", out, cyc, w); run<15>("v_div_fmas_f32", out, cyc, w); run<16>("v_div_scale_f32", out, cyc, w);
    }
    return 0;
}
