// Probe (r06): does v_exp_f32 (a "quarter-rate" transcendental) occupy the SIMD's vector ALU for its 16 clk, or a pipe of its own that
// ordinary VALU instructions can issue beside?  Decides whether computing part of the attention kernel's exponentials as a
// polynomial on packed FMAs can gain anything: per 2 results a degree-3 polynomial with magic-number
// range reduction costs 8 full-rate instructions = 16 clk per result - exactly what v_exp_f32 costs if it blocks the VALU.
// Straight-line blocks, independent chains, 1 / 2 / 3 waves per SIMD:
//   E: 4 v_exp_f32          F: 12 v_fma_f32          M: 4 v_exp_f32 interleaved with 12 v_fma_f32         P: 4 v_exp + 6 v_pk_fma_f32
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void __launch_bounds__(1024) probe(float* out, int iters, long long* clk) {
    float e0 = threadIdx.x * 1e-3f, e1 = e0 + 0.1f, e2 = e0 + 0.2f, e3 = e0 + 0.3f;
    float f0 = e0, f1 = e1, f2 = e2, f3 = e3, b = 0.999f, c = 0.001f;
    typedef float f2_t __attribute__((ext_vector_type(2)));
    f2_t p0 = {e0, e1}, p1 = {e2, e3}, pb = {b, b}, pc = {c, c};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) asm volatile(".rept 256\n v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n .endr" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3));
        if constexpr (MODE == 1) asm volatile(".rept 256\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                                              " v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                                              " v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n .endr"
                                              : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b), "v"(c));
        if constexpr (MODE == 2) asm volatile(".rept 256\n v_exp_f32 %0, %0\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n"
                                              " v_exp_f32 %1, %1\n v_fma_f32 %7, %7, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n"
                                              " v_exp_f32 %2, %2\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n v_fma_f32 %4, %4, %8, %9\n"
                                              " v_exp_f32 %3, %3\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n .endr"
                                              : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b), "v"(c));
        if constexpr (MODE == 3) asm volatile(".rept 256\n v_exp_f32 %0, %0\n v_pk_fma_f32 %4, %4, %6, %7\n v_exp_f32 %1, %1\n v_pk_fma_f32 %5, %5, %6, %7\n v_pk_fma_f32 %4, %4, %6, %7\n"
                                              " v_exp_f32 %2, %2\n v_pk_fma_f32 %5, %5, %6, %7\n v_exp_f32 %3, %3\n v_pk_fma_f32 %4, %4, %6, %7\n v_pk_fma_f32 %5, %5, %6, %7\n .endr"
                                              : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(p0), "+v"(p1) : "v"(pb), "v"(pc));
        if constexpr (MODE == 4) asm volatile(".rept 256\n v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n"
                                              " v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n .endr" : "+v"(p0), "+v"(p1) : "v"(pb), "v"(pc));
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    if (e0 + e1 + e2 + e3 + f0 + f1 + f2 + f3 + p0[0] + p1[1] == 123.456f) out[0] = e0;
}
template <int MODE> void run(const char* what, int n_instr, int waves, long long* clk, float* out) {
    const int iters = 16;
    for (int rep = 0; rep < 3; ++rep) probe<MODE><<<256, 64 * waves>>>(out, iters, clk);
    hipDeviceSynchronize();
    long long h[256 * 16]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 256; ++i) for (int w = 0; w < waves; ++w) s += h[i * 16 + w];
    const double per_block = s / 256 / waves / ((double)iters * 256);
    printf("%-44s %2d waves per CU (%d per SIMD): %7.1f shader clk per block of %2d instructions per wave = %5.2f clk per instruction\n",
           what, waves, waves / 4, per_block, n_instr, per_block / n_instr);
}
int main() {
    long long* clk; float* out; hipMalloc(&clk, 256 * 16 * 8); hipMalloc(&out, 64);
    for (int waves : {4, 8, 12}) {
        run<0>("E: 4 v_exp_f32", 4, waves, clk, out);
        run<1>("F: 12 v_fma_f32", 12, waves, clk, out);
        run<2>("M: 4 v_exp_f32 + 12 v_fma_f32 interleaved", 16, waves, clk, out);
        run<4>("K: 6 v_pk_fma_f32", 6, waves, clk, out);
        run<3>("P: 4 v_exp_f32 + 6 v_pk_fma_f32 interleaved", 10, waves, clk, out);
    }
    return 0;
}
