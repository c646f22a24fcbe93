// Probe (r03): straight-line VALU code of FOOT KB executed `iters` times by 1 or 2 waves per SIMD: clocks per instruction.
// Question: what do the 60-80 KB contraction kernels (fully unrolled setup / epilogue, each line executed once per workgroup)
// pay for instruction fetch when a single wave per SIMD runs them?
#include <hip/hip_runtime.h>
#include <cstdio>
#ifdef INDEP
#define BLK16KB(dep) asm volatile(".rept 512\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n .endr" : "+v"(dep), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
#else
#define BLK16KB(dep) asm volatile(".rept 2048\n v_fma_f32 %0, %0, %1, %2\n .endr" : "+v"(dep) : "v"(b), "v"(c));
#endif
template <int KB16>
__global__ void __launch_bounds__(512) straight(float* out, int iters, long long* clk) {
    extern __shared__ char lds[];
    float a = threadIdx.x, b = 1.0001f, c = 0.5f, a1 = a + 1, a2 = a + 2, a3 = a + 3;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        BLK16KB(a)
        if constexpr (KB16 > 1) { BLK16KB(a) }
        if constexpr (KB16 > 2) { BLK16KB(a) }
        if constexpr (KB16 > 3) { BLK16KB(a) }
        if constexpr (KB16 > 4) { BLK16KB(a) }
        if constexpr (KB16 > 5) { BLK16KB(a) }
        if constexpr (KB16 > 6) { BLK16KB(a) }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
    if (a + a1 + a2 + a3 == 123.456f) out[0] = a;
}
template <int KB16> void run(int waves, long long* clk, float* out) {
    const int iters = 8;
    hipFuncSetAttribute((const void*)straight<KB16>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int rep = 0; rep < 3; ++rep) straight<KB16><<<256, 64 * waves, 100 * 1024>>>(out, iters, clk);
    hipDeviceSynchronize();
    long long h[256]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 256; ++i) s += h[i];
    const double per = s / 256 / ((double)iters * KB16 * 2048);
    printf("footprint %3d KB, %d waves per CU (%d per SIMD): %6.2f clk per instruction per wave\n", KB16 * 16, waves, waves / 4, per);
}
int main() {
    long long* clk; float* out; hipMalloc(&clk, 4096); hipMalloc(&out, 64);
    for (int waves : {4, 8}) {
        run<1>(waves, clk, out); run<2>(waves, clk, out); run<3>(waves, clk, out); run<4>(waves, clk, out); run<5>(waves, clk, out); run<6>(waves, clk, out); run<7>(waves, clk, out);
    }
    return 0;
}
