// Probe (r03): what does dispatching a contraction-sized grid cost when the workgroups do (almost) nothing?
// 5440 workgroups (GEGLU 139264 x 2560 in 256 x 256 tiles) of 512 / 256 threads with 130 KB / 75 KB of LDS each, body =
// read the kernel arguments + one barrier.  Also: the same grid as a PERSISTENT launch (256 workgroups looping 21.25 times).
#include <hip/hip_runtime.h>
#include <cstdio>
struct Args { const void* p[7]; int v[40]; };
__global__ void __launch_bounds__(512) empty512(Args a, int* sink) {
    extern __shared__ char lds[];
    if (threadIdx.x == 0) lds[0] = (char)a.v[blockIdx.x & 31];
    __syncthreads();
    if (a.v[39] == 12345 && threadIdx.x == 0) sink[blockIdx.x] = lds[0];
}
__global__ void __launch_bounds__(256) empty256(Args a, int* sink) {
    extern __shared__ char lds[];
    if (threadIdx.x == 0) lds[0] = (char)a.v[blockIdx.x & 31];
    __syncthreads();
    if (a.v[39] == 12345 && threadIdx.x == 0) sink[blockIdx.x] = lds[0];
}
__global__ void __launch_bounds__(512) persistent512(Args a, int* sink, int tiles) {
    extern __shared__ char lds[];
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        if (threadIdx.x == 0) lds[0] = (char)a.v[t & 31];
        __syncthreads();
        if (a.v[39] == 12345 && threadIdx.x == 0) sink[t] = lds[0];
    }
}
int main() {
    int* sink; hipMalloc(&sink, 1 << 20);
    Args a = {};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)empty512, hipFuncAttributeMaxDynamicSharedMemorySize, 133120);
    hipFuncSetAttribute((const void*)empty256, hipFuncAttributeMaxDynamicSharedMemorySize, 75776);
    hipFuncSetAttribute((const void*)persistent512, hipFuncAttributeMaxDynamicSharedMemorySize, 133120);
    const int tiles = 5440;
    for (int mode = 0; mode < 4; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) empty512<<<tiles, 512, 133120>>>(a, sink);
            if (mode == 1) empty256<<<tiles, 256, 75776>>>(a, sink);
            if (mode == 2) persistent512<<<256, 512, 133120>>>(a, sink, tiles);
            if (mode == 3) empty512<<<544, 512, 133120>>>(a, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        const char* n[4] = {"5440 workgroups x 512 threads, 130 KB LDS (1 per CU)", "5440 workgroups x 256 threads, 74 KB LDS (2 per CU)",
                            "256 persistent workgroups looping over 5440 tiles", "544 workgroups x 512 threads (a 139264 x 320 launch)"};
        printf("%-62s %8.1f us\n", n[mode], best * 1e3);
    }
    return 0;
}
