// Probe (r03): how fast can ONE CU pull L2-resident operand tiles?  Every workgroup (1 per CU, 256 of them) re-reads its own
// 512 KB slice (stays in L2 / MALL) REP times with
//   mode 0: buffer_load_dwordx4 ... lds   (LDS-DMA, 1 KiB per wave instruction, what the contraction kernels use)
//   mode 1: global_load_dwordx4 -> VGPR   (16 B per lane; data discarded through an asm sink)
//   mode 2: global_load_dwordx4 -> VGPR -> ds_write_b128 (register staging)
// with 4 or 8 waves per workgroup, 8 loads in flight per wave.  Prints bytes / clock / CU (s_memtime) and TB/s over the chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(512) fetch(const char* src, size_t slice, int rep, long long* clk, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const char* base = src + (size_t)blockIdx.x * slice;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)slice, 0x00020000);
    const int pieces = (int)(slice >> 10);                      // 1 KiB pieces
    u32x4 acc = {0, 0, 0, 0};
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rep; ++r) {
        for (int p = wave; p < pieces; p += nw * 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int pc = p + u * nw;
                const unsigned off = (unsigned)pc * 1024u + lane * 16u;
                if (MODE == 0) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + ((wave * 8 + u) & 63) * 1024), 16, off, 0, 0, 0);
                } else {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(base + off);
                    if (MODE == 2) { *reinterpret_cast<u32x4*>(lds + ((wave * 8 + u) & 63) * 1024 + lane * 16) = v; asm volatile("" ::: "memory"); }
                    else { acc[0] ^= v[0]; acc[1] ^= v[1]; acc[2] ^= v[2]; acc[3] ^= v[3]; }
                }
            }
        }
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
    if (acc[0] == 0x12345678u) sink[threadIdx.x] = acc[1] ^ acc[2] ^ acc[3];
}

int main() {
    const size_t slice = (size_t)(getenv("SLICE_KB") ? atoi(getenv("SLICE_KB")) : 64) << 10;
    const int ncu = 256, rep = (int)((20u << 20) / slice);
    char* src; hipMalloc(&src, slice * ncu); hipMemset(src, 1, slice * ncu);
    long long* clk; hipMalloc(&clk, ncu * 8);
    unsigned* sink; hipMalloc(&sink, 4096);
    long long h[256];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"LDS-DMA (buffer_load_dwordx4 lds)", "global_load_dwordx4 -> VGPR", "global_load_dwordx4 -> VGPR -> ds_write_b128"};
    for (int threads = 256; threads <= 512; threads += 256)
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e9f;
            for (int it = 0; it < 3; ++it) {
                hipEventRecord(e0);
                if (mode == 0) fetch<0><<<ncu, threads, 65536>>>(src, slice, rep, clk, sink);
                if (mode == 1) fetch<1><<<ncu, threads, 65536>>>(src, slice, rep, clk, sink);
                if (mode == 2) fetch<2><<<ncu, threads, 65536>>>(src, slice, rep, clk, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            hipMemcpy(h, clk, ncu * 8, hipMemcpyDeviceToHost);
            double mean = 0; for (int i = 0; i < ncu; ++i) mean += (double)h[i]; mean /= ncu;
            const double bytes = (double)slice * rep;
            printf("%d waves/CU  %-46s %6.1f B/clk/CU   %6.2f TB/s chip   (%.0f us)\n", threads / 64, names[mode], bytes / mean, bytes * ncu / (best * 1e-3) / 1e12, best * 1e3);
        }
    return 0;
}
