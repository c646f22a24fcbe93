// Probe (r03): what does the epilogue's STORE PATTERN cost?  Writes a [M][N] fp16 matrix (no compute) with
//   0: the contraction epilogue's pattern - lane (ec = lane&31, eh = lane>>5) owns row ec of a 32x32 block and writes its
//      columns 16*eh .. +15 as two 16-byte stores (two separate instructions): every store instruction touches 32 rows,
//      two 16-byte chunks 32 bytes apart per row;
//   1: same ownership after a permlane32-swap style exchange: the pair (ec,0),(ec,1) writes 32 contiguous bytes per row per
//      instruction;
//   2: fully coalesced: consecutive lanes write consecutive 16-byte chunks (8 lanes = one 128-byte line), as after an LDS
//      transpose of the wave's tile;
//   3: pattern 0 with nontemporal stores.
// Tiling mirrors the kernel: 256-thread workgroups own 256 x 256 tiles, wave (wm, wn) a 128 x 128 quarter, blocks (i, j).
// Build: hipcc --offload-arch=gfx950 -O3 scripts/probe/store_pattern.hip -o gpurun_out/store_pattern ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256) store_kernel(unsigned short* out, int M, int N, int tiles_n) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
    const int m0 = tile_m * 256 + wm * 128, n0 = tile_n * 256 + wn * 128;
    const int ec = lane & 31, eh = lane >> 5;
    const u32x4 v = {0x3c003c00u + lane, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u + blockIdx.x};
    if (MODE == 2) {
        // wave tile 128 x 128 fp16 = 128 rows x 256 bytes: one instruction = 4 rows x 256 bytes
#pragma unroll 4
        for (int it = 0; it < 32; ++it) {
            const int row = m0 + it * 4 + (lane >> 4), col = n0 + (lane & 15) * 8;
            if (row < M) *reinterpret_cast<u32x4*>(out + (size_t)row * N + col) = v;
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + i * 32 + ec;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                int col;
                if (MODE == 1) col = n0 + j * 32 + 16 * q + 8 * eh;      // pair writes 32 contiguous bytes
                else col = n0 + j * 32 + 16 * eh + 8 * q;
                if (m < M) {
                    u32x4* dst = reinterpret_cast<u32x4*>(out + (size_t)m * N + col);
                    if (MODE == 3) __builtin_nontemporal_store(v, dst); else *dst = v;
                }
            }
    }
}

int main() {
    const int M = 139264, N = 1280;
    unsigned short* out;
    hipMalloc(&out, (size_t)M * N * 2);
    const int tiles_n = N / 256, tiles = (M / 256) * tiles_n;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[4] = {"row-per-lane, 2 x 16 B 32 B apart (epilogue today)", "pair writes 32 contiguous bytes", "fully coalesced 128-byte lines", "epilogue pattern, nontemporal"};
    for (int mode = 0; mode < 4; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) store_kernel<0><<<tiles, 256>>>(out, M, N, tiles_n);
            if (mode == 1) store_kernel<1><<<tiles, 256>>>(out, M, N, tiles_n);
            if (mode == 2) store_kernel<2><<<tiles, 256>>>(out, M, N, tiles_n);
            if (mode == 3) store_kernel<3><<<tiles, 256>>>(out, M, N, tiles_n);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        printf("mode %d  %-52s %8.1f us  %6.2f TB/s\n", mode, names[mode], best * 1e3, (double)M * N * 2 / (best * 1e-3) / 1e12);
    }
    return 0;
}
