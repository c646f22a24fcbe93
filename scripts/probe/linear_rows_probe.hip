// Probe (r06): where does aa::linear_rows_kernel (320-channel rows in registers) spend its time at 139264 rows?  Times the kernel's VAR forms and
// workgroup sizes on random data (timing only - parity is tests/test_linear_rows.py through the library):
//   VAR bit 0: no output stores   bit 1: the weight pieces fetch nothing   bit 2: no fragment reads, no MFMAs   bit 3: stores / residual loads a token per lane   bit 4: x straight into registers   bit 5: split workgroups last
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffast-math -I animate_anything_amd/csrc/kernels/device -I animate_anything_amd/csrc/kernels
//        -I animate_anything_amd/csrc -I include scripts/probe/linear_rows_probe.hip -o scripts/probe/bin/linear_rows_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#include "kernels/linear_rows.h"
#include "linear_rows_stream.h"

template <int NW, int VAR>
static float run(const char* what, AaLinearRows d, int split) {
    const int tiles = (int)((d.rows + 32 * NW - 1) / (32 * NW)), nq = d.n_out / 32, slots = 512;
    int n_full = tiles / slots * slots, n_split = 1;
    if (split && tiles - n_full > 0)
        for (int s = 2; s <= nq && (tiles - n_full) * s <= slots; ++s)
            if (nq % s == 0) n_split = s;
    if (n_split == 1) n_full = tiles;
    const dim3 grid(n_full + (tiles - n_full) * n_split), block(64 * NW);
    auto k = aa::linear_rows_kernel<f16_t, 320, NW, VAR>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, aa::lr_lds_bytes(NW));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, grid, block, aa::lr_lds_bytes(NW), 0, d, n_full, n_split);
    std::vector<float> ts;
    for (int i = 0; i < 15; ++i) {
        hipEventRecord(e0); hipLaunchKernelGGL(k, grid, block, aa::lr_lds_bytes(NW), 0, d, n_full, n_split); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms * 1e3f);
    }
    std::sort(ts.begin(), ts.end());
    if (hipGetLastError() != hipSuccess) printf("  !! launch error\n");
    printf("  rows %lld n_out %4d res %d flags>>8 %d | NW %d VAR %3d split %d (grid %5u): %7.1f us  %s\n", (long long)d.rows, d.n_out, d.residual ? 1 : 0, d.flags >> 8, NW, VAR, n_split, grid.x,
           ts[ts.size() / 2], what);
    fflush(stdout);
    return ts[ts.size() / 2];
}

__global__ void fill_halfs(_Float16* p, long long n, unsigned seed) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        unsigned v = (unsigned)i * 2654435761u + seed; v ^= v >> 15; v *= 2246822519u; v ^= v >> 13;
        p[i] = (_Float16)(((int)(v & 1023) - 512) * (1.0f / 1024.0f));
    }
}

template <int DEPTH>
static float run_stream(AaLinearRows d, int grid_wgs) {
    auto k = aa::linear_rows_stream_kernel<f16_t, 320, DEPTH>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, aa::lrs_lds_bytes());
    const dim3 grid(grid_wgs), block(256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, grid, block, aa::lrs_lds_bytes(), 0, d);
    std::vector<float> ts;
    for (int i = 0; i < 15; ++i) {
        hipEventRecord(e0); hipLaunchKernelGGL(k, grid, block, aa::lrs_lds_bytes(), 0, d); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms * 1e3f);
    }
    std::sort(ts.begin(), ts.end());
    if (hipGetLastError() != hipSuccess) printf("  !! launch error\n");
    printf("  rows %lld n_out %4d res 0 | PERSISTENT one workgroup per CU, next tile's x prefetched, %d fragment reads in flight (grid %d): %7.1f us\n", (long long)d.rows, d.n_out, DEPTH, grid_wgs, ts[ts.size() / 2]);
    fflush(stdout);
    return ts[ts.size() / 2];
}

int main() {
    const long long rows = 139264;
    void *x, *res, *out, *w;
    hipMalloc(&x, rows * 320 * 2); hipMalloc(&res, rows * 960 * 2); hipMalloc(&out, rows * 960 * 2); hipMalloc(&w, 30 * AA_LR_STAGE_BYTES);
    void* out2; hipMalloc(&out2, rows * 960 * 2);
    fill_halfs<<<1024, 256>>>((_Float16*)x, rows * 320, 1u); fill_halfs<<<1024, 256>>>((_Float16*)res, rows * 960, 2u);
    fill_halfs<<<64, 256>>>((_Float16*)w, 30LL * AA_LR_STAGE_BYTES / 2, 3u);
    hipDeviceSynchronize();
    for (int pass = 0; pass < 2; ++pass)
        for (int n_out : {320, 960})
            for (int with_res = 1; with_res >= 0; --with_res) {
                AaLinearRows d{};
                d.x = x; d.residual = with_res ? res : nullptr; d.out = out; d.w = w; d.rows = rows; d.channels = 320; d.n_out = n_out;
                d.ldx = 320; d.ld_res = n_out; d.ldo = n_out; d.normalize = 0; d.ln_eps = 1e-5f; d.dtype = AA_F16; d.flags = 0;
                run<4, 0>("the kernel", d, 1);
                if (!with_res) {
                    AaLinearRows d2 = d; d2.out = out2;
                    hipMemset(out2, 0, rows * n_out * 2);
                    run_stream<3>(d2, 256); run_stream<7>(d2, 256);
                    hipDeviceSynchronize();
                    std::vector<unsigned short> a(rows * n_out), b(rows * n_out);
                    hipMemcpy(a.data(), out, a.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(b.data(), out2, b.size() * 2, hipMemcpyDeviceToHost);
                    long long bad = 0; for (size_t i = 0; i < a.size(); ++i) bad += a[i] != b[i];
                    printf("  persistent against the kernel: %lld of %zu output halfs differ\n", bad, a.size());
                }
                run<4, 8>("stores / residual loads a token per lane (16 bytes of another row in every lane)", d, 1);
                run<4, 32>("split workgroups last in the grid", d, 1);
                run<4, 16>("x straight into registers (16 bytes per lane and row)", d, 1);
                run<4, 0>("no stage split of the last round", d, 0);
                run<4, 1>("no output stores", d, 1);
                run<4, 2>("weight pieces fetch nothing", d, 1);
                run<4, 4>("no fragment reads, no MFMAs", d, 1);
                run<4, 6>("no weights, no fragment reads, no MFMAs", d, 1);
                run<4, 7>("x loads (+ residual) and barriers only", d, 1);
                run<4, 23>("x straight into registers: x loads (+ residual) and barriers only", d, 1);
            }
    return 0;
}
