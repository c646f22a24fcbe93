// libaa_mi355.so: HIP launch layer for the C ABI in include/aa_mi355.h (gfx950 only).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC (see animate_anything_amd/build.py).
#include <hip/hip_runtime.h>

namespace aa {
// Kernels above 64 KiB of dynamic LDS (the 128x128 contraction tile uses 72 KiB of the CU's 160 KiB)
// must opt in once per function.
template <typename K>
static void ensure_lds(K kernel, size_t lds_bytes) {
    static thread_local const void* done[256];
    static thread_local int n_done = 0;
    if (lds_bytes <= 64 * 1024) return;
    const void* key = reinterpret_cast<const void*>(kernel);
    for (int i = 0; i < n_done; ++i) if (done[i] == key) return;
    (void)hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (n_done < 256) done[n_done++] = key;
}
}  // namespace aa

#define AA_LAUNCH(kernel, grid, block, lds, stream, ...)                                        \
    do {                                                                                        \
        aa::ensure_lds(kernel, (size_t)(lds));                                                  \
        hipLaunchKernelGGL(kernel, grid, block, (size_t)(lds), (hipStream_t)(stream), __VA_ARGS__); \
    } while (0)

static const char* aa_post_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? nullptr : hipGetErrorString(e);
}
#define AA_POST_LAUNCH() aa_post_launch()

#include "aa_api_impl.h"
