// libaa_mi355.so: HIP launch layer for the C ABI in include/aa_mi355.h (gfx950 only).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC (see animate_anything_amd/build.py).
#include <hip/hip_runtime.h>

namespace aa {
// Kernels above 64 KiB of dynamic LDS (the 128x128 contraction tile uses 72 KiB of the CU's 160 KiB)
// must opt in once per function.
// The attribute is per (device, function): the cache is keyed on both (one process may drive several GPUs), and a
// failure to raise the limit is reported through the launch error path instead of being dropped.
static thread_local hipError_t g_lds_error = hipSuccess;
template <typename K>
static void ensure_lds(K kernel, size_t lds_bytes) {
    struct Done { const void* fn; int dev; };
    static thread_local Done done[1024];
    static thread_local int n_done = 0;
    if (lds_bytes <= 64 * 1024) return;
    const void* key = reinterpret_cast<const void*>(kernel);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    for (int i = 0; i < n_done; ++i) if (done[i].fn == key && done[i].dev == dev) return;
    const hipError_t e = hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { g_lds_error = e; return; }
    if (n_done < 1024) done[n_done++] = Done{key, dev};
}
}  // namespace aa

#define AA_LAUNCH(kernel, grid, block, lds, stream, ...)                                        \
    do {                                                                                        \
        aa::ensure_lds(kernel, (size_t)(lds));                                                  \
        hipLaunchKernelGGL(kernel, grid, block, (size_t)(lds), (hipStream_t)(stream), __VA_ARGS__); \
    } while (0)

static const char* aa_post_launch() {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && aa::g_lds_error != hipSuccess) { e = aa::g_lds_error; aa::g_lds_error = hipSuccess; }
    return e == hipSuccess ? nullptr : hipGetErrorString(e);
}
#define AA_POST_LAUNCH() aa_post_launch()


#include "aa_api_impl.h"
