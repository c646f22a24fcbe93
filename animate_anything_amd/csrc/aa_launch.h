// HIP launch layer shared by the translation units of libaa_mi355.so (aa_api.hip: the C ABI; aa_tiles.hip: one group of the
// contraction tile table each).
#pragma once
#include <hip/hip_runtime.h>

namespace aa {
// Kernels above 64 KiB of dynamic LDS (the 128x128 contraction tile uses 72 KiB of the CU's 160 KiB)
// must opt in once per function.
// The attribute is per (device, function): the cache is keyed on both (one process may drive several GPUs), and a
// failure to raise the limit is reported through the launch error path instead of being dropped.
extern thread_local hipError_t g_lds_error;          // defined in aa_api.hip, read by its post-launch check
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
