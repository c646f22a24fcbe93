// libaa_mi355.so: the C ABI in include/aa_mi355.h on top of the HIP launch layer (gfx950 only).
// Build: animate_anything_amd/build.py - this unit plus AA_TU_GROUPS x aa_tiles.hip, compiled in parallel with
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC, linked into one shared object.
#include "aa_launch.h"

namespace aa { thread_local hipError_t g_lds_error = hipSuccess; }

static const char* aa_post_launch() {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && aa::g_lds_error != hipSuccess) { e = aa::g_lds_error; aa::g_lds_error = hipSuccess; }
    return e == hipSuccess ? nullptr : hipGetErrorString(e);
}
#define AA_POST_LAUNCH() aa_post_launch()

#include "aa_api_impl.h"
