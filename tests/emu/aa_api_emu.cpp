// CPU build of the C ABI (include/aa_mi355.h) on top of the SIMT emulator in tests/emu/dev.h.
// TEST INFRASTRUCTURE ONLY: same entry points, same kernel sources, host pointers instead of device
// pointers.  Built by tests/emu/build_emu.py with the host clang; never loaded by the product package.
#include "dev.h"

#define AA_LAUNCH(kernel, grid, block, lds, stream, ...) \
    emu::launch(grid, block, (size_t)(lds), [=]() { kernel(__VA_ARGS__); })
#define AA_POST_LAUNCH() ((const char*)nullptr)

#include "aa_api_impl.h"
