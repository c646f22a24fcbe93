// One group of the contraction tile table for the emulator build (the host-side twin of csrc/aa_tiles.hip): build_emu.py compiles
// this file once per group, in parallel with aa_api_emu.cpp (a single unit took 13 minutes).  TEST INFRASTRUCTURE ONLY.
#include "dev.h"

#define AA_LAUNCH(kernel, grid, block, lds, stream, ...) \
    emu::launch(grid, block, (size_t)(lds), [=]() { kernel(__VA_ARGS__); })
#define AA_POST_LAUNCH() ((const char*)nullptr)
#define AA_TU_TILES_ONLY
#include "aa_api_impl.h"

template bool aa::cg_launch_cfg_group<f16_t, AA_TU_GROUP>(int, const AaConvGemm&, int, int, void*, int);
template bool aa::cg_launch_cfg_group<bf16_t, AA_TU_GROUP>(int, const AaConvGemm&, int, int, void*, int);
