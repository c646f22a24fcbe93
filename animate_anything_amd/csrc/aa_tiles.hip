// One group of the contraction tile table (aa_api_impl.h: cg_launch_cfg_group<T, AA_TU_GROUP>): the kernels of the tiles i with
// i % AA_TU_GROUPS == AA_TU_GROUP and their launchers, for both storage types.  build.py compiles this file once per group.
#include "aa_launch.h"
#define AA_POST_LAUNCH() ((const char*)nullptr)
#define AA_TU_TILES_ONLY
#include "aa_api_impl.h"

template bool aa::cg_launch_cfg_group<f16_t, AA_TU_GROUP>(int, const AaConvGemm&, int, int, void*, int);
template bool aa::cg_launch_cfg_group<bf16_t, AA_TU_GROUP>(int, const AaConvGemm&, int, int, void*, int);
