// C-ABI entry points of libaa_mi355.so (declared in include/aa_mi355.h): argument validation, tile
// configuration choice and kernel launches.  The translation unit that includes this file defines
//   AA_LAUNCH(kernel, grid, block, lds_bytes, stream, args...)
// (csrc/aa_api.hip: a HIP launch on the caller's stream; tests/emu/aa_api_emu.cpp: the CPU SIMT emulator)
// and AA_POST_LAUNCH() returning a failure string or nullptr.
#pragma once
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include "aa_mi355.h"
#include "kernels/conv_gemm.h"
#include "kernels/conv_gemm_dma.h"
#include "kernels/conv_slab.h"
#include "kernels/conv_gemm_x.h"
#include "kernels/norm.h"
#include "kernels/attention.h"
#include "kernels/seq_attention.h"
#include "kernels/ff_fused.h"
#include "kernels/linear_rows.h"
#include "kernels/glue.h"

namespace aa {

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

static int finish(const char* what) {
    const char* e = AA_POST_LAUNCH();
    if (e) return fail(AA_E_HIP, "%s: %s", what, e);
    return AA_OK;
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Tile configurations of the LDS-DMA contraction kernel.  `rate` is the relative throughput of a tile
// shape once the CUs are full (measured: the L2 -> LDS stream limits the narrow tiles); the chooser
// minimises rounds(tiles / resident slots) * tile work / rate over the shapes that divide the packed width.
// slab: conv_slab.h (3x3 stride-1 halo-slab kernel); x: conv_gemm_x.h; sched: what else tells two entries of one shape apart
// (aa_conv_gemm_tile_flags) - bit 0 staggered, bit 1 spread DMA issue of the compiled kernel, bits 6.. DP3 | DP0 << 4 | DP1 << 8
struct CgCfg { int bm, bn, wm, wn, bk, stages, per_cu; float rate; int slab = 0; int x = 0; int sched = 0; };
constexpr int cg_dp(int dp3, int dp0, int dp1) { return (dp3 | dp0 << 4 | dp1 << 8) << 6; }
static const CgCfg kCgCfgs[] = {
    {128, 64, 2, 2, 64, 2, 3, 0.78f},     // 0
    {128, 128, 2, 2, 64, 2, 2, 0.90f},    // 1
    {192, 256, 3, 2, 64, 2, 1, 1.05f},    // 2
    {256, 256, 4, 2, 64, 2, 1, 1.10f},    // 3
    {256, 320, 4, 2, 64, 2, 1, 1.10f},    // 4
    {192, 320, 3, 2, 64, 2, 1, 1.05f},    // 5
    {256, 320, 4, 2, 32, 4, 1, 1.15f},    // 6  deep ring (3 tiles in flight)
    {256, 256, 4, 2, 32, 4, 1, 1.15f},    // 7
    {128, 128, 2, 2, 32, 4, 2, 0.95f},    // 8
    {128, 64, 2, 2, 32, 4, 3, 0.80f},     // 9
    {192, 320, 3, 2, 32, 4, 1, 1.10f},    // 10
    {128, 320, 2, 2, 32, 2, 2, 1.10f},    // 11 two independent 4-wave workgroups per CU (phases de-synchronise)
    {128, 256, 2, 2, 32, 2, 2, 1.05f},    // 12
    {128, 256, 2, 2, 64, 2, 1, 1.00f},    // 13
    {256, 320, 4, 2, 64, 2, 1, 1.30f, 0, 0, 1},    // 14 staggered: the second wave of every SIMD multiplies before it issues its DMA share
    {256, 256, 4, 2, 64, 2, 1, 1.25f, 0, 0, 1},    // 15 staggered
    {128, 128, 2, 2, 32, 2, 3, 0.90f},    // 16 three small-LDS workgroups per CU
    {128, 64, 2, 2, 32, 2, 4, 0.78f},     // 17 four per CU
    {64, 128, 2, 2, 32, 2, 4, 0.70f},     // 18 short tiles for the small-M levels (more workgroups)
    {64, 64, 2, 2, 32, 2, 6, 0.60f},      // 19
    {64, 256, 2, 2, 32, 2, 3, 0.80f},     // 20
    {256, 320, 4, 2, 32, 4, 1, 1.20f, 0, 0, 1},    // 21 staggered, deep ring
    {256, 256, 4, 2, 32, 4, 1, 1.20f, 0, 0, 1},    // 22 staggered, deep ring
    {192, 256, 2, 4, 64, 2, 1, 1.05f},    // 23 eight waves of 96x64: 230 tiles for the 8704-row (16x16) level = one 90 % round
    {192, 256, 2, 4, 64, 2, 1, 1.05f, 0, 0, 1},    // 24 staggered
    {128, 256, 2, 4, 32, 2, 2, 1.00f},    // 25 eight waves of 64x64, two workgroups per CU
    // "spread": every wave issues its DMA share of the next tile in pieces in front of the k sub-steps of its multiply
    {256, 320, 4, 2, 64, 2, 1, 1.30f, 0, 0, 2},    // 26 spread
    {256, 256, 4, 2, 64, 2, 1, 1.25f, 0, 0, 2},    // 27 spread
    {192, 256, 2, 4, 64, 2, 1, 1.05f, 0, 0, 2},    // 28 spread
    {256, 320, 4, 2, 32, 4, 1, 1.20f, 0, 0, 2},    // 29 spread, deep ring
    {128, 320, 2, 2, 32, 2, 2, 1.10f, 0, 0, 2},    // 30 spread, two workgroups per CU
    {128, 128, 2, 2, 64, 2, 2, 0.90f, 0, 0, 2},    // 31 spread
    {192, 320, 3, 2, 64, 2, 1, 1.05f, 0, 0, 2},    // 32 spread
    {128, 256, 2, 2, 32, 2, 2, 1.05f, 0, 0, 2},    // 33 spread
    // halo-slab 3x3 kernel (conv_slab.h): the activations of a 32-channel unit are staged once for all nine taps
    {256, 320, 4, 2, 32, 3, 1, 1.00f, 1}, // 34
    {256, 256, 4, 2, 32, 3, 1, 1.00f, 1}, // 35
    // hand-placed instruction stream, accumulators in a[0:255] (conv_gemm_x.h): one wave per SIMD, 128x128 / 128x160 per wave
    {256, 256, 2, 2, 64, 2, 1, 1.45f, 0, 1, cg_dp(8, 8, 0)}, // 36 DMA pieces of the tile after next under sub-steps 3 / 0 (8 + 8)
    {256, 320, 2, 2, 64, 2, 1, 1.45f, 0, 1, cg_dp(11, 7, 0)}, // 37 (11 + 7)
    {256, 256, 2, 2, 64, 2, 1, 1.40f, 0, 1, cg_dp(6, 5, 5)}, // 38 pieces spread over sub-steps 3 / 0 / 1 (6 + 5 + 5)
    {256, 320, 2, 2, 64, 2, 1, 1.40f, 0, 1, cg_dp(7, 6, 5)}, // 39 (7 + 6 + 5)
    // the same stream with two waves per SIMD (64x128 per wave)
    {256, 256, 4, 2, 64, 2, 1, 1.35f, 0, 1, cg_dp(2, 2, 2)}, // 40
    // deep ring: four stages of 32 K, three of them in flight, counted waits (conv_gemm_x.h "BK = 32")
    {256, 256, 2, 2, 32, 4, 1, 1.50f, 0, 1, cg_dp(4, 0, 0)}, // 41 one wave per SIMD
    {256, 320, 2, 2, 32, 4, 1, 1.50f, 0, 1, cg_dp(5, 0, 0)}, // 42
    {256, 256, 4, 2, 32, 4, 1, 1.45f, 0, 1, cg_dp(2, 0, 0)}, // 43 two waves per SIMD
    // two 4-wave workgroups per CU (64x128 per wave, a[0:127]), three-slot ring: one multiplies while the other is in its epilogue
    {128, 256, 2, 2, 32, 3, 2, 1.20f, 0, 1, cg_dp(3, 0, 0)}, // 44
    {256, 128, 4, 1, 32, 3, 2, 1.20f, 0, 1, cg_dp(3, 0, 0)}, // 45
    // the shapes the 34 x 16 x 16 and 34 x 8 x 8 levels (8704 / 2176 rows) fill the chip with
    {192, 256, 2, 2, 64, 2, 1, 1.30f, 0, 1, cg_dp(5, 5, 4)}, // 46 one wave per SIMD, 96x128 per wave (a[0:191])
    {128, 128, 2, 2, 64, 2, 2, 1.00f, 0, 1, cg_dp(2, 2, 2)}, // 47 two 4-wave workgroups per CU, 64x64 per wave (a[0:63])
    {128, 128, 2, 2, 32, 4, 2, 1.00f, 0, 1, cg_dp(2, 0, 0)}, // 48 the same on the deep ring
    // 139264 rows = 725.3 tiles of 192: three rounds of 3/4-size tiles and no leftover launch, against two rounds of 256-row tiles
    // plus a leftover launch that costs ~0.7 of a round
    {192, 320, 2, 2, 64, 2, 1, 1.35f, 0, 1, cg_dp(6, 5, 5)}, // 49 one wave per SIMD, 96x160 per wave (a[0:239])
};
constexpr int kNumCgCfgs = sizeof(kCgCfgs) / sizeof(kCgCfgs[0]);

static thread_local int g_tile_override = -2;   // per calling thread.  -2: read AA_FORCE_CFG once; -1: automatic; >= 0: forced index
static int cg_force_cfg() {
    if (g_tile_override == -2) { const char* e = getenv("AA_FORCE_CFG"); g_tile_override = e ? atoi(e) : -1; }
    return g_tile_override;
}

// The halo-slab kernel handles 3x3 / stride 1 / pad 1 convolutions whose tiles are whole image rows of one image.
static bool cg_slab_ok(const AaConvGemm& d, const CgCfg& c) {
    const int hw = d.h_in * d.w_in;
    return d.kh == 3 && d.kw == 3 && d.stride == 1 && d.pad_h == 1 && d.pad_w == 1 && d.h_virt == d.h_in && d.w_virt == d.w_in &&
           d.h_out == d.h_in && d.w_out == d.w_in && d.k_order == 1 && d.c0 % 64 == 0 && d.c1 % 64 == 0 && !d.geglu && !d.bias_per_row &&
           c.bm % d.w_in == 0 && hw % c.bm == 0 && (c.bm / d.w_in + 2) * (d.w_in + 2) <= CS_SLAB_ROWS && !(d.debug & 8);
}

// The hand-scheduled tiles (conv_gemm_x.h) do not carry the nearest-neighbour resize of Upsample2D.
static thread_local unsigned g_x_disabled = 0;   // bit i: table entry 36 + i is not offered (aa_set_tile_override(-100 - mask): bisecting aid)
static bool cg_x_ok(const AaConvGemm& d) {
    if (!(d.h_virt == d.h_in && d.w_virt == d.w_in)) return false;
    {   // their K walk is (64-channel chunk, tap, channel); one validity bit per tap in a 32-bit mask; source offsets are
        // pixel * bytes-per-pixel through v_mad_u32_u24, counted from pad_h rows + pad_w pixels in front of the tensor, bit 31 = "zero"
        const int taps = d.kh * d.kw;
        if (taps > 1 && d.k_order != 1) return false;
        if (taps > 31) return false;
        const int64_t px = (int64_t)d.n_img * d.h_in * d.w_in + (int64_t)d.pad_h * d.w_in + d.pad_w + (int64_t)d.kh * d.w_in + d.kw;
        if (px >= ((int64_t)1 << 24) || px * (d.c0 > d.c1 ? d.c0 : d.c1) * 2 >= ((int64_t)1 << 31)) return false;
    }
    if (d.rowvec) {          // their epilogue reads the row vector through a buffer descriptor: 32-bit byte offsets below 2^31
        const int64_t M = (int64_t)d.n_img * d.h_out * d.w_out;
        const int64_t ld = d.rowvec_ld ? d.rowvec_ld : d.n_out;
        if (((M - 1) / (d.rowvec_div > 0 ? d.rowvec_div : 1) + 1) * ld * 2 >= ((int64_t)1 << 31)) return false;
    }
    return true;
}

// Can table entry i carry out call d at all (the one predicate behind the automatic choice, a forced tile, aa_conv_gemm_tile_ok and
// the tile of a split-off last round - ADVICE r04: the last of these had its own, incomplete copy and handed folded-LayerNorm
// calls to a tile that cannot start its accumulators from the fold's terms).
static bool cg_tile_fits(const AaConvGemm& d, int i) {
    const CgCfg& c = kCgCfgs[i];
    if (d.n_pad % c.bn) return false;
    if (d.geglu && (c.bn / c.wn) % 64) return false;          // value / gate blocks pair up inside one wavefront
    if (c.slab && !cg_slab_ok(d, c)) return false;
    if (c.x && (!cg_x_ok(d) || (i >= 36 && ((g_x_disabled >> (i - 36)) & 1u)))) return false;
    // the LayerNorm fold starts the accumulators of the hand-scheduled tiles from its rank-1 terms: one-wave-per-SIMD tiles with a
    // 64-wide K step only (cgx_ln_ok); the compiled tiles apply the full formula in their epilogue
    if (c.x && d.ln_stats && !cgx_ln_ok(c.bk, c.wm, c.wn, c.per_cu)) return false;
    return true;
}

static int cg_choose(const AaConvGemm& d, int M) {
    int best = -1;
    double best_cost = 0.0;
    const int forced = d.tile >= 0 ? d.tile : cg_force_cfg();
    for (int i = 0; i < kNumCgCfgs; ++i) {
        const CgCfg& c = kCgCfgs[i];
        if (!cg_tile_fits(d, i)) continue;
        if (forced == i) return i;
        const double tiles = (double)((M + c.bm - 1) / c.bm) * (d.n_pad / c.bn);
        const double slots = 256.0 * c.per_cu;
        const double rounds = (double)(long long)((tiles + slots - 1) / slots);
        const double cost = rounds * c.bm * c.bn * c.per_cu / c.rate;
        if (best < 0 || cost < best_cost) { best = i; best_cost = cost; }
    }
    return best;
}

// Split-K factor: few output tiles but a long K loop (the small-M levels) leave most CUs idle behind a serial
// chain of K steps; spread the K range over up to 8 workgroups per tile (fp32 partials + a reduce launch).
static int cg_splits(const AaConvGemm& d, int M, const CgCfg& c) {
    if (d.geglu || c.slab) return 1;
    const int tiles = ((M + c.bm - 1) / c.bm) * (d.n_pad / c.bn);
    const int slots = 256 * c.per_cu;
    const int nk = d.k_pad / c.bk;
    if (d.k_splits >= 1) return d.k_splits <= nk ? d.k_splits : nk;      // the caller's (autotuned) choice
    if (tiles * 2 > slots || nk < 32) return 1;
    int s = slots / tiles;
    if (s > 8) s = 8;
    if (s > nk / 8) s = nk / 8;
    return s < 1 ? 1 : s;
}

static bool cg_dma_ok(const AaConvGemm& d) {
    const int n_cols = d.geglu ? d.n_out / 2 : d.n_out;
    return (d.c0 + d.c1) % 64 == 0 && d.c0 % 64 == 0 && d.kh <= 8 && d.kw <= 8 && d.out_dtype == d.dtype && n_cols % 8 == 0 &&
           d.ldo % 8 == 0 && aligned16(d.out) && (!d.residual || (d.ldr % 8 == 0 && aligned16(d.residual))) &&
           d.n_out % 8 == 0 && (!d.bias || d.bias_per_row || aligned16(d.bias)) && (!d.rowvec || aligned16(d.rowvec)) &&
           // operands are addressed with 32-bit byte offsets through buffer descriptors; offsets >= 2^31 mean "zero"
           (int64_t)d.n_img * d.h_in * d.w_in * (d.c0 > d.c1 ? d.c0 : d.c1) * 2 < ((int64_t)1 << 31) &&
           (int64_t)d.n_pad * d.k_pad * 2 < ((int64_t)1 << 31) &&
           // a scattered output grid (out_sy / out_sx) is addressed through one descriptor over the whole grid
           (!cgd_out_mapped(d) || (int64_t)d.n_img * d.h_out * d.w_out * (d.out_sy > 1 ? d.out_sy : 1) * (d.out_sx > 1 ? d.out_sx : 1) * d.ldo * 2 < ((int64_t)1 << 31) - 32);
}

// How one aa_conv_gemm call is carried out on the LDS-DMA path (shared by aa_conv_gemm_workspace and the launcher).
struct CgPlan {
    int cfg;            // tile table index
    int splits;         // > 1: the whole call is split along K (fp32 partials + reduce)
    int m_main;         // rows [0, m_main) run with `cfg` in full residency rounds
    int tail_cfg;       // rows [m_main, M): tile table index ...
    int tail_splits;    // ... and K split count (> 1: big tile, partials of the tail rows only)
    size_t workspace;   // bytes of fp32 scratch the plan needs (0: none)
    int tickets;        // > 0: the K-split launch (whole call or tail) finishes inside the kernel with this many ticket counters
};

// A K-split launch of `cfg` over rows [m_begin, M) can finish inside the kernel (AaConvGemm.tickets): hand-scheduled tiles only
// (the compiled kernels keep partials + reduce launch), no folded LayerNorm (its rank-1 terms live in the reduce kernel).
static int cg_ticket_count(const AaConvGemm& d, int cfg, int m_begin, int M) {
    if (!d.tickets || cfg < 0 || !kCgCfgs[cfg].x || d.ln_stats || (d.debug & 8)) return 0;
    const CgCfg& c = kCgCfgs[cfg];
    if (!cgx_ticket_ok(c.bm, c.bn, c.wm, c.wn, c.per_cu)) return 0;
    const int wgs = ((M - m_begin + c.bm - 1) / c.bm) * (d.n_pad / c.bn);
    return wgs <= d.tickets_len ? wgs : 0;
}

static CgPlan cg_plan(const AaConvGemm& d, int M, bool have_workspace_or_query) {
    CgPlan p = {cg_choose(d, M), 1, M, -1, 1, 0, 0};
    if (p.cfg < 0) return p;
    const CgCfg& c = kCgCfgs[p.cfg];
    const bool probe = (d.debug & 8) != 0;               // phase probe: the workspace holds time stamps, one plain launch
    if (probe) return p;
    p.splits = have_workspace_or_query ? cg_splits(d, M, c) : 1;
    if (p.splits > 1) { p.workspace = (size_t)p.splits * M * d.n_pad * 4; p.tickets = cg_ticket_count(d, p.cfg, 0, M); return p; }
    p.splits = 1;
    // Tiles of one or two workgroups per CU run in lock-step rounds of 256 (512); a sparsely filled last round wastes most of
    // the chip (the 139264-row level is 2.125 rounds of 256-row tiles AND of 128-row tiles at two per CU).
    // Split it off: full rounds with the big tile; the remaining rows either with the same tile split along K (long K:
    // the leftover tiles x splits fill the chip for nk / splits steps) or with a small (2-3 per CU) tile.
    if (c.per_cu > 2) return p;
    const int tiles_n = d.n_pad / c.bn;
    const int tiles_m = (M + c.bm - 1) / c.bm;
    const int cus = ((d.debug & 4) ? 2 : 256) * c.per_cu;   // resident slots.  debug bit 4: pretend a 2-CU chip (exercises the split in tests)
    const int rounds = tiles_m * tiles_n / cus;
    const int rem = tiles_m * tiles_n - rounds * cus;
    if (!(rounds >= 1 && rem > 0 && rem * 8 < cus * 5)) return p;
    {   // a leftover launch costs ~25 us (launch, first operands, a short epilogue-bound kernel: 21 us measured at the 64x64 level):
        // only worth it when the empty part of the sparse round would cost more.  One K step of a 256 x 256 x 64 tile takes
        // ~1.3 us on a CU, a tile's fixed costs (setup, first stage, epilogue, turnover) ~8 us (scripts/phase_probe_x.py).
        const double t_round = (double)(d.k_pad / c.bk) * 1.3 * c.bm * c.bn / 65536.0 * (c.bk / 64.0) * c.per_cu + 8.0;
        if (!(d.debug & 4) && (1.0 - (double)rem / cus) * t_round < 25.0) return p;
    }
    p.m_main = (rounds * cus / tiles_n) * c.bm;
    if (p.m_main >= M) { p.m_main = M; return p; }
    const int nk = d.k_pad / c.bk;
    const int tail_tiles = ((M - p.m_main + c.bm - 1) / c.bm) * tiles_n;
    int ts = c.per_cu == 1 ? cus / tail_tiles : 1;        // workgroups available per leftover tile (K splits: one-per-CU tiles only)
    if (ts > 8) ts = 8;                                   // (each split costs a round trip of fp32 partials)
    if (ts > nk / 24) ts = nk / 24;                       // measured: K loops of <= 45 steps are better off with small tiles
    if (have_workspace_or_query && ts >= 2 && !d.geglu) {
        p.tail_cfg = c.slab ? (c.bn == 320 ? 14 : 15) : p.cfg;       // (the slab kernel does not split K: its im2col twin does)
        p.tail_splits = ts;
        p.workspace = (size_t)ts * (M - p.m_main) * d.n_pad * 4;
        p.tickets = cg_ticket_count(d, p.tail_cfg, p.m_main, M);
        return p;
    }
    const int small[3] = {47, 1, 0};                       // 128x128 (hand-scheduled, then compiled), 128x64
    for (int k = 0; k < 3 && p.tail_cfg < 0; ++k)
        if (cg_tile_fits(d, small[k])) p.tail_cfg = small[k];
    if (p.tail_cfg < 0) p.tail_cfg = p.cfg;               // no small tile fits (wide GEGLU): big tile again
    return p;
}

template <typename T, int BM, int BN, int WM, int WN, int BK, int STAGES, int PER_CU, bool STAGGER = false, bool SPREAD = false>
static void cg_launch_dma(const AaConvGemm& d, int m_begin, int m_end, int splits, void* stream) {
    const int tiles_n = d.n_pad / BN;
    const dim3 grid(((m_end - m_begin + BM - 1) / BM) * tiles_n, splits), block(64 * WM * WN);
    AA_LAUNCH((conv_gemm_dma_kernel<T, BM, BN, WM, WN, BK, STAGES, PER_CU, STAGGER, SPREAD>), grid, block, cgd_lds_bytes(BM, BN, BK, STAGES), stream, d, m_end, tiles_n, m_begin, splits);
}

template <typename T, int BM, int BN, int WM, int WN>
static void cg_launch_slab(const AaConvGemm& d, int m_begin, int m_end, void* stream) {
    const int tiles_n = d.n_pad / BN;
    const dim3 grid(((m_end - m_begin) / BM) * tiles_n), block(64 * WM * WN);
    AA_LAUNCH((conv3x3_slab_kernel<T, BM, BN, WM, WN>), grid, block, cs_lds_bytes(BN), stream, d, m_end, tiles_n, m_begin);
}

template <typename T, int BM, int BN, int WM, int WN, int BK, int DP3, int DP0, int DP1, int RING = (BK == 64 ? 2 : 4), int PER_CU = 1>
static void cg_launch_x(const AaConvGemm& d, int m_begin, int m_end, int splits, void* stream) {
    const int tiles_n = d.n_pad / BN;
    const dim3 grid(((m_end - m_begin + BM - 1) / BM) * tiles_n, splits), block(64 * WM * WN);
    if (d.kh * d.kw == 1 && d.stride == 1 && d.pad_h == 0 && d.pad_w == 0)
        AA_LAUNCH((conv_gemm_x_kernel<T, BM, BN, WM, WN, BK, DP3, DP0, DP1, RING, PER_CU, true>), grid, block, cgx_lds_bytes(BM, BN, BK, RING, WM, WN, PER_CU), stream, d, m_end, tiles_n, m_begin, splits);
    else
        AA_LAUNCH((conv_gemm_x_kernel<T, BM, BN, WM, WN, BK, DP3, DP0, DP1, RING, PER_CU, false>), grid, block, cgx_lds_bytes(BM, BN, BK, RING, WM, WN, PER_CU), stream, d, m_end, tiles_n, m_begin, splits);
}

// The contraction kernels are compiled in AA_TU_GROUPS translation units (build.py compiles them in parallel: one unit took 5+
// minutes): tile-table entry i lives in unit i % AA_TU_GROUPS (csrc/aa_tiles.hip, -DAA_TU_GROUP=g), which instantiates
// cg_launch_cfg_group<T, g> explicitly; everywhere else only its declaration is seen.  AA_TU_GROUPS == 1 (the emulator build,
// ablation builds): everything in the including unit.
#ifndef AA_TU_GROUPS
#define AA_TU_GROUPS 1
#endif
template <typename T, int G>
bool cg_launch_cfg_group(int cfg, const AaConvGemm& d, int m_begin, int m_end, void* stream, int splits) {
#define AA_TILE(i, ...) case i: if constexpr (i % AA_TU_GROUPS == G) { __VA_ARGS__; return true; } else return false;
    switch (cfg) {
        AA_TILE(0, cg_launch_dma<T, 128, 64, 2, 2, 64, 2, 3>(d, m_begin, m_end, splits, stream))
        AA_TILE(1, cg_launch_dma<T, 128, 128, 2, 2, 64, 2, 2>(d, m_begin, m_end, splits, stream))
        AA_TILE(2, cg_launch_dma<T, 192, 256, 3, 2, 64, 2, 1>(d, m_begin, m_end, splits, stream))
        AA_TILE(3, cg_launch_dma<T, 256, 256, 4, 2, 64, 2, 1>(d, m_begin, m_end, splits, stream))
        AA_TILE(4, cg_launch_dma<T, 256, 320, 4, 2, 64, 2, 1>(d, m_begin, m_end, splits, stream))
        AA_TILE(5, cg_launch_dma<T, 192, 320, 3, 2, 64, 2, 1>(d, m_begin, m_end, splits, stream))
        AA_TILE(6, cg_launch_dma<T, 256, 320, 4, 2, 32, 4, 1>(d, m_begin, m_end, splits, stream))
        AA_TILE(7, cg_launch_dma<T, 256, 256, 4, 2, 32, 4, 1>(d, m_begin, m_end, splits, stream))
        AA_TILE(8, cg_launch_dma<T, 128, 128, 2, 2, 32, 4, 2>(d, m_begin, m_end, splits, stream))
        AA_TILE(9, cg_launch_dma<T, 128, 64, 2, 2, 32, 4, 3>(d, m_begin, m_end, splits, stream))
        AA_TILE(10, cg_launch_dma<T, 192, 320, 3, 2, 32, 4, 1>(d, m_begin, m_end, splits, stream))
        AA_TILE(11, cg_launch_dma<T, 128, 320, 2, 2, 32, 2, 2>(d, m_begin, m_end, splits, stream))
        AA_TILE(12, cg_launch_dma<T, 128, 256, 2, 2, 32, 2, 2>(d, m_begin, m_end, splits, stream))
        AA_TILE(13, cg_launch_dma<T, 128, 256, 2, 2, 64, 2, 1>(d, m_begin, m_end, splits, stream))
        AA_TILE(14, cg_launch_dma<T, 256, 320, 4, 2, 64, 2, 1, true>(d, m_begin, m_end, splits, stream))
        AA_TILE(15, cg_launch_dma<T, 256, 256, 4, 2, 64, 2, 1, true>(d, m_begin, m_end, splits, stream))
        AA_TILE(16, cg_launch_dma<T, 128, 128, 2, 2, 32, 2, 3>(d, m_begin, m_end, splits, stream))
        AA_TILE(17, cg_launch_dma<T, 128, 64, 2, 2, 32, 2, 4>(d, m_begin, m_end, splits, stream))
        AA_TILE(18, cg_launch_dma<T, 64, 128, 2, 2, 32, 2, 4>(d, m_begin, m_end, splits, stream))
        AA_TILE(19, cg_launch_dma<T, 64, 64, 2, 2, 32, 2, 6>(d, m_begin, m_end, splits, stream))
        AA_TILE(20, cg_launch_dma<T, 64, 256, 2, 2, 32, 2, 3>(d, m_begin, m_end, splits, stream))
        AA_TILE(21, cg_launch_dma<T, 256, 320, 4, 2, 32, 4, 1, true>(d, m_begin, m_end, splits, stream))
        AA_TILE(22, cg_launch_dma<T, 256, 256, 4, 2, 32, 4, 1, true>(d, m_begin, m_end, splits, stream))
        AA_TILE(23, cg_launch_dma<T, 192, 256, 2, 4, 64, 2, 1>(d, m_begin, m_end, splits, stream))
        AA_TILE(24, cg_launch_dma<T, 192, 256, 2, 4, 64, 2, 1, true>(d, m_begin, m_end, splits, stream))
        AA_TILE(25, cg_launch_dma<T, 128, 256, 2, 4, 32, 2, 2>(d, m_begin, m_end, splits, stream))
        AA_TILE(26, cg_launch_dma<T, 256, 320, 4, 2, 64, 2, 1, false, true>(d, m_begin, m_end, splits, stream))
        AA_TILE(27, cg_launch_dma<T, 256, 256, 4, 2, 64, 2, 1, false, true>(d, m_begin, m_end, splits, stream))
        AA_TILE(28, cg_launch_dma<T, 192, 256, 2, 4, 64, 2, 1, false, true>(d, m_begin, m_end, splits, stream))
        AA_TILE(29, cg_launch_dma<T, 256, 320, 4, 2, 32, 4, 1, false, true>(d, m_begin, m_end, splits, stream))
        AA_TILE(30, cg_launch_dma<T, 128, 320, 2, 2, 32, 2, 2, false, true>(d, m_begin, m_end, splits, stream))
        AA_TILE(31, cg_launch_dma<T, 128, 128, 2, 2, 64, 2, 2, false, true>(d, m_begin, m_end, splits, stream))
        AA_TILE(32, cg_launch_dma<T, 192, 320, 3, 2, 64, 2, 1, false, true>(d, m_begin, m_end, splits, stream))
        AA_TILE(33, cg_launch_dma<T, 128, 256, 2, 2, 32, 2, 2, false, true>(d, m_begin, m_end, splits, stream))
        AA_TILE(34, cg_launch_slab<T, 256, 320, 4, 2>(d, m_begin, m_end, stream))
        AA_TILE(35, cg_launch_slab<T, 256, 256, 4, 2>(d, m_begin, m_end, stream))
        AA_TILE(36, cg_launch_x<T, 256, 256, 2, 2, 64, 8, 8, 0>(d, m_begin, m_end, splits, stream))
        AA_TILE(37, cg_launch_x<T, 256, 320, 2, 2, 64, 11, 7, 0>(d, m_begin, m_end, splits, stream))
        AA_TILE(38, cg_launch_x<T, 256, 256, 2, 2, 64, 6, 5, 5>(d, m_begin, m_end, splits, stream))
        AA_TILE(39, cg_launch_x<T, 256, 320, 2, 2, 64, 7, 6, 5>(d, m_begin, m_end, splits, stream))
        AA_TILE(40, cg_launch_x<T, 256, 256, 4, 2, 64, 2, 2, 2>(d, m_begin, m_end, splits, stream))
        AA_TILE(41, cg_launch_x<T, 256, 256, 2, 2, 32, 4, 0, 0>(d, m_begin, m_end, splits, stream))
        AA_TILE(42, cg_launch_x<T, 256, 320, 2, 2, 32, 5, 0, 0>(d, m_begin, m_end, splits, stream))
        AA_TILE(43, cg_launch_x<T, 256, 256, 4, 2, 32, 2, 0, 0>(d, m_begin, m_end, splits, stream))
        AA_TILE(44, cg_launch_x<T, 128, 256, 2, 2, 32, 3, 0, 0, 3, 2>(d, m_begin, m_end, splits, stream))
        AA_TILE(45, cg_launch_x<T, 256, 128, 4, 1, 32, 3, 0, 0, 3, 2>(d, m_begin, m_end, splits, stream))
        AA_TILE(46, cg_launch_x<T, 192, 256, 2, 2, 64, 5, 5, 4>(d, m_begin, m_end, splits, stream))
        AA_TILE(47, cg_launch_x<T, 128, 128, 2, 2, 64, 2, 2, 2, 2, 2>(d, m_begin, m_end, splits, stream))
        AA_TILE(48, cg_launch_x<T, 128, 128, 2, 2, 32, 2, 0, 0, 4, 2>(d, m_begin, m_end, splits, stream))
        AA_TILE(49, cg_launch_x<T, 192, 320, 2, 2, 64, 6, 5, 5>(d, m_begin, m_end, splits, stream))
        default: return false;
    }
#undef AA_TILE
}
#if AA_TU_GROUPS > 1
#define AA_X(g) extern template bool cg_launch_cfg_group<f16_t, g>(int, const AaConvGemm&, int, int, void*, int); \
                extern template bool cg_launch_cfg_group<bf16_t, g>(int, const AaConvGemm&, int, int, void*, int);
AA_X(0) AA_X(1) AA_X(2) AA_X(3) AA_X(4) AA_X(5) AA_X(6) AA_X(7)
#undef AA_X
static_assert(AA_TU_GROUPS == 8, "aa_tiles.hip is compiled once per group: keep build.py and this list in step");
#endif

template <typename T>
static bool cg_launch_cfg(int cfg, const AaConvGemm& d, int m_begin, int m_end, void* stream, int splits = 1) {
    if (cfg < 0) return false;
#if AA_TU_GROUPS > 1
    switch (cfg % AA_TU_GROUPS) {
        case 0: return cg_launch_cfg_group<T, 0>(cfg, d, m_begin, m_end, stream, splits);
        case 1: return cg_launch_cfg_group<T, 1>(cfg, d, m_begin, m_end, stream, splits);
        case 2: return cg_launch_cfg_group<T, 2>(cfg, d, m_begin, m_end, stream, splits);
        case 3: return cg_launch_cfg_group<T, 3>(cfg, d, m_begin, m_end, stream, splits);
        case 4: return cg_launch_cfg_group<T, 4>(cfg, d, m_begin, m_end, stream, splits);
        case 5: return cg_launch_cfg_group<T, 5>(cfg, d, m_begin, m_end, stream, splits);
        case 6: return cg_launch_cfg_group<T, 6>(cfg, d, m_begin, m_end, stream, splits);
        default: return cg_launch_cfg_group<T, 7>(cfg, d, m_begin, m_end, stream, splits);
    }
#else
    return cg_launch_cfg_group<T, 0>(cfg, d, m_begin, m_end, stream, splits);
#endif
}

#ifdef AA_TU_TILES_ONLY             // (csrc/aa_tiles.hip stops here: the launchers of one group, nothing of the API)
}  // namespace aa
#else
template <typename T>
static int conv_gemm_t(const AaConvGemm& d, void* stream) {
    const int M = (int)((int64_t)d.n_img * d.h_out * d.w_out);
    // LDS-DMA fast path: K tiles never straddle a filter tap / concat source, output rows are 16-byte chunks
    if (cg_dma_ok(d)) {
        CgPlan pl = cg_plan(d, M, d.workspace != nullptr);
        if (pl.cfg < 0) return fail(AA_E_SHAPE, "conv_gemm: no tile shape divides n_pad=%d (geglu=%d)", d.n_pad, d.geglu);
        if (pl.workspace > (size_t)d.workspace_bytes) pl = cg_plan(d, M, false);       // scratch too small: plan without it
        auto reduce = [&](int m_begin, int splits) {
            int64_t blocks = ((int64_t)(M - m_begin) * (d.n_out / 8) + 255) / 256;
            if (blocks > 4096) blocks = 4096;
            AA_LAUNCH((splitk_reduce_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, stream, d, M, m_begin, splits);
        };
        // the kernels finish a K split themselves iff they see a ticket array: hand it over only where the plan counted on it
        AaConvGemm dk = d;
        if (!pl.tickets) dk.tickets = nullptr;
        if (pl.splits > 1) {
            cg_launch_cfg<T>(pl.cfg, dk, 0, M, stream, pl.splits);
            if (!pl.tickets) reduce(0, pl.splits);
            return finish("conv_gemm");
        }
        cg_launch_cfg<T>(pl.cfg, dk, 0, pl.m_main, stream);
        if (pl.m_main < M) {
            AaConvGemm tail = dk;
            tail.tile = -1;
            cg_launch_cfg<T>(pl.tail_cfg, tail, pl.m_main, M, stream, pl.tail_splits);
            if (pl.tail_splits > 1 && !pl.tickets) reduce(pl.m_main, pl.tail_splits);
        }
        return finish("conv_gemm");
    }
    if (d.geglu) return fail(AA_E_SHAPE, "conv_gemm: GEGLU needs the LDS-DMA path (channels %% 64 == 0, 16-byte rows)");
    if (d.k_order) return fail(AA_E_SHAPE, "conv_gemm: chunk-major weights need the LDS-DMA path (channels %% 64 == 0, 16-byte rows)");
    // generic gather path (odd channel counts: conv_in2, conv_out, VAE stem / head, fp32 scores)
    const bool bn128 = (d.n_pad % 128 == 0);
    const int bn = bn128 ? 128 : 64;
    const int tiles_n = d.n_pad / bn;
    const dim3 grid(((M + CG_BM - 1) / CG_BM) * tiles_n), block(CG_THREADS);
    if (bn128) AA_LAUNCH((conv_gemm_kernel<T, 128>), grid, block, cg_lds_bytes(128), stream, d, M, tiles_n);
    else       AA_LAUNCH((conv_gemm_kernel<T, 64>), grid, block, cg_lds_bytes(64), stream, d, M, tiles_n);
    return finish("conv_gemm");
}

template <typename T>
static int groupnorm_t(const AaGroupNorm& d, float* ws, int chunks, int apply_chunks, void* stream) {
    const int C = d.c0 + d.c1;
    const int S = C / 8;
    const int rpp = S <= GN_THREADS ? GN_THREADS / S : 1;
    AA_LAUNCH((groupnorm_stats_kernel<T>), dim3(chunks, d.n_groups_img), dim3(GN_THREADS), (size_t)C * 8 * (1 + rpp), stream, d, ws, chunks);
    const size_t lds = ((size_t)2 * C + 2 * (GN_THREADS + d.num_groups)) * 4;
    AA_LAUNCH((groupnorm_apply_kernel<T>), dim3(apply_chunks, d.n_groups_img), dim3(GN_THREADS), lds, stream, d, (const float*)ws, chunks, apply_chunks);
    return finish("groupnorm");
}

// ---- single-pass GroupNorm (norm.h: groupnorm_fused_kernel): which channel block / rows per thread, if one workgroup can
// hold every token of an image group at all.  Whether it beats the two-kernel pair is the caller's measurement (ops.py).
struct GnfPlan { int gb, nr, parts, grid; };
static bool gnf_plan(const AaGroupNorm& d, GnfPlan& best) {
    const int C = d.c0 + d.c1, G = d.num_groups, cg = C / G;
    static const int kNR[] = {8, 16, 24, 32, 48};
    bool found = false;
    long best_est = 0;
    for (int gb = 1; gb <= G; gb *= 2) {
        if (G % gb || (gb * cg) % 8 || (gb * cg) / 8 > GNF_THREADS) continue;
        const int cw = gb * cg, sw = cw / 8, rpp = GNF_THREADS / sw;
        if (rpp * sw * 4 < GNF_THREADS * 3) continue;                      // more than a quarter of the threads idle
        if (gnf_lds_bytes(cw, gb) > 64 * 1024) continue;
        for (int k = 0; k < 5; ++k) {
            const int nr = kNR[k], pt = rpp * nr;
            if (pt < d.tokens_per_group) continue;                         // one workgroup per (image group, channel block)
            const long grid = (long)d.n_groups_img * (G / gb);
            if (grid > 65535L * 16) break;
            // estimated time (ns; calibrated on the r03q measurements): a workgroup needs a fixed ~8 us (launch share, two memory
            // round trips, the reductions) plus its bytes at ~25 GB/s; a grid beyond what is resident at once runs in rounds;
            // row segments below 160 bytes waste DRAM bursts
            const long resident = (nr <= 16 ? 2 : 1) * 256L;              // launch bounds: 4 (2) waves per SIMD = 2 (1) workgroups per CU
            const long rounds = (grid + resident - 1) / resident;
            long est = rounds * (8000 + (long)d.tokens_per_group * cw * 4 / 25);
            if (cw * 2 < 160) est = est * 160 / (cw * 2);
            if (grid < 128) est += (128 - grid) * 100;                      // too few workgroups to pull the HBM rate
            if (!found || est < best_est) { best_est = est; best = GnfPlan{gb, nr, 1, (int)grid}; found = true; }
            break;                                                          // (a larger nr only costs registers)
        }
    }
    return found;
}

template <typename T>
static void gnf_launch(const AaGroupNorm& d, const GnfPlan& pl, void* stream) {
    const int cw = pl.gb * ((d.c0 + d.c1) / d.num_groups);
    const dim3 grid(1, d.num_groups / pl.gb, d.n_groups_img), block(GNF_THREADS);
    const size_t lds = gnf_lds_bytes(cw, pl.gb);
    switch (pl.nr) {
        case 8:  AA_LAUNCH((groupnorm_fused_kernel<T, 8>), grid, block, lds, stream, d, pl.gb); break;
        case 16: AA_LAUNCH((groupnorm_fused_kernel<T, 16>), grid, block, lds, stream, d, pl.gb); break;
        case 24: AA_LAUNCH((groupnorm_fused_kernel<T, 24>), grid, block, lds, stream, d, pl.gb); break;
        case 32: AA_LAUNCH((groupnorm_fused_kernel<T, 32>), grid, block, lds, stream, d, pl.gb); break;
        default: AA_LAUNCH((groupnorm_fused_kernel<T, 48>), grid, block, lds, stream, d, pl.gb); break;
    }
}

static thread_local bool g_gn_two_pass = false;      // aa_set_groupnorm_two_pass(1): the statistics + apply kernel pair (tests, A/B timing)
static int gn_chunks(const AaGroupNorm& d) {
    // aim for >= ~1024 workgroups, at least 16 tokens each, at most 256 chunks per image group (every apply workgroup
    // re-reduces its image group's chunk partials: measured 54 us at 256 chunks vs 94 us at 1024 for the clip-wide
    // statistics of the 64x64 level, r02 GPU call F)
    // (round 5: FLOOR - 34 images x ceil(1024 / 34) = 1054 workgroups put a fifth workgroup on 30 of the 256 CUs while the rest hold four:
    //  the kernel then lasts 5 / 4.12 of its balanced time; 34 x 30 = 1020 fit four per CU)
    int want = 1024 / d.n_groups_img;                    //  measured -5 ... -7 % on every per-image norm, profiles/r05k_groupnorm_chunks.txt)
    if (want < 1) want = 1;
    int cap = (d.tokens_per_group + 15) / 16;
    if (cap > 256) cap = 256;
    int c = want < cap ? want : cap;
    return c < 1 ? 1 : c;
}

template <typename T>
static int attention_t(const AaAttention& d, void* stream) {
    const int nseq = d.n_outer * d.n_inner;
    if (d.head_dim == 8 || d.head_dim == 80) {
        const dim3 grid((d.q_len + 255) / 256, d.heads, nseq);
        const size_t lds = (size_t)2 * 256 * d.head_dim * sizeof(T);
        if (d.head_dim == 8) AA_LAUNCH((attention_small_kernel<T, 8>), grid, dim3(256), lds, stream, d);
        else                 AA_LAUNCH((attention_small_kernel<T, 80>), grid, dim3(256), lds, stream, d);
        return finish("attention");
    }
    if (d.q_len >= 128 && d.kv_len > AT_KT && d.kv_len <= ATS_KEYS && !d.causal && !(d._pad & 4)) {
        // short key sequence, many queries (text cross-attention): K / V fragments resident in registers, each wave walks `qb` 32-query
        // blocks (AaAttention._pad bit 2 = the general kernel instead: A/B).  qb: enough waves to fill the chip twice over, at most 8 blocks
        const int n_qb = (d.q_len + 31) / 32;
        int qb = 8;
        while (qb > 1 && (int64_t)((n_qb + 4 * qb - 1) / (4 * qb)) * d.heads * nseq < 1024) qb >>= 1;
        const dim3 grid((n_qb + 4 * qb - 1) / (4 * qb), d.heads, nseq);
        AA_LAUNCH((attention_shortkv_kernel<T>), grid, dim3(256), 2 * AT_TILE_BYTES, stream, d, qb);
    } else if (d.q_len >= 512 && d.kv_len >= 512 && (d._pad & 8)) {
        // AaAttention._pad bit 3 (experiment, round 6): long sequences with two 32-query blocks per wave at two waves per SIMD.  Measured SLOWER
        // than one block per wave at three waves per SIMD - 1058 against 883 us at 34 x 5 x 4096 tokens, 169 against 159 at 34 x 10 x 1024
        // (profiles/r06g_attention_two_blocks_per_wave.txt): at 249 registers only two V^T fragment pairs can be in flight, every O^T MFMA
        // waits for its LDS read, and two waves per SIMD do not cover that.  Kept for the tests that walk its slow path; never the default.
        const dim3 grid((d.q_len + 255) / 256, d.heads, nseq);
        AA_LAUNCH((attention_kernel<T, 4, 64, 1, 2>), grid, dim3(256), attn_lds_bytes(d.kv_len), stream, d);
    } else if (d.q_len > 64) {
        const dim3 grid((d.q_len + 127) / 128, d.heads, nseq);
        AA_LAUNCH((attention_kernel<T, 4>), grid, dim3(256), attn_lds_bytes(d.kv_len), stream, d);
    } else {
        const dim3 grid((d.q_len + 31) / 32, d.heads, nseq);
        if (d.kv_len <= 32 && d.q_len <= 32 && nseq >= 4096) {          // many short sequences: eight (four) per workgroup, one per wave
            const dim3 grid8(1, d.heads, (nseq + 7) / 8);              // (measured, 64x64 level: 109 -> 93.5 -> 89.0 us; 32x32 level: 45 -> 37.6 / 40.1 us)
            AA_LAUNCH((attention_kernel<T, 1, 32, 8>), grid8, dim3(512), 8 * (AT_TILE_BYTES / 2), stream, d);
        } else if (d.kv_len <= 32 && d.q_len <= 32 && nseq >= 256) {
            const dim3 grid4(1, d.heads, (nseq + 3) / 4);
            AA_LAUNCH((attention_kernel<T, 1, 32, 4>), grid4, dim3(256), 4 * (AT_TILE_BYTES / 2), stream, d);
        } else if (d.kv_len <= 32) AA_LAUNCH((attention_kernel<T, 1, 32>), grid, dim3(64), AT_TILE_BYTES / 2, stream, d);
        else                AA_LAUNCH((attention_kernel<T, 1>), grid, dim3(64), attn_lds_bytes(d.kv_len), stream, d);
    }
    return finish("attention");
}

}  // namespace aa

extern "C" {

int aa_version(void) { return AA_VERSION; }
int aa_conv_gemm_tile_info(int idx, int32_t info[7]) {
    if (idx < 0 || idx >= aa::kNumCgCfgs || !info) return -1;
    const aa::CgCfg& c = aa::kCgCfgs[idx];
    info[0] = c.bm; info[1] = c.bn; info[2] = c.wm; info[3] = c.wn; info[4] = c.bk; info[5] = c.stages; info[6] = c.per_cu;
    return 0;
}
int aa_conv_gemm_tile_ok(const AaConvGemm* d, int idx) {
    using namespace aa;
    if (!d || idx < 0 || idx >= kNumCgCfgs || !cg_dma_ok(*d)) return 0;
    return cg_tile_fits(*d, idx) ? 1 : 0;
}
void aa_set_tile_override(int cfg) {
    if (cfg <= -100) { aa::g_x_disabled = (unsigned)(-100 - cfg); return; }     // bisecting aid: withdraw hand-scheduled tiles (bit i = entry 36 + i)
    aa::g_tile_override = cfg < 0 ? -1 : cfg;
}
const char* aa_last_error(void) { return aa::g_err; }

size_t aa_conv_gemm_workspace(const AaConvGemm* d) {
    using namespace aa;
    if (!d || !cg_dma_ok(*d)) return 0;
    const int M = (int)((int64_t)d->n_img * d->h_out * d->w_out);
    return cg_plan(*d, M, true).workspace;
}

int aa_conv_gemm_row_stats_parts(const AaConvGemm* d) {
    using namespace aa;
    if (!d || !cg_dma_ok(*d) || d->geglu || d->rowvec || d->act != AA_ACT_NONE || d->bias_per_row || d->ln_stats || cgd_out_mapped(*d)) return 0;
    const int M = (int)((int64_t)d->n_img * d->h_out * d->w_out);
    CgPlan pl = cg_plan(*d, M, d->workspace != nullptr);
    if (pl.cfg < 0) return 0;
    if (pl.workspace > (size_t)d->workspace_bytes) pl = cg_plan(*d, M, false);
    const CgCfg& c = kCgCfgs[pl.cfg];
    // only the branch-free epilogue forms of the hand-scheduled tiles emit them, in ONE launch that covers every row
    if (!c.x || (pl.splits > 1 && !pl.tickets) || pl.m_main < M || (d->debug & 8)) return 0;      // (a K split that finishes in the kernel runs the same epilogue)
    return (d->n_pad / c.bn) * c.wn;
}

int aa_conv_gemm_row_coef_ok(const AaConvGemm* d) {
    using namespace aa;
    if (!d || !cg_dma_ok(*d)) return 0;
    AaConvGemm q = *d;                                   // asked as a plain statistics call
    q.row_coef = nullptr;
    const int parts = aa_conv_gemm_row_stats_parts(&q);
    if (parts <= 0) return 0;
    const int M = (int)((int64_t)d->n_img * d->h_out * d->w_out);
    CgPlan pl = cg_plan(q, M, q.workspace != nullptr);
    if (pl.workspace > (size_t)q.workspace_bytes) pl = cg_plan(q, M, false);
    const CgCfg& c = kCgCfgs[pl.cfg];
    // one launch, one column tile: the tile's waves hold the whole row between them; the plain K loop only (a K split that finishes in
    // the kernel runs its epilogue in ONE of the tile's workgroups: the others never reach the exchange barrier - and it is not needed here)
    return (d->n_pad == c.bn && parts == c.wn && pl.splits == 1 && pl.m_main >= M && c.wm * c.wn * (c.bm / c.wm) * 8 <= CGX_COEF_BYTES) ? 1 : 0;
}

static bool cg_plan_of(const AaConvGemm* d, aa::CgPlan& pl, int& M) {
    using namespace aa;
    if (!d || !cg_dma_ok(*d)) return false;
    M = (int)((int64_t)d->n_img * d->h_out * d->w_out);
    pl = cg_plan(*d, M, d->workspace != nullptr);
    if (pl.cfg < 0) return false;
    if (pl.workspace > (size_t)d->workspace_bytes) pl = cg_plan(*d, M, false);
    return true;
}

int aa_conv_gemm_reduce_launches(const AaConvGemm* d) {
    aa::CgPlan pl; int M;
    if (!cg_plan_of(d, pl, M) || pl.tickets) return 0;
    if (pl.splits > 1) return 1;
    return (pl.m_main < M && pl.tail_splits > 1) ? 1 : 0;
}

int aa_conv_gemm_tickets(const AaConvGemm* d) {
    aa::CgPlan pl; int M;
    if (!d) return 0;
    // (asked with a stand-in array: how many counters WOULD the call use)
    AaConvGemm q = *d;
    static int32_t probe_array;
    if (!q.tickets) { q.tickets = &probe_array; q.tickets_len = 0x7fffffff; }
    if (!cg_plan_of(&q, pl, M)) return 0;
    return pl.tickets;
}

int aa_conv_gemm_launch_count(const AaConvGemm* d) {
    using namespace aa;
    if (!d) return 0;
    if (!cg_dma_ok(*d)) return 1;
    CgPlan pl; int M;
    if (!cg_plan_of(d, pl, M)) return 0;
    int n = 1;
    if (pl.splits == 1 && pl.m_main < M) n += 1;
    return n + aa_conv_gemm_reduce_launches(d);
}

int aa_conv_gemm_tile_flags(int idx) {
    if (idx < 0 || idx >= aa::kNumCgCfgs) return -1;
    const aa::CgCfg& c = aa::kCgCfgs[idx];
    return (c.slab ? 1 : 0) | (c.x ? 2 : 0) | (c.sched << 2);
}

int aa_conv_gemm(const AaConvGemm* d, void* stream) {
    using namespace aa;
    if (!d) return fail(AA_E_SHAPE, "conv_gemm: null descriptor");
    const int ctot = d->c0 + d->c1;
    if (d->c0 <= 0 || d->c0 % 8 || d->c1 % 8 || (d->c1 > 0 && !d->a1))
        return fail(AA_E_SHAPE, "conv_gemm: channels must be multiples of 8 (c0=%d c1=%d)", d->c0, d->c1);
    if (d->n_pad % 64 || d->k_pad % CG_BK || d->k_pad < d->kh * d->kw * ctot || d->n_out > d->n_pad)
        return fail(AA_E_SHAPE, "conv_gemm: bad packed extents n_pad=%d k_pad=%d (K=%d, n_out=%d)", d->n_pad, d->k_pad, d->kh * d->kw * ctot, d->n_out);
    if (d->n_img <= 0 || d->h_out <= 0 || d->w_out <= 0 || d->kh <= 0 || d->kw <= 0 || d->stride <= 0 || d->rowvec_div <= 0)
        return fail(AA_E_SHAPE, "conv_gemm: bad geometry");
    if (d->rowvec_ld < 0 || d->rowvec_ld % 8 || (d->rowvec_ld && d->rowvec_ld < d->n_out)) return fail(AA_E_SHAPE, "conv_gemm: rowvec_ld=%d must be 0 or a multiple of 8 >= n_out", d->rowvec_ld);
    if (d->geglu && (d->geglu != 32 || d->n_out != d->n_pad || d->bias_per_row))
        return fail(AA_E_SHAPE, "conv_gemm: GEGLU packs (32 value | 32 gate) column blocks, n_out == n_pad (geglu=%d n_out=%d)", d->geglu, d->n_out);
    if ((int64_t)d->n_img * d->h_out * d->w_out >= (int64_t)1 << 31) return fail(AA_E_SHAPE, "conv_gemm: M overflows int32");
    if (!aligned16(d->a0) || !aligned16(d->a1) || !aligned16(d->w)) return fail(AA_E_ALIGN, "conv_gemm: operands must be 16-byte aligned");
    if (d->out_dtype != AA_F32 && d->out_dtype != d->dtype) return fail(AA_E_DTYPE, "conv_gemm: out_dtype must be dtype or f32");
    if (d->act != AA_ACT_NONE && d->act != AA_ACT_SILU) return fail(AA_E_SHAPE, "conv_gemm: activation %d is not fused here (AA_ACT_NONE / AA_ACT_SILU)", d->act);
    if (d->out_sy < 0 || d->out_sx < 0 || d->out_oy < 0 || d->out_ox < 0 || d->out_oy >= (d->out_sy > 1 ? d->out_sy : 1) || d->out_ox >= (d->out_sx > 1 ? d->out_sx : 1))
        return fail(AA_E_SHAPE, "conv_gemm: bad output grid mapping (out_sy=%d out_sx=%d out_oy=%d out_ox=%d)", d->out_sy, d->out_sx, d->out_oy, d->out_ox);
    if (d->ln_stats) {
        if (!d->ln_cols || d->bias || d->rowvec || d->residual || d->act != AA_ACT_NONE || d->bias_per_row || d->c1 ||
            d->kh * d->kw != 1 || d->out_scale != 1.0f || (d->acc_scale != 0.0f && d->acc_scale != 1.0f) || !cg_dma_ok(*d) || cgd_out_mapped(*d))
            return fail(AA_E_SHAPE, "conv_gemm: the LayerNorm fold (ln_stats) is a plain or GEGLU linear call: ln_cols, no bias / row vector / residual / activation / scales");
        if (d->ln_parts < 0 || d->ln_parts > 64 || (d->ln_parts > 0 && !(d->ln_eps > 0.0f)))
            return fail(AA_E_SHAPE, "conv_gemm: ln_parts=%d (0: finalised coefficients, 1..64: raw partial sums + ln_eps > 0)", d->ln_parts);
    }
    if (d->row_coef) {
        if (d->row_stats || d->row_stats_parts || !(d->row_coef_eps > 0.0f) || !aligned16(d->row_coef) || !aa_conv_gemm_row_coef_ok(d))
            return fail(AA_E_SHAPE, "conv_gemm: row_coef needs a statistics-emitting call whose tile spans the output row (aa_conv_gemm_row_coef_ok), "
                                    "row_stats NULL, row_coef_eps > 0, 16-byte aligned coefficients");
    }
    if (d->row_stats && d->row_stats_parts != aa_conv_gemm_row_stats_parts(d))
        return fail(AA_E_SHAPE, "conv_gemm: row_stats_parts=%d, this call emits %d partial statistics per row (aa_conv_gemm_row_stats_parts)", d->row_stats_parts, aa_conv_gemm_row_stats_parts(d));
    if (cgd_out_mapped(*d) && !cg_dma_ok(*d)) return fail(AA_E_SHAPE, "conv_gemm: a scattered output grid needs the LDS-DMA path (channels %% 64, 16-byte rows, < 2 GiB)");
    // a tile named in the descriptor is a demand, not a hint: a call it cannot carry out fails (it used to fall through to the
    // automatic choice, so a "forced tile" test could pass without running that tile).  The thread-local override of
    // aa_set_tile_override stays a preference ("every following call whose packed width it divides").
    if (d->tile >= 0 && !aa_conv_gemm_tile_ok(d, d->tile))
        return fail(AA_E_SHAPE, "conv_gemm: tile %d (AaConvGemm.tile) cannot carry out this call (aa_conv_gemm_tile_ok == 0)", d->tile);
    if (d->dtype == AA_F16) return conv_gemm_t<f16_t>(*d, stream);
    if (d->dtype == AA_BF16) return conv_gemm_t<bf16_t>(*d, stream);
    return fail(AA_E_DTYPE, "conv_gemm: unsupported dtype %d", d->dtype);
}

size_t aa_groupnorm_workspace(const AaGroupNorm* d) {
    return (size_t)d->n_groups_img * aa::gn_chunks(*d) * d->num_groups * 2 * sizeof(float);
}

void aa_set_groupnorm_two_pass(int on) { aa::g_gn_two_pass = on != 0; }
int aa_groupnorm_plan(const AaGroupNorm* d, int32_t info[4]) {
    aa::GnfPlan pl;
    if (!d || !info || aa::g_gn_two_pass || !aa::gnf_plan(*d, pl)) return 0;
    info[0] = pl.gb; info[1] = pl.nr; info[2] = pl.parts; info[3] = pl.grid;
    return 1;
}

int aa_groupnorm(const AaGroupNorm* d, void* workspace, size_t workspace_bytes, void* stream) {
    using namespace aa;
    if (!d) return fail(AA_E_SHAPE, "groupnorm: null descriptor");
    const int C = d->c0 + d->c1;
    if (d->c0 <= 0 || d->c0 % 8 || d->c1 % 8 || d->num_groups <= 0 || C % d->num_groups || d->num_groups > GN_THREADS)
        return fail(AA_E_SHAPE, "groupnorm: bad channels c0=%d c1=%d groups=%d", d->c0, d->c1, d->num_groups);
    if (d->n_groups_img <= 0 || d->tokens_per_group <= 0) return fail(AA_E_SHAPE, "groupnorm: bad token geometry");
    if (!aligned16(d->x0) || !aligned16(d->x1) || !aligned16(d->y)) return fail(AA_E_ALIGN, "groupnorm: operands must be 16-byte aligned");
    const size_t need = aa_groupnorm_workspace(d);
    if (!workspace || workspace_bytes < need) return fail(AA_E_WORKSPACE, "groupnorm: workspace %zu < %zu bytes", workspace_bytes, need);
    if (d->dtype != AA_F16 && d->dtype != AA_BF16) return fail(AA_E_DTYPE, "groupnorm: unsupported dtype %d", d->dtype);
    GnfPlan pl;
    if (!g_gn_two_pass && gnf_plan(*d, pl)) {
        if (d->dtype == AA_F16) gnf_launch<f16_t>(*d, pl, stream);
        else gnf_launch<bf16_t>(*d, pl, stream);
        return finish("groupnorm");
    }
    const int chunks = gn_chunks(*d);
    if (d->dtype == AA_F16) return groupnorm_t<f16_t>(*d, (float*)workspace, chunks, chunks, stream);
    return groupnorm_t<bf16_t>(*d, (float*)workspace, chunks, chunks, stream);
}

int aa_groupnorm_coef(const AaGroupNorm* d, void* workspace, size_t workspace_bytes, float* coef, void* stream) {
    using namespace aa;
    if (!d || !coef) return fail(AA_E_SHAPE, "groupnorm_coef: null descriptor / coef");
    const int C = d->c0 + d->c1;
    if (d->c0 <= 0 || d->c0 % 8 || d->c1 % 8 || d->num_groups <= 0 || C % d->num_groups || d->num_groups > GN_THREADS)
        return fail(AA_E_SHAPE, "groupnorm_coef: bad channels c0=%d c1=%d groups=%d", d->c0, d->c1, d->num_groups);
    if (d->n_groups_img <= 0 || d->tokens_per_group <= 0) return fail(AA_E_SHAPE, "groupnorm_coef: bad token geometry");
    if (d->silu) return fail(AA_E_SHAPE, "groupnorm_coef: a norm with SiLU behind it is not affine");
    if (!aligned16(d->x0) || !aligned16(d->x1) || !aligned16(coef)) return fail(AA_E_ALIGN, "groupnorm_coef: operands must be 16-byte aligned");
    const size_t need = aa_groupnorm_workspace(d);
    if (!workspace || workspace_bytes < need) return fail(AA_E_WORKSPACE, "groupnorm_coef: workspace %zu < %zu bytes", workspace_bytes, need);
    if (d->dtype != AA_F16 && d->dtype != AA_BF16) return fail(AA_E_DTYPE, "groupnorm_coef: unsupported dtype %d", d->dtype);
    const int chunks = gn_chunks(*d), S = C / 8, rpp = S <= GN_THREADS ? GN_THREADS / S : 1;
    const size_t lds = (size_t)2 * (GN_THREADS + d->num_groups) * 4;
    float* ws = (float*)workspace;
    if (d->dtype == AA_F16) {
        AA_LAUNCH((groupnorm_stats_kernel<f16_t>), dim3(chunks, d->n_groups_img), dim3(GN_THREADS), (size_t)C * 8 * (1 + rpp), stream, *d, ws, chunks);
        AA_LAUNCH((groupnorm_coef_kernel<f16_t>), dim3(d->n_groups_img), dim3(GN_THREADS), lds, stream, *d, (const float*)ws, chunks, coef);
    } else {
        AA_LAUNCH((groupnorm_stats_kernel<bf16_t>), dim3(chunks, d->n_groups_img), dim3(GN_THREADS), (size_t)C * 8 * (1 + rpp), stream, *d, ws, chunks);
        AA_LAUNCH((groupnorm_coef_kernel<bf16_t>), dim3(d->n_groups_img), dim3(GN_THREADS), lds, stream, *d, (const float*)ws, chunks, coef);
    }
    return finish("groupnorm_coef");
}

int aa_layernorm(const void* x, const void* gamma, const void* beta, void* y, int64_t rows, int32_t channels,
                 float eps, int32_t dtype, void* stream) {
    using namespace aa;
    if (rows <= 0 || channels <= 0 || channels % 8 || channels > 2048) return fail(AA_E_SHAPE, "layernorm: rows=%lld channels=%d", (long long)rows, channels);
    if (!aligned16(x) || !aligned16(y) || !aligned16(gamma) || !aligned16(beta)) return fail(AA_E_ALIGN, "layernorm: operands must be 16-byte aligned");
    const int sj = channels <= 512 ? 1 : (channels <= 1024 ? 2 : 4);      // 16-byte slots per lane and row
    const int per_wg = 4 * (4 / sj);                                        // rows per workgroup (4 waves)
    const dim3 grid((unsigned)((rows + per_wg - 1) / per_wg)), block(256);
    if (dtype != AA_F16 && dtype != AA_BF16) return fail(AA_E_DTYPE, "layernorm: unsupported dtype %d", dtype);
#define AA_LN(T, SJ) AA_LAUNCH((layernorm_kernel<T, SJ>), grid, block, 0, stream, (const T*)x, (const T*)gamma, (const T*)beta, (T*)y, rows, channels, eps)
    if (dtype == AA_F16) { if (sj == 1) AA_LN(f16_t, 1); else if (sj == 2) AA_LN(f16_t, 2); else AA_LN(f16_t, 4); }
    else                 { if (sj == 1) AA_LN(bf16_t, 1); else if (sj == 2) AA_LN(bf16_t, 2); else AA_LN(bf16_t, 4); }
#undef AA_LN
    return finish("layernorm");
}

int aa_ln_finalize(const float* stats, int32_t parts, float* coef, int64_t rows, int32_t channels, float eps, void* stream) {
    using namespace aa;
    if (!stats || !coef || parts <= 0 || rows <= 0 || channels <= 0) return fail(AA_E_SHAPE, "ln_finalize: rows=%lld parts=%d channels=%d", (long long)rows, parts, channels);
    if (!aligned16(coef)) return fail(AA_E_ALIGN, "ln_finalize: coef must be 16-byte aligned");
    const dim3 grid((unsigned)((rows + 255) / 256)), block(256);
    AA_LAUNCH(ln_finalize_kernel<0>, grid, block, 0, stream, stats, coef, rows, (int)parts, 1.0f / (float)channels, eps);
    return finish("ln_finalize");
}

int aa_attention(const AaAttention* d, void* stream) {
    using namespace aa;
    if (!d) return fail(AA_E_SHAPE, "attention: null descriptor");
    if (d->head_dim != 64 && d->head_dim != 8 && d->head_dim != 80) return fail(AA_E_SHAPE, "attention: head_dim must be 64, 8 or 80 (got %d)", d->head_dim);
    if (d->q_len <= 0 || d->kv_len <= 0 || d->heads <= 0 || d->n_outer <= 0 || d->n_inner <= 0) return fail(AA_E_SHAPE, "attention: bad lengths");
    const AaAttnOperand* ops[4] = {&d->q, &d->k, &d->v, &d->o};
    for (const AaAttnOperand* x : ops) {
        if (!aligned16(x->ptr) || x->ld % 8 || x->col0 % 8) return fail(AA_E_ALIGN, "attention: operand rows must be 16-byte aligned");
        if (x->outer_div <= 0) return fail(AA_E_SHAPE, "attention: outer_div must be >= 1");
        if (x->seq_mod < 0) return fail(AA_E_SHAPE, "attention: seq_mod must be >= 0");
    }
    if (d->q.seq_mod || d->o.seq_mod) return fail(AA_E_SHAPE, "attention: seq_mod addressing is for K / V only");
    if (d->causal && d->head_dim != 64) return fail(AA_E_SHAPE, "attention: causal masking is implemented for head_dim 64");
    // K / V are read through buffer descriptors with 32-bit byte offsets; offsets >= 2^31 mean "zero"
    if (d->head_dim == 64 && (attn_extent_bytes(d->k, d->n_outer, d->n_inner, d->kv_len) >= ((int64_t)1 << 31) ||
        attn_extent_bytes(d->v, d->n_outer, d->n_inner, d->kv_len) >= ((int64_t)1 << 31)))
        return fail(AA_E_SHAPE, "attention: K / V operand must stay below 2 GiB");
    if (d->dtype == AA_F16) return attention_t<f16_t>(*d, stream);
    if (d->dtype == AA_BF16) return attention_t<bf16_t>(*d, stream);
    return fail(AA_E_DTYPE, "attention: unsupported dtype %d", d->dtype);
}

int aa_seq_self_attention_ok(const AaSeqSelfAttn* d) {
    using namespace aa;
    if (!d) return 0;
    if (d->channels != 320 && d->channels != 512 && d->channels != 640) return 0;
    if (d->seq_len < 1 || d->seq_len > 32 || d->n_outer <= 0 || d->n_inner <= 0) return 0;
    if (d->ldx % 8 || d->ldo % 8 || d->ldx < d->channels || d->ldo < d->channels) return 0;
    if (d->x_bytes <= 0 || d->o_bytes <= 0 || d->x_bytes >= ((int64_t)1 << 31) || d->o_bytes >= ((int64_t)1 << 31)) return 0;
    if (d->dtype != AA_F16 && d->dtype != AA_BF16) return 0;
    return 1;
}

int aa_seq_self_attention(const AaSeqSelfAttn* d, void* stream) {
    using namespace aa;
    if (!d) return fail(AA_E_SHAPE, "seq_self_attention: null descriptor");
    if (!aa_seq_self_attention_ok(d))
        return fail(AA_E_SHAPE, "seq_self_attention: unsupported call (channels %d, seq_len %d, ldx %d, ldo %d, dtype %d)", d->channels, d->seq_len, d->ldx, d->ldo, d->dtype);
    if (!aligned16(d->x) || !aligned16(d->w) || !aligned16(d->o) || !aligned16(d->w_bias))
        return fail(AA_E_ALIGN, "seq_self_attention: operands must be 16-byte aligned");
    if (d->normalize && !(d->ln_eps > 0.0f)) return fail(AA_E_SHAPE, "seq_self_attention: normalize needs ln_eps > 0");
    if (d->pre_w) {
        if (!d->pre_out || !aligned16(d->pre_w) || !aligned16(d->pre_out) || !aligned16(d->pre_bias) || !aligned16(d->pre_residual))
            return fail(AA_E_ALIGN, "seq_self_attention: the projection in front needs pre_out and 16-byte aligned operands");
        const int64_t rows_ext = d->x_bytes / (2 * (int64_t)d->ldx);
        if (d->ld_pre % 8 || d->ld_pre < d->channels || rows_ext * d->ld_pre * 2 >= ((int64_t)1 << 31) ||
            (d->pre_residual && (d->ld_res % 8 || d->ld_res < d->channels || rows_ext * d->ld_res * 2 >= ((int64_t)1 << 31))))
            return fail(AA_E_SHAPE, "seq_self_attention: bad row pitch / extent of pre_out / pre_residual");
    } else if (d->pre_bias || d->pre_residual || d->pre_out) return fail(AA_E_SHAPE, "seq_self_attention: pre_bias / pre_residual / pre_out without pre_w");
    const int per = (32 * SA_NW) / d->seq_len;
    const int64_t n_seq = (int64_t)d->n_outer * d->n_inner;
    const dim3 grid((unsigned)((n_seq + per - 1) / per)), block(64 * SA_NW);
    const size_t lds = sa_lds_bytes(d->channels);
#define AA_SA(T, C_) AA_LAUNCH((seq_self_attention_kernel<T, C_>), grid, block, lds, stream, *d)
    if (d->dtype == AA_F16) {
        if (d->channels == 320) AA_SA(f16_t, 320); else if (d->channels == 512) AA_SA(f16_t, 512); else AA_SA(f16_t, 640);
    } else {
        if (d->channels == 320) AA_SA(bf16_t, 320); else if (d->channels == 512) AA_SA(bf16_t, 512); else AA_SA(bf16_t, 640);
    }
#undef AA_SA
    return finish("seq_self_attention");
}

int aa_ff_fused_ok(const AaFFFused* d) {
    if (!d || d->channels != 320 || d->rows <= 0) return 0;
    if (d->ldx % 8 || d->ldo % 8 || d->ld_outer % 8 || d->ldx < d->channels || d->ldo < d->channels || (d->outer && d->ld_outer < d->channels)) return 0;
    const int64_t lim = (int64_t)1 << 31;
    if (d->rows * d->ldx * 2 >= lim || d->rows * d->ldo * 2 >= lim || (d->outer && d->rows * d->ld_outer * 2 >= lim)) return 0;
    if (d->dtype != AA_F16 && d->dtype != AA_BF16) return 0;
    return 1;
}

int aa_ff_fused(const AaFFFused* d, void* stream) {
    using namespace aa;
    if (!d) return fail(AA_E_SHAPE, "ff_fused: null descriptor");
    if (!aa_ff_fused_ok(d)) return fail(AA_E_SHAPE, "ff_fused: unsupported call (channels %d, rows %lld, dtype %d)", d->channels, (long long)d->rows, d->dtype);
    if (!d->x || !d->out || !d->w) return fail(AA_E_SHAPE, "ff_fused: x, out, w are required");
    if (!aligned16(d->x) || !aligned16(d->outer) || !aligned16(d->out) || !aligned16(d->w)) return fail(AA_E_ALIGN, "ff_fused: operands must be 16-byte aligned");
    if (d->normalize && !(d->ln_eps > 0.0f)) return fail(AA_E_SHAPE, "ff_fused: normalize needs ln_eps > 0");
    const dim3 grid((unsigned)((d->rows + 32 * FF_NW - 1) / (32 * FF_NW))), block(64 * FF_NW);
    const int abl = d->flags >> 8;             // timing ablations (scripts/bench_ff_fused.py --ablate): fp16 instantiations only
#define AA_FF(ABL_) case ABL_: AA_LAUNCH((ff_fused_kernel<f16_t, 320, ABL_>), grid, block, ff_lds_bytes(), stream, *d); break
    if (abl && d->dtype == AA_F16) {
        switch (abl) { AA_FF(1); AA_FF(4); AA_FF(8); AA_FF(16); AA_FF(32); AA_FF(33); AA_FF(24); AA_FF(61); AA_FF(256); AA_FF(64); AA_FF(128); AA_FF(512);
                       default: return fail(AA_E_SHAPE, "ff_fused: no instantiation for ablation %d", abl); }
    } else if (d->dtype == AA_F16) AA_LAUNCH((ff_fused_kernel<f16_t, 320>), grid, block, ff_lds_bytes(), stream, *d);
    else                    AA_LAUNCH((ff_fused_kernel<bf16_t, 320>), grid, block, ff_lds_bytes(), stream, *d);
#undef AA_FF
    return finish("ff_fused");
}

int aa_linear_rows_ok(const AaLinearRows* d) {
    if (!d || d->channels != 320 || d->rows <= 0 || d->n_out <= 0 || d->n_out % 32) return 0;
    if (d->ldx % 8 || d->ldo % 8 || d->ld_res % 8 || d->ldx < d->channels || d->ldo < d->n_out || (d->residual && d->ld_res < d->n_out)) return 0;
    const int64_t lim = (int64_t)1 << 31;
    if (d->rows * d->ldx * 2 >= lim || d->rows * d->ldo * 2 >= lim || (d->residual && d->rows * d->ld_res * 2 >= lim)) return 0;
    if (d->dtype != AA_F16 && d->dtype != AA_BF16) return 0;
    if (d->row_affine && (d->rows_per_group <= 0 || d->rows_per_group % 32)) return 0;
    return 1;
}

int aa_linear_rows(const AaLinearRows* d, void* stream) {
    using namespace aa;
    if (!d) return fail(AA_E_SHAPE, "linear_rows: null descriptor");
    if (!aa_linear_rows_ok(d)) return fail(AA_E_SHAPE, "linear_rows: unsupported call (channels %d, n_out %d, rows %lld, dtype %d)", d->channels, d->n_out, (long long)d->rows, d->dtype);
    if (!d->x || !d->out || !d->w) return fail(AA_E_SHAPE, "linear_rows: x, out, w are required");
    if (!aligned16(d->x) || !aligned16(d->residual) || !aligned16(d->out) || !aligned16(d->w) || !aligned16(d->row_affine)) return fail(AA_E_ALIGN, "linear_rows: operands must be 16-byte aligned");
    if (d->normalize && !(d->ln_eps > 0.0f)) return fail(AA_E_SHAPE, "linear_rows: normalize needs ln_eps > 0");
    // Two workgroups per CU run in lock-step rounds of 512 tiles (139264 rows = 1088 tiles = 2.125 rounds: the last round would hold 64 tiles and
    // cost as much as a full one).  The tiles behind the last full round are split over their stages - n_split workgroups of nq / n_split
    // stages each, as many as still fit one round (flags bit 0: pretend a 2-CU chip - exercises the split in tests; bit 1: no split).
    const int tiles = (int)((d->rows + 32 * LR_NW - 1) / (32 * LR_NW)), nq = d->n_out / 32;
    const int slots = ((d->flags & 1) ? 2 : 256) * 2;
    int n_full = (d->flags & 2) ? tiles : tiles / slots * slots, n_split = 1;
    if (tiles - n_full > 0)
        for (int s = 2; s <= nq && (tiles - n_full) * s <= slots; ++s)
            if (nq % s == 0) n_split = s;
    if (n_split == 1) n_full = tiles;
    const dim3 grid((unsigned)(n_full + (tiles - n_full) * n_split)), block(64 * LR_NW);
    if (d->dtype == AA_F16) AA_LAUNCH((linear_rows_kernel<f16_t, 320>), grid, block, lr_lds_bytes(), stream, *d, n_full, n_split);
    else                    AA_LAUNCH((linear_rows_kernel<bf16_t, 320>), grid, block, lr_lds_bytes(), stream, *d, n_full, n_split);
    return finish("linear_rows");
}

int aa_softmax_rows(const float* x, void* y, int64_t rows, int32_t cols, int32_t x_ld, int32_t y_ld, int32_t dtype, void* stream) {
    using namespace aa;
    if (rows <= 0 || cols <= 0 || x_ld < cols || y_ld < cols) return fail(AA_E_SHAPE, "softmax_rows: bad shape");
    const dim3 grid((unsigned)rows), block(256);
    if (dtype == AA_F16) AA_LAUNCH((softmax_rows_kernel<f16_t>), grid, block, 64, stream, x, (f16_t*)y, cols, x_ld, y_ld);
    else if (dtype == AA_BF16) AA_LAUNCH((softmax_rows_kernel<bf16_t>), grid, block, 64, stream, x, (bf16_t*)y, cols, x_ld, y_ld);
    else return fail(AA_E_DTYPE, "softmax_rows: unsupported dtype %d", dtype);
    return finish("softmax_rows");
}

int aa_cfg_dpm_step(const AaDpmStep* d, void* stream) {
    using namespace aa;
    if (!d || d->n <= 0) return fail(AA_E_SHAPE, "cfg_dpm_step: bad size");
    int64_t blocks = (d->n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    const dim3 grid((unsigned)blocks), block(256);
    if (d->dtype == AA_F16) AA_LAUNCH((cfg_dpm_step_kernel<f16_t>), grid, block, 0, stream, *d);
    else if (d->dtype == AA_BF16) AA_LAUNCH((cfg_dpm_step_kernel<bf16_t>), grid, block, 0, stream, *d);
    else return fail(AA_E_DTYPE, "cfg_dpm_step: unsupported dtype %d", d->dtype);
    return finish("cfg_dpm_step");
}

int aa_timestep_embedding(const float* t, void* out, int32_t n, int32_t dim, int32_t dtype, void* stream) {
    using namespace aa;
    if (!t || !out || n <= 0 || dim <= 0 || dim % 2) return fail(AA_E_SHAPE, "timestep_embedding: n=%d dim=%d", n, dim);
    const dim3 grid((unsigned)((n * (dim / 2) + 255) / 256)), block(256);
    if (dtype == AA_F16) AA_LAUNCH((timestep_embed_kernel<f16_t>), grid, block, 0, stream, t, (f16_t*)out, n, dim);
    else if (dtype == AA_BF16) AA_LAUNCH((timestep_embed_kernel<bf16_t>), grid, block, 0, stream, t, (bf16_t*)out, n, dim);
    else return fail(AA_E_DTYPE, "timestep_embedding: unsupported dtype %d", dtype);
    return finish("timestep_embedding");
}

int aa_pack_latents(const AaPackLatents* d, void* stream) {
    using namespace aa;
    if (!d || !d->sample || !d->cond || !d->out) return fail(AA_E_SHAPE, "pack_latents: null operand");
    if (d->batch <= 0 || d->sample_batch <= 0 || d->cond_batch <= 0 || d->frames <= 0 || d->hw <= 0 || d->channels <= 0 ||
        d->channels + (d->mask ? 1 : 0) > 8 || (d->mask && d->mask_batch <= 0))
        return fail(AA_E_SHAPE, "pack_latents: bad geometry (batch=%d channels=%d frames=%d hw=%d)", d->batch, d->channels, d->frames, d->hw);
    if (!aligned16(d->out)) return fail(AA_E_ALIGN, "pack_latents: out must be 16-byte aligned");
    if (d->sample_dtype != AA_F32 && d->sample_dtype != d->dtype) return fail(AA_E_DTYPE, "pack_latents: sample dtype must be f32 or the storage dtype");
    int64_t blocks = ((int64_t)d->batch * (d->frames + 1) * d->hw + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    const dim3 grid((unsigned)blocks), block(256);
    if (d->dtype == AA_F16) AA_LAUNCH((pack_latents_kernel<f16_t>), grid, block, 0, stream, *d);
    else if (d->dtype == AA_BF16) AA_LAUNCH((pack_latents_kernel<bf16_t>), grid, block, 0, stream, *d);
    else return fail(AA_E_DTYPE, "pack_latents: unsupported dtype %d", d->dtype);
    return finish("pack_latents");
}

int aa_cfg_dpm_step_tokens(const AaDpmStepTok* d, void* stream) {
    using namespace aa;
    if (!d || !d->eps_tokens || !d->latents || !d->x0_prev) return fail(AA_E_SHAPE, "cfg_dpm_step_tokens: null operand");
    if (d->clips <= 0 || d->channels <= 0 || d->frames <= 0 || d->hw <= 0 || d->eps_ld < d->channels || d->next_t_count < 0 || d->next_t_count > 256)
        return fail(AA_E_SHAPE, "cfg_dpm_step_tokens: bad geometry");
    int64_t blocks = ((int64_t)d->clips * d->channels * d->frames * d->hw + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    const dim3 grid((unsigned)blocks), block(256);
    if (d->dtype == AA_F16) AA_LAUNCH((cfg_dpm_step_tok_kernel<f16_t>), grid, block, 0, stream, *d);
    else if (d->dtype == AA_BF16) AA_LAUNCH((cfg_dpm_step_tok_kernel<bf16_t>), grid, block, 0, stream, *d);
    else return fail(AA_E_DTYPE, "cfg_dpm_step_tokens: unsupported dtype %d", d->dtype);
    return finish("cfg_dpm_step_tokens");
}

int aa_blend(const AaBlend* d, void* stream) {
    using namespace aa;
    if (!d || !d->x || !d->out) return fail(AA_E_SHAPE, "blend: null operand");
    if (d->act < AA_ACT_NONE || d->act > AA_ACT_QUICK_GELU) return fail(AA_E_SHAPE, "blend: unknown activation %d", d->act);
    if (d->rows <= 0 || d->channels <= 0 || d->channels % 8 || (d->rowvec && d->rowvec_div <= 0) || d->rowvec_mod < 0 ||
        (d->rowvec_ld && d->rowvec_ld < d->channels) || d->rowvec_ld % 8)
        return fail(AA_E_SHAPE, "blend: bad geometry (rows=%lld channels=%d)", (long long)d->rows, d->channels);
    if (!aligned16(d->x) || !aligned16(d->out) || (d->y && !aligned16(d->y)) || (d->rowvec && !aligned16(d->rowvec)))
        return fail(AA_E_ALIGN, "blend: operands must be 16-byte aligned");
    int64_t blocks = (d->rows * (d->channels / 8) + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    const dim3 grid((unsigned)blocks), block(256);
    if (d->dtype == AA_F16) AA_LAUNCH((blend_kernel<f16_t>), grid, block, 0, stream, *d);
    else if (d->dtype == AA_BF16) AA_LAUNCH((blend_kernel<bf16_t>), grid, block, 0, stream, *d);
    else return fail(AA_E_DTYPE, "blend: unsupported dtype %d", d->dtype);
    return finish("blend");
}

int aa_pack_frames(const AaPackFrames* d, void* stream) {
    using namespace aa;
    if (!d || !d->out) return fail(AA_E_SHAPE, "pack_frames: null operand");
    int channels = 0;
    for (int s = 0; s < 3; ++s)
        if (d->src[s]) {
            if (d->src_channels[s] <= 0 || d->src_batch[s] <= 0) return fail(AA_E_SHAPE, "pack_frames: source %d has no channels / batch", s);
            channels += d->src_channels[s];
        }
    if (d->batch <= 0 || d->frames <= 0 || d->hw <= 0 || channels <= 0 || (d->out_channels != 8 && d->out_channels != 16) || channels > d->out_channels)
        return fail(AA_E_SHAPE, "pack_frames: bad geometry (batch=%d frames=%d hw=%d channels=%d -> %d)", d->batch, d->frames, d->hw, channels, d->out_channels);
    if (!aligned16(d->out)) return fail(AA_E_ALIGN, "pack_frames: out must be 16-byte aligned");
    int64_t blocks = ((int64_t)d->batch * d->frames * d->hw + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    const dim3 grid((unsigned)blocks), block(256);
    const bool wide = d->out_channels == 16;
    if (d->dtype == AA_F16) { if (wide) AA_LAUNCH((pack_frames_kernel<f16_t, 16>), grid, block, 0, stream, *d); else AA_LAUNCH((pack_frames_kernel<f16_t, 8>), grid, block, 0, stream, *d); }
    else if (d->dtype == AA_BF16) { if (wide) AA_LAUNCH((pack_frames_kernel<bf16_t, 16>), grid, block, 0, stream, *d); else AA_LAUNCH((pack_frames_kernel<bf16_t, 8>), grid, block, 0, stream, *d); }
    else return fail(AA_E_DTYPE, "pack_frames: unsupported dtype %d", d->dtype);
    return finish("pack_frames");
}

int aa_cfg_euler_step_tokens(const AaEulerStepTok* d, void* stream) {
    using namespace aa;
    if (!d || !d->v_tokens || !d->latents) return fail(AA_E_SHAPE, "cfg_euler_step_tokens: null operand");
    if (d->clips <= 0 || d->channels <= 0 || d->frames <= 0 || d->hw <= 0 || d->ld < d->channels || d->next_t_count < 0 || d->next_t_count > 256)
        return fail(AA_E_SHAPE, "cfg_euler_step_tokens: bad geometry");
    int64_t blocks = ((int64_t)d->clips * d->channels * d->frames * d->hw + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    const dim3 grid((unsigned)blocks), block(256);
    if (d->dtype == AA_F16) AA_LAUNCH((cfg_euler_step_tok_kernel<f16_t>), grid, block, 0, stream, *d);
    else if (d->dtype == AA_BF16) AA_LAUNCH((cfg_euler_step_tok_kernel<bf16_t>), grid, block, 0, stream, *d);
    else return fail(AA_E_DTYPE, "cfg_euler_step_tokens: unsupported dtype %d", d->dtype);
    return finish("cfg_euler_step_tokens");
}

}  // extern "C"
#endif  // AA_TU_TILES_ONLY
