// A linear layer over 320-channel token rows with the rows in REGISTERS (see include/aa_mi355.h: aa_linear_rows):
//     out = [LayerNorm](x) W^T + b (+ residual)
// the K = C projections of the 320-channel transformers (diffusers Transformer2DModel / TransformerTemporalModel.proj_in, Attention.to_q / to_out[0],
// the Q|K|V projection of the spatial self-attention; reference models/unet_3d_blocks.py:287,446,681 and :379,526,759 via diffusers).  As contractions
// of the tile family these calls were bound by per-tile fixed costs (a 192 x 320 tile with K = 320 is five K steps between a cold prologue and an
// epilogue that owns the CU alone: 67 us for 267 MB at the 64x64 level, 42 us at the HBM rate).  This kernel is the first phase of ff_fused.h on
// its own, at TWO workgroups per CU:
//  * a 4-wave workgroup owns 128 token rows, wave w rows 32 w .. + 31: x = 80 registers in MFMA operand layout, fetched once (through LDS: whole
//    cache lines), optionally normalised in place (fp32 statistics; gamma / beta folded into W / b by the host) - no row statistics have to come
//    from the producer;
//  * the weights stream L2 -> LDS as host-packed stage images of 32 output channels x all 320 K (20 KB, XOR-swizzled, + the 32 bias rows as (hi, lo)
//    for the matrix pipe: a 21st k-slice), three-stage ring, counted vmcnt, one raw barrier per stage; every stage finishes 32 output channels of the
//    wave's 32 tokens: + residual, rounded, stored - no accumulators live across stages (176 registers, 80 KB of LDS: two workgroups per CU);
//  * the tiles behind the last full round of the chip are split over their stages (aa_linear_rows: n_full, n_split).
// Measured (profiles/r06_linear_rows_probe.txt, 139264 rows): 320 -> 320 56-60 us without / 65 with a residual (tile family 69 / 83), 320 -> 960
// 134 (175); the x fetch, the stores and the multiply phase of a workgroup add up instead of overlapping (x only 27 us, + stores 39, + weights and
// MFMAs 48 on constant data), and neither three waves per SIMD, nor later stores, nor workgroups started half a life apart changed that.
#pragma once
#include "dev.h"
#include "aa_mi355.h"

namespace aa {

constexpr int LR_NW = 4;
constexpr int LR_CHUNK_BYTES = 4096;              // [32 rows][64 K]: 128-byte rows, 16-byte slots XOR-swizzled with (row >> 1) & 7
constexpr int LR_BIAS_OFF = 5 * LR_CHUNK_BYTES;   // [32 rows][8]: (bias hi, bias lo, 0 x 6), then 512 bytes of zeros
constexpr int LR_STAGE_BYTES = LR_BIAS_OFF + 1024;
constexpr int LR_RING = 3;
constexpr int LR_OT_STRIDE = 80;                  // a row of a wave's [32 tokens][32 channels] output tile in LDS: 64 bytes + 16 (bank spread)
constexpr int LR_OT_BYTES = 32 * LR_OT_STRIDE;
constexpr int LR_X_BYTES = 20480;                 // a wave's 32 rows x 320 channels on their way to the registers: five [32][64] chunks like a weight stage's
// the weight ring + 1 KiB that takes the pieces that fetch nothing (NW not a divisor of 20); in front of the first stage the same memory stages x
// ... behind it the waves' output tiles (the epilogue's transposition)
__host__ __device__ constexpr int lr_lds_bytes(int nw = LR_NW) {
    return nw * LR_X_BYTES > LR_RING * LR_STAGE_BYTES + 1024 + nw * LR_OT_BYTES ? nw * LR_X_BYTES : LR_RING * LR_STAGE_BYTES + 1024 + nw * LR_OT_BYTES;
}

// VAR: timing-only forms for scripts/probe/linear_rows_probe.hip (bit 0: no output stores, bit 1: the weight pieces fetch nothing, bit 2: no fragment
// reads and no MFMAs); bit 3: stores / residual loads a token per lane; bit 4: x straight into registers; bit 5: the split workgroups last in the grid.  (Measured and removed: a stage's stores at the top of the next stage, 192-row workgroups at three waves per SIMD, half of the first round started late - profiles/r06_linear_rows_probe.txt.)
template <typename T, int C, int NW = LR_NW, int VAR = 0>
__global__ void __launch_bounds__(64 * NW, NW / 2) linear_rows_kernel(const AaLinearRows p, const int n_full, const int n_split) {
    static_assert(C == 320, "one stage = five 32 x 64 chunks: 320 channels");
    constexpr int NKS = C / 16;
    constexpr bool EXTRA = 20 % NW == 0;          // the 20 weight KB split evenly over the waves, wave 0 adds the bias KB; otherwise all 21 KB round-robin
    constexpr int PPW = EXTRA ? 20 / NW : (21 + NW - 1) / NW;     // LDS-DMA instructions per stage and wave (1 KiB each: 8 rows of a chunk)
    static_assert(PPW <= 5, "the pieces of stage q + 2 ride behind every fourth of the 20 k-slices");
    constexpr unsigned OOB = 0x80000000u;
    char* ring = dyn_smem();
    char* dump = ring + LR_RING * LR_STAGE_BYTES;
    char* otile = dump + 1024 + wave_id() * LR_OT_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = wave_id();
    const int c = lane & 31, h = lane >> 5;
    // workgroups < n_full own a row tile and all its output channels; the others - the tiles of the last, partly filled round of the chip - a tile
    // and 1 / n_split of its stages each (x is then fetched n_split times, from L2): the tail of the launch is n_split times shorter
    // The split workgroups come FIRST in the grid: they are short, so the full workgroups that take over their slots run half a workgroup's life
    // behind the ones that started at 0 - otherwise all resident workgroups fetch x at the same time (HBM saturated, matrix pipe idle) and then
    // all multiply at the same time (HBM idle), round after round.
    const int n_tail = ((int)gridDim.x - n_full);
    const int over = (VAR & 32) ? (int)blockIdx.x - n_full : (int)blockIdx.x < n_tail ? (int)blockIdx.x : -1;       // (VAR bit 5: the split workgroups last)
    const int tile = over < 0 ? ((VAR & 32) ? (int)blockIdx.x : (int)blockIdx.x - n_tail) : n_full + over / n_split;
    const int64_t row = (int64_t)tile * (32 * NW) + 32 * wave + c;
    const bool row_ok = row < p.rows;
    const int nq_all = p.n_out / 32;              // stages per tile
    const int q0 = over < 0 ? 0 : (over % n_split) * (nq_all / n_split);
    const int nq = over < 0 ? nq_all : q0 + nq_all / n_split;          // this workgroup's stages: q0 .. nq - 1

    const BufRsrc r_x = make_rsrc(p.x, (unsigned)(p.rows * p.ldx * 2));
    const BufRsrc r_res = make_rsrc(p.residual, p.residual ? (unsigned)(p.rows * p.ld_res * 2) : 0u);
    const BufRsrc r_o = make_rsrc(p.out, (unsigned)(p.rows * p.ldo * 2));
    const BufRsrc r_w = make_rsrc(p.w, (unsigned)(nq_all * LR_STAGE_BYTES));
    const unsigned lane16 = (unsigned)(lane * 16);

    // fragment addresses inside a [32][64] chunk (weights and x alike): row c, slot (4 nbl + 2 h + s) ^ ((c >> 1) & 7): one register per k-slice of the chunk
    unsigned wa[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) wa[v] = (unsigned)(c * 128 + ((((v >> 1) * 4 + 2 * h + (v & 1)) ^ ((c >> 1) & 7)) << 4));

    // ---- x first.  A lane's fragments are 16 bytes of ITS row per k-slice: fetched straight into registers every load instruction touches 32 cache
    // lines for 1 KiB (measured: 89 MB in 30 us, the L1 refetches every line for each of its four k-slices).  Instead the wave's rows go through LDS
    // like the weights: 20 LDS-DMA pieces of 8 rows x 128 bytes (whole cache lines; 16-byte slots XOR-swizzled on the source side), then 20
    // conflict-free fragment reads.  The staging area is the memory of the weight ring, which is idle until x sits in the registers.
    u32x4 xf[NKS];
    if constexpr ((VAR & 16) != 0) {
        const unsigned xb = row_ok ? (unsigned)(row * p.ldx * 2) + (unsigned)(16 * h * 2) : OOB;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) xf[ks] = buf_load16(r_x, xb + (unsigned)((32 * (ks >> 1) + 8 * (ks & 1)) * 2));
    } else {
        char* xs = ring + wave * LR_X_BYTES;
        const int r8 = lane >> 3, s8 = lane & 7;                     // piece (rg, ch): lane = (row 8 rg + r8, slot s8) of the chunk
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int64_t prow = (int64_t)tile * (32 * NW) + 32 * wave + 8 * rg + r8;
            const unsigned lo = prow < p.rows ? (unsigned)(prow * p.ldx * 2) + (unsigned)((s8 ^ ((4 * rg + (r8 >> 1)) & 7)) << 4) : OOB;
#pragma unroll
            for (int ch = 0; ch < 5; ++ch) async_copy16_buf_s(r_x, lo, (unsigned)(ch * 128), xs + ch * LR_CHUNK_BYTES + rg * 1024);
        }
        dma_wait<0>();
        wave_sync();
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) lds_read16_async(xf[ks], xs + (ks >> 2) * LR_CHUNK_BYTES + wa[ks & 3]);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) lds_wait<0>(xf[ks]);
        block_barrier();                                             // (the ring's first stages overlap other waves' staging areas)
    }
    // ---- the weight stream: stage q = KB 21 q .. of `w`; piece pi = wave + 4 j (j < 5) is KB pi of the stage, wave 0 adds KB 20 (the bias rows)
    // piece j of this wave for stage q: KB wave + NW j of the stage (behind the last stage, and KB >= 21: a piece that fetches nothing keeps the counts uniform)
    auto piece = [&](int q, int j) __attribute__((always_inline)) {
        const int pi = wave + NW * j;
        const bool real = q < nq && pi < 21 && !(VAR & 2);
        async_copy16_buf_s(r_w, real ? lane16 : OOB, (unsigned)((real ? q : 0) * LR_STAGE_BYTES + (pi < 21 ? pi : 0) * 1024),
                           pi < 21 ? ring + (q % LR_RING) * LR_STAGE_BYTES + pi * 1024 : dump);
    };
    auto bias_piece = [&](int q) __attribute__((always_inline)) {
        const bool real = q < nq && !(VAR & 2);
        async_copy16_buf_s(r_w, real ? lane16 : OOB, (unsigned)((real ? q : 0) * LR_STAGE_BYTES + 20 * 1024), ring + (q % LR_RING) * LR_STAGE_BYTES + 20 * 1024);
    };
    auto stage_pieces = [&](int q) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < PPW; ++j) piece(q, j);
        if (EXTRA && wave == 0) bias_piece(q);
    };
    stage_pieces(q0);
    stage_pieces(q0 + 1);

    // the bias k-slice's B operand: k = 0, 1 (lanes h = 0) multiply the rows' (hi, lo)
    const u32x4 xone = u32x4{h == 0 ? ones_pair(T()) : 0u, 0u, 0u, 0u};
    const unsigned wb = (unsigned)(LR_BIAS_OFF + c * 16);

    // ---- a GroupNorm in front (aa_groupnorm_coef): x <- x * scale[g] + shift[g] per channel, rounded to the storage type - what groupnorm_apply_kernel
    // would have written and this kernel read back.  A wave's 32 rows lie in one group (rows_per_group % 32 == 0): the coefficients are wave-uniform
    // (scalar loads); lane half h holds channels 32 (ks >> 1) + 8 (ks & 1) + 16 h .. + 7 of k-slice ks.
    if (p.row_affine) {
        const int64_t r0 = (int64_t)tile * (32 * NW) + 32 * wave;                               // (blockIdx and wave_id(): uniform)
        const int64_t grp = (r0 < p.rows ? r0 : p.rows - 1) / p.rows_per_group;                  // (a wave behind the last row reads the last group's, never past the table)
        const float* cf = p.row_affine + grp * 2 * C;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            Pack8<T> v; v.raw = xf[ks];
            const int c0 = 32 * (ks >> 1) + 8 * (ks & 1);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float sc = h ? cf[c0 + 16 + e] : cf[c0 + e], sh = h ? cf[C + c0 + 16 + e] : cf[C + c0 + e];
                v.e[e] = (T)((float)v.e[e] * sc + sh);
            }
            xf[ks] = v.raw;
        }
    }
    // ---- x~ = (x - mean) * rstd in place (LayerNorm's gamma / beta live in W / b)
    if (p.normalize) {
        float sum = 0.0f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int k = 0; k < 4; ++k) sum = dot2_f32(T(), xf[ks][k], ones_pair(T()), sum);
        const float mean = wave_sum_halves(sum) * (1.0f / (float)C);
        const f32x2 nmean2 = f32x2{-mean, -mean};
        f32x2 var2 = f32x2{0.0f, 0.0f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            Pack8<T> v; v.raw = xf[ks];
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const f32x2 d = f32x2{(float)v.e[e], (float)v.e[e + 1]} + nmean2;
                var2 = __builtin_elementwise_fma(d, d, var2);
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum_halves(var2[0] + var2[1]) * (1.0f / (float)C) + p.ln_eps);
        const f32x2 rstd2 = f32x2{rstd, rstd}, off2 = f32x2{-mean * rstd, -mean * rstd};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            Pack8<T> v; v.raw = xf[ks];
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const f32x2 y = __builtin_elementwise_fma(f32x2{(float)v.e[e], (float)v.e[e + 1]}, rstd2, off2);
                v.e[e] = (T)y[0]; v.e[e + 1] = (T)y[1];
            }
            xf[ks] = v.raw;
        }
    }

    f32x16 zero16;
#pragma unroll
    for (int e = 0; e < 16; ++e) zero16[e] = 0.0f;
    // Output (and residual) addressing.  The accumulators hold a token per lane (16 channels): stored from there every lane of a store instruction
    // writes 16 bytes of ANOTHER row - 64 address cycles per instruction, the kernel's stores (and residual loads) were bound by that, not by
    // bytes (139264 x 320: + 12 us each).  Instead a stage's [32 tokens][32 channels] go through a wave-private LDS tile, rounded, and come back
    // four lanes to a row: lane l = (row l >> 2 (+ 16 j), 16 bytes l & 3) - a quarter of the address cycles; the residual is fetched in the
    // same layout and added to the ROUNDED product (the order of the reference: to_out's output is a tensor of the storage type, then + residual).
    // (VAR bit 3: the token-per-lane form.)
    constexpr bool ROWMAJOR = !(VAR & 8);
    const int64_t row2 = (int64_t)tile * (32 * NW) + 32 * wave + (lane >> 2);          // (+ 16 j)
    unsigned ob, rb, ob2[2], rb2[2];
    if constexpr (ROWMAJOR) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool ok = row2 + 16 * j < p.rows;
            ob2[j] = ok ? (unsigned)((row2 + 16 * j) * p.ldo * 2) + (unsigned)(16 * (lane & 3)) : OOB;
            rb2[j] = (ok && p.residual) ? (unsigned)((row2 + 16 * j) * p.ld_res * 2) + (unsigned)(16 * (lane & 3)) : OOB;
        }
    } else {
        ob = row_ok ? (unsigned)(row * p.ldo * 2) + (unsigned)(16 * h * 2) : OOB;
        rb = (row_ok && p.residual) ? (unsigned)(row * p.ld_res * 2) + (unsigned)(16 * h * 2) : OOB;
    }
    char* ot_w = otile + c * LR_OT_STRIDE + 32 * h;                                      // this lane's 2 x 16 bytes of the tile (t = 0, 1: + 16 t)
    const char* ot_r = otile + (lane >> 2) * LR_OT_STRIDE + 16 * (lane & 3);             // (+ 16 j rows)

    for (int q = q0; q < nq; ++q) {
        // stage top: my pieces of stage q have landed (loads return in order: at most the PPW (+ 1) pieces of stage q + 1 - the newest loads - may be
        // outstanding; stores in flight can only make this wait longer), everyone's have, every wave is done with stage q - 1 whose buffer takes q + 2
        if (EXTRA && wave == 0) dma_wait<PPW + 1>(); else dma_wait<PPW>();
        block_barrier();
        // this stage's residual (lane (token c, h): channels 32 q + 16 h .. + 15), in front of the younger pieces
        u32x4 res[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if constexpr (ROWMAJOR) res[t] = buf_load16(r_res, rb2[t] + (unsigned)(64 * q));                     // (no residual: zeros)
            else res[t] = buf_load16(r_res, rb + (unsigned)((32 * q + 8 * t) * 2));
        }
        const char* st = ring + (q % LR_RING) * LR_STAGE_BYTES;
        const char* b4[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) b4[v] = st + wa[v];
        const char* bb = st + wb;
        u32x4 wf[4];
        auto rd = [&](auto u_) __attribute__((always_inline)) {
            constexpr int u = decltype(u_)::value, ch = u >> 2, v = u & 3, set = u & 3;
            if constexpr (u < NKS) lds_read16_async_off<ch * LR_CHUNK_BYTES>(wf[set], b4[v]);
            else lds_read16_async_off<0>(wf[set], bb);
        };
        if constexpr (!(VAR & 4)) { rd(IntTag<0>()); rd(IntTag<1>()); rd(IntTag<2>()); }
        f32x16 acc;
        if constexpr ((VAR & 4) != 0) acc = zero16;
        static_for<NKS + 1>([&](auto u_) __attribute__((always_inline)) {
            constexpr int u = decltype(u_)::value, set = u & 3;
            if constexpr (!(VAR & 4)) {
                if constexpr (u + 3 < NKS + 1) { rd(IntTag<u + 3>()); lds_wait<3>(wf[set]); }
                else if constexpr (u + 2 < NKS + 1) lds_wait<2>(wf[set]);
                else if constexpr (u + 1 < NKS + 1) lds_wait<1>(wf[set]);
                else lds_wait<0>(wf[set]);
                if constexpr (u == 0) acc = mfma_32x32x16(T(), wf[set], xf[0], zero16);
                else if constexpr (u < NKS) acc = mfma_32x32x16(T(), wf[set], xf[u], acc);
                else acc = mfma_32x32x16(T(), wf[set], xone, acc);
            } else if constexpr (u < NKS) asm volatile("" ::"v"(xf[u]));
            sched_fence();
            // the pieces of stage q + 2, one behind every fourth k-slice (wave 0: the bias KB behind the second)
            if constexpr ((u & 3) == 3 && u < NKS && (u >> 2) < PPW) piece(q + 2, u >> 2);
            else if constexpr (u == 1 && EXTRA) {
                if (wave == 0) bias_piece(q + 2);
            }
            sched_fence();
        });
        // rounded, (+ residual), stored: register 8 t + e = channel 32 q + 16 h + 8 t + e of the lane's token (the host packs the rows in that order)
        if constexpr (ROWMAJOR) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                Pack8<T> v;
#pragma unroll
                for (int e = 0; e < 8; ++e) v.e[e] = (T)acc[8 * t + e];
                lds_write16_async(ot_w + 16 * t, v.raw);
            }
            wave_sync();
            u32x4 o[2];
            lds_read16_async(o[0], ot_r);
            lds_read16_async(o[1], ot_r + 16 * LR_OT_STRIDE);
            lds_wait<1>(o[0]);
            lds_wait<0>(o[1]);
            wave_sync();                                                 // (emulator: every lane has read before the next stage's writes)
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) {
                Pack8<T> r, v;
                r.raw = res[j2]; v.raw = o[j2];
                if (p.residual) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v.e[e] = (T)((float)v.e[e] + (float)r.e[e]);
                }
                if constexpr (!(VAR & 1)) buf_store16(r_o, ob2[j2] + (unsigned)(64 * q), v.raw);
                else asm volatile("" ::"v"(v.raw));
            }
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                Pack8<T> r, v;
                r.raw = res[t];
#pragma unroll
                for (int e = 0; e < 8; ++e) v.e[e] = (T)(acc[8 * t + e] + (float)r.e[e]);
                if constexpr (!(VAR & 1)) buf_store16(r_o, ob + (unsigned)((32 * q + 8 * t) * 2), v.raw);
                else asm volatile("" ::"v"(v.raw));
            }
        }
    }
    dma_wait<0>();                  // (the last two stages' pieces fetched nothing, but they do write LDS)
}

}  // namespace aa
