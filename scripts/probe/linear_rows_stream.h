// PROTOTYPE (round 6, probe only - not part of the library): aa::linear_rows_kernel as ONE persistent workgroup per CU that walks row tiles,
// with the NEXT tile's x on its way through a second LDS area while the current tile is multiplied.  Question it answers: do the x fetch and the
// stage loop of linear_rows_kernel - which add up there (profiles/r06_linear_rows_probe.txt) - overlap when a workgroup prefetches?
// No residual (a residual fetched per stage would sit in the same in-order load queue with one stage of slack), no LayerNorm / GroupNorm step:
// timing of the 320 -> n_out projection without a residual only.
//   LDS: weight ring 3 x 21504 | 1 KiB dump | 4 output tiles x 2560 | 4 x-staging areas x 20480 = 157696 bytes: one workgroup per CU.
//   Per item (= row tile, all stages): x staging -> registers; stage 0 issues the x pieces of the NEXT item behind its weight pieces; the weight
//   pieces of stage s + 2 ride in stage s across item boundaries (the stream is the same for every tile), so only the first item has a cold start.
//   Loads return in order: the x pieces (HBM) sit between weight pieces (L2), so they have two stage times to land before a younger weight
//   piece is needed; items need >= 4 stages.
#pragma once
#include "kernels/linear_rows.h"

namespace aa {

__host__ __device__ constexpr int lrs_lds_bytes() { return LR_RING * LR_STAGE_BYTES + 1024 + 4 * LR_OT_BYTES + 4 * LR_X_BYTES; }

template <typename T, int C, int DEPTH = 3>
__global__ void __launch_bounds__(256, 1) linear_rows_stream_kernel(const AaLinearRows p) {
    static_assert(C == 320, "one stage = five 32 x 64 chunks");
    constexpr int NW = 4, NKS = C / 16, PPW = 5;
    constexpr unsigned OOB = 0x80000000u;
    char* ring = dyn_smem();
    char* dump = ring + LR_RING * LR_STAGE_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = wave_id();
    char* otile = dump + 1024 + wave * LR_OT_BYTES;
    char* xs = dump + 1024 + NW * LR_OT_BYTES + wave * LR_X_BYTES;
    const int c = lane & 31, h = lane >> 5;
    const int nq = p.n_out / 32;
    const int tiles = (int)((p.rows + 32 * NW - 1) / (32 * NW));

    const BufRsrc r_x = make_rsrc(p.x, (unsigned)(p.rows * p.ldx * 2));
    const BufRsrc r_o = make_rsrc(p.out, (unsigned)(p.rows * p.ldo * 2));
    const BufRsrc r_w = make_rsrc(p.w, (unsigned)(nq * LR_STAGE_BYTES));
    const unsigned lane16 = (unsigned)(lane * 16);

    unsigned wa[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) wa[v] = (unsigned)(c * 128 + ((((v >> 1) * 4 + 2 * h + (v & 1)) ^ ((c >> 1) & 7)) << 4));
    const unsigned wb = (unsigned)(LR_BIAS_OFF + c * 16);
    const u32x4 xone = u32x4{h == 0 ? ones_pair(T()) : 0u, 0u, 0u, 0u};
    f32x16 zero16;
#pragma unroll
    for (int e = 0; e < 16; ++e) zero16[e] = 0.0f;

    // the 20 x pieces of this wave's 32 rows of `tile` (a tile behind the last one: pieces that fetch nothing keep the counts uniform)
    auto x_pieces = [&](int tile) __attribute__((always_inline)) {
        const int r8 = lane >> 3, s8 = lane & 7;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int64_t prow = (int64_t)tile * (32 * NW) + 32 * wave + 8 * rg + r8;
            const unsigned lo = (tile < tiles && prow < p.rows) ? (unsigned)(prow * p.ldx * 2) + (unsigned)((s8 ^ ((4 * rg + (r8 >> 1)) & 7)) << 4) : OOB;
#pragma unroll
            for (int ch = 0; ch < 5; ++ch) async_copy16_buf_s(r_x, lo, (unsigned)(ch * 128), xs + ch * LR_CHUNK_BYTES + rg * 1024);
        }
    };
    // piece j of this wave for weight stage qw into ring slot `slot` (real = false: fetches nothing)
    auto piece = [&](int qw, int slot, int j, bool real) __attribute__((always_inline)) {
        const int pi = wave + NW * j;
        async_copy16_buf_s(r_w, real ? lane16 : OOB, (unsigned)((real ? qw : 0) * LR_STAGE_BYTES + pi * 1024), ring + slot * LR_STAGE_BYTES + pi * 1024);
    };
    auto bias_piece = [&](int qw, int slot, bool real) __attribute__((always_inline)) {
        async_copy16_buf_s(r_w, real ? lane16 : OOB, (unsigned)((real ? qw : 0) * LR_STAGE_BYTES + 20 * 1024), ring + slot * LR_STAGE_BYTES + 20 * 1024);
    };

    int item = (int)blockIdx.x;
    if (item >= tiles) return;
    // cold start: x of the first item, then the weight pieces of its stages 0 and 1
    x_pieces(item);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int j = 0; j < PPW; ++j) piece(s, s, j, true);
        if (wave == 0) bias_piece(s, s, true);
    }
    if (wave == 0) dma_wait<2 * (PPW + 1)>(); else dma_wait<2 * PPW>();           // x has landed (the two stages may still be in flight)

    int sc = 0;                                       // running stage counter: ring slot = sc % 3
    for (; item < tiles; item += (int)gridDim.x) {
        const int next = item + (int)gridDim.x;
        const bool has_next = next < tiles;
        // ---- x: staging area -> registers (this wave's own area; the x pieces of the next item overwrite it at the end of stage 0)
        wave_sync();
        u32x4 xf[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) lds_read16_async(xf[ks], xs + (ks >> 2) * LR_CHUNK_BYTES + wa[ks & 3]);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) lds_wait<0>(xf[ks]);
        wave_sync();
        const int64_t row2 = (int64_t)item * (32 * NW) + 32 * wave + (lane >> 2);
        unsigned ob2[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) ob2[j] = row2 + 16 * j < p.rows ? (unsigned)((row2 + 16 * j) * p.ldo * 2) + (unsigned)(16 * (lane & 3)) : OOB;
        char* ot_w = otile + c * LR_OT_STRIDE + 32 * h;
        const char* ot_r = otile + (lane >> 2) * LR_OT_STRIDE + 16 * (lane & 3);

        for (int q = 0; q < nq; ++q, ++sc) {
            // stage top: the pieces of this stage have landed.  Younger loads that may be in flight: the pieces of the next stage, and - at the top of
            // stages 1 and 2 - the 20 x pieces issued at the end of stage 0
            if (q == 1 || q == 2) { if (wave == 0) dma_wait<PPW + 1 + 20>(); else dma_wait<PPW + 20>(); }
            else { if (wave == 0) dma_wait<PPW + 1>(); else dma_wait<PPW>(); }
            block_barrier();
            const char* st = ring + (sc % LR_RING) * LR_STAGE_BYTES;
            const char* b4[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) b4[v] = st + wa[v];
            const char* bb = st + wb;
            // the weight stage two ahead: of this item, or of the next one (the same stream), or nothing behind the last item
            const int q2 = q + 2 < nq ? q + 2 : q + 2 - nq;
            const bool real2 = q + 2 < nq || has_next;
            const int slot2 = (sc + 2) % LR_RING;
            u32x4 wf[DEPTH + 1];
            auto rd = [&](auto u_) __attribute__((always_inline)) {
                constexpr int u = decltype(u_)::value, ch = u >> 2, v = u & 3, set = u % (DEPTH + 1);
                if constexpr (u < NKS) lds_read16_async_off<ch * LR_CHUNK_BYTES>(wf[set], b4[v]);
                else lds_read16_async_off<0>(wf[set], bb);
            };
            static_for<DEPTH>([&](auto u_) __attribute__((always_inline)) { rd(u_); });
            f32x16 acc;
            static_for<NKS + 1>([&](auto u_) __attribute__((always_inline)) {
                constexpr int u = decltype(u_)::value, set = u % (DEPTH + 1);
                if constexpr (u + DEPTH < NKS + 1) { rd(IntTag<u + DEPTH>()); lds_wait<DEPTH>(wf[set]); }
                else lds_wait<NKS - u>(wf[set]);
                if constexpr (u == 0) acc = mfma_32x32x16(T(), wf[set], xf[0], zero16);
                else if constexpr (u < NKS) acc = mfma_32x32x16(T(), wf[set], xf[u], acc);
                else acc = mfma_32x32x16(T(), wf[set], xone, acc);
                sched_fence();
                if constexpr ((u & 3) == 3 && u < NKS) piece(q2, slot2, u >> 2, real2);
                else if constexpr (u == 1) {
                    if (wave == 0) bias_piece(q2, slot2, real2);
                }
                sched_fence();
            });
            if (q == 0) x_pieces(next);               // (behind this stage's weight pieces: see the waits at the top of stages 1 .. 3)
            // rounded, through the wave's LDS tile, stored four lanes to a row
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
            wave_sync();
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) buf_store16(r_o, ob2[j2] + (unsigned)(64 * q), o[j2]);
        }
    }
    dma_wait<0>();
}

}  // namespace aa
