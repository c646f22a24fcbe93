// Self-attention over SHORT sequences with the projections inside (see include/aa_mi355.h: aa_seq_self_attention):
//     O = softmax(q k^T * scale) v,   [q | k | v] = LayerNorm(x) [Wq | Wk | Wv]^T     (LayerNorm's gamma / beta folded into the packed weights / a bias)
// for sequences of at most 32 positions - the frames of one pixel in diffusers' TransformerTemporalModel
// (reference models/unet_3d_blocks.py:379,526,759; models/unet_3d_condition_mask.py:433-437).  Q, K and V never
// leave the chip: at the 64x64 level the three-launch form (Q|K|V contraction 158 us, 17-key attention 89 us) wrote and
// re-read 267 MB per call for 0.06 TFLOP of attention.
//
// A workgroup owns P = floor(32 NW / L) whole sequences = 32 NW tile rows in (sequence, position) order; wave w owns rows
// 32 w .. 32 w + 31 (one MFMA row block) and ALL output columns:
//  * x: the wave's 32 rows are fetched ONCE, straight from global memory into registers, already in MFMA operand layout
//    (lane (row c, half h) holds channels 32 nb + 16 h + 8 s .. + 7 as k-slice (nb, s)), normalised in place (fp32
//    statistics over the lane's half row + one v_permlane32_swap) and kept for all heads: C / 4 registers per lane;
//  * weights: per head three passes - the 64 to_k rows, the 64 to_v rows, the 64 to_q rows - stream L2 -> LDS by LDS-DMA
//    in stages of [64 rows][160 / 256 / 320 K] through a two-stage ring (one barrier per stage); every wave reads every weight
//    fragment (1.0 LDS fragment reads per MFMA = half of the LDS rate the matrix pipe would need at full speed);
//  * K and Q come out of the TRANSPOSED product (weights = MFMA A operand): lane (token, h) holds 16 d of its token - as
//    they stand (rounded to the storage type, 8 registers of 2) the A operand (K) / B operand (Q) of S^T = K Q^T;
//    V comes out of the STRAIGHT product (activations = A operand): lane (d, h) holds 16 tokens of its column - as they
//    stand the A operand V^T of O^T = V^T P^T.  No transposing read anywhere;
//  * sequences straddle the waves' row blocks, so K / V^T operand registers are parked in LDS (8 KB per block) and wave w
//    multiplies its queries against blocks w-1, w, w+1 (L <= 32); keys of other sequences are masked in the softmax
//    (one pass: every key is present; base 2, fp32);
//  * O^T leaves as 16 consecutive channels per lane (the to_v rows of a head are permuted at pack time).
// C = 320: 4-wave workgroups (7 sequences of 17 = 119 of 128 rows), TWO per CU (256 registers per lane, 72 KB of LDS each): they drift out
// of phase, so one's x fetch / softmax / stores run under the other's MFMAs, and 1171 half-size tiles fill three rounds of 512 slots where
// 547 eight-wave tiles left the third round of 256 slots 14 % full (first form, r06a: 167 us per call at the 64x64 level; the three-launch
// form 262).  C = 512 / 640: 4 waves, one per SIMD (x alone is 128 / 160 registers).
#pragma once
#include "dev.h"
#include "aa_mi355.h"

namespace aa {

// Weight stages in LDS: chunks of [64 rows][KC K] (row = 2 KC bytes, its 16-byte slots XOR-swizzled with the row's 256-byte bank row so
// that the fragment read - row = lane & 31, one slot - is conflict-free), SC chunks per stage, two stages.
constexpr int SA_NW = 4;
__host__ __device__ constexpr int sa_chunk_k(int channels) { return channels == 320 ? 32 : 64; }
__host__ __device__ constexpr int sa_stage_chunks(int channels) { return channels % 320 == 0 ? 5 : 4; }
__host__ __device__ constexpr int sa_per_cu(int channels) { return channels == 320 ? 2 : 1; }
__host__ __device__ constexpr int sa_lds_bytes(int channels) { return 2 * sa_stage_chunks(channels) * 128 * sa_chunk_k(channels) + SA_NW * 8192; }

// first row of sequence `seq` in x / o (rows of the token matrix)
__device__ __forceinline__ int64_t sa_seq_row(const AaSeqSelfAttn& p, int seq) {
    const int o = seq / p.n_inner, i = seq - o * p.n_inner;
    return (int64_t)o * p.outer_stride + (int64_t)i * p.inner_stride;
}

template <typename T, int C, int KC = sa_chunk_k(C), int SC = sa_stage_chunks(C), int PER_CU = sa_per_cu(C)>
__global__ void __launch_bounds__(64 * SA_NW, PER_CU) seq_self_attention_kernel(const AaSeqSelfAttn p) {
    constexpr int NW = SA_NW;
    constexpr int H = C / 64, NKS = C / 16, NKC = C / KC, SPP = NKC / SC;
    static_assert(C % 64 == 0 && NKC % SC == 0 && (KC == 64 || KC == 32), "channels");
    constexpr int CHUNK_BYTES = 128 * KC;       // 64 rows
    constexpr int RB = 2 * KC;                  // bytes per chunk row
    constexpr int SPR = KC / 8;                 // 16-byte slots per row
    constexpr int RPB = 256 / RB;               // rows per 256-byte bank row
    constexpr int RPP = 1024 / RB;              // rows per LDS-DMA piece (1 KiB)
    constexpr int PPC = 64 / RPP;               // pieces per chunk
    constexpr int SPC = KC / 16;                // k-slices per chunk
    constexpr int STAGE_BYTES = SC * CHUNK_BYTES;
    constexpr int PIECES = SC * PPC, PPW = PIECES / NW;          // LDS-DMA instructions per stage, per wave
    static_assert(PIECES % NW == 0, "DMA share");
    constexpr unsigned OOB = 0x80000000u;
    char* ring = dyn_smem();
    char* xch = ring + 2 * STAGE_BYTES;         // [NW row blocks][8 slices][64 lanes][16 B]: slices 0..3 K operands (k = d), 4..7 V^T operands (d block, key slice)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = wave_id();
    const int c = lane & 31, h = lane >> 5;
    const int L = p.seq_len;
    const int P = (32 * NW) / L;
    const int n_seq = p.n_outer * p.n_inner;
    const int s0 = blockIdx.x * P;
    const int rows_valid = min(P, n_seq - s0) * L;
    const bool wave_live = 32 * wave < rows_valid;

    // ---- this lane's row: sequence, position, token row
    const int r = 32 * wave + c;
    const bool row_ok = r < rows_valid;
    const int rs = r / L, rp = r - rs * L;
    const int64_t trow = row_ok ? sa_seq_row(p, s0 + rs) + (int64_t)rp * p.pos_stride : 0;
    // keys of this lane's sequence, relative to the first row of block wave-1 and to the lane's half: key register e of block n
    // (n = 0, 1, 2 = blocks wave-1, wave, wave+1) sits at 32 n + (e & 3) + 8 (e >> 2) + 4 h of that scale
    const int key_lo = rs * L - 32 * (wave - 1) - 4 * h;
    const bool from_prev = rs * L < 32 * wave;              // the lane's sequence starts in the block in front (else it may run into the block behind)
    const int key_lo_side = from_prev ? key_lo : key_lo - 64;

    const BufRsrc r_x = make_rsrc(p.x, (unsigned)p.x_bytes);
    const BufRsrc r_o = make_rsrc(p.o, (unsigned)p.o_bytes);
    const BufRsrc r_w = make_rsrc(p.w, (unsigned)((int64_t)H * 3 * 64 * C * 2));

    // ---- weight DMA: piece pi = wave + NW j of a stage covers rows RPP (pi % PPC) .. of chunk pi / PPC; lane: row (lane / SPR), slot (lane % SPR)
    unsigned w_lane[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int pi = wave + NW * j, kc = pi / PPC, rg = pi % PPC;
        const int row = RPP * rg + lane / SPR;
        const int slot = (lane % SPR) ^ ((row / RPB) & (SPR - 1));
        w_lane[j] = (unsigned)((row * C + kc * KC + slot * 8) * 2);
    }
    // Stages: first the NPRE stages of the projection in front (pre_w: C / 64 passes of 64 rows), then per head the K, V and Q passes.
    constexpr int NQH = H * 3 * SPP;
    const bool has_pre = p.pre_w != nullptr;
    const int NPRE = has_pre ? (C / 64) * SPP : 0;
    const int NQ = NPRE + NQH;
    const BufRsrc r_pw = make_rsrc(p.pre_w, has_pre ? (unsigned)((int64_t)C * C * 2) : 0u);
    auto issue = [&](int q, int buf) __attribute__((always_inline)) {            // stage q: pass q / SPP of its stream ((head * 3 + pass) of the main one), K range q % SPP
        const bool pre = q < NPRE;
        const int qq = pre ? q : q - NPRE;
        const int hp = qq / SPP, sub = qq - hp * SPP;
        const unsigned uni = (unsigned)((hp * 64 * C + sub * SC * KC) * 2);
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int pi = wave + NW * j;
            async_copy16_buf_s(pre ? r_pw : r_w, w_lane[j], uni, ring + buf * STAGE_BYTES + (pi / PPC) * CHUNK_BYTES + (pi % PPC) * 1024);
        }
    };
    // AaSeqSelfAttn.flags: timing ablations (results are garbage): 1 no attention phase, 2 no weight DMA behind the first two stages, 4 no projection
    // MFMAs, 8 no x fetch / LayerNorm, 16 no output stores, 32 no row normalisation
    const int dbg = p.flags;
    issue(0, 0);

    u32x4 xf[NKS];          // x fragments (B operand of the transposed products, A operand of the straight one): k-slice ks = 2 nb + s
    // waves behind the tile's last row do no matrix work; their operand slots must hold finite values (p = 0 times V)
    if (!wave_live) {
#pragma unroll
        for (int u = 0; u < 8; ++u) *reinterpret_cast<u32x4*>(xch + wave * 8192 + u * 1024 + lane * 16) = u32x4{0u, 0u, 0u, 0u};
    }

    // ---- weight fragment addresses: row c (+ 32 j) of a chunk, slot (4 nbl + 2 h + s) ^ swizzle(c): one register per k-slice (nbl, s) of a chunk
    unsigned wa[SPC];
#pragma unroll
    for (int v = 0; v < SPC; ++v) wa[v] = (unsigned)(c * RB + ((((v >> 1) * 4 + 2 * h + (v & 1)) ^ ((c / RPB) & (SPR - 1))) << 4));

    const float sl2e = p.scale * 1.4426950408889634f;
    f32x16 zero16;
#pragma unroll
    for (int e = 0; e < 16; ++e) zero16[e] = 0.0f;

    f32x16 acc[2];
    // one stage: SC chunks = SPC SC k-slices, two MFMAs each; the fragment reads of slice u + 1 are in flight under the MFMAs of slice u
    auto stage = [&](auto pass_, auto sub_, const char* st, const u32x4 (&xop)[NKS]) __attribute__((always_inline)) {
        constexpr int pass = decltype(pass_)::value, sub = decltype(sub_)::value;
        constexpr int NU = SPC * SC;
        u32x4 wf[3][2];             // fragment reads run two k-slices ahead of the MFMAs (three register sets)
        const char* b4[SPC];
#pragma unroll
        for (int v = 0; v < SPC; ++v) b4[v] = st + wa[v];
        auto rd = [&](auto u_) __attribute__((always_inline)) {
            constexpr int u = decltype(u_)::value, kc = u / SPC, v = u % SPC, set = u % 3;
            lds_read16_async_off<kc * CHUNK_BYTES>(wf[set][0], b4[v]);
            lds_read16_async_off<kc * CHUNK_BYTES + 32 * RB>(wf[set][1], b4[v]);
        };
        rd(IntTag<0>());
        if constexpr (NU > 1) rd(IntTag<1>());
        static_for<NU>([&](auto u_) __attribute__((always_inline)) {
            constexpr int u = decltype(u_)::value, set = u % 3, ks = sub * SC * SPC + u;
            if constexpr (u + 2 < NU) { rd(IntTag<u + 2>()); lds_wait<4>(wf[set][0]); }
            else if constexpr (u + 1 < NU) lds_wait<2>(wf[set][0]);
            else lds_wait<0>(wf[set][0]);
            lds_pin(wf[set][1]);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if constexpr (pass == 1) acc[j] = mfma_32x32x16(T(), xop[ks], wf[set][j], acc[j]);     // V[token][d]
                else                     acc[j] = mfma_32x32x16(T(), wf[set][j], xop[ks], acc[j]);     // K^T / Q^T [d][token]
            }
        });
    };
    // A pass's bias (W beta of the folded LayerNorm: `w_bias`; the bias of the projection in front: `pre_bias`; fp32, packed row order) is
    // FETCHED at the top of the pass and ADDED behind it: starting the accumulators from it put an L2 round trip in front of the first MFMA
    // of every pass (15 per tile: the "barriers and loops only" ablation of r06f took 41 us).  Transposed passes (K, Q, projection in front):
    // register r of block j is packed row 32 j + (r & 3) + 8 (r >> 2) + 4 h - four 16-byte loads per block; straight pass (V): the lane's
    // column 32 j + c in every register.
    f32x16 pbias[2];
    auto fetch_bias = [&](auto pass_, const float* b) __attribute__((always_inline)) {
        constexpr int pass = decltype(pass_)::value;
        acc[0] = zero16; acc[1] = zero16;
        if (b == nullptr) { pbias[0] = zero16; pbias[1] = zero16; return; }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if constexpr (pass == 1) {
                const float v = b[32 * j + c];
#pragma unroll
                for (int e = 0; e < 16; ++e) pbias[j][e] = v;
            } else {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(b + 32 * j + 8 * g4 + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) pbias[j][4 * g4 + e] = v[e];
                }
            }
        }
    };
    // the accumulators of a pass as operand registers: slice 2 j + t = registers 8 t .. 8 t + 7 of block j, rounded to the storage type
    auto to_operands = [&](u32x4 (&op)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                Pack8<T> v;
#pragma unroll
                for (int e = 0; e < 8; ++e) v.e[e] = (T)(acc[j][8 * t + e] + pbias[j][8 * t + e]);
                op[2 * j + t] = v.raw;
            }
    };

    // ---- x: fetched once, straight into operand layout.  With a projection in front (pre_w) the fetched rows are ITS input; it runs here as
    // C / 64 transposed passes of 64 output channels - lane (token, h) of block j then holds channels 64 pp + 32 j + 16 h .. + 15 (the rows of
    // pre_w are packed in that order), exactly the k-slices 2 (2 pp + j), + 1 of the operand layout - plus the residual;
    // the result x' is stored (pre_out) and attended over.
    int q = 0;
    {
        const unsigned xb = (row_ok && !(dbg & 8)) ? (unsigned)(trow * p.ldx * 2) + (unsigned)(16 * h * 2) : OOB;
        if (has_pre) {
            u32x4 xin[NKS];
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) xin[ks] = buf_load16(r_x, xb + (unsigned)((32 * (ks >> 1) + 8 * (ks & 1)) * 2));
            const int64_t rows_ext = p.x_bytes / (2 * (int64_t)p.ldx);
            const BufRsrc r_res = make_rsrc(p.pre_residual, p.pre_residual ? (unsigned)(rows_ext * p.ld_res * 2) : 0u);
            const BufRsrc r_po = make_rsrc(p.pre_out, (unsigned)(rows_ext * p.ld_pre * 2));
            const unsigned rb = (row_ok && p.pre_residual) ? (unsigned)(trow * p.ld_res * 2) + (unsigned)(16 * h * 2) : OOB;
            const unsigned pb = (row_ok && !(dbg & 16)) ? (unsigned)(trow * p.ld_pre * 2) + (unsigned)(16 * h * 2) : OOB;
            static_for<C / 64>([&](auto pp_) __attribute__((always_inline)) {
                constexpr int pp = decltype(pp_)::value;
                u32x4 res[4];
                static_for<SPP>([&](auto sub_) __attribute__((always_inline)) {
                    // (the first stage of a later pass: its pieces were waited for in front of the previous pass's stores, see below)
                    if (sub_.value == 0 && pp > 0) lds_wait_all(); else mem_wait_all();
                    block_barrier();
                    if (q + 1 < NQ && !((dbg & 2) && q >= 1)) issue(q + 1, (q + 1) & 1);
                    if constexpr (sub_.value == 0) fetch_bias(IntTag<0>(), p.pre_bias ? p.pre_bias + pp * 64 : nullptr);
                    if constexpr (sub_.value == SPP - 1) {          // the residual under this pass's 64 channels travels while its last stage multiplies (no residual: zeros)
#pragma unroll
                        for (int u = 0; u < 4; ++u) res[u] = buf_load16(r_res, rb + (unsigned)((64 * pp + 32 * (u >> 1) + 8 * (u & 1)) * 2));
                    }
                    if (wave_live && !(dbg & 4)) stage(IntTag<0>(), sub_, ring + (q & 1) * STAGE_BYTES, xin);
                    ++q;
                });
                // x' of this pass's 64 channels: + residual, rounded, stored.  It is NOT kept: x (80 registers) and x' (80) side by side left the
                // multiply stages no room (hipcc spilled ~100 registers, fetching the residual through scratch one load at a time) - the tile's
                // x' is read back from L2 behind the last pass.
                dma_wait<0>();          // the residual - and the next stage's weight pieces (issued a stage ago): only the stores below stay outstanding
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        Pack8<T> v; v.raw = res[2 * j + t];
#pragma unroll
                        for (int e = 0; e < 8; ++e) v.e[e] = (T)(acc[j][8 * t + e] + pbias[j][8 * t + e] + (float)v.e[e]);
                        buf_store16(r_po, pb + (unsigned)((64 * pp + 32 * j + 8 * t) * 2), v.raw);
                    }
            });
            dma_wait<0>();              // my stores have reached L2 (nobody else wrote or cached these rows): read the tile's x' back
            const unsigned pl = row_ok ? (unsigned)(trow * p.ld_pre * 2) + (unsigned)(16 * h * 2) : OOB;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) xf[ks] = buf_load16(r_po, pl + (unsigned)((32 * (ks >> 1) + 8 * (ks & 1)) * 2));
        } else {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) xf[ks] = buf_load16(r_x, xb + (unsigned)((32 * (ks >> 1) + 8 * (ks & 1)) * 2));
        }
    }
    if (p.normalize && !(dbg & (8 | 32))) {
        // Row normalisation of nn.LayerNorm (fp32 statistics, biased variance; diffusers BasicTransformerBlock.norm1 / norm2): x~ = (x - mean) * rstd,
        // rounded to the storage type.  The affine part lives in the operands: gamma is folded into the packed weights (W' = W diag(gamma)),
        // beta enters as the bias W beta the accumulators start from (`w_bias`) - the same re-association as the folded LayerNorm of
        // aa_conv_gemm (ln_cols).  Packed arithmetic: the sums through dot products on 16-bit pairs (exact products, fp32 sums), the
        // centred squares and the normalisation on fp32 pairs (v_pk_add / v_pk_fma): 4.5 vector instructions per element where the plain
        // form (convert, subtract, multiply, gamma, beta, convert per element + two passes of statistics) took 11.5 - it was 28 us of a
        // 159 us call at the 64x64 level (profiles/r06_seq_attention_ablations.txt).
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
    for (int head = 0; head < H; ++head) {
        u32x4 qop[4];
        static_for<3>([&](auto pass_) __attribute__((always_inline)) {
            constexpr int pass = decltype(pass_)::value;
            static_for<SPP>([&](auto sub_) __attribute__((always_inline)) {
                // my pieces of stage q have landed, my operand slots are written.  (The first stage of a later head: its pieces were waited
                // for in front of the previous head's output stores - those stay in flight across this barrier instead of being drained here)
                if (pass == 0 && sub_.value == 0 && (head > 0 || has_pre)) lds_wait_all(); else mem_wait_all();
                block_barrier();                     // everyone's have; every wave is done with stage q - 1 and with the operand slots of the previous head
                if (q + 1 < NQ && !((dbg & 2) && q >= 1)) issue(q + 1, (q + 1) & 1);
                if constexpr (sub_.value == 0) fetch_bias(pass_, p.w_bias ? p.w_bias + (head * 3 + pass) * 64 : nullptr);
                if (wave_live && !(dbg & 4)) stage(pass_, sub_, ring + (q & 1) * STAGE_BYTES, xf);
                ++q;
            });
            if (wave_live) {
                if constexpr (pass < 2) {
                    u32x4 op[4];
                    to_operands(op);
#pragma unroll
                    for (int u = 0; u < 4; ++u) *reinterpret_cast<u32x4*>(xch + wave * 8192 + (4 * pass + u) * 1024 + lane * 16) = op[u];
                } else to_operands(qop);
            }
        });
        if (!wave_live || (dbg & 1)) { dma_wait<0>(); continue; }
        // (every wave wrote its K operands before the barrier of the V pass and its V^T operands before the barrier of the Q pass)
        // ---- S^T = K Q^T against the row blocks wave-1, wave, wave+1
        // (all twelve fragment reads go out at once - a block that does not exist reads a neighbour's slot and is not multiplied - and are
        //  consumed block by block behind counted waits: one LDS latency per head instead of three)
        f32x16 sacc[3];
        bool have[3];
        const char* xslot[3];
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            const int kb = wave - 1 + n;
            have[n] = kb >= 0 && kb < NW && 32 * kb < rows_valid;
            xslot[n] = xch + (have[n] ? kb : wave) * 8192 + lane * 16;
            sacc[n] = zero16;
        }
        {
            u32x4 kf[3][4];
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                lds_read16_async_off<0>(kf[n][0], xslot[n]);
                lds_read16_async_off<1024>(kf[n][1], xslot[n]);
                lds_read16_async_off<2048>(kf[n][2], xslot[n]);
                lds_read16_async_off<3072>(kf[n][3], xslot[n]);
            }
            static_for<3>([&](auto n_) __attribute__((always_inline)) {
                constexpr int n = decltype(n_)::value;
                lds_wait<4 * (2 - n)>(kf[n][0]);
                lds_pin(kf[n][1]); lds_pin(kf[n][2]); lds_pin(kf[n][3]);
                if (have[n]) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) sacc[n] = mfma_32x32x16(T(), kf[n][u], qop[u], sacc[n]);
                }
            });
        }
        // ---- softmax over the keys of the lane's own sequence (base 2, one pass).  L <= 32: the sequence lies in this block and ONE of its
        // neighbours - per lane the other neighbour's 16 scores are dropped before the exponentials (32 instead of 48 per lane)
        f32x16 side;
#pragma unroll
        for (int e = 0; e < 16; ++e) side[e] = from_prev ? sacc[0][e] : sacc[2][e];
        float mx = -1.0e30f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int off = (e & 3) + 8 * (e >> 2);
            const bool ok_s = (unsigned)(off - key_lo_side) < (unsigned)L;
            const bool ok_o = (unsigned)(off + 32 - key_lo) < (unsigned)L;
            const float ts = ok_s ? side[e] * sl2e : -1.0e30f;
            const float to = ok_o ? sacc[1][e] * sl2e : -1.0e30f;
            side[e] = ts;
            sacc[1][e] = to;
            mx = fmaxf(mx, fmaxf(ts, to));
        }
        mx = wave_max_halves(mx);
        float ps = 0.0f;
        u32x4 pop[3][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            Pack8<T> pk_s, pk_o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float p_s = fast_exp2(side[8 * t + e] - mx), p_o = fast_exp2(sacc[1][8 * t + e] - mx);
                ps += p_s + p_o;
                pk_s.e[e] = (T)p_s;
                pk_o.e[e] = (T)p_o;
            }
            pop[1][t] = pk_o.raw;
#pragma unroll
            for (int k = 0; k < 4; ++k) { pop[0][t][k] = from_prev ? pk_s.raw[k] : 0u; pop[2][t][k] = from_prev ? 0u : pk_s.raw[k]; }
        }
        const float inv = 1.0f / wave_sum_halves(ps);
        // ---- O^T = V^T P^T
        f32x16 oacc[2];
        oacc[0] = zero16; oacc[1] = zero16;
        {
            u32x4 vf[3][4];
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                lds_read16_async_off<4096>(vf[n][0], xslot[n]);
                lds_read16_async_off<4096 + 1024>(vf[n][1], xslot[n]);
                lds_read16_async_off<4096 + 2048>(vf[n][2], xslot[n]);
                lds_read16_async_off<4096 + 3072>(vf[n][3], xslot[n]);
            }
            static_for<3>([&](auto n_) __attribute__((always_inline)) {
                constexpr int n = decltype(n_)::value;
                lds_wait<4 * (2 - n)>(vf[n][0]);
                lds_pin(vf[n][1]); lds_pin(vf[n][2]); lds_pin(vf[n][3]);
                if (have[n]) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int t = 0; t < 2; ++t) oacc[j] = mfma_32x32x16(T(), vf[n][2 * j + t], pop[n][t], oacc[j]);
                }
            });
        }
        dma_wait<0>();                  // the next stage's weight pieces (issued a stage ago) - from here on only the stores below are outstanding
        // ---- lane (token c, h): channels head * 64 + 32 j + 16 h .. + 15 (the to_v rows were permuted for this at pack time)
        const unsigned ob = (row_ok && !(dbg & 16)) ? (unsigned)(trow * p.ldo * 2) + (unsigned)((head * 64 + 16 * h) * 2) : OOB;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                Pack8<T> v;
#pragma unroll
                for (int e = 0; e < 8; ++e) v.e[e] = (T)(oacc[j][8 * t + e] * inv);
                buf_store16(r_o, ob + (unsigned)((32 * j + 8 * t) * 2), v.raw);
            }
    }
}

}  // namespace aa
