// FeedForward of a transformer block with the wrapper's proj_out behind it, as ONE kernel (see include/aa_mi355.h: aa_ff_fused):
//     out = [ GEGLU(LayerNorm(x) W1^T + b1) | x ] Wm^T + bm + outer          Wm = [Wp W2 | Wp],  bm = Wp b2 + bp
// i.e. diffusers BasicTransformerBlock.norm3 -> FeedForward (GEGLU, Linear) -> + x, then Transformer2DModel / TransformerTemporalModel
// .proj_out -> + the transformer's input (reference models/unet_3d_blocks.py:287,446,681 / :379,526,759 via diffusers).  The GEGLU
// activation [tokens, 4 C] - 356 MB written and read back per transformer at the 64x64 level, for the two contractions around it
// (350 + 183 us there) - never leaves the chip.
//
// Round 5 rejected this fusion on paper (LDS fragment reads per MFMA rise to 1.0, "one 1-KiB fragment per 8 clk per CU"); ds_read_b128
// delivers 256 B/clk (MI355X_MICROARCH.md, LDS), i.e. one fragment per 4 clk against one MFMA per 8 clk per CU, and aa_seq_self_attention
// runs its projections at 1.0 reads per MFMA.  This kernel is built the same way:
//  * a 4-wave workgroup (one wave per SIMD: up to 512 registers per lane) owns 128 consecutive token rows, wave w rows 32 w .. + 31 and ALL
//    320 output channels: out^T accumulators = 10 blocks of 32 channels = 160 registers, x = 80 registers in operand layout (fetched once);
//  * first the x part of the merged tail, out^T = Wm_x x^T (5 passes of 64 output channels), on the raw rows; then x is normalised in
//    place (fp32 statistics, LayerNorm's gamma / beta live in W1 / b1: pack_ff_fused) and the hidden axis is walked in chunks of 32 units:
//    value^T | gate^T = W1_chunk x~^T (one transposed pass of 64 rows: 40 MFMAs), h = value * gelu(gate) in registers - rounded to the storage
//    type these ARE two B-operand k-slices of the ff-out product - and after every second chunk out^T += Wm_h[:, 64 units] h^T (40 MFMAs);
//  * every weight matrix streams L2 -> LDS by LDS-DMA in 41 KB stages ([64 rows][320 K] for W1 and Wm_x, [320 rows][64 K] for Wm_h, both as
//    five [64][64] chunks with source-side XOR swizzle, + 1 KB of bias rows) through a three-stage ring: two stages in flight, counted vmcnt,
//    one raw barrier per stage; 65 stages per tile;
//  * the biases ride on the MATRIX pipe: a pass of 64 weight rows has a 21st k-slice whose A operand holds (hi, lo) of each row's fp32 bias in
//    the storage type (hi + lo = the bias to 2^-22 / 2^-16 relative) and whose B operand is (1, 1, 0, ...) - two more MFMAs per pass instead of
//    8 loads, 32 accumulator writes and 32 live registers per pass; the accumulators START at the pass's first MFMA (constant-zero source).
#pragma once
#include "dev.h"
#include "aa_mi355.h"
#include "conv_gemm.h"      // gelu_erf_2

namespace aa {

constexpr int FF_NW = 4;
constexpr int FF_CHUNK_BYTES = 8192;        // [64 rows][64 K]: 128-byte rows, 16-byte slots XOR-swizzled with (row >> 1) & 7
constexpr int FF_BIAS_OFF = 5 * FF_CHUNK_BYTES;   // [64 rows][8]: the rows' (bias hi, bias lo, 0 x 6)
constexpr int FF_STAGE_BYTES = FF_BIAS_OFF + 1024;
constexpr int FF_RING = 3;
__host__ __device__ constexpr int ff_lds_bytes() { return FF_RING * FF_STAGE_BYTES; }

// ABL: timing ablations compiled as separate instantiations (AaFFFused.flags >> 8; results are garbage): 1 no GELU arithmetic, 4 no LDS-DMA pieces behind
// the first two stages, 8 no fragment reads, 16 no MFMAs, 32 no bias k-slice;
// experiments (results valid): 256: the fine side-work placement (GELU in thirds of a pair behind every MFMA); 64: coarse, two pairs at a time;
// 128: no rotation of the DMA piece order between workgroups; 512: x / outer / out accessed with the non-temporal hint
template <typename T, int C, int ABL = 0>
__global__ void __launch_bounds__(64 * FF_NW, 1) ff_fused_kernel(const AaFFFused p) {
    static_assert(C == 320, "one stage = five 64 x 64 chunks: 320 channels");
    constexpr int NW = FF_NW, NKS = C / 16, HID = 4 * C, NCH = HID / 32;       // k-slices of x; hidden units; chunks of 32 hidden units
    constexpr int NB = C / 32;                                                   // output channel blocks per wave
    constexpr int PPW = 40 / NW;                                                 // weight LDS-DMA instructions per stage and wave (1 KiB each: 8 rows of a chunk)
    constexpr int GP = (ABL & 64) ? 2 : 8;                                       // GELU pairs per side-work slot (coarse placement)
    constexpr unsigned OOB = 0x80000000u;
    static_assert(NCH % 2 == 0 && 40 % NW == 0 && NW == 4, "shape");
    char* ring = dyn_smem();

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = wave_id();
    const int c = lane & 31, h = lane >> 5;
    const int64_t row = (int64_t)blockIdx.x * (32 * NW) + 32 * wave + c;
    const bool row_ok = row < p.rows;

    const BufRsrc r_x = make_rsrc(p.x, (unsigned)(p.rows * p.ldx * 2));
    const BufRsrc r_res = make_rsrc(p.outer, p.outer ? (unsigned)(p.rows * p.ld_outer * 2) : 0u);
    const BufRsrc r_o = make_rsrc(p.out, (unsigned)(p.rows * p.ldo * 2));
    constexpr int NQ = 5 + 3 * (NCH / 2);
    // ---- the weight stream.  The host (ops.pack_ff_fused) lays the operands out as the SEQUENCE OF LDS STAGE IMAGES the kernel consumes, 65 of
    // FF_STAGE_BYTES each, in stage order (see "the schedule" below): five [64][64] chunks - K chunks of 64 weight rows (Wm_x, W1: [64 rows][320 K])
    // or row chunks of 64 K (Wm_h: [320 rows][64 K]) - with the 16-byte slots of row r XOR-swizzled by (r >> 1) & 7, then 1 KB of bias rows
    // ([64][8]: (hi, lo, 0 x 6), zeros behind a Wm_h stage).  Streaming a stage = copying 41 consecutive KB: piece pi = wave + 4 j (j < 10) is KB
    // pi of the image, lane l its bytes 16 l .. + 15; wave 0 adds KB 40 (the biases).  One descriptor, no address arithmetic per stage.
    const BufRsrc r_w = make_rsrc(p.w, (unsigned)(NQ * FF_STAGE_BYTES));
    const unsigned lane16 = (unsigned)(lane * 16);
    // All workgroups walk the same stream nearly in step: without a rotation every CU of an XCD asks its L2 for the same KB at the same time
    // (one channel busy, fifteen idle).  Workgroup b starts its walk of a stage's 40 KB at KB 13 b mod 40.
    const int rot = (ABL & 128) ? 0 : (int)((blockIdx.x * 13u) % 40u);
    int pi_of[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) { const int v = wave + NW * j + rot; pi_of[j] = v >= 40 ? v - 40 : v; }

    // ---- x fragments (B operand of every transposed product): k-slice ks = 2 nb + s = channels 32 nb + 16 h + 8 s .. + 7
    u32x4 xf[NKS];
    {
        const unsigned xb = row_ok ? (unsigned)(row * p.ldx * 2) + (unsigned)(16 * h * 2) : OOB;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const unsigned o = xb + (unsigned)((32 * (ks >> 1) + 8 * (ks & 1)) * 2);
            xf[ks] = (ABL & 512) ? buf_load16_nt(r_x, o) : buf_load16(r_x, o);
        }
    }
    // the bias k-slice's B operand: k = 0, 1 (lanes h = 0) multiply the rows' (hi, lo)
    const u32x4 xone = u32x4{h == 0 ? ones_pair(T()) : 0u, 0u, 0u, 0u};
    // fragment addresses inside a stage: row c (+ 32 j) of a chunk, slot (4 nbl + 2 h + s) ^ ((c >> 1) & 7): one register per k-slice of a 64-K chunk
    unsigned wa[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) wa[v] = (unsigned)(c * 128 + ((((v >> 1) * 4 + 2 * h + (v & 1)) ^ ((c >> 1) & 7)) << 4));
    const unsigned wb = (unsigned)(FF_BIAS_OFF + c * 16);                       // bias entry of row c (both half-waves read it; h = 1 multiplies zeros)

    f32x16 zero16;
#pragma unroll
    for (int e = 0; e < 16; ++e) zero16[e] = 0.0f;
    f32x16 oacc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) oacc[nb] = zero16;

    // AaFFFused.flags bit 1 (2): timing ablation - the weight DMA behind the first two stages fetches nothing (out-of-range offsets: the
    // instructions are still issued); flags >> 8: the ABL instantiations
    const bool do_dma = !(p.flags & 2);

    // ---- the schedule.  One wave per SIMD: nothing hides a wave's own stalls, so everything that is not an MFMA is threaded BETWEEN the MFMAs
    // of a pass ("side work" behind the two MFMAs of a k-slice / the four of a row block):
    //   * the LDS-DMA pieces of stage q + 2 (an LDS-DMA instruction holds the wave's issue for ~60-180 clk: ten in a burst at the top of a
    //     stage left the matrix pipe idle for that long),
    //   * the GELU of the PREVIOUS chunk (value * gelu(gate), 8 packed pairs per lane): chunk A(g)'s runs inside pass B(g), B(g)'s inside
    //     A(g + 1) - two accumulator sets alternate.
    // Stage order: X0..X4 (Wm_x), A(0), B(0), then for g = 1..19: A(g), H(g - 1), B(g), and H(19) at the end (A / B(g) = W1 chunks 2 g / 2 g + 1,
    // H(g) = Wm_h[:, 64 g ..]): H(g - 1) sits behind A(g) because the GELU of B(g - 1) finishes inside A(g).
    int q = 0;
    unsigned nxt_lane = OOB, nxt_uni = 0;       // stage q + 2, set at the top of stage q (behind the last stage: pieces that fetch nothing)
    auto piece = [&](unsigned lane_off, unsigned uni, int buf, int pi) __attribute__((always_inline)) {
        async_copy16_buf_s(r_w, lane_off, uni + (unsigned)(pi * 1024), ring + buf * FF_STAGE_BYTES + pi * 1024);
    };
    {   // prologue: stages 0 and 1 whole
#pragma unroll
        for (int j = 0; j < PPW; ++j) piece(lane16, 0u, 0, pi_of[j]);
        if (wave == 0) piece(lane16, 0u, 0, 40);
#pragma unroll
        for (int j = 0; j < PPW; ++j) piece(lane16, (unsigned)FF_STAGE_BYTES, 1, pi_of[j]);
        if (wave == 0) piece(lane16, (unsigned)FF_STAGE_BYTES, 1, 40);
    }
    // stage top: my pieces of stage q have landed (the younger stage's stay in flight), everyone's have, every wave is done with stage q - 1,
    // whose buffer takes stage q + 2 piece by piece during this stage
    auto stage_top = [&]() __attribute__((always_inline)) {
        if constexpr (ABL & 4) dma_wait<0>();
        else if (wave == 0) dma_wait<PPW + 1>();
        else dma_wait<PPW>();
        block_barrier();
        const bool on = q + 2 < NQ;
        nxt_lane = (on && do_dma) ? lane16 : OOB;
        nxt_uni = on ? (unsigned)((q + 2) * FF_STAGE_BYTES) : 0u;
    };
    auto dma_side = [&](int j) __attribute__((always_inline)) {
        if constexpr (!(ABL & 4)) {
            if (j < PPW) piece(nxt_lane, nxt_uni, (q + 2) % FF_RING, pi_of[j]);
            else if (wave == 0) piece(nxt_lane, nxt_uni, (q + 2) % FF_RING, 40);      // (wave 0 only: the other waves' vmcnt budget is PPW)
        }
    };
    // one packed pair of a chunk's 16 values per lane: pair i = registers 8 t + e, e + 1 (t = i >> 2, e = 2 (i & 3)) of the value / gate blocks ->
    // dword i & 3 of B-operand k-slice t
    auto gelu_pair = [&](const f32x16 (&a)[2], u32x4 (&hp)[2], auto i_) __attribute__((always_inline)) {
        constexpr int i = decltype(i_)::value, t = i >> 2, e = 2 * (i & 3);
        const f32x2 val = f32x2{a[0][8 * t + e], a[0][8 * t + e + 1]};
        const f32x2 gat = f32x2{a[1][8 * t + e], a[1][8 * t + e + 1]};
        const f32x2 y = (ABL & 1) ? val : val * gelu_erf_2(gat);
        typedef T t2 __attribute__((ext_vector_type(2)));
        hp[t][i & 3] = __builtin_bit_cast(unsigned, __builtin_convertvector(y, t2));          // (one v_cvt_pk_*_f32, round to nearest even)
    };
    // Side work of an A / B pass, by SLOT: slot 2 u + j sits behind MFMA j of k-slice u (42 slots).  An MFMA occupies the matrix pipe for 32 clk
    // and the wave issues in order: vector work queued behind a SECOND MFMA waits for the first one to leave the pipe, and a long run of vector
    // work leaves the pipe idle - so the work is cut into units of <= 10 instructions, one per slot:
    //   FINE (ABL 256; measured slower, 502 against 447 us): the GELU of pair i in three units - U0 clamp / square / polynomial, U1 exp2 + 1, reciprocal, U2 products, rounding -
    //   at slots 2 + 3 i + k (consecutive slots continue the SAME pair: its chain is spaced by an MFMA issue, neighbouring pairs never wait
    //   for each other); the ten weight pieces at slots 26 .. 35, the bias piece at 36;
    //   coarse (default): all eight pairs behind k-slice 2 (the compiler interleaves their chains), pieces behind the odd k-slices.
    constexpr bool FINE = (ABL & 256) != 0;
    struct GeluState { f32x2 val[8], gat[8], t[8]; };
    auto side_dma = [&](auto s_) __attribute__((always_inline)) {
        constexpr int sl = decltype(s_)::value;
        if constexpr (FINE) {
            if constexpr (sl >= 26 && sl < 26 + PPW) dma_side(sl - 26);
            else if constexpr (sl == 26 + PPW) dma_side(PPW);
        } else if constexpr (sl & 1) {
            constexpr int u = sl >> 1;
            if constexpr (u < NKS && (u & 1)) dma_side(u >> 1);
            else if constexpr (u == 18) dma_side(PPW);
        }
    };
    auto side_gelu = [&](auto s_, const f32x16 (&a)[2], u32x4 (&hp)[2], GeluState& gs) __attribute__((always_inline)) {
        constexpr int sl = decltype(s_)::value;
        if constexpr (FINE) {
            if constexpr (sl >= 2 && sl < 26) {
                constexpr int i = (sl - 2) / 3, k = (sl - 2) % 3, t = i >> 2, e = 2 * (i & 3);
                if constexpr (k == 0) {
                    gs.val[i] = f32x2{a[0][8 * t + e], a[0][8 * t + e + 1]};
                    gs.gat[i] = f32x2{a[1][8 * t + e], a[1][8 * t + e + 1]};
                    const f32x2 xc = f32x2{clamp_f(gs.gat[i][0], -8.0f, 8.0f), clamp_f(gs.gat[i][1], -8.0f, 8.0f)};
                    const f32x2 x2 = xc * xc;
                    const f32x2 c5 = f32x2{0.001014264184050262f, 0.001014264184050262f}, c3 = f32x2{-0.10677573084831238f, -0.10677573084831238f},
                                c1 = f32x2{-2.301121234893799f, -2.301121234893799f};
                    gs.t[i] = xc * __builtin_elementwise_fma(__builtin_elementwise_fma(c5, x2, c3), x2, c1);      // (conv_gemm.h gelu_erf_2, in three parts)
                } else if constexpr (k == 1) {
                    const f32x2 d = f32x2{fast_exp2(gs.t[i][0]), fast_exp2(gs.t[i][1])} + f32x2{1.0f, 1.0f};
                    gs.t[i] = f32x2{fast_rcp(d[0]), fast_rcp(d[1])};
                } else {
                    const f32x2 y = (ABL & 1) ? gs.val[i] : gs.val[i] * (gs.gat[i] * gs.t[i]);
                    union { T e2[2]; unsigned u; } pk;
                    pk.e2[0] = (T)y[0]; pk.e2[1] = (T)y[1];
                    hp[t][i & 3] = pk.u;
                }
            }
        } else if constexpr (sl & 1) {
            constexpr int u = sl >> 1;
            if constexpr (u >= 2 && u < 18 && (u - 2) % (2 * GP) == 0)
                static_for<GP>([&](auto k_) __attribute__((always_inline)) { gelu_pair(a, hp, IntTag<(u - 2) / 2 + decltype(k_)::value>()); });
        }
    };
    // a transposed pass over all 320 K (+ the bias k-slice) of a [64 rows][320 K] stage: a[j] (j = 0, 1: rows 32 j .. + 31) = W x^T + b, fragment
    // reads two k-slices ahead; side(u) runs behind the MFMAs of k-slice u
    auto pass64 = [&](const char* st, f32x16 (&a)[2], auto&& side) __attribute__((always_inline)) {
        constexpr int NK1 = (ABL & 32) ? NKS : NKS + 1;
        u32x4 wf[3][2];
        const char* b4[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) b4[v] = st + wa[v];
        const char* bb = st + wb;
        auto rd = [&](auto u_) __attribute__((always_inline)) {
            constexpr int u = decltype(u_)::value, ch = u >> 2, v = u & 3, set = u % 3;
            if constexpr (u < NKS) {
                lds_read16_async_off<ch * FF_CHUNK_BYTES>(wf[set][0], b4[v]);
                lds_read16_async_off<ch * FF_CHUNK_BYTES + 4096>(wf[set][1], b4[v]);
            } else {
                lds_read16_async_off<0>(wf[set][0], bb);
                lds_read16_async_off<512>(wf[set][1], bb);
            }
        };
        if constexpr (ABL & 8) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 2; ++j) wf[i][j] = xf[i + j]; }
        else { rd(IntTag<0>()); rd(IntTag<1>()); }
        static_for<NK1>([&](auto u_) __attribute__((always_inline)) {
            constexpr int u = decltype(u_)::value, set = u % 3;
            if constexpr (!(ABL & 8)) {
                if constexpr (u + 2 < NK1) { rd(IntTag<u + 2>()); lds_wait<4>(wf[set][0]); }
                else if constexpr (u + 1 < NK1) lds_wait<2>(wf[set][0]);
                else lds_wait<0>(wf[set][0]);
                lds_pin(wf[set][1]);
            }
            static_for<2>([&](auto j_) __attribute__((always_inline)) {
                constexpr int j = decltype(j_)::value;
                if constexpr (!(ABL & 16)) {
                    // (the accumulators START at k-slice 0: a constant-zero third source, no initialisation)
                    if constexpr (u == 0) a[j] = mfma_32x32x16(T(), wf[set][j], xf[0], zero16);
                    else if constexpr (u < NKS) a[j] = mfma_32x32x16(T(), wf[set][j], xf[u], a[j]);
                    else a[j] = mfma_32x32x16(T(), wf[set][j], xone, a[j]);
                } else if constexpr (u == 0) a[j] = zero16;
                sched_fence();
                side(IntTag<2 * u + j>());
                sched_fence();
            });
        });
    };
    // out^T += Wm_h[:, a pair's 64 units] h^T: 10 row blocks x 4 k-slices; k-slice (cc, t) sits at slot 4 cc + 2 h + t of the stage's 64 K
    auto passH = [&](const char* st, const u32x4 (&hp)[2][2]) __attribute__((always_inline)) {
        const char* b4[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) b4[v] = st + wa[v];
        u32x4 wf[3][4];
        auto rd = [&](auto i_) __attribute__((always_inline)) {          // read i: row block nb = i, all four k-slices
            constexpr int i = decltype(i_)::value, set = i % 3;
            lds_read16_async_off<(i >> 1) * FF_CHUNK_BYTES + (i & 1) * 4096>(wf[set][0], b4[0]);
            lds_read16_async_off<(i >> 1) * FF_CHUNK_BYTES + (i & 1) * 4096>(wf[set][1], b4[1]);
            lds_read16_async_off<(i >> 1) * FF_CHUNK_BYTES + (i & 1) * 4096>(wf[set][2], b4[2]);
            lds_read16_async_off<(i >> 1) * FF_CHUNK_BYTES + (i & 1) * 4096>(wf[set][3], b4[3]);
        };
        if constexpr (ABL & 8) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) wf[i][j] = xf[i + j]; }
        else { rd(IntTag<0>()); rd(IntTag<1>()); }
        static_for<NB>([&](auto nb_) __attribute__((always_inline)) {
            constexpr int nb = decltype(nb_)::value, set = nb % 3;
            if constexpr (!(ABL & 8)) {
                if constexpr (nb + 2 < NB) { rd(IntTag<nb + 2>()); lds_wait<8>(wf[set][0]); }
                else if constexpr (nb + 1 < NB) lds_wait<4>(wf[set][0]);
                else lds_wait<0>(wf[set][0]);
                lds_pin(wf[set][1]); lds_pin(wf[set][2]); lds_pin(wf[set][3]);
            }
            static_for<4>([&](auto v_) __attribute__((always_inline)) {
                constexpr int v = decltype(v_)::value;
                if constexpr (!(ABL & 16)) oacc[nb] = mfma_32x32x16(T(), wf[set][v], hp[v >> 1][v & 1], oacc[nb]);
                sched_fence();
                if constexpr (v == 1) dma_side(nb);                                   // (one weight piece per row block, the bias piece at the end)
                if constexpr (nb == NB - 1 && v == 3) dma_side(PPW);
                sched_fence();
            });
        });
    };

    f32x16 accx[2], accy[2];
    // ---- 1. out^T = Wm_x x^T + bm: five passes of 64 output channels on the raw rows
    static_for<5>([&](auto pp_) __attribute__((always_inline)) {
        constexpr int pp = decltype(pp_)::value;
        stage_top();
        pass64(ring + (q % FF_RING) * FF_STAGE_BYTES, accx, side_dma);
        ++q;
        oacc[2 * pp] = accx[0]; oacc[2 * pp + 1] = accx[1];
    });

    // ---- 2. x~ = (x - mean) * rstd in place (LayerNorm's gamma / beta live in W1 / b1)
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

    // ---- 3. the hidden axis in pairs of 32-unit chunks
    u32x4 hop[2][2];                // h of a pair's two chunks as B-operand k-slices: slice t of chunk cc = registers 8 t .. 8 t + 7 of its value / gate blocks
    GeluState gs;
    // A(g): accx accumulates; the GELU of B(g - 1) (accy -> hop[1]) rides along (g >= 1)
    auto stage_a = [&](auto first_) __attribute__((always_inline)) {
        constexpr bool first = decltype(first_)::value;
        stage_top();
        if constexpr (first) pass64(ring + (q % FF_RING) * FF_STAGE_BYTES, accx, side_dma);
        else pass64(ring + (q % FF_RING) * FF_STAGE_BYTES, accx, [&](auto s_) __attribute__((always_inline)) { side_dma(s_); side_gelu(s_, accy, hop[1], gs); });
        ++q;
    };
    // B(g): accy accumulates; the GELU of A(g) (accx -> hop[0]) rides along
    auto stage_b = [&]() __attribute__((always_inline)) {
        stage_top();
        pass64(ring + (q % FF_RING) * FF_STAGE_BYTES, accy, [&](auto s_) __attribute__((always_inline)) { side_dma(s_); side_gelu(s_, accx, hop[0], gs); });
        ++q;
    };
    auto stage_h = [&]() __attribute__((always_inline)) {
        stage_top();
        passH(ring + (q % FF_RING) * FF_STAGE_BYTES, hop);
        ++q;
    };
    stage_a(BoolTag<true>());
    stage_b();
    for (int g = 1; g < NCH / 2; ++g) {
        stage_a(BoolTag<false>());
        stage_h();
        stage_b();
    }
    static_for<8>([&](auto i_) __attribute__((always_inline)) { gelu_pair(accy, hop[1], i_); });
    stage_h();
    dma_wait<0>();                  // (the last two stages' pieces fetched nothing, but they do write LDS)

    // ---- 4. + the transformer's input, rounded, stored: lane (token c, h) holds channels 32 nb + 16 h .. + 15 of block nb (the rows of Wm are packed
    // in that order)
    const unsigned ob = row_ok ? (unsigned)(row * p.ldo * 2) + (unsigned)(16 * h * 2) : OOB;
    const unsigned rb = (row_ok && p.outer) ? (unsigned)(row * p.ld_outer * 2) + (unsigned)(16 * h * 2) : OOB;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            Pack8<T> r, v;
            r.raw = (ABL & 512) ? buf_load16_nt(r_res, rb + (unsigned)((32 * nb + 8 * t) * 2)) : buf_load16(r_res, rb + (unsigned)((32 * nb + 8 * t) * 2));          // (no outer residual: zeros)
#pragma unroll
            for (int e = 0; e < 8; ++e) v.e[e] = (T)(oacc[nb][8 * t + e] + (float)r.e[e]);
            if constexpr (ABL & 512) buf_store16_nt(r_o, ob + (unsigned)((32 * nb + 8 * t) * 2), v.raw);
            else buf_store16(r_o, ob + (unsigned)((32 * nb + 8 * t) * 2), v.raw);
        }
}

}  // namespace aa
