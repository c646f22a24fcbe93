// Implicit-GEMM convolution / linear kernel, LDS-DMA edition (fast path of aa_conv_gemm).
//
// A BM x BN output tile per workgroup, K step BK (64 or 32), WM x WN wavefronts each owning a
// (BM/WM) x (BN/WN) block of v_mfma_f32_32x32x16 accumulators, a ring of STAGES LDS buffers.
// Both operand tiles travel HBM/L2 -> LDS with `buffer_load_dwordx4 ... lds` (no VGPR round trip, no ds_write):
//  * every wave instruction deposits 64 lanes x 16 B, lane-linear = RPI tile rows of BK*2 bytes; the im2col
//    gather, the conv halo and the M / K tails are expressed in the per-lane SOURCE offset of a buffer load
//    (halo lanes are out of range of the descriptor and deposit zeros);
//  * bank conflicts of the ds_read_b128 fragment reads are removed by an XOR swizzle applied on the source
//    side (lane of row r, 16-byte position s fetches k-slot s ^ swz(r)) and undone on the read side;
//  * requires the K tile to sit inside one filter tap and one concat source ((c0+c1) % 64 == 0 and
//    c0 % 64 == 0): tap / source / channel base are then wave-uniform scalars and the per-row pixel offsets
//    are recomputed only when the tap changes;
//  * pipeline: STAGES-1 tiles are in flight ahead of the one being multiplied; the rings wait with a COUNTED
//    `s_waitcnt vmcnt(n)` (only the oldest tile must have landed) and a raw `s_barrier` (a __syncthreads() would
//    drain every DMA); one barrier per K step.  STAGGER variants let the second wave of every SIMD issue its DMA
//    share in the middle of its multiply, so the CU-wide DMA issue queue never holds all eight waves at once;
//  * the MFMA runs "transposed" (weights = A operand, rows permuted), which leaves 16 consecutive output columns
//    of one row in each lane's accumulator registers: the epilogue (bias from an LDS slice, time-embedding row
//    vector, SiLU, GEGLU value*gelu(gate) inside one lane, residual, scale) needs no LDS transpose and no
//    barrier and stores 16-byte pieces straight from registers.
#pragma once
#include "dev.h"
#include "aa_mi355.h"
#include "conv_gemm.h"

namespace aa {

__host__ __device__ inline int cgd_lds_bytes(int bm, int bn, int bk, int stages) {
    return stages * (bm + bn) * bk * 2 + 1024 + 1024;             // + dummy DMA landing zone + the tile's bias slice
}

// ---- epilogue, straight from the accumulators (no LDS tile, no barrier): shared by the contraction kernels.  The weight
// fragments were the MFMA "A" operand with their rows permuted by pi, so lane (ec, eh) holds out[m = ec][16*eh + e] in
// acc[i][j][e] - 16 consecutive columns = two 16-byte stores per block; bias (from LDS), time-embedding row vector, SiLU,
// GEGLU (value block j, gate block j+1 sit in the same lane) and the residual are applied on the way out.
//   m_wave  first output row of this wave's blocks          n_wave  first (packed) output column of this wave's blocks
//   sBiasW  the bias slice of those columns in LDS
// ---- the same epilogue for the combinations that carry the step (plain / time-embedding row vector / residual / GEGLU),
// without a branch inside: output, residual and row vector go through buffer descriptors whose range check drops row
// tails (the descriptor ends behind the last valid row of this wave) and column tails (offset forced out of range), so a
// wave runs straight through MI x NI blocks; GELU on packed fp32 pairs.  BIAS = false: the caller started the
// accumulators from the bias.  Everything else (SiLU, per-row bias, GEGLU with a residual ...) takes cgd_epilogue_g's
// general path below.
// Output row of GEMM row m: m itself, or - AaConvGemm.out_sy / out_sx > 1 - pixel (y * out_sy + out_oy, x * out_sx + out_ox) of an
// [n_img, h_out * out_sy, w_out * out_sx] grid (one parity class of the four 2x2 convolutions that carry out Upsample2D).
__host__ __device__ __forceinline__ bool cgd_out_mapped(const AaConvGemm& p) { return p.out_sy > 1 || p.out_sx > 1; }
__host__ __device__ __forceinline__ int64_t cgd_out_row(const AaConvGemm& p, const int m) {
    if (!cgd_out_mapped(p)) return m;
    const int sy = p.out_sy > 1 ? p.out_sy : 1, sx = p.out_sx > 1 ? p.out_sx : 1;
    const int x = m % p.w_out, t = m / p.w_out, y = t % p.h_out, img = t / p.h_out;
    return ((int64_t)img * (p.h_out * sy) + (y * sy + p.out_oy)) * (p.w_out * sx) + (x * sx + p.out_ox);
}

// LayerNorm folded into the consuming contraction (AaConvGemm.ln_stats = per row (-mean, sqrt(var + eps), rstd, 0), written by
// aa_ln_finalize from the producer's partial sums):  LN(x) W^T + b = a * (x W'^T) + nm * colsum(W') + b',  a = rstd, nm = -mean * rstd.
struct LnRow { float a, nm, nmean, sd; };
// rstd of a lane's row of every block row, handed from a kernel's accumulator start to its epilogue BY VALUE (a pointer to the
// caller's array parks it in scratch)
template <int MI> struct LnRstd { float v[MI]; bool on; };
__device__ __forceinline__ LnRow cgd_ln_row(const AaConvGemm& p, const int m) {
    if (p.ln_parts > 0) {
        // ln_stats = the PRODUCER's partial (sum, sum of squares) pairs, [M][ln_parts][2]: finalised here, in the arithmetic and the
        // order of ln_finalize_kernel (norm.h) - the 99 finalize launches of a step (4.6 us each, r04g) disappear
        const float* st = p.ln_stats + (int64_t)m * (p.ln_parts * 2);
        float s = 0.0f, q = 0.0f;
        if (p.ln_parts == 2) {                    // a producer tile that spans the row (320 channels): ONE 16-byte load per row, as with coefficients
            const f32x4 v = *reinterpret_cast<const f32x4*>(st);
            s = v[0] + v[2]; q = v[1] + v[3];
        } else if (p.ln_parts == 10) {            // five 128-column (256-column) tiles of two waves: 640 (1280) channels - five independent loads
            f32x4 v[5];
#pragma unroll
            for (int t = 0; t < 5; ++t) v[t] = *reinterpret_cast<const f32x4*>(st + 4 * t);
#pragma unroll
            for (int t = 0; t < 5; ++t) { s += v[t][0]; q += v[t][1]; s += v[t][2]; q += v[t][3]; }
        } else {
            // (correct for any count, but a chain of dependent loads: measured +12 ... +23 % on the consuming kernels at 10 parts,
            //  r04h - callers with other part counts keep the aa_ln_finalize launch)
            for (int t = 0; t < p.ln_parts; ++t) {
                const f32x2 v = *reinterpret_cast<const f32x2*>(st + 2 * t);
                s += v[0]; q += v[1];
            }
        }
        const float inv_c = 1.0f / (float)p.c0;
        const float mean = s * inv_c;
        const float var = fmaxf(q * inv_c - mean * mean, 0.0f);
        const float sd = sqrtf(var + p.ln_eps);
        const float rstd = 1.0f / sd;
        return LnRow{rstd, -mean * rstd, -mean, sd};
    }
    const f32x4 c = *reinterpret_cast<const f32x4*>(p.ln_stats + (int64_t)m * 4);
    return LnRow{c[2], c[0] * c[2], c[0], c[1]};
}

// sCoef (STATS with AaConvGemm.row_coef, version 107): LDS scratch [waves][MI * 32][2] fp32 of a workgroup whose tile spans the output row, WN_ its
// waves per tile row: the waves park their partial (sum, sum of squares) there, meet at a barrier, and the first wave of every tile row adds them in
// wave order and stores the finished coefficients - the arithmetic and order of ln_finalize_kernel (norm.h), one launch fewer per LayerNorm.
template <typename T, int MI, int NI, bool GEGLU, bool RV, bool POST, bool BIAS, bool LNF, bool STATS, typename Get>
__device__ __forceinline__ void cgd_epilogue_fast(const AaConvGemm& p, const int M, Get&& get, const int m_wave, const int n_wave, const T* sBiasW, const int part, const LnRstd<MI> ln_rstd,
                                                  float* sCoef = nullptr, const int WN_ = 1) {
    // ln_rstd (LNF): rstd of this lane's row of every block row, still in the caller's registers from its accumulator start
    static_assert(!GEGLU || (NI % 2 == 0 && !RV && !POST), "GEGLU pairs value block j with gate block j + 1");
    static_assert(!LNF || (!RV && !POST && !BIAS && !STATS), "the LayerNorm fold covers the plain and the GEGLU form");
    static_assert(!STATS || !GEGLU, "row statistics: plain / residual forms");
    constexpr unsigned OOB = 0x80000000u;
    // a column past the tensor's width: out of range of every descriptor used here (all shorter than 2^31 - 32 bytes), and - unlike
    // OOB + OOB - still out of range when added to the OOB row offset of a scattered output grid (r04: 2^31 + 2^31 wrapped to 0)
    constexpr unsigned COL_OOB = 0x7ffffff0u;
    constexpr int NJ = GEGLU ? NI / 2 : NI;                 // output blocks per block row
    const int lane = threadIdx.x & 63;
    const int ec = lane & 31, eh = lane >> 5;
    const int n_cols = GEGLU ? (p.n_out >> 1) : p.n_out;
    const int rows_ok = max(0, min(MI * 32, M - m_wave));
    const bool mapped = cgd_out_mapped(p);                  // rows scatter over a bigger output grid: whole-tensor descriptor, absolute rows
    const BufRsrc r_out = mapped ? make_rsrc(p.out, (unsigned)((int64_t)p.n_img * p.h_out * p.w_out * (p.out_sy > 1 ? p.out_sy : 1) * (p.out_sx > 1 ? p.out_sx : 1) * p.ldo * 2))
                                 : make_rsrc(reinterpret_cast<const T*>(p.out) + (int64_t)m_wave * p.ldo, (unsigned)(rows_ok * p.ldo) * 2u);
    const T* resid = reinterpret_cast<const T*>(p.residual);
    const BufRsrc r_res = make_rsrc(resid ? resid + (int64_t)m_wave * p.ldr : resid, resid ? (unsigned)(rows_ok * p.ldr) * 2u : 0u);
    const BufRsrc r_rv = make_rsrc(p.rowvec, 0x7fffffffu);
    const float acc_scale = p.acc_scale != 0.0f ? p.acc_scale : 1.0f;
    const float out_scale = p.out_scale;
    // byte offset of this lane's columns inside an output (= residual) row, per output block and 16-byte half
    unsigned coff[NJ][2];
#pragma unroll
    for (int h = 0; h < NJ; ++h)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int nc = (GEGLU ? (n_wave >> 1) : n_wave) + h * 32 + 16 * eh + 8 * q;
            coff[h][q] = nc + 8 <= n_cols ? (unsigned)nc * 2u : COL_OOB;
        }
    // the row vector (else the residual) of a whole block row is fetched one block row AHEAD: a wave that is alone on its SIMD
    // has nothing else to run while a load is in flight
    u32x4 pre[2][NJ][2];
    auto prefetch = [&](auto i_) __attribute__((always_inline)) {
        constexpr int i = decltype(i_)::value;
        const unsigned row = (unsigned)(i * 32 + ec);
        if constexpr (RV) {
            const int mc = min(m_wave + (int)row, M - 1);
            const unsigned rv_row = (unsigned)((mc / p.rowvec_div) * (p.rowvec_ld ? p.rowvec_ld : p.n_out)) * 2u;
#pragma unroll
            for (int h = 0; h < NJ; ++h)
#pragma unroll
                for (int q = 0; q < 2; ++q) pre[i & 1][h][q] = buf_load16(r_rv, rv_row + coff[h][q]);
        } else if constexpr (POST) {
            const unsigned r_row = row * (unsigned)p.ldr * 2u;
#pragma unroll
            for (int h = 0; h < NJ; ++h)
#pragma unroll
                for (int q = 0; q < 2; ++q) pre[i & 1][h][q] = buf_load16(r_res, r_row + coff[h][q]);
        }
    };
    prefetch(IntTag<0>());
    // LayerNorm fold: rstd of this lane's row of every block row
    float ln_a[LNF ? MI : 1];
    if constexpr (LNF) {
#pragma unroll
        for (int i = 0; i < MI; ++i) ln_a[i] = ln_rstd.on ? ln_rstd.v[i] : cgd_ln_row(p, min(m_wave + i * 32 + ec, M - 1)).a;
    }
    // row statistics of the STORED values (sum, sum of squares over this wave's columns): two scalars per lane, fed by packed dot
    // products on the 16-bit output pairs (exact products, fp32 sums); the two half-waves hold the two 16-column halves of a row
    // and are joined once per block row.  (Not matrix-core products: see dev.h dot2_f32.)
    static_for<MI>([&](auto i_) __attribute__((always_inline)) {
        constexpr int i = decltype(i_)::value;
        const unsigned row = (unsigned)(i * 32 + ec);
        float st_sum = 0.0f, st_sq = 0.0f;
        const unsigned o_row = !mapped ? row * (unsigned)p.ldo * 2u
                                       : ((int)row < rows_ok ? (unsigned)(cgd_out_row(p, m_wave + (int)row) * p.ldo) * 2u : OOB);
        if constexpr (i + 1 < MI) prefetch(IntTag<i + 1>());
        static_for<NJ>([&](auto h_) __attribute__((always_inline)) {
            constexpr int h = decltype(h_)::value, j = GEGLU ? 2 * h : h;
            float v[16];
            {
                const f32x16 a = get(IntTag<i>(), IntTag<j>());
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = a[e];
            }
            if constexpr (BIAS) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    Pack8<T> b; b.raw = *reinterpret_cast<const u32x4*>(sBiasW + (j * 32 + 16 * eh + 8 * q));
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[8 * q + e] += (float)b.e[e];
                }
            }
            if constexpr (RV) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    Pack8<T> r; r.raw = pre[i & 1][h][q];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[8 * q + e] += (float)r.e[e];
                }
            }
            auto ln_apply = [&](float (&x)[16], const int) __attribute__((always_inline)) {       // the accumulation started from the fold's
              if constexpr (LNF) {                                                                  // rank-1 terms (conv_gemm_x.h): x *= rstd
#pragma unroll
                for (int e = 0; e < 16; ++e) x[e] *= ln_a[i];       // (scalar on purpose: as packed pairs this cost the GEGLU epilogue 0.9 k clk per tile - r04pk probe)
              }
            };
            if constexpr (LNF) ln_apply(v, j);
            if constexpr (GEGLU) {
                float g[16];
                {
                    const f32x16 a = get(IntTag<i>(), IntTag<j + 1>());
#pragma unroll
                    for (int e = 0; e < 16; ++e) g[e] = a[e];
                }
                if constexpr (LNF) ln_apply(g, j + 1);
                if constexpr (BIAS) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        Pack8<T> b; b.raw = *reinterpret_cast<const u32x4*>(sBiasW + ((j + 1) * 32 + 16 * eh + 8 * q));
#pragma unroll
                        for (int e = 0; e < 8; ++e) g[8 * q + e] += (float)b.e[e];
                    }
                }
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const f32x2 y = (AA_X_ABLATE & 8) ? f32x2{g[e], g[e + 1]} : gelu_erf_2(f32x2{g[e], g[e + 1]});      // (ablation build: no GELU)
                    v[e] *= y[0]; v[e + 1] *= y[1];
                }
            }
            if constexpr (POST) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    Pack8<T> r;
                    if constexpr (RV) r.raw = buf_load16(r_res, row * (unsigned)p.ldr * 2u + coff[h][q]); else r.raw = pre[i & 1][h][q];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[8 * q + e] = (v[8 * q + e] * acc_scale + (float)r.e[e]) * out_scale;
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                Pack8<T> o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o.e[e] = (T)v[8 * q + e];
                if constexpr ((AA_X_ABLATE & 16) != 0) { if (o.raw[0] == 0x12345678u) buf_store16(r_out, o_row + coff[h][q], o.raw); }   // (ablation build: no stores)
                else buf_store16(r_out, o_row + coff[h][q], o.raw);
                if constexpr (STATS) {
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        st_sum = dot2_f32(T(), o.raw[w], ones_pair(T()), st_sum);
                        st_sq = dot2_f32(T(), o.raw[w], o.raw[w], st_sq);
                    }
                }
            }
        });
        if constexpr (STATS) {
            const float row_sum = wave_sum_halves(st_sum), row_sq = wave_sum_halves(st_sq);      // the other 16 columns of every block
            const int m = m_wave + (int)row;
            if (sCoef != nullptr) {                      // (the whole workgroup takes this branch or none: p.row_coef)
                if (eh == 0) {
                    float* dst = sCoef + ((int)(threadIdx.x >> 6) * (MI * 32) + (int)row) * 2;
                    dst[0] = row_sum;
                    dst[1] = row_sq;
                }
            } else if (eh == 0 && m < M) {
                float* dst = p.row_stats + ((int64_t)m * p.row_stats_parts + part) * 2;
                dst[0] = row_sum;
                dst[1] = row_sq;
            }
        }
    });
    if constexpr (STATS) {
        if (sCoef != nullptr) {
            __syncthreads();                             // every wave of the tile has parked its partial sums
            const int wave = (int)(threadIdx.x >> 6);
            if (wave % WN_ == 0 && eh == 0) {            // first wave of a tile row: its rows, the partials of waves wave .. wave + WN_ - 1 in order
                const float inv_c = 1.0f / (float)p.n_out;
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int r = i * 32 + ec, m = m_wave + r;
                    float s_ = 0.0f, q_ = 0.0f;
                    for (int t = 0; t < WN_; ++t) {
                        const float* src = sCoef + ((wave + t) * (MI * 32) + r) * 2;
                        s_ += src[0]; q_ += src[1];
                    }
                    const float mean = s_ * inv_c;
                    const float var = fmaxf(q_ * inv_c - mean * mean, 0.0f);
                    const float sd = sqrtf(var + p.row_coef_eps);
                    if (m < M) *reinterpret_cast<f32x4*>(p.row_coef + (int64_t)m * 4) = f32x4{-mean, sd, 1.0f / sd, 0.0f};
                }
            }
        }
    }
}

// BIAS_FOLDED: the accumulators were started from the bias (conv_gemm_x.h) and sBiasW holds zeros.
// part: this wave's slot of AaConvGemm.row_stats (-1: the caller cannot emit them).  ln_rstd.on: the caller has started its
// accumulators from the rank-1 terms of a folded LayerNorm (AaConvGemm.ln_stats): its epilogue form only scales by rstd.
template <typename T, int MI, int NI, bool BIAS_FOLDED = false, typename Get>
__device__ __forceinline__ void cgd_epilogue_g(const AaConvGemm& p, const int M, Get&& get, const int m_wave, const int n_wave, const T* sBiasW,
                                               const int part = -1, const LnRstd<MI> ln_rstd = LnRstd<MI>{{}, false}, float* sCoef = nullptr, const int WN_ = 1) {
    // get(IntTag<i>, IntTag<j>) -> the 16 accumulators of 32x32 block (i, j) of this lane (an array element, or a read-out
    // of the literal accumulation registers of conv_gemm_x.h)
    const int lane = threadIdx.x & 63;
    const int ec = lane & 31, eh = lane >> 5;              // lane owns output row ec, columns 16*eh .. +15 of a block
    const T* rowvec = reinterpret_cast<const T*>(p.rowvec);
    const T* resid = reinterpret_cast<const T*>(p.residual);
    const T* bias = reinterpret_cast<const T*>(p.bias);
    T* out = reinterpret_cast<T*>(p.out);
    const int n_cols = p.geglu ? (p.n_out >> 1) : p.n_out;
    const float acc_scale = p.acc_scale != 0.0f ? p.acc_scale : 1.0f;
    const bool post = resid || p.out_scale != 1.0f || acc_scale != 1.0f;
    const bool silu = p.act == AA_ACT_SILU;
    const bool pre_is_rv = rowvec != nullptr;             // the prefetch registers carry the row vector, else the residual
    const bool lnf = p.ln_stats != nullptr;               // LayerNorm folded into this contraction (host: no bias / row vector / residual with it)
    const bool coef = p.row_coef != nullptr && part >= 0 && sCoef != nullptr;      // (host: only where the tile spans the row - aa_conv_gemm_row_coef_ok)
    const bool stats = (p.row_stats != nullptr || coef) && part >= 0;
    float* const sC = coef ? sCoef : nullptr;
    // the combinations that carry the step take the branch-free forms - where the accumulators sit in the accumulation registers
    // (conv_gemm_x.h): next to 128-160 accumulators in VGPRs the extra offsets spill (measured: 160-220 dwords of scratch)
    // (a folded LayerNorm takes its rstd-only form only where THIS kernel instance started its accumulators from the fold's rank-1
    //  terms - ln_rstd.on; anything else falls through to the general path below, which applies the whole formula: ADVICE r04)
    if (BIAS_FOLDED && !silu && !p.bias_per_row && !(lnf && !ln_rstd.on)) {
        if (lnf) {
            if (p.geglu) {
                if constexpr (NI % 2 == 0) { cgd_epilogue_fast<T, MI, NI, true, false, false, false, true, false>(p, M, get, m_wave, n_wave, sBiasW, part, ln_rstd); return; }
            } else { cgd_epilogue_fast<T, MI, NI, false, false, false, false, true, false>(p, M, get, m_wave, n_wave, sBiasW, part, ln_rstd); return; }
        } else if (p.geglu) {
            if constexpr (NI % 2 == 0) { if (!rowvec && !post) { cgd_epilogue_fast<T, MI, NI, true, false, false, !BIAS_FOLDED, false, false>(p, M, get, m_wave, n_wave, sBiasW, part, ln_rstd); return; } }
        } else if (rowvec) {
            if (post) cgd_epilogue_fast<T, MI, NI, false, true, true, !BIAS_FOLDED, false, false>(p, M, get, m_wave, n_wave, sBiasW, part, ln_rstd);
            else cgd_epilogue_fast<T, MI, NI, false, true, false, !BIAS_FOLDED, false, false>(p, M, get, m_wave, n_wave, sBiasW, part, ln_rstd);
            return;
        } else if (stats) {
            if (post) cgd_epilogue_fast<T, MI, NI, false, false, true, !BIAS_FOLDED, false, true>(p, M, get, m_wave, n_wave, sBiasW, part, ln_rstd, sC, WN_);
            else cgd_epilogue_fast<T, MI, NI, false, false, false, !BIAS_FOLDED, false, true>(p, M, get, m_wave, n_wave, sBiasW, part, ln_rstd, sC, WN_);
            return;
        } else {
            if (post) cgd_epilogue_fast<T, MI, NI, false, false, true, !BIAS_FOLDED, false, false>(p, M, get, m_wave, n_wave, sBiasW, part, ln_rstd);
            else cgd_epilogue_fast<T, MI, NI, false, false, false, !BIAS_FOLDED, false, false>(p, M, get, m_wave, n_wave, sBiasW, part, ln_rstd);
            return;
        }
    }
#define AA_ZERO4 (u32x4{0u, 0u, 0u, 0u})          /* a prvalue: `c ? arr[i] : zero_variable` would select between ADDRESSES and pin arr in scratch */
    auto col_of = [&](int j) __attribute__((always_inline)) { return p.geglu ? (n_wave >> 1) + (j >> 1) * 32 + 16 * eh : n_wave + j * 32 + 16 * eh; };
    static_for<MI>([&](auto i_) __attribute__((always_inline)) {
        constexpr int i = decltype(i_)::value;
        const int m = m_wave + i * 32 + ec;
        const bool m_ok = m < M;
        const int mc = m_ok ? m : M - 1;
        const float brow = (p.bias_per_row && bias) ? (float)bias[mc] : 0.0f;
        const T* rv = rowvec ? rowvec + (int64_t)(mc / p.rowvec_div) * (p.rowvec_ld ? p.rowvec_ld : p.n_out) : nullptr;
        const T* rs = resid ? resid + (int64_t)mc * p.ldr : nullptr;
        const int64_t o_row_g = cgd_out_row(p, mc) * p.ldo;
        const LnRow lr = lnf ? cgd_ln_row(p, mc) : LnRow{1.0f, 0.0f, 0.0f, 1.0f};
        // one block-row of row-vector (or residual) pieces is fetched up front so their latency overlaps
        u32x4 pre[NI][2];
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int n_rv = n_wave + j * 32 + 16 * eh + 8 * q, n_rs = col_of(j) + 8 * q;
                const bool ok = pre_is_rv ? (n_rv + 8 <= p.n_out) : (rs != nullptr && !(p.geglu && (j & 1)) && n_rs + 8 <= n_cols);
                const T* src = pre_is_rv ? rv + n_rv : rs + n_rs;
                pre[j][q] = ok ? *reinterpret_cast<const u32x4*>(src) : AA_ZERO4;
            }
        auto finish_block = [&](int j, float (&v)[2][8]) __attribute__((always_inline)) {   // residual / scale in fp32, ONE rounding, the two stores of a 32x32 block
            const int nc = col_of(j);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const u32x4 pv = pre[j][q];               // (read unconditionally: a load in only one branch gets merged with the
                if (post) {                               //  global load of the other into one load through a selected POINTER)
                    Pack8<T> r; r.raw = AA_ZERO4;
                    if (rs) { if (pre_is_rv) { if (nc + 8 * q + 8 <= n_cols) r.raw = *reinterpret_cast<const u32x4*>(rs + nc + 8 * q); } else r.raw = pv; }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[q][e] = (v[q][e] * acc_scale + (float)r.e[e]) * p.out_scale;
                }
                Pack8<T> o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o.e[e] = (T)v[q][e];
                if ((AA_X_ABLATE & 16) ? (o.raw[0] == 0x12345678u && m_ok) : (m_ok && nc + 8 * q + 8 <= n_cols)) *reinterpret_cast<u32x4*>(out + o_row_g + nc + 8 * q) = o.raw;   // (ablation build: no stores)
            }
        };
        auto block_f32 = [&](auto j_, float (&v)[2][8]) __attribute__((always_inline)) {     // accumulators + bias (+ row vector) (+ SiLU) in fp32
            constexpr int j = decltype(j_)::value;
            const f32x16 a = get(IntTag<i>(), IntTag<j>());
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                Pack8<T> b; b.raw = *reinterpret_cast<const u32x4*>(sBiasW + (j * 32 + 16 * eh + 8 * q));
                const u32x4 pv = pre[j][q];
                Pack8<T> r; r.raw = AA_ZERO4;
                if (pre_is_rv) r.raw = pv;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[q][e] = a[8 * q + e] + (float)b.e[e] + brow;
                if (lnf && !ln_rstd.on) {                 // rstd * acc - rstd * mean * colsum(W') + b'  (nothing was folded into the accumulator start)
                    const float* cs = p.ln_cols + n_wave + j * 32 + 16 * eh + 8 * q;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[q][e] = fmaf(v[q][e], lr.a, fmaf(lr.nm, cs[e], cs[p.n_pad + e]));
                }
                if (pre_is_rv) {                          // uniform branches once per eight values, not once per value
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[q][e] += (float)r.e[e];
                }
                if (silu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[q][e] = silu_f(v[q][e]);
                }
            }
        };
        if (p.geglu) {
            if constexpr (NI % 2 == 0) {
                static_for<NI / 2>([&](auto h_) __attribute__((always_inline)) {
                    constexpr int j = 2 * decltype(h_)::value;
                    // value * gelu(gate) in fp32 on the accumulators, ONE rounding to the storage type
                    float val[2][8], gate[2][8];
                    block_f32(IntTag<j>(), val);
                    block_f32(IntTag<j + 1>(), gate);
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int e = 0; e < 8; ++e) val[q][e] *= (AA_X_ABLATE & 8) ? gate[q][e] : gelu_erf_f(gate[q][e]);      // (ablation build: no GELU)
                    finish_block(j, val);
                });
            }
        } else {
            static_for<NI>([&](auto j_) __attribute__((always_inline)) {
                float v[2][8];
                block_f32(j_, v);
                finish_block(decltype(j_)::value, v);
            });
        }
    });
#undef AA_ZERO4
}

template <typename T, int MI, int NI>
__device__ __forceinline__ void cgd_epilogue(const AaConvGemm& p, const int M, f32x16 (&acc)[MI][NI], const int m_wave, const int n_wave,
                                             const T* sBiasW) {
    cgd_epilogue_g<T, MI, NI, false>(p, M, [&](auto i_, auto j_) __attribute__((always_inline)) -> const f32x16& { return acc[decltype(i_)::value][decltype(j_)::value]; },
                              m_wave, n_wave, sBiasW);
}

// PER_CU = workgroups meant to be co-resident on a CU (register budget: 512 / (PER_CU * waves per SIMD)).
// SPREAD: a wave's DMA share of the next tile is not issued in one burst but in pieces in front of the first k sub-steps of
// its multiply (each piece queues in the CU-wide LDS-DMA issue path while the wave's previous MFMAs still occupy the matrix pipe).
template <typename T, int BM, int BN, int WM, int WN, int BK, int STAGES, int PER_CU, bool STAGGER, bool SPREAD = false>
__global__ void __launch_bounds__(64 * WM * WN, (PER_CU * WM * WN + 3) / 4) conv_gemm_dma_kernel(const AaConvGemm p, const int M, const int tiles_n, const int m_begin, const int k_splits) {
    // rows [m_begin, M) of the output are tiled by this launch (a launch may cover only part of the rows:
    // the host splits off a sparsely filled last round of big tiles and runs it with small tiles)
    constexpr int NW = WM * WN;
    constexpr int THREADS = 64 * NW;
    constexpr int MI = BM / WM / 32;     // 32-row accumulator blocks per wave
    constexpr int NI = BN / WN / 32;     // 32-column accumulator blocks per wave
    constexpr int ROWB = BK * 2;         // bytes per LDS tile row (128 or 64)
    constexpr int SPR = BK / 8;          // 16-byte slots per row (8 or 4)
    constexpr int RPI = 64 / SPR;        // tile rows deposited by one wave DMA instruction (8 or 16)
    constexpr int RPB = 256 / ROWB;      // tile rows per 256-byte LDS bank row (2 or 4)
    constexpr int GA = BM / RPI, GB = BN / RPI;          // DMA groups of the activation / weight tile
    constexpr int AJ = (GA + NW - 1) / NW;               // DMA instructions per wave (group = wave + NW*j)
    constexpr int BJ = (GB + NW - 1) / NW;
    constexpr int PER_TILE = AJ + BJ;                    // every wave issues exactly this many per tile
    constexpr int DIST = STAGES - 1;                     // tiles in flight ahead of the multiply
    constexpr int STAGE_BYTES = (BM + BN) * ROWB;
    // accumulators + two fragment sets fit beside the rest.  (One-wave-per-SIMD tiles with 128x160 per wave - 0.45 LDS
    // fragment reads per MFMA instead of 0.7 - need the accumulators in AGPRs beyond 256 registers: hipcc keeps them in
    // VGPRs and spills 2.4 KB of scratch instead; measured unusable, not in the table.)
    constexpr bool FRAG_ASM = (MI * NI * 16 + 2 * (MI + NI) * 4) <= 184;
    static_assert(BM % (32 * WM) == 0 && BN % (32 * WN) == 0 && BM % RPI == 0 && BN % RPI == 0, "tile shape");
    static_assert(BK == 64 || BK == 32, "K step");
    static_assert(STAGES >= 2 && STAGES <= 4 && (STAGES - 2) * PER_TILE <= 63, "pipeline depth");
    char* smem = dyn_smem();
    char* dummy = smem + STAGES * STAGE_BYTES;           // where surplus (guarded-out) DMA instructions land
    T* sBias = reinterpret_cast<T*>(dummy + 1024);       // bias of the tile's BN columns (the K-loop barriers publish it)
    static_assert(BN * 2 <= 1024, "bias slice");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = wave_id();          // scalar (SGPR): everything derived from it stays on the scalar unit
    const int wm = wave / WN, wn = wave % WN;

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int tile_m = logical / tiles_n;
    const int tile_n = logical - tile_m * tiles_n;

    // debug bit 8 (phase probe): thread 0 leaves shader-clock stamps in workspace[bid][8]
    auto stamp = [&](int slot) {
        if ((p.debug & 8) && tid == 0) reinterpret_cast<long long*>(p.workspace)[(int64_t)blockIdx.x * 8 + slot] = clock_now();
    };
    stamp(0);
    const int ctot = p.c0 + p.c1;
    // split-K: blockIdx.y owns K steps [kbase, kbase + nk) and leaves raw fp32 partial sums in the workspace
    const int nk_all = p.k_pad / BK;
    const int k_per = (nk_all + k_splits - 1) / k_splits;
    const int kbase = blockIdx.y * k_per;
    const int nk = max(0, min(k_per, nk_all - kbase));
    const bool resize = (p.h_virt != p.h_in) || (p.w_virt != p.w_in);
    const bool linear = p.kh * p.kw == 1 && p.stride == 1 && p.pad_h == 0 && p.pad_w == 0 && !resize;

    // ---- DMA geometry: this lane feeds LDS rows ((wave + NW*j)*RPI + lane/SPR), 16-byte position lane%SPR ----
    // Operands are read through buffer descriptors (base in SGPRs, one 32-bit byte offset per lane): halo, tail and
    // K-padding lanes carry an offset >= 2^31, which the range check of the descriptor turns into 16 zero bytes -
    // no zero page, no 64-bit per-lane pointers (the host guarantees every operand is < 2 GiB).
    constexpr unsigned OOB = 0x80000000u;
    const BufRsrc r_a0 = make_rsrc(p.a0, (unsigned)((int64_t)p.n_img * p.h_in * p.w_in * p.c0 * 2));
    const BufRsrc r_a1 = make_rsrc(p.a1, p.c1 ? (unsigned)((int64_t)p.n_img * p.h_in * p.w_in * p.c1 * 2) : 0u);
    const BufRsrc r_w = make_rsrc(p.w, (unsigned)((int64_t)p.n_pad * p.k_pad * 2));
#ifdef AA_PHASE_PROBE
    const int swm = (p.debug & 16) ? 0 : SPR - 1;        // probe build, debug bit 16: no XOR swizzle (LDS bank conflicts of the fragment reads)
    const bool dry = p.debug & 32;                       // probe build, debug bit 32: every DMA piece out of range (issue cost without memory traffic)
#else
    constexpr int swm = SPR - 1;
    constexpr bool dry = false;
#endif
    const bool two_src = p.c1 != 0;                      // channel-concatenated input (up blocks): a second offset set
    const int lrow = lane / SPR, lpos = lane % SPR;
    // per fed activation row: byte offset of its top-left tap pixel (+ this lane's swizzled k-slot) in a0 / a1, and which
    // taps read a real pixel: bits 0..7 = rows dy inside the image, bits 8..15 = columns dx (0 for rows outside the tile)
    unsigned ctr0[AJ], ctr1[AJ], vmask[AJ];
    struct RowCoords { int img, iy, ix; bool ok; };
    auto row_coords = [&](int j) __attribute__((always_inline)) {
        const int rr = (wave + NW * j) * RPI + lrow;
        const int m = m_begin + tile_m * BM + rr;
        const bool ok = m < M && rr < BM;
        const int mm = ok ? m : 0;
        if (linear) return RowCoords{0, 0, mm, ok};                               // 1x1 / nn.Linear: the row IS the pixel
        const int x = mm % p.w_out;
        const int t = mm / p.w_out;
        const int y = t % p.h_out;
        return RowCoords{t / p.h_out, y * p.stride - p.pad_h, x * p.stride - p.pad_w, ok};
    };
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const RowCoords rc = row_coords(j);
        const int img = rc.img, iy = rc.iy, ix = rc.ix;
        const bool ok = rc.ok;
        const int rr = (wave + NW * j) * RPI + lrow;
        const int slot8 = (lpos ^ ((rr / RPB) & swm)) * 8;
        const int pix = (img * p.h_in + iy) * p.w_in + ix;                        // may be "negative": only used for in-range taps
        ctr0[j] = (unsigned)(pix * p.c0 + slot8) * 2u;
        ctr1[j] = (unsigned)(pix * p.c1 + slot8) * 2u;
        // taps dy with 0 <= iy + dy < h_virt form an interval [lo, hi) (same for dx): two shifts instead of a loop
        const int ylo = max(0, -iy), yhi = max(ylo, min(p.kh, p.h_virt - iy));
        const int xlo = max(0, -ix), xhi = max(xlo, min(p.kw, p.w_virt - ix));
        const unsigned my = ((1u << yhi) - 1u) ^ ((1u << ylo) - 1u), mx = ((1u << xhi) - 1u) ^ ((1u << xlo) - 1u);
        vmask[j] = ok ? (my | (mx << 8)) : 0u;
    }
    // weight panel of this tile: byte offsets into the packed tensor
    unsigned wb[BJ];
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int rr = (wave + NW * j) * RPI + lrow;
        wb[j] = (unsigned)((tile_n * BN + rr) * p.k_pad + (lpos ^ ((rr / RPB) & swm)) * 8) * 2u;
    }

    const int taps = p.kh * p.kw;

    // K position of the NEXT issue() call as wave-uniform scalars, advanced incrementally (issue() is called for
    // consecutive K tiles): tap-major order (tap, channel) or, for multi-tap filters packed chunk-major,
    // (64-channel chunk, tap, channel) - consecutive K steps then re-read the same activation slab shifted by one tap,
    // which keeps it L2-resident across the 9 taps.
    int n_tap, n_dy, n_dx, n_cb;
    {
        const int k0 = kbase * BK;
        if (p.k_order) { const int unit = k0 >> 6; const int chunk = unit / taps; n_tap = unit - chunk * taps; n_cb = chunk * 64 + (k0 & 63); }
        else           { n_tap = k0 / ctot; n_cb = k0 - n_tap * ctot; }
        n_dy = n_tap / p.kw; n_dx = n_tap - n_dy * p.kw;
    }
    int cur_tap = -1;
    unsigned pb0[AJ], pb1[AJ];           // byte offset of each fed row's source pixel for the current tap (OOB = halo / tail)
    // state of the tile being issued (issue_prepare -> issue_dma pieces)
    bool is_src1 = false;
    unsigned is_ccb = 0u, is_kb = 0u;
    int is_buf = 0;
    auto issue_prepare = [&](int kt, int buf) {
        const int tap = n_tap, cb = n_cb, dy = n_dy, dx = n_dx;
        // advance to the next K tile (selects, not `++x` in branches: those get tail-merged into one increment through
        // a selected POINTER, which pins the counters in scratch)
        {
            const bool wrap_x = n_dx + 1 == p.kw;
            if (p.k_order) {
                const bool half = BK == 32 && !(n_cb & 32);              // first half of a 64-channel unit: same tap
                const bool last_tap = n_tap + 1 == taps;
                const int cb_unit = (n_cb & ~63) + (last_tap ? 64 : 0);
                const int t1 = last_tap ? 0 : n_tap + 1;
                const int y1 = last_tap ? 0 : (wrap_x ? n_dy + 1 : n_dy);
                const int x1 = (last_tap || wrap_x) ? 0 : n_dx + 1;
                n_cb = half ? n_cb + 32 : cb_unit;
                n_tap = half ? n_tap : t1; n_dy = half ? n_dy : y1; n_dx = half ? n_dx : x1;
            } else {
                const bool last_c = n_cb + BK >= ctot;
                n_cb = last_c ? 0 : n_cb + BK;
                n_tap = last_c ? n_tap + 1 : n_tap;
                n_dy = (last_c && wrap_x) ? n_dy + 1 : n_dy;
                n_dx = last_c ? (wrap_x ? 0 : n_dx + 1) : n_dx;
            }
        }
        if (tap != cur_tap) {
            cur_tap = tap;
            const bool tap_ok = tap < taps;
            if (!resize) {                                     // (branches hoisted out of the per-row loops)
                const int d = dy * p.w_in + dx;
                const unsigned d0 = (unsigned)(d * p.c0) * 2u;
#pragma unroll
                for (int j = 0; j < AJ; ++j) {
                    const bool ok = tap_ok && (((vmask[j] >> dy) & (vmask[j] >> (8 + dx))) & 1u);
                    pb0[j] = ok ? ctr0[j] + d0 : OOB;
                }
                if (two_src) {
                    const unsigned d1 = (unsigned)(d * p.c1) * 2u;
#pragma unroll
                    for (int j = 0; j < AJ; ++j) {
                        const bool ok = tap_ok && (((vmask[j] >> dy) & (vmask[j] >> (8 + dx))) & 1u);
                        pb1[j] = ok ? ctr1[j] + d1 : OOB;
                    }
                }
            } else {                                           // nearest-neighbour resize in front (Upsample2D)
#pragma unroll
                for (int j = 0; j < AJ; ++j) {
                    const bool ok = tap_ok && (((vmask[j] >> dy) & (vmask[j] >> (8 + dx))) & 1u);
                    const RowCoords rc = row_coords(j);
                    const int sy = ((rc.iy + dy) * p.h_in) / p.h_virt, sx = ((rc.ix + dx) * p.w_in) / p.w_virt;
                    const int pix = (rc.img * p.h_in + sy) * p.w_in + sx;
                    const int rr = (wave + NW * j) * RPI + lrow;
                    const int slot8 = (lpos ^ ((rr / RPB) & swm)) * 8;
                    pb0[j] = ok ? (unsigned)(pix * p.c0 + slot8) * 2u : OOB;
                    pb1[j] = ok ? (unsigned)(pix * p.c1 + slot8) * 2u : OOB;
                }
            }
        }
        is_src1 = cb >= p.c0;
        is_ccb = (unsigned)(is_src1 ? cb - p.c0 : cb) * 2u;
        is_kb = (unsigned)((kbase + kt) * BK) * 2u;
        is_buf = buf;
    };
    // DMA instructions [J0, J1) of the prepared tile: indices < AJ feed activation rows, the rest weight rows.
    // every wave issues exactly PER_TILE instructions per tile (vmcnt bookkeeping): groups past the tile edge read
    // out of range and land in the dummy zone (wave-uniform choice)
    auto issue_dma = [&](auto j0_, auto j1_) {
        constexpr int J0 = decltype(j0_)::value, J1 = decltype(j1_)::value;
        const BufRsrc ra = is_src1 ? r_a1 : r_a0;
        char* a = smem + is_buf * STAGE_BYTES + wave * RPI * ROWB;
        char* b = smem + is_buf * STAGE_BYTES + BM * ROWB + wave * RPI * ROWB;
#pragma unroll
        for (int jj = J0; jj < J1; ++jj) {
            if (jj < AJ) {
                const int j = jj;
                const bool real = (GA % NW == 0) || (wave + NW * j < GA);
                async_copy16_buf(ra, (real && !dry) ? (is_src1 ? pb1[j] : pb0[j]) + is_ccb : OOB, real ? a + j * NW * RPI * ROWB : dummy);
            } else {
                const int j = jj - AJ;
                const bool real = (GB % NW == 0) || (wave + NW * j < GB);
                async_copy16_buf(r_w, (real && !dry) ? wb[j] + is_kb : OOB, real ? b + j * NW * RPI * ROWB : dummy);
            }
        }
    };
    auto issue = [&](int kt, int buf) {
        issue_prepare(kt, buf);
        issue_dma(IntTag<0>(), IntTag<PER_TILE>());
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // fragment read offsets (bytes): row = base + (lane&31), k-slot ks*2 + (lane>>5), un-swizzled per row
    const int frow = lane & 31, fh = lane >> 5;
    int a_off[MI], b_off[NI], a_swz[MI], b_swz[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) { const int rr = wm * (BM / WM) + i * 32 + frow; a_off[i] = rr * ROWB; a_swz[i] = (rr / RPB) & swm; }
    // weights are the MFMA "A" operand (rows -> accumulator registers); MFMA row r is fed tile row pi(r) =
    // (r&3) + 4*(r>>3) + 16*((r>>2)&1), which makes the 16 registers of a lane 16 consecutive output columns
    const int prow = (frow & 3) + 4 * (frow >> 3) + 16 * ((frow >> 2) & 1);
#pragma unroll
    for (int j = 0; j < NI; ++j) { const int rr = wn * (BN / WN) + j * 32 + prow; b_off[j] = BM * ROWB + rr * ROWB; b_swz[j] = (rr / RPB) & swm; }

    // Fragment reads.  FRAG_ASM: two register sets, the ds_reads of sub-step ks+1 are hand-issued before the
    // MFMAs of sub-step ks and waited with a counted lgkmcnt (LDS returns in order: "at most MI+NI outstanding"
    // == the older set has landed), so LDS latency hides under the matrix pipe.  Otherwise plain C++ loads
    // (hipcc keeps ONE set and waits lgkmcnt(0) each sub-step) - for shapes whose accumulators leave no room.
    auto frag_ptr_a = [&](const char* st, int ks, int i) { return st + a_off[i] + (((ks * 2 + fh) ^ a_swz[i]) << 4); };
    auto frag_ptr_b = [&](const char* st, int ks, int j) { return st + b_off[j] + (((ks * 2 + fh) ^ b_swz[j]) << 4); };
    constexpr int KS = BK / 16;
    // SPREAD: the pieces go in front of the first SP sub-steps (the last sub-step of a 2-stage ring stays free: its data
    // would have too little time to land before the next barrier)
    constexpr int SP = (KS >= 4 && STAGES == 2) ? KS - 1 : KS;
    auto issue_piece = [&](auto ks_, int kt_next) {
        constexpr int ks = decltype(ks_)::value;
        if constexpr (ks < SP) {
            constexpr int J0 = PER_TILE * ks / SP, J1 = PER_TILE * (ks + 1) / SP;
            issue_dma(IntTag<J0>(), IntTag<J1>());        // (the tile was prepared at the end of the previous K step)
        }
    };
    // multiply K tile `buf`; before sub-step `issue_at` the wave issues its DMA share of tile `kt_next` (if any)
    auto compute = [&](int buf, int issue_at, int kt_next, bool more) {
        const char* st = smem + buf * STAGE_BYTES;
        if constexpr (FRAG_ASM) {
            u32x4 fa[2][MI], fb[2][NI];
            auto issue_frags = [&](int ks, int set) {
#pragma unroll
                for (int i = 0; i < MI; ++i) lds_read16_async(fa[set][i], frag_ptr_a(st, ks, i));
#pragma unroll
                for (int j = 0; j < NI; ++j) lds_read16_async(fb[set][j], frag_ptr_b(st, ks, j));
            };
            issue_frags(0, 0);
            auto substep = [&](auto ks_) __attribute__((always_inline)) {
                constexpr int ks = decltype(ks_)::value;
                constexpr int set = ks & 1;
                if constexpr (SPREAD) { if (more) issue_piece(ks_, kt_next); }
                else if (more && ks == issue_at) issue(kt_next, kt_next % STAGES);
                if constexpr (ks + 1 < KS) { issue_frags(ks + 1, set ^ 1); lds_wait<MI + NI>(fa[set][0]); }
                else lds_wait<0>(fa[set][0]);
#pragma unroll
                for (int i = 1; i < MI; ++i) lds_pin(fa[set][i]);
#pragma unroll
                for (int j = 0; j < NI; ++j) lds_pin(fb[set][j]);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j] = mfma_32x32x16(T(), fb[set][j], fa[set][i], acc[i][j]);
            };
            substep(IntTag<0>());
            substep(IntTag<1>());
            if constexpr (KS == 4) { substep(IntTag<2>()); substep(IntTag<3>()); }
        } else {
            auto piece = [&](auto ks_) __attribute__((always_inline)) { if constexpr (SPREAD) { if (more) issue_piece(ks_, kt_next); } };
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if constexpr (SPREAD) {
                    if (ks == 0) piece(IntTag<0>());
                    if (ks == 1) piece(IntTag<1>());
                    if (KS == 4 && ks == 2) piece(IntTag<2>());
                    if (KS == 4 && ks == 3) piece(IntTag<3>());
                } else if (more && ks == issue_at) issue(kt_next, kt_next % STAGES);
                u32x4 fa[MI], fb[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i) fa[i] = *reinterpret_cast<const u32x4*>(frag_ptr_a(st, ks, i));
#pragma unroll
                for (int j = 0; j < NI; ++j) fb[j] = *reinterpret_cast<const u32x4*>(frag_ptr_b(st, ks, j));
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j] = mfma_32x32x16(T(), fb[j], fa[i], acc[i][j]);
            }
        }
    };

    // -DAA_PHASE_PROBE build (scripts/phase_probe.py): wave 0 sums the cycles it spends waiting for DMA, at the
    // barrier, issuing DMA and in the multiply over the K loop -> workspace[65536 + bid][0..3]
#ifdef AA_PHASE_PROBE
    long long pt[4] = {0, 0, 0, 0};
    long long plast = clock_now();
#define AA_TICK(k) { const long long now_ = clock_now(); pt[k] += now_ - plast; plast = now_; }
#else
#define AA_TICK(k)
#endif
    stamp(1);
    {
#pragma unroll
        for (int t = 0; t < DIST; ++t)
            if (t < nk) issue(t, t);
        // SPREAD: the address arithmetic of the NEXT tile to issue (K position, per-row tap offsets) runs at the END of a K
        // step - behind the wave's last MFMAs, in time it would otherwise spend at the barrier - not at its start
        if constexpr (SPREAD) { if (DIST < nk) issue_prepare(DIST, DIST % STAGES); }
        // the tile's bias slice goes to LDS BEHIND the first operand tiles: its global load would otherwise stall the first
        // wave for a memory latency before it has issued its DMA share (the K-loop barriers publish the slice)
        if (tid < BN / 8) {
            const int n = tile_n * BN + tid * 8;
            u32x4 b = u32x4{0u, 0u, 0u, 0u};
            if (p.bias && !p.bias_per_row && n + 8 <= p.n_out) b = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.bias) + n);
            *reinterpret_cast<u32x4*>(sBias + tid * 8) = b;
        }
        for (int kt = 0; kt < nk; ++kt) {
            // tile kt must have landed; the (up to DIST-1) younger tiles may stay in flight
            const int younger = min(nk, kt + DIST) - (kt + 1);
            if (DIST >= 3 && younger == 2) dma_wait<2 * PER_TILE>();
            else if (DIST >= 2 && younger >= 1) dma_wait<(DIST >= 2 ? PER_TILE : 0)>();
            else dma_wait<0>();
            AA_TICK(0)
            block_barrier();                 // everyone's share of tile kt landed; buffer (kt-1)%STAGES is free
            AA_TICK(1)
            // STAGGER: the second wave of every SIMD (waves NW/2..) issues its DMA share in the MIDDLE of its multiply (the
            // data still has half a K step to land), its partner before its own: a SIMD then always has one wave feeding the matrix pipe while the other sits in the (CU-wide,
            // ~20 clk per instruction) LDS-DMA issue queue - instead of eight waves queueing there together.
            int issue_at = (STAGGER && wave >= NW / 2) ? KS / 2 : 0;
            if (STAGGER && (p.debug & 0x1000)) issue_at = (wave >= NW / 2) ? (p.debug >> 10) & 3 : (p.debug >> 8) & 3;   // probe: issue points
            const bool more = kt + DIST < nk && !(p.debug & 1);
            if (!(p.debug & 2)) compute(kt % STAGES, issue_at, kt + DIST, more);
            else if (more) issue(kt + DIST, (kt + DIST) % STAGES);
            if constexpr (SPREAD) { if (kt + 1 + DIST < nk) issue_prepare(kt + 1 + DIST, (kt + 1 + DIST) % STAGES); }
            AA_TICK(3)
        }
    }
#ifdef AA_PHASE_PROBE
    if ((p.debug & 8) && tid == 0)
        for (int k = 0; k < 4; ++k) reinterpret_cast<long long*>(p.workspace)[((int64_t)65536 + blockIdx.x) * 8 + k] = pt[k];
#endif
#undef AA_TICK

    stamp(2);
    const int ec = lane & 31, eh = lane >> 5;              // epilogue: lane owns output row ec, columns 16*eh .. +15 of a block
    const int m_tile = m_begin + tile_m * BM;
    const int n_wave = tile_n * BN + wn * (BN / WN);
    if (k_splits > 1) {
        // partial sums straight from the accumulator layout: ws[split][m][n] fp32, 64 contiguous bytes per lane and block
        // (rows are numbered from m_begin: a split-K tail launch keeps partials of its own rows only)
        float* ws = reinterpret_cast<float*>(p.workspace) + ((int64_t)blockIdx.y * (M - m_begin) - m_begin) * p.n_pad;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = m_tile + wm * (BM / WM) + i * 32 + ec;
            if (m >= M) continue;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                float* dst = ws + (int64_t)m * p.n_pad + n_wave + j * 32 + 16 * eh;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<f32x4*>(dst + 4 * q) = f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
            }
        }
        return;
    }

    cgd_epilogue<T, MI, NI>(p, M, acc, m_tile + wm * (BM / WM), n_wave, sBias + wn * (BN / WN));
    stamp(5);
}

}  // namespace aa

namespace aa {
// Split-K finish: sum the fp32 partials of `splits` K ranges and apply the usual epilogue (bias, row vector,
// activation, residual, scale); one 16-byte output chunk per thread.
template <typename T>
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const AaConvGemm p, const int M, const int m_begin, const int splits) {
    const int cpr = p.n_out >> 3;
    const int rows = M - m_begin;                         // the partials cover rows [m_begin, M)
    const int64_t total = (int64_t)rows * cpr;
    const float* ws = reinterpret_cast<const float*>(p.workspace);
    const T* bias = reinterpret_cast<const T*>(p.bias);
    const T* rowvec = reinterpret_cast<const T*>(p.rowvec);
    const T* resid = reinterpret_cast<const T*>(p.residual);
    T* out = reinterpret_cast<T*>(p.out);
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < total; c += (int64_t)gridDim.x * 256) {
        const int ml = (int)(c / cpr), n = (int)(c - (int64_t)ml * cpr) * 8;
        const int m = m_begin + ml;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.0f;
        for (int sp = 0; sp < splits; ++sp) {
            const f32x4* src = reinterpret_cast<const f32x4*>(ws + ((int64_t)sp * rows + ml) * p.n_pad + n);
            const f32x4 a = src[0], b = src[1];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += a[e]; v[4 + e] += b[e]; }
        }
        // uniform branches once per eight values (not once per value)
        if (p.ln_stats) {
            const LnRow lr = cgd_ln_row(p, m);
            const float* cs = p.ln_cols + n;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], lr.a, fmaf(lr.nm, cs[e], cs[p.n_pad + e]));
        }
        if (bias) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)bias[p.bias_per_row ? m : n + e];
        }
        if (rowvec) {
            const T* rv = rowvec + (int64_t)(m / p.rowvec_div) * (p.rowvec_ld ? p.rowvec_ld : p.n_out) + n;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
        }
        if (p.act == AA_ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
        }
        if (p.acc_scale != 0.0f) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= p.acc_scale;
        }
        if (resid) {                                                 // (fp32 until the single final rounding, like the fused epilogue)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)resid[(int64_t)m * p.ldr + n + e];
        }
        Pack8<T> o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.e[e] = (T)(v[e] * p.out_scale);
        *reinterpret_cast<u32x4*>(out + cgd_out_row(p, m) * p.ldo + n) = o.raw;
    }
}
}  // namespace aa
