// Implicit-GEMM convolution / linear kernel, LDS-DMA edition (fast path of aa_conv_gemm).
//
// A BM x BN output tile per workgroup, K step BK (64 or 32), WM x WN wavefronts each owning a
// (BM/WM) x (BN/WN) block of v_mfma_f32_32x32x16 accumulators, a ring of STAGES LDS buffers.
// Both operand tiles travel HBM/L2 -> LDS with `global_load_lds_dwordx4` (no VGPR round trip, no ds_write):
//  * every wave instruction deposits 64 lanes x 16 B, lane-linear = RPI tile rows of BK*2 bytes; the im2col
//    gather, the conv halo and the M / K tails are expressed in the per-lane SOURCE address (halo lanes read
//    a 16-byte zero page);
//  * bank conflicts of the ds_read_b128 fragment reads are removed by an XOR swizzle applied on the source
//    side (lane of row r, 16-byte position s fetches k-slot s ^ swz(r)) and undone on the read side;
//  * requires the K tile to sit inside one filter tap and one concat source ((c0+c1) % 64 == 0 and
//    c0 % 64 == 0): tap / source / channel base are then wave-uniform scalars and the per-row pixel offsets
//    are recomputed only when the tap changes;
//  * pipeline: STAGES-1 tiles are in flight ahead of the one being multiplied.  The measured limiter of the
//    2-stage version is the DMA round trip under load (~1.5-2.5 us against a ~1.2 us MFMA phase), so the
//    deeper rings wait with a COUNTED `s_waitcnt vmcnt(n)` (only the oldest tile must have landed) and a raw
//    `s_barrier` (a __syncthreads() would drain every DMA); one barrier per K step;
//  * epilogue: bias / time-embedding row vector / SiLU in registers, tile parked in LDS as storage dtype
//    (one accumulator block-row of every wave at a time), read back row-major so GEGLU pairing (value | gate
//    halves of the tile), residual loads and output stores are full 16-byte, row-contiguous.
#pragma once
#include "dev.h"
#include "aa_mi355.h"
#include "conv_gemm.h"

namespace aa {

__host__ __device__ inline int cgd_lds_bytes(int bm, int bn, int bk, int stages) {
    const int operands = stages * (bm + bn) * bk * 2 + 1024;     // + dummy DMA landing zone
    const int staging = 128 * (bn + 8) * 2;                      // >= WM*32 rows of the epilogue staging tile
    return operands > staging ? operands : staging;
}

// PER_CU = workgroups meant to be co-resident on a CU (register budget: 512 / (PER_CU * waves per SIMD)).
template <typename T, int BM, int BN, int WM, int WN, int BK, int STAGES, int PER_CU, bool STAGGER>
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
    constexpr bool FRAG_ASM = (MI * NI * 16 + 2 * (MI + NI) * 4) <= 184;   // accumulators + two fragment sets fit beside the rest
    static_assert(BM % (32 * WM) == 0 && BN % (32 * WN) == 0 && BM % RPI == 0 && BN % RPI == 0, "tile shape");
    static_assert(BK == 64 || BK == 32, "K step");
    static_assert(STAGES >= 2 && STAGES <= 4 && (STAGES - 2) * PER_TILE <= 63, "pipeline depth");
    char* smem = dyn_smem();
    char* dummy = smem + STAGES * STAGE_BYTES;           // where surplus (guarded-out) DMA instructions land

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
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
    const int lrow = lane / SPR, lpos = lane % SPR;
    int row_img[AJ], row_iy[AJ], row_ix[AJ], kslot[AJ];
    bool row_ok[AJ];
    int aoff[AJ];                        // element offset of this lane's 16-byte piece inside the source, -1 = zero page
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int rr = (wave + NW * j) * RPI + lrow;
        const int m = m_begin + tile_m * BM + rr;
        row_ok[j] = m < M && rr < BM;
        const int mm = row_ok[j] ? m : 0;
        if (linear) {                    // 1x1 / nn.Linear: the row IS the pixel, no (img, y, x) decomposition
            row_img[j] = 0; row_iy[j] = 0; row_ix[j] = mm;
        } else {
            const int x = mm % p.w_out;
            const int t = mm / p.w_out;
            const int y = t % p.h_out;
            row_img[j] = t / p.h_out;
            row_iy[j] = y * p.stride - p.pad_h;
            row_ix[j] = x * p.stride - p.pad_w;
        }
        kslot[j] = lpos ^ ((rr / RPB) % SPR);
        aoff[j] = -1;
    }
    // weight panel of this tile: 32-bit element offsets (a packed panel is far below 2^31 elements)
    const T* wtile = reinterpret_cast<const T*>(p.w) + (int64_t)tile_n * BN * p.k_pad;
    int woff[BJ];
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int rr = (wave + NW * j) * RPI + lrow;
        woff[j] = rr * p.k_pad + (lpos ^ ((rr / RPB) % SPR)) * 8;
    }
    const T* zero = reinterpret_cast<const T*>(zero_page());

    int cur_tap = -1;
    int pixel[AJ];                       // source pixel of each fed row for the current tap (-1 = halo / tail)
    auto issue = [&](int kt, int buf) {
        // K order (wave-uniform scalars): tap-major (tap, channel) or, for multi-tap filters packed
        // chunk-major, (64-channel chunk, tap, channel) - consecutive K steps then re-read the same
        // activation slab shifted by one tap, which keeps it L2-resident across the 9 taps.
        const int k0 = (kbase + kt) * BK;
        int tap, cb;
        if (p.k_order) { const int taps = p.kh * p.kw; const int unit = k0 >> 6; const int chunk = unit / taps;
                         tap = unit - chunk * taps; cb = chunk * 64 + (k0 & 63); }
        else           { tap = k0 / ctot; cb = k0 - tap * ctot; }
        if (tap != cur_tap) {
            cur_tap = tap;
            const int dy = tap / p.kw, dx = tap - dy * p.kw;
            const bool tap_ok = tap < p.kh * p.kw;
#pragma unroll
            for (int j = 0; j < AJ; ++j) {
                const int iy = row_iy[j] + dy, ix = row_ix[j] + dx;
                if (linear) { pixel[j] = (tap_ok && row_ok[j]) ? ix : -1; continue; }
                const bool ok = tap_ok && row_ok[j] && (unsigned)iy < (unsigned)p.h_virt && (unsigned)ix < (unsigned)p.w_virt;
                int sy = iy, sx = ix;
                if (resize) { sy = (iy * p.h_in) / p.h_virt; sx = (ix * p.w_in) / p.w_virt; }
                pixel[j] = ok ? (row_img[j] * p.h_in + sy) * p.w_in + sx : -1;
            }
        }
        const T* src; int cs, cc;
        if (cb < p.c0) { src = reinterpret_cast<const T*>(p.a0); cs = p.c0; cc = cb; }
        else           { src = reinterpret_cast<const T*>(p.a1); cs = p.c1; cc = cb - p.c0; }
#pragma unroll
        for (int j = 0; j < AJ; ++j) aoff[j] = pixel[j] >= 0 ? pixel[j] * cs + cc + kslot[j] * 8 : -1;
        char* a = smem + buf * STAGE_BYTES + wave * RPI * ROWB;
        char* b = smem + buf * STAGE_BYTES + BM * ROWB + wave * RPI * ROWB;
        // every wave issues exactly PER_TILE instructions (vmcnt bookkeeping): groups past the tile edge are
        // pointed at the zero page / the dummy landing zone (wave-uniform choice)
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const bool real = (GA % NW == 0) || (wave + NW * j < GA);
            const T* g = (real && aoff[j] >= 0) ? src + aoff[j] : zero;
            async_copy16(g, real ? a + j * NW * RPI * ROWB : dummy);
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const bool real = (GB % NW == 0) || (wave + NW * j < GB);
            async_copy16(real ? wtile + woff[j] + (kbase + kt) * BK : zero, real ? b + j * NW * RPI * ROWB : dummy);
        }
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
    for (int i = 0; i < MI; ++i) { const int rr = wm * (BM / WM) + i * 32 + frow; a_off[i] = rr * ROWB; a_swz[i] = (rr / RPB) % SPR; }
#pragma unroll
    for (int j = 0; j < NI; ++j) { const int rr = wn * (BN / WN) + j * 32 + frow; b_off[j] = BM * ROWB + rr * ROWB; b_swz[j] = (rr / RPB) % SPR; }

    // Fragment reads.  FRAG_ASM: two register sets, the ds_reads of sub-step ks+1 are hand-issued before the
    // MFMAs of sub-step ks and waited with a counted lgkmcnt (LDS returns in order: "at most MI+NI outstanding"
    // == the older set has landed), so LDS latency hides under the matrix pipe.  Otherwise plain C++ loads
    // (hipcc keeps ONE set and waits lgkmcnt(0) each sub-step) - for shapes whose accumulators leave no room.
    auto frag_ptr_a = [&](const char* st, int ks, int i) { return st + a_off[i] + (((ks * 2 + fh) ^ a_swz[i]) << 4); };
    auto frag_ptr_b = [&](const char* st, int ks, int j) { return st + b_off[j] + (((ks * 2 + fh) ^ b_swz[j]) << 4); };
    auto compute = [&](int buf) {
        const char* st = smem + buf * STAGE_BYTES;
        constexpr int KS = BK / 16;
        if constexpr (FRAG_ASM) {
            u32x4 fa[2][MI], fb[2][NI];
            auto issue_frags = [&](int ks, int set) {
#pragma unroll
                for (int i = 0; i < MI; ++i) lds_read16_async(fa[set][i], frag_ptr_a(st, ks, i));
#pragma unroll
                for (int j = 0; j < NI; ++j) lds_read16_async(fb[set][j], frag_ptr_b(st, ks, j));
            };
            issue_frags(0, 0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int set = ks & 1;
                if (ks + 1 < KS) { issue_frags(ks + 1, set ^ 1); lds_wait<MI + NI>(fa[set][0]); }
                else lds_wait<0>(fa[set][0]);
#pragma unroll
                for (int i = 1; i < MI; ++i) lds_pin(fa[set][i]);
#pragma unroll
                for (int j = 0; j < NI; ++j) lds_pin(fb[set][j]);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j] = mfma_32x32x16(T(), fa[set][i], fb[set][j], acc[i][j]);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                u32x4 fa[MI], fb[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i) fa[i] = *reinterpret_cast<const u32x4*>(frag_ptr_a(st, ks, i));
#pragma unroll
                for (int j = 0; j < NI; ++j) fb[j] = *reinterpret_cast<const u32x4*>(frag_ptr_b(st, ks, j));
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j] = mfma_32x32x16(T(), fa[i], fb[j], acc[i][j]);
            }
        }
    };

    stamp(1);
    if constexpr (STAGGER) {
        // Two wave groups (first / second wave of every SIMD) run half a K step apart: while one group
        // multiplies tile kt the other only issues / waits for DMA, so the matrix pipe never sees both waves
        // parked at the same barrier.  Two barriers per K step, two LDS buffers:
        //   interval 2k  : both issue their share of tile kt+1 -> buffer (kt+1)&1 (its last reader, the late
        //                  group's multiply of tile kt-1, finished before the previous barrier); EARLY multiplies kt
        //   interval 2k+1: LATE multiplies tile kt; both drain their own DMA share before the closing barrier.
        static_assert(!STAGGER || (STAGES == 2 && NW == 8), "stagger needs 8 waves and a 2-buffer ring");
        const bool late = wave >= NW / 2;
        issue(0, 0);
        dma_wait<0>();
        block_barrier();
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
            if (!late) compute(kt & 1);
            block_barrier();
            if (late) compute(kt & 1);
            dma_wait<0>();
            block_barrier();
        }
    } else {
#pragma unroll
        for (int t = 0; t < DIST; ++t)
            if (t < nk) issue(t, t);
        for (int kt = 0; kt < nk; ++kt) {
            // tile kt must have landed; the (up to DIST-1) younger tiles may stay in flight
            const int younger = min(nk, kt + DIST) - (kt + 1);
            if (DIST >= 3 && younger == 2) dma_wait<2 * PER_TILE>();
            else if (DIST >= 2 && younger >= 1) dma_wait<(DIST >= 2 ? PER_TILE : 0)>();
            else dma_wait<0>();
            block_barrier();                 // everyone's share of tile kt landed; buffer (kt-1)%STAGES is free
            if (kt + DIST < nk && !(p.debug & 1)) issue(kt + DIST, (kt + DIST) % STAGES);
            if (!(p.debug & 2)) compute(kt % STAGES);
        }
    }

    stamp(2);
    if (k_splits > 1) {
        // partial sums straight from the accumulator layout (col = lane&31, 4-row groups): ws[split][m][n] fp32
        float* ws = reinterpret_cast<float*>(p.workspace) + (int64_t)blockIdx.y * M * p.n_pad;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int n = tile_n * BN + wn * (BN / WN) + j * 32 + (lane & 31);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m_begin + tile_m * BM + wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    if (m < M) ws[(int64_t)m * p.n_pad + n] = acc[i][j][e];
                }
            }
        return;
    }

    // ---- epilogue: pass i stages accumulator block-row i of EVERY wave (WM*32 tile rows) through an LDS
    // tile [WM*32][BN+8] of storage dtype; acc[i] is dead after pass i, so register pressure only falls ----
    constexpr int LDE = BN + 8;
    constexpr int PROWS = WM * 32;                       // tile rows staged per pass
    T* sE = reinterpret_cast<T*>(smem);
    const T* bias = reinterpret_cast<const T*>(p.bias);
    const T* rowvec = reinterpret_cast<const T*>(p.rowvec);
    const T* resid = reinterpret_cast<const T*>(p.residual);
    T* out = reinterpret_cast<T*>(p.out);
    const int col_l = lane & 31, row_l = 4 * (lane >> 5);
    const int m_tile = m_begin + tile_m * BM;
    const int g0 = m_tile / p.rowvec_div;
    const int g_edge = (g0 + 1) * p.rowvec_div;
    // rows of one tile fall into at most two row-vector groups when rowvec_div >= BM (always true on the
    // UNet path: rowvec_div = frames*H*W); otherwise divide per element.
    const bool two_groups = p.rowvec_div >= BM;
    const int n_cols = p.geglu ? (p.n_out >> 1) : p.n_out;

    // per-column terms are the same in every pass: hoist them
    float bcol[NI], rv0[NI], rv1[NI];
    int nbj[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int n = min(tile_n * BN + wn * (BN / WN) + j * 32 + col_l, p.n_pad - 1);
        nbj[j] = p.geglu ? n : min(n, p.n_out - 1);       // GEGLU bias is packed like the weights
        bcol[j] = (bias && !p.bias_per_row) ? (float)bias[nbj[j]] : 0.0f;
        rv0[j] = 0.0f; rv1[j] = 0.0f;
        if (rowvec && two_groups) {
            const int gmax = (M - 1) / p.rowvec_div;
            rv0[j] = (float)rowvec[(int64_t)min(g0, gmax) * p.n_out + nbj[j]];
            rv1[j] = (float)rowvec[(int64_t)min(g0 + 1, gmax) * p.n_out + nbj[j]];
        }
    }

    // row-major read-back of one staged pass, CPR = 16-byte chunks per output row (compile-time: no divides)
    auto read_back = [&](int ps, auto cpr_tag) {
        constexpr int CPR = decltype(cpr_tag)::value;
        const int col0 = tile_n * CPR * 8;
        for (int c = tid; c < PROWS * CPR; c += THREADS) {
            const int rl = c / CPR, ch = c - rl * CPR;
            const int m = m_tile + (rl >> 5) * (BM / WM) + ps * 32 + (rl & 31), n = col0 + ch * 8;
            if (m >= M || n >= n_cols) continue;
            Pack8<T> v; v.raw = *reinterpret_cast<const u32x4*>(sE + rl * LDE + ch * 8);
            if (CPR * 8 != BN) {                          // GEGLU: value half | gate half of the tile
                Pack8<T> gt; gt.raw = *reinterpret_cast<const u32x4*>(sE + rl * LDE + CPR * 8 + ch * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) v.e[e] = (T)((float)v.e[e] * gelu_erf_f((float)gt.e[e]));
            }
            if (resid || p.out_scale != 1.0f) {
                Pack8<T> rs; rs.raw = u32x4{0u, 0u, 0u, 0u};
                if (resid) rs.raw = *reinterpret_cast<const u32x4*>(resid + (int64_t)m * p.ldr + n);
#pragma unroll
                for (int e = 0; e < 8; ++e) v.e[e] = (T)(((float)v.e[e] + (float)rs.e[e]) * p.out_scale);
            }
            *reinterpret_cast<u32x4*>(out + (int64_t)m * p.ldo + n) = v.raw;
        }
    };

#pragma unroll
    for (int ps = 0; ps < MI; ++ps) {
        __syncthreads();                                  // previous users of the LDS region are done
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int nl = wn * (BN / WN) + j * 32 + col_l;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rl = wm * 32 + (e & 3) + 8 * (e >> 2) + row_l;               // staged row
                const int m = min(m_tile + wm * (BM / WM) + ps * 32 + (e & 3) + 8 * (e >> 2) + row_l, M - 1);
                float v = acc[ps][j][e] + bcol[j];
                if (p.bias_per_row && bias) v += (float)bias[m];
                if (rowvec) v += two_groups ? (m < g_edge ? rv0[j] : rv1[j]) : (float)rowvec[(int64_t)(m / p.rowvec_div) * p.n_out + nbj[j]];
                if (p.act == AA_ACT_SILU) v = silu_f(v);
                sE[rl * LDE + nl] = (T)v;
            }
        }
        __syncthreads();
        if (ps == 0) stamp(3);
        if (p.geglu) read_back(ps, IntTag<BN / 16>());
        else read_back(ps, IntTag<BN / 8>());
        if (ps == 0) stamp(4);
    }
    stamp(5);
}

}  // namespace aa

namespace aa {
// Split-K finish: sum the fp32 partials of `splits` K ranges and apply the usual epilogue (bias, row vector,
// activation, residual, scale); one 16-byte output chunk per thread.
template <typename T>
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const AaConvGemm p, const int M, const int splits) {
    const int cpr = p.n_out >> 3;
    const int64_t total = (int64_t)M * cpr;
    const float* ws = reinterpret_cast<const float*>(p.workspace);
    const T* bias = reinterpret_cast<const T*>(p.bias);
    const T* rowvec = reinterpret_cast<const T*>(p.rowvec);
    const T* resid = reinterpret_cast<const T*>(p.residual);
    T* out = reinterpret_cast<T*>(p.out);
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < total; c += (int64_t)gridDim.x * 256) {
        const int m = (int)(c / cpr), n = (int)(c - (int64_t)m * cpr) * 8;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.0f;
        for (int sp = 0; sp < splits; ++sp) {
            const f32x4* src = reinterpret_cast<const f32x4*>(ws + ((int64_t)sp * M + m) * p.n_pad + n);
            const f32x4 a = src[0], b = src[1];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += a[e]; v[4 + e] += b[e]; }
        }
        Pack8<T> o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float x = v[e];
            if (bias) x += (float)bias[p.bias_per_row ? m : n + e];
            if (rowvec) x += (float)rowvec[(int64_t)(m / p.rowvec_div) * p.n_out + n + e];
            if (p.act == AA_ACT_SILU) x = silu_f(x);
            x = (float)(T)x;                                       // same rounding point as the fused epilogue
            if (resid) x += (float)resid[(int64_t)m * p.ldr + n + e];
            o.e[e] = (T)(x * p.out_scale);
        }
        *reinterpret_cast<u32x4*>(out + (int64_t)m * p.ldo + n) = o.raw;
    }
}
}  // namespace aa
