// Implicit-GEMM convolution / linear kernel with a HAND-PLACED instruction stream ("x" tiles of the tile table).
//
// Same operand path as conv_gemm_dma.h (buffer-descriptor LDS-DMA, source-side XOR swizzle, transposed
// v_mfma_f32_32x32x16 whose accumulator registers are 16 consecutive output columns, register-direct epilogue),
// but the K loop is not left to the compiler's scheduler:
//  * the accumulators live in the accumulation half of the unified register file under LITERAL names
//    (a[16 b : 16 b + 15] for 32x32 block b; blocks 16.. of the 320-column tile in compiler-allocated VGPRs):
//    a 4-wave workgroup (one wave per SIMD, 512 registers per lane) owns a 256 x 256 or 256 x 320 tile with
//    128 x 128 / 128 x 160 outputs per wave = 0.5 / 0.45 fragment reads per MFMA instead of 0.75 / 0.7, and the
//    compiler's register allocator never sees them (no spills, no copies);
//  * every MFMA, fragment read, DMA piece, wait and barrier is one `asm volatile` statement, so their program
//    order IS the issue order: the gap behind each MFMA carries at most one fragment read of the next sub-step
//    or one LDS-DMA piece of the tile after next (the CU's texture-address unit takes one 1-KiB piece per 16 clk:
//    pieces issued in a burst hold every wave in its queue with the matrix pipe idle; one piece per 32-clk MFMA
//    slot never queues);
//  * the barrier of a K step sits BEFORE its last sub-step: once the fragments of sub-step 3 are in registers the
//    stage is free, so the DMA of the tile after next and the first fragment reads of the next tile are issued
//    under the 16 - 20 MFMAs of sub-step 3 and the matrix pipe does not drain at the step boundary.
// Per K step and wave (256 x 256): 64 MFMA, 32 ds_read_b128, 16 DMA pieces, one barrier.
#pragma once
#include "dev.h"
#include "aa_mi355.h"
#include "conv_gemm_dma.h"

namespace aa {

// the tiles that can start their accumulators from a folded LayerNorm (one wave per SIMD, K step 64) keep a private copy of the two
// column vectors of each wave's columns behind the bias slice: [wave][2][256] fp32 (an LDS-DMA piece deposits 1 KiB)
__host__ __device__ constexpr bool cgx_ln_ok(int bk, int wm, int wn, int per_cu) { return wm * wn * per_cu <= 4 && bk == 64; }
// + 4 KiB behind everything else: where the waves of a row-spanning tile exchange their row statistics (AaConvGemm.row_coef, version 107:
// [waves][rows of a wave][2] fp32 - 4 x 128 or 8 x 64 rows)
constexpr int CGX_COEF_BYTES = 4096;
__host__ __device__ inline int cgx_lds_main_bytes(int bm, int bn, int bk, int ring, int wm, int wn, int per_cu) {
    return cgd_lds_bytes(bm, bn, bk, ring) + (cgx_ln_ok(bk, wm, wn, per_cu) ? wm * wn * 2048 : 0);
}
__host__ __device__ inline int cgx_lds_bytes(int bm, int bn, int bk, int ring, int wm, int wn, int per_cu) {
    return cgx_lds_main_bytes(bm, bn, bk, ring, wm, wn, per_cu) + CGX_COEF_BYTES;
}

// Tiles whose K-split launches can finish inside the kernel (AaConvGemm.tickets): the ones the small-M levels split along K - 128 x 128
// (two workgroups per CU, 64 accumulation registers) and 192 x 256.  The finish is a second instance of every epilogue form: it is
// compiled into these three tiles only (in all of them it doubled the library and added two minutes to the build, for a path that
// measured slower than the reduce launch), and never into tiles with 128 literal accumulation registers next to 128 VGPRs, where the
// extra pressure makes hipcc park values in accumulation registers it believes free (tests/test_abi.py counts v_accvgpr_write).
__host__ __device__ constexpr bool cgx_ticket_ok(int bm, int bn, int wm, int wn, int per_cu) {
    return (bm / wm / 32) * (bn / wn / 32) <= 4 || (bm == 192 && bn == 256 && (wm * wn * per_cu + 3) / 4 == 1);
}

// DP3 / DP0 / DP1: LDS-DMA pieces (per wave) of the tile after next issued under sub-step 3 of a K step and under
// sub-steps 0 / 1 of the following one (the rest under sub-step 2); activations first (they may come from HBM).
//
// BK = 32 ("deep ring"): four stages of 32 K each.  Stage s is multiplied in two sub-steps; once the fragments of its second
// sub-step are in registers its LDS slot is free, so with one barrier per stage the slots hold stages s+1 .. s+4: THREE
// stages (96 KB at 256 x 256) are in flight while one is multiplied, every piece has two stage times to land, and the wait
// in front of a barrier is a counted vmcnt (the two youngest stages stay in flight).  The 2-stage BK = 64 ring has one tile
// in flight for about half of each K step: its operand stream is latency-bound (measured: the DMA stream alone takes as
// long as the MFMA stream alone, r03 ablation).  DP3 = pieces of stage s+4 issued under the second sub-step of stage s
// (the rest go out under the first sub-step of stage s+1).
// RING = slots of the deep ring (4, or 3 for the tiles that run two workgroups per CU: 2 x 3 x 24 KB at 128 x 256);
// (Also measured and not kept, r03i: per-wave SKEWED DMA gaps - wave w issues behind MFMAs w, w+4, ... so that the CU's address unit
// never sees two pieces at once: 0 ... -4 %.)
// (A persistent variant - one workgroup per CU slot walking the tiles - was measured too: +-2 % on every shape, r03h; dispatching
// 5440 empty workgroups costs 6.5 us, scripts/probe/dispatch_cost.hip.  Not kept.)
// PER_CU = workgroups meant to be co-resident on a CU (two 4-wave workgroups drift out of phase: one multiplies while the
// other runs its epilogue - the VALU-heavy GEGLU epilogue is as long as a K = 320 loop).
// LIN: the call is a plain linear layer / 1x1 convolution (one tap, stride 1, no padding): the per-row pixel arithmetic (two
// integer divisions per fed row), the tap masks and the tap walk of the K position are compiled out of the setup.
template <typename T, int BM, int BN, int WM, int WN, int BK, int DP3, int DP0, int DP1, int RING = (BK == 64 ? 2 : 4), int PER_CU = 1, bool LIN = false>
__global__ void __launch_bounds__(64 * WM * WN, (PER_CU * WM * WN + 3) / 4) conv_gemm_x_kernel(const AaConvGemm p, const int M, const int tiles_n, const int m_begin, const int k_splits) {
    constexpr int STAGES = RING;
    static_assert(BK == 64 ? RING == 2 : (RING == 3 || RING == 4), "ring depth");
    constexpr int NW = WM * WN;
    constexpr int MI = BM / WM / 32;     // 32-row accumulator blocks per wave
    constexpr int NI = BN / WN / 32;     // 32-column accumulator blocks per wave
    constexpr int NB = MI * NI;          // accumulator blocks per wave: block i * NI + j
    constexpr int ROWB = BK * 2;         // bytes per LDS tile row
    constexpr int SPR = BK / 8;          // 16-byte slots per row
    constexpr int RPI = 64 / SPR;        // tile rows deposited by one wave DMA instruction
    constexpr int RPB = 256 / ROWB;      // tile rows per 256-byte LDS bank row
    constexpr int GA = BM / RPI, GB = BN / RPI;
    constexpr int AJ = (GA + NW - 1) / NW;
    constexpr int BJ = (GB + NW - 1) / NW;
    constexpr int PER_TILE = AJ + BJ;
    constexpr int DP2 = PER_TILE - DP3 - DP0 - DP1;
    constexpr int STAGE_BYTES = (BM + BN) * ROWB;
    constexpr int R = MI + NI;           // fragment reads per sub-step
    constexpr int KS = BK / 16;
    static_assert(BM % (32 * WM) == 0 && BN % (32 * WN) == 0 && BM % RPI == 0 && BN % RPI == 0, "tile shape");
    static_assert(NB <= ACC_BLOCKS && R <= NB, "accumulator file");
    static_assert(BK == 64 || BK == 32, "K step");
    static_assert(DP2 >= 0 && DP3 >= 0 && DP0 >= 0 && DP1 >= 0 && (BK == 64 || (DP0 == 0 && DP1 == 0)), "DMA schedule");
    static_assert(BN * 2 <= 1024, "bias slice");
    static_assert(AJ <= NB, "one activation row offset per MFMA gap of a sub-step");
    char* smem = dyn_smem();
    char* dummy = smem + STAGES * STAGE_BYTES;
    T* sBias = reinterpret_cast<T*>(dummy + 1024);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = wave_id();
    const int wm = wave / WN, wn = wave % WN;

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    // An XCD walks a contiguous range of `logical`.  Row-major order made its 32 resident tiles 32 COLUMN tiles of one row tile: the row tile's
    // activations were shared, but 32 different weight tiles streamed through its 4-MB L2 per round (the 16x16-level GEGLU, 34 x 40 tiles of
    // 256 x 256, K = 1280: 1040 MB of fabric traffic per call against 138 MB of operands, profiles/r06l_traffic_by_shape.json).  Grouped order
    // (panels of `gm` row tiles, column by column inside a panel): the resident tiles form a gm x (32 / gm) block, the gm activation tiles of the
    // panel stay in L2 while it sweeps the columns, and every weight tile is fetched once per panel and shared by gm workgroups.  gm from the
    // bytes of a row tile's activations (<= ~2.5 MB per panel, <= 8), and only for WIDE outputs (>= 8 column tiles): with a few column tiles the whole
    // weight matrix sits in L2 and row-major already streams the activations once - grouping measured +10-15 % fabric traffic there
    // (profiles/r06m_traffic_by_shape_grouped_everywhere.json).  AaConvGemm.debug bit 4 (16): row-major everywhere (A/B).
    int tile_m, tile_n;
    {
        const int tiles_m = nwg / tiles_n;
        const int a_tile_bytes = BM * (p.c0 + p.c1) * 2;
        int gm = ((p.debug & 16) || tiles_n < 8) ? 1 : min(8, max(1, (5 << 19) / a_tile_bytes));
        gm = min(gm, tiles_m);
        const int panel = gm * tiles_n, pidx = logical / panel;
        const int rows_here = min(gm, tiles_m - pidx * gm);
        const int within = logical - pidx * panel;
        tile_n = within / rows_here;
        tile_m = pidx * gm + (within - tile_n * rows_here);
    }

    // debug bit 8 (scripts/phase_probe_x.py): thread 0 leaves stamps in workspace[bid][8]: shader clock at entry (0), before the
    // first DMA piece (1), with the first stage landed (6), behind the K loop (2), at exit (5); 100 MHz wall clock at entry (7) and
    // exit (4); which CU ran it (3)
    auto stamp = [&](int slot) __attribute__((always_inline)) {
        if ((p.debug & 8) && tid == 0) reinterpret_cast<long long*>(p.workspace)[(int64_t)blockIdx.x * 8 + slot] = clock_now();
    };
    auto stamp_wall = [&](int slot, int id_slot) __attribute__((always_inline)) {
        if ((p.debug & 8) && tid == 0) {
            long long* w = reinterpret_cast<long long*>(p.workspace) + (int64_t)blockIdx.x * 8;
            w[slot] = wall_now();
            if (id_slot >= 0) w[id_slot] = hw_id();
        }
    };
    stamp(0);
    stamp_wall(7, 3);
    const int ctot = p.c0 + p.c1;
    const int nk_all = p.k_pad / BK;
    const int k_per = (nk_all + k_splits - 1) / k_splits;
    const int kbase = blockIdx.y * k_per;
    const int nk = max(0, min(k_per, nk_all - kbase));
    // (convolutions behind a nearest-neighbour resize - Upsample2D - stay with conv_gemm_dma.h: aa_conv_gemm_tile_ok)
    const bool linear = LIN || (p.kh * p.kw == 1 && p.stride == 1 && p.pad_h == 0 && p.pad_w == 0);
    const int kh_ = LIN ? 1 : p.kh, kw_ = LIN ? 1 : p.kw;

    // ---- DMA geometry (as conv_gemm_dma.h): this lane feeds LDS rows ((wave + NW*j)*RPI + lane/SPR), 16-byte position lane%SPR
    constexpr unsigned OOB = 0x80000000u;
    // (the activation descriptors start `margin` pixels in front of the tensors: see pixi[] below; lanes that would read there
    // carry an out-of-range offset)
    const int margin_px = linear ? 0 : p.pad_h * p.w_in + p.pad_w;
    const BufRsrc r_a0 = make_rsrc(static_cast<const char*>(p.a0) - (int64_t)margin_px * p.c0 * 2, (unsigned)(((int64_t)p.n_img * p.h_in * p.w_in + margin_px) * p.c0 * 2));
    const BufRsrc r_a1 = make_rsrc(p.c1 ? static_cast<const char*>(p.a1) - (int64_t)margin_px * p.c1 * 2 : nullptr, p.c1 ? (unsigned)(((int64_t)p.n_img * p.h_in * p.w_in + margin_px) * p.c1 * 2) : 0u);
    const BufRsrc r_w = make_rsrc(p.w, (unsigned)((int64_t)p.n_pad * p.k_pad * 2));
    constexpr int swm = SPR - 1;
    const bool two_src = p.c1 != 0;
    const int lrow = lane / SPR, lpos = lane % SPR;
    // The lane's swizzled 16-byte k-slot is the same for every piece (RPI * NW rows apart).
    // Per fed activation row: the pixel under tap (0, 0), counted from `margin` pixels in front of the tensor (>= 0 for every
    // row), and one bit per tap that does NOT read a real pixel (halo, rows behind the tile; bits >= taps are set).  The source
    // offset of a piece is then three VALU instructions per row and tap, with no branch (im2col_offset): they ride in MFMA gaps.
    // (Round 3 recomputed the offsets in a cluster of ~75 compiler-scheduled, exec-masked instructions behind every barrier -
    // with the tap changing every K step (k_order 1) that cluster drained the matrix pipe for ~15 % of a K step.)
    unsigned pixi[AJ], inv[AJ];
    struct RowCoords { int img, iy, ix; bool ok; };
    auto row_coords = [&](int j) __attribute__((always_inline)) {
        const int rr = (wave + NW * j) * RPI + lrow;
        const int m = m_begin + tile_m * BM + rr;
        const bool ok = m < M && rr < BM;
        const int mm = ok ? m : 0;
        if (linear) return RowCoords{0, 0, mm, ok};
        const int x = mm % p.w_out;
        const int t = mm / p.w_out;
        const int y = t % p.h_out;
        return RowCoords{t / p.h_out, y * p.stride - p.pad_h, x * p.stride - p.pad_w, ok};
    };
    static_assert((RPI * NW / RPB) % SPR == 0, "the swizzle term must not depend on the piece index");
    const unsigned slot16 = (unsigned)((lpos ^ (((wave * RPI + lrow) / RPB) & swm)) * 16);
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const RowCoords rc = row_coords(j);
        const int iy = rc.iy, ix = rc.ix;
        pixi[j] = (unsigned)((rc.img * p.h_in + iy) * p.w_in + ix + margin_px);
        const int ylo = max(0, -iy), yhi = max(ylo, min(kh_, p.h_virt - iy));
        const int xlo = max(0, -ix), xhi = max(xlo, min(kw_, p.w_virt - ix));
        const unsigned mx = ((1u << xhi) - 1u) ^ ((1u << xlo) - 1u);
        unsigned valid = 0u;
        for (int dy = ylo; dy < yhi; ++dy) valid |= mx << (dy * kw_);
        inv[j] = rc.ok ? ~valid : ~0u;
    }
    // weight panel: this lane's row of piece 0; piece j is NW * RPI rows further (a wave-uniform offset)
    const unsigned wb0 = (unsigned)((tile_n * BN + wave * RPI + lrow) * p.k_pad) * 2u + slot16;
    const unsigned wb_step = (unsigned)(NW * RPI * p.k_pad) * 2u;
    const int taps = kh_ * kw_;

    // K position of the next prepare() call (wave-uniform scalars, advanced incrementally): filter tap, its column, the tap's pixel
    // offset from tap (0, 0), channel base.  The packed K axis is walked as (64-channel chunk, tap, channel) - AaConvGemm.k_order 1,
    // which for one tap is k_order 0 as well (the host offers these tiles to nothing else: cg_x_ok).
    int n_tap, n_dx, n_toff, n_cb;
    {
        const int k0 = kbase * BK;
        const int unit = k0 >> 6, chunk = unit / taps;
        n_tap = unit - chunk * taps;
        n_cb = chunk * 64 + (k0 & 63);
        const int dy = n_tap / kw_;
        n_dx = n_tap - dy * kw_;
        n_toff = dy * p.w_in + n_dx;
    }
    const int wrap_d = p.w_in - kw_ + 1;                    // pixel offset from the last tap of a filter row to the first of the next
    int cur_src = -1;
    unsigned pb[AJ];                     // byte offset of each fed row's source pixel for the prepared (tap, source) (bit 31 = halo / tail)
    bool is_src1 = false;
    unsigned is_ccb = 0u, is_kb = 0u;
    int is_buf = 0;
    unsigned s_c2 = 0u, s_sh = 0u;       // prepared (tap, source): bytes per pixel of the source; 31 - tap
    unsigned v_t = 0u;                   // slot16 + byte offset of the tap's pixel from tap (0, 0)
    auto prep_row = [&](auto j_) __attribute__((always_inline)) {
        constexpr int j = decltype(j_)::value;
        im2col_offset(pb[j], pixi[j], s_c2, v_t, s_sh, inv[j]);
    };
    // wave-uniform part of prepare(): advances the K position, leaves what the DMA pieces of tile kt need besides pb[]
    auto prep_scalars = [&](int kt, int buf) __attribute__((always_inline)) {
        const int tap = n_tap, cb = n_cb, toff = n_toff;
        {
            const bool last_tap = n_tap + 1 == taps;
            const bool wrap_x = n_dx + 1 == kw_;
            const int t1 = last_tap ? 0 : n_tap + 1;
            const int o1 = last_tap ? 0 : n_toff + (wrap_x ? wrap_d : 1);
            const int x1 = (last_tap || wrap_x) ? 0 : n_dx + 1;
            if constexpr (BK == 32) {
                const bool stay = !(n_cb & 32);                          // first half of a 64-channel unit: same tap
                n_cb = stay ? n_cb + 32 : (n_cb & ~63) + (last_tap ? 64 : 0);
                n_tap = stay ? n_tap : t1; n_toff = stay ? n_toff : o1; n_dx = stay ? n_dx : x1;
            } else {
                n_cb += last_tap ? 64 : 0;
                n_tap = t1; n_toff = o1; n_dx = x1;
            }
        }
        is_src1 = cb >= p.c0;
        s_c2 = (unsigned)(is_src1 ? p.c1 : p.c0) * 2u;
        s_sh = (unsigned)(31 - min(tap, 31));                          // taps past the filter (K padding) hit a set bit: zeros
        v_t = slot16 + (unsigned)toff * s_c2;
        is_ccb = (unsigned)(is_src1 ? cb - p.c0 : cb) * 2u;
        is_kb = (unsigned)((kbase + kt) * BK) * 2u;
        is_buf = buf;
        if constexpr (LIN) {             // one tap: the offsets only change with the source (once per K loop of a two-source call)
            const int src = is_src1 ? 1 : 0;
            if (src != cur_src) { cur_src = src; static_for<AJ>(prep_row); }
        }
    };
    // prologue form: scalars and every row at once (inside the K loop the rows ride in MFMA gaps: substep())
    auto prepare = [&](int kt, int buf) __attribute__((always_inline)) {
        prep_scalars(kt, buf);
        if constexpr (!LIN) static_for<AJ>(prep_row);
    };
    // DMA piece JJ of the prepared tile: pieces < AJ feed activation rows, the rest weight rows.  The wave-uniform parts
    // of the source address (channel chunk; K position and piece row of the weights) travel in the scalar offset.
    auto dma_piece = [&](auto jj_) __attribute__((always_inline)) {
        constexpr int jj = decltype(jj_)::value;
        if constexpr (jj < AJ) {
            constexpr int j = jj;
            const bool real = (GA % NW == 0) || (wave + NW * j < GA);
            char* a = smem + is_buf * STAGE_BYTES + wave * RPI * ROWB;
            async_copy16_buf_s(is_src1 ? r_a1 : r_a0, real ? pb[j] : OOB, is_ccb, real ? a + j * NW * RPI * ROWB : dummy);
        } else {
            constexpr int j = jj - AJ;
            const bool real = (GB % NW == 0) || (wave + NW * j < GB);
            char* b = smem + is_buf * STAGE_BYTES + BM * ROWB + wave * RPI * ROWB;
            async_copy16_buf_s(r_w, real ? wb0 : OOB, is_kb + j * wb_step, real ? b + j * NW * RPI * ROWB : dummy);
        }
    };
    auto dma_range = [&](auto j0_, auto j1_) __attribute__((always_inline)) {
        constexpr int J0 = decltype(j0_)::value, J1 = decltype(j1_)::value;
        static_for<J1 - J0>([&](auto t) __attribute__((always_inline)) { dma_piece(IntTag<J0 + decltype(t)::value>()); });
    };

    // ---- fragment read addresses: row = base + (lane&31), k-slot ks*2 + (lane>>5), un-swizzled per row.  A row is 128 (64)
    // bytes and every stage starts on a multiple of that, so the byte address of sub-step ks is (address of sub-step 0) ^ (ks << 5):
    // one register per fragment row block, the sub-step is an XOR constant inside the read statement
    const int frow = lane & 31, fh = lane >> 5;
    int a_off[MI], w_off[NI];
    auto fragment_offsets = [&]() __attribute__((always_inline)) {        // (called behind the first DMA issue: nothing waits for it)
#pragma unroll
        for (int i = 0; i < MI; ++i) { const int rr = wm * (BM / WM) + i * 32 + frow; a_off[i] = rr * ROWB + ((fh ^ ((rr / RPB) & swm)) << 4); }
        const int prow = (frow & 3) + 4 * (frow >> 3) + 16 * ((frow >> 2) & 1);      // permuted weight rows: conv_gemm_dma.h
#pragma unroll
        for (int j = 0; j < NI; ++j) { const int rr = wn * (BN / WN) + j * 32 + prow; w_off[j] = BM * ROWB + rr * ROWB + ((fh ^ ((rr / RPB) & swm)) << 4); }
    };
    static_assert(STAGE_BYTES % ROWB == 0 && ROWB % 64 == 0 && (BM * ROWB) % 256 == 0, "XOR addressing of the k sub-steps");

    AccFile af;
    if constexpr (PER_CU > 1) wave_priority<2>();               // K loop above the co-resident workgroup's epilogue
    // The accumulation starts from the bias (this lane's 16 columns of every column block, the same for all row blocks): the
    // epilogue adds nothing.  Fetched here, ahead of the first operand pieces; written into the accumulators once those are on
    // their way.  Split-K partials and per-row biases start from zero (a descriptor of length 0 loads zeros).
    const bool bias_folded = k_splits == 1 && !p.bias_per_row;
    const BufRsrc r_bias = make_rsrc(p.bias, (p.bias && bias_folded) ? (unsigned)p.n_out * 2u : 0u);
    // LayerNorm folded into this contraction (AaConvGemm.ln_stats):  out = rstd * (x W'^T - mean * colsum(W') + b' / rstd).  The
    // bracket's two rank-1 terms are where the accumulation STARTS (2 VALU per accumulator, spent while the first operand stage is
    // still on its way - in the epilogue the same arithmetic cost the K = 320 GEGLU / Q|K|V calls 15-20 %, r04d); the epilogue only
    // multiplies by rstd.  Row m of block row i is this lane's row (lane & 31) in the accumulator layout.
    // (one wave per SIMD only: the two-waves-per-SIMD tiles have 128 registers for everything outside the accumulators and would
    // spill the prefetched column vectors; the host does not offer them a folded call - aa_conv_gemm_tile_ok)
    constexpr bool LN_OK = cgx_ln_ok(BK, WM, WN, PER_CU);
    const bool ln_start = LN_OK && p.ln_stats != nullptr && k_splits == 1;
    // The row coefficients (16 bytes per row) are fetched HERE, with the bias and ahead of the first operand pieces; the column
    // vectors of this wave's columns travel by LDS-DMA into a private LDS copy, as the OLDEST pieces of the wave: the counted
    // wait in front of start_from_bias() covers them while the operand stages stay in flight (a global load issued where its value is
    // needed costs a K = 320 tile 1-2 us of its ~13 - r04e: the fold lost 30 % on GEGLU / Q|K|V that way; prefetching into registers
    // spills: 160 registers on the 320-column tiles).
    constexpr int LNW = BN / WN;                         // columns of a wave
    float* sLnW = reinterpret_cast<float*>(dummy + 2048) + wave * 512;
    float ln_nmean[LN_OK ? MI : 1], ln_sd[LN_OK ? MI : 1], ln_rstd[LN_OK ? MI : 1];
    u32x4 bias_raw[NI][2];
    if (!ln_start) {                     // (a folded call has no bias)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                bias_raw[j][q] = buf_load16(r_bias, (unsigned)(tile_n * BN + wn * (BN / WN) + j * 32 + 16 * (lane >> 5) + 8 * q) * 2u);
    }
    if constexpr (LN_OK) if (ln_start) {
        static_assert(!LN_OK || LNW * 4 <= 1024, "one LDS-DMA piece per column vector");
        const BufRsrc r_ln = make_rsrc(p.ln_cols, (unsigned)(2 * p.n_pad) * 4u);
        const unsigned c0b = (unsigned)(tile_n * BN + wn * LNW) * 4u + (unsigned)lane * 16u;
        const bool in = lane * 4 < LNW;
        async_copy16_buf(r_ln, in ? c0b : OOB, sLnW);                                       // colsum(W')
        async_copy16_buf(r_ln, in ? c0b + (unsigned)p.n_pad * 4u : OOB, sLnW + 256);       // b' (each piece deposits 64 x 16 bytes)
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const LnRow r = cgd_ln_row(p, min(m_begin + tile_m * BM + wm * (BM / WM) + i * 32 + (lane & 31), M - 1));
            ln_nmean[i] = r.nmean; ln_sd[i] = r.sd; ln_rstd[i] = r.a;
        }
    }
    auto start_from_bias = [&]() __attribute__((always_inline)) {
        static_for<NI>([&](auto j_) __attribute__((always_inline)) {
            constexpr int j = decltype(j_)::value;
            if (LN_OK && ln_start) {
              if constexpr (LN_OK) {
                f32x16 cs, bp;                                 // this lane's 16 columns of block j (the wave's own DMA: its counted wait suffices)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const f32x4 c4 = *reinterpret_cast<const f32x4*>(sLnW + j * 32 + 16 * (lane >> 5) + 4 * q4);
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(sLnW + 256 + j * 32 + 16 * (lane >> 5) + 4 * q4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { cs[4 * q4 + e] = c4[e]; bp[4 * q4 + e] = b4[e]; }
                }
                static_for<MI>([&](auto i_) __attribute__((always_inline)) {
                    constexpr int i = decltype(i_)::value;
                    f32x16 v;
#pragma unroll
                    for (int e = 0; e < 16; e += 2) {           // packed fp32 pairs: half the multiply / fma issues of this VALU-bound prologue
                        const f32x2 t = f32x2{bp[e], bp[e + 1]} * f32x2{ln_sd[i], ln_sd[i]};          // (r04 phase probe: the scalar form was 512 of its ~770 instructions)
                        const f32x2 r = __builtin_elementwise_fma(f32x2{ln_nmean[i], ln_nmean[i]}, f32x2{cs[e], cs[e + 1]}, t);
                        v[e] = r[0]; v[e + 1] = r[1];
                    }
                    acc_init<i * NI + j>(af, v);
                });
              }
            } else {
                f32x16 b;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    Pack8<T> h; h.raw = bias_raw[j][q];
#pragma unroll
                    for (int e = 0; e < 8; ++e) b[8 * q + e] = (float)h.e[e];
                }
                static_for<MI>([&](auto i_) __attribute__((always_inline)) { acc_init<decltype(i_)::value * NI + j>(af, b); });
            }
        });
    };

    u32x4 fa[2][MI], fw[2][NI];
    // read number rd of sub-step ks of the stage at `st` into fragment set `set` (order of first use: a0, w0 .. w(NI-1), a1 ..)
    auto frag_read = [&](const char* st, auto ks_, auto set_, auto rd_) __attribute__((always_inline)) {
        constexpr int ks = decltype(ks_)::value, set = decltype(set_)::value, rd = decltype(rd_)::value;
        if constexpr (rd == 0)       lds_read16_xor(fa[set][0], st + a_off[0], IntTag<(ks << 5)>());
        else if constexpr (rd <= NI) lds_read16_xor(fw[set][rd - 1], st + w_off[rd - 1], IntTag<(ks << 5)>());
        else                         lds_read16_xor(fa[set][rd - NI], st + a_off[rd - NI], IntTag<(ks << 5)>());
    };
    // One sub-step: NB MFMAs out of fragment set ks&1; the gap behind MFMA g carries read g of the NEXT sub-step (g < R,
    // from stage `st_rd`, sub-step KSN) and / or DMA piece (DJ0 + g - G0) of the prepared tile for g in [G0, G0 + DN).
    // PREP: the gaps behind the first AJ MFMAs also carry the source offsets of the prepared tile's activation rows (its pieces
    // follow in later gaps of this sub-step or in the next K step).
    auto substep = [&](auto ks_, auto ksn_, const char* st_rd, const bool do_read, auto dj0_, auto dn_, auto prep_) __attribute__((always_inline)) {
        constexpr int ks = decltype(ks_)::value, ksn = decltype(ksn_)::value, set = ks & 1;
        constexpr int DJ0 = decltype(dj0_)::value, DN = decltype(dn_)::value;
        constexpr bool PREP = decltype(prep_)::value && !LIN;
        constexpr int G0 = (DN <= NB - R) ? R : NB - DN;              // DMA pieces prefer the read-free gaps at the end
        static_for<NB>([&](auto g_) __attribute__((always_inline)) {
            constexpr int g = decltype(g_)::value, i = g / NI, j = g % NI;
            acc_mfma<i * NI + j>(af, T(), fw[set][j], fa[set][i]);
            if constexpr (g < R) { if (do_read) frag_read(st_rd, IntTag<ksn>(), IntTag<set ^ 1>(), IntTag<g>()); }
            if constexpr (PREP && g < AJ) prep_row(IntTag<g>());
            if constexpr (DN > 0 && g >= G0 && g < G0 + DN) dma_piece(IntTag<DJ0 + g - G0>());
        });
        lds_wait_all();
    };
    // K step kt: HAS_NEXT = tile kt+1 exists (the rest of its DMA goes out under sub-steps 0..2, its first fragments are read
    // under sub-step 3), HAS_NEXT2 = tile kt+2 exists (its first DP3 pieces go out under sub-step 3)
    auto kstep = [&](int kt, auto has_next_, auto has_next2_) __attribute__((always_inline)) {
        constexpr bool HAS_NEXT = decltype(has_next_)::value, HAS_NEXT2 = decltype(has_next2_)::value;
        const char* st = smem + (kt & 1) * STAGE_BYTES;
        const char* st_next = smem + ((kt + 1) & 1) * STAGE_BYTES;
        substep(IntTag<0>(), IntTag<1>(), st, true, IntTag<DP3>(), IntTag<HAS_NEXT ? DP0 : 0>(), BoolTag<false>());
        substep(IntTag<1>(), IntTag<2>(), st, true, IntTag<DP3 + DP0>(), IntTag<HAS_NEXT ? DP1 : 0>(), BoolTag<false>());
        substep(IntTag<2>(), IntTag<3>(), st, true, IntTag<DP3 + DP0 + DP1>(), IntTag<HAS_NEXT ? DP2 : 0>(), BoolTag<false>());
        if constexpr (HAS_NEXT) {
            if constexpr (!(AA_X_ABLATE & 64)) dma_wait<0>();    // my pieces of tile kt+1 landed ...
            if constexpr (!(AA_X_ABLATE & 32)) block_barrier();  // ... everyone's did, and every wave holds its last fragments of tile kt
        }
        if constexpr (HAS_NEXT2) prep_scalars(kt + 2, kt & 1);   // stage kt&1 is free from here on; every piece of tile kt+1 is out: pb[] may change
        substep(IntTag<3>(), IntTag<0>(), st_next, HAS_NEXT, IntTag<0>(), IntTag<HAS_NEXT2 ? DP3 : 0>(), BoolTag<HAS_NEXT2>());
    };

    // YOUNGER = operand pieces this wave has issued since the fold's two column-vector pieces: those must have landed, these stay in flight
    auto put_bias = [&](auto younger_) __attribute__((always_inline)) {       // the general epilogue path adds a bias slice from LDS: zeros here, the
        if (tid < BN / 8) *reinterpret_cast<u32x4*>(sBias + tid * 8) = u32x4{0u, 0u, 0u, 0u};     // accumulators already carry the bias
        if constexpr (LN_OK) { if (ln_start) { dma_wait<decltype(younger_)::value>(); wave_sync(); } }      // (the pieces were deposited by all lanes of this wave)
        start_from_bias();
    };
    if constexpr (BK == 64) {
        if (nk > 0) {
            prepare(0, 0);
            stamp(1);
            dma_range(IntTag<0>(), IntTag<PER_TILE>());
            if (nk > 1) { prepare(1, 1); dma_range(IntTag<0>(), IntTag<DP3>()); }
            fragment_offsets();
            if (nk > 1) put_bias(IntTag<PER_TILE + DP3>()); else put_bias(IntTag<PER_TILE>());
            if (nk > 1) dma_wait<DP3>(); else dma_wait<0>();
            block_barrier();
            stamp(6);
            static_for<R>([&](auto rd) __attribute__((always_inline)) { frag_read(smem, IntTag<0>(), IntTag<0>(), rd); });
            lds_wait_all();
            int kt = 0;
            for (; kt + 2 < nk; ++kt) kstep(kt, BoolTag<true>(), BoolTag<true>());
            if (nk >= 2) { kstep(kt, BoolTag<true>(), BoolTag<false>()); ++kt; }
            kstep(kt, BoolTag<false>(), BoolTag<false>());
        } else {
            put_bias(IntTag<0>());
            __syncthreads();
        }
    } else {
        // ---- deep ring: stage s lives in slot s % RING; after the barrier of step s the slots hold stages s+1 .. s+RING.
        // HN = stage s+1 exists; HA = stage s+RING-1 exists (the rest of its pieces go out under the first sub-step);
        // HB = stage s+RING exists (prepared behind the barrier, first DPB pieces under the second sub-step);
        // NIF = stages among s+2 .. s+RING-1 that exist = stages still in flight when stage s+1 is waited for.
        constexpr int DPB = DP3, DPA = PER_TILE - DP3;
        auto stage_step = [&](int s, auto hn_, auto ha_, auto hb_, auto nif_) __attribute__((always_inline)) {
            constexpr bool HN = decltype(hn_)::value, HA = decltype(ha_)::value, HB = decltype(hb_)::value;
            constexpr int NIF = decltype(nif_)::value;
            const char* st = smem + (s % RING) * STAGE_BYTES;
            const char* st_next = smem + ((s + 1) % RING) * STAGE_BYTES;
            substep(IntTag<0>(), IntTag<1>(), st, true, IntTag<DPB>(), IntTag<HA ? DPA : 0>(), BoolTag<false>());
            if constexpr (HN) {
                if constexpr (!(AA_X_ABLATE & 64)) dma_wait<PER_TILE * NIF>();      // my pieces of stage s+1 landed; younger stages stay in flight
                if constexpr (!(AA_X_ABLATE & 32)) block_barrier();                 // everyone's did; every wave holds its last fragments of stage s
            }
            if constexpr (HB) prep_scalars(s + RING, s % RING);         // slot s % RING is free from here on
            substep(IntTag<1>(), IntTag<0>(), st_next, HN, IntTag<0>(), IntTag<HB ? DPB : 0>(), BoolTag<HB>());
        };
        if (nk > 0) {
            // stages 0 .. RING-2 in full, the first pieces of stage RING-1 (= the second sub-step of a "stage -1")
            stamp(1);
            static_for<RING - 1>([&](auto k_) __attribute__((always_inline)) {
                constexpr int k = decltype(k_)::value;
                if (nk > k) { prepare(k, k); dma_range(IntTag<0>(), IntTag<PER_TILE>()); }
            });
            if (nk > RING - 1) { prepare(RING - 1, RING - 1); dma_range(IntTag<0>(), IntTag<DPB>()); }
            fragment_offsets();
            put_bias(IntTag<0>());                                      // (deep-ring tiles carry no folded LayerNorm: cgx_ln_ok)
            if (nk > RING - 1) dma_wait<(RING - 2) * PER_TILE + DPB>();
            else if (RING == 4 && nk == 3) dma_wait<2 * PER_TILE>();
            else if (nk >= 2) dma_wait<PER_TILE>();                     // (nk == 2, or RING == 3 and nk == 2)
            else dma_wait<0>();
            block_barrier();
            stamp(6);
            static_for<R>([&](auto rd) __attribute__((always_inline)) { frag_read(smem, IntTag<0>(), IntTag<0>(), rd); });
            lds_wait_all();
            int s = 0;
            for (; s + RING < nk; ++s) stage_step(s, BoolTag<true>(), BoolTag<true>(), BoolTag<true>(), IntTag<RING - 2>());
            // the last RING steps: stage s+RING, then s+RING-1, ... no longer exist
            if constexpr (RING == 4) {
                if (nk >= 4) { stage_step(s, BoolTag<true>(), BoolTag<true>(), BoolTag<false>(), IntTag<2>()); ++s; }
                if (nk >= 3) { stage_step(s, BoolTag<true>(), BoolTag<false>(), BoolTag<false>(), IntTag<1>()); ++s; }
                if (nk >= 2) { stage_step(s, BoolTag<true>(), BoolTag<false>(), BoolTag<false>(), IntTag<0>()); ++s; }
            } else {
                if (nk >= 3) { stage_step(s, BoolTag<true>(), BoolTag<true>(), BoolTag<false>(), IntTag<1>()); ++s; }
                if (nk >= 2) { stage_step(s, BoolTag<true>(), BoolTag<false>(), BoolTag<false>(), IntTag<0>()); ++s; }
            }
            stage_step(s, BoolTag<false>(), BoolTag<false>(), BoolTag<false>(), IntTag<0>());
        } else {
            put_bias(IntTag<0>());
            __syncthreads();
        }
    }
    LnRstd<MI> ln_scale;
    ln_scale.on = ln_start;
#pragma unroll
    for (int i = 0; i < MI; ++i) ln_scale.v[i] = LN_OK ? ln_rstd[LN_OK ? i : 0] : 1.0f;
    acc_settle();                                                // MFMA results visible to v_accvgpr_read
    stamp(2);
    if constexpr (PER_CU > 1) wave_priority<0>();

    const int ec = lane & 31, eh = lane >> 5;
    const int m_tile = m_begin + tile_m * BM;
    const int n_wave = tile_n * BN + wn * (BN / WN);
    if (k_splits > 1) {
        float* ws = reinterpret_cast<float*>(p.workspace) + ((int64_t)blockIdx.y * (M - m_begin) - m_begin) * p.n_pad;
        static_for<MI>([&](auto i_) __attribute__((always_inline)) {
            constexpr int i = decltype(i_)::value;
            const int m = m_tile + wm * (BM / WM) + i * 32 + ec;
            static_for<NI>([&](auto j_) __attribute__((always_inline)) {
                constexpr int j = decltype(j_)::value;
                const f32x16 a = acc_get<i * NI + j>(af);
                if (m < M) {
                    float* dst = ws + (int64_t)m * p.n_pad + n_wave + j * 32 + 16 * eh;
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) *reinterpret_cast<f32x4*>(dst + 4 * qq) = f32x4{a[4 * qq], a[4 * qq + 1], a[4 * qq + 2], a[4 * qq + 3]};
                }
            });
        });
        if constexpr (!cgx_ticket_ok(BM, BN, WM, WN, PER_CU)) return;
        else {
        if (p.tickets == nullptr) return;                        // the host follows up with splitk_reduce_kernel
        // ---- finish inside the kernel (AaConvGemm.tickets): the LAST of this tile's k_splits workgroups to get here sums the
        // partials in split order - its own from the accumulation registers (bit-identical to what it stored), the others from
        // the workspace - and runs the usual epilogue.  Release / acquire as in the classic last-block reduction: fence behind the
        // partial stores of every wave, barrier, one ticket per workgroup, fence in front of the reads.  The counter goes back
        // to zero for the next launch (nobody else touches it after the last ticket was drawn).
        int* last_flag = reinterpret_cast<int*>(dummy);
        __threadfence();
        __syncthreads();
        if (tid == 0) {
            const int ticket = atomicAdd(p.tickets + blockIdx.x, 1);
            const int last = ticket == k_splits - 1;
            if (last) p.tickets[blockIdx.x] = 0;
            *last_flag = last;
        }
        __syncthreads();
        if (*last_flag == 0) return;
        __threadfence();
        const int my_split = blockIdx.y;
        const float* ws0 = reinterpret_cast<const float*>(p.workspace) - (int64_t)m_begin * p.n_pad;
        const int64_t split_stride = (int64_t)(M - m_begin) * p.n_pad;
        const BufRsrc r_cb = make_rsrc(p.bias, (p.bias && !p.bias_per_row) ? (unsigned)p.n_out * 2u : 0u);      // (per-row biases: general epilogue)
        auto get_sum = [&](auto i_, auto j_) __attribute__((always_inline)) {
            constexpr int i = decltype(i_)::value, j = decltype(j_)::value;
            const int m = min(m_tile + wm * (BM / WM) + i * 32 + ec, M - 1);               // (rows behind the tensor: any finite value, never stored)
            const float* src = ws0 + (int64_t)m * p.n_pad + n_wave + j * 32 + 16 * eh;
            const f32x16 own = acc_get<i * NI + j>(af);
            f32x16 v;
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = 0.0f;
            for (int sp = 0; sp < k_splits; ++sp) {
                if (sp == my_split) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] += own[e];
                } else {
                    const f32x4* q4 = reinterpret_cast<const f32x4*>(src + sp * split_stride);
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        const f32x4 t = q4[qq];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[4 * qq + e] += t[e];
                    }
                }
            }
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {                     // the accumulators of a split started from zero: the column bias joins here
                Pack8<T> b; b.raw = buf_load16(r_cb, (unsigned)(n_wave + j * 32 + 16 * eh + 8 * qq) * 2u);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[8 * qq + e] += (float)b.e[e];
            }
            return v;
        };
        LnRstd<MI> no_ln;
        no_ln.on = false;
#pragma unroll
        for (int i = 0; i < MI; ++i) no_ln.v[i] = 1.0f;
        cgd_epilogue_g<T, MI, NI, true>(p, M, get_sum, m_tile + wm * (BM / WM), n_wave, sBias + wn * (BN / WN), p.row_stats ? tile_n * WN + wn : -1, no_ln);
        return;
        }
    }
    static_assert(NW * (BM / WM) * 8 <= CGX_COEF_BYTES, "row-statistics exchange area");
    float* sCoef = reinterpret_cast<float*>(smem + cgx_lds_main_bytes(BM, BN, BK, RING, WM, WN, PER_CU));
    cgd_epilogue_g<T, MI, NI, true>(p, M, [&](auto i_, auto j_) __attribute__((always_inline)) { return acc_get<decltype(i_)::value * NI + decltype(j_)::value>(af); },
                              m_tile + wm * (BM / WM), n_wave, sBias + wn * (BN / WN),
                              (p.row_stats || p.row_coef) ? tile_n * WN + wn : -1, ln_scale, (p.row_coef && tiles_n == 1) ? sCoef : nullptr, WN);
    stamp(5);
    stamp_wall(4, -1);
}

}  // namespace aa
