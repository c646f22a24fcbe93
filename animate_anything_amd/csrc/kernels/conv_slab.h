// 3x3 stride-1 convolution with HALO-SLAB staging (fast path of aa_conv_gemm for ResnetBlock2D.conv1/conv2 and the
// Upsample-free 3x3 convolutions of the 64x64 / 32x32 / 16x16 levels).
//
// The im2col kernel (conv_gemm_dma.h) re-fetches every activation row nine times, once per filter tap: per K step of a
// 256x320 tile the LDS-DMA moves (256 + 320) rows, and the CU-wide DMA issue path + the L2 -> LDS stream are what the matrix
// pipe waits for.  Here a tile is BM consecutive output pixels = R = BM / W whole image rows of one image, and the
// activations of one 32-channel unit are staged ONCE as a slab of (R + 2) x (W + 2) pixels (halo included, out-of-image
// pixels arrive as zeros through the descriptor range check); the nine taps are nine VIEWS of that slab: the fragment read
// of tap (dy, dx) reads slab row  pixel_row + dy*(W+2) + dx.  Per nine K steps the DMA moves (R+2)(W+2) + 9*BN rows instead
// of 9*(BM + BN): -37 % instructions and bytes for the 256x320 tile at W = 64.
//
//  * K order: (32-channel unit u, tap t): the weights are the chunk-major packing of ops.pack_weight (k_order 1:
//    (64-channel chunk, tap, 64 channels)), read as (chunk u/2, tap, half u%2);
//  * slab ring: two halves; unit u+1 is fetched into the half that unit u-1 used, one DMA instruction per wave during the
//    first taps of unit u;  weight ring: three stages, the tile of step k+2 is issued at step k;
//  * one raw s_barrier per K step, counted vmcnt (in-order: "everything but the youngest N landed");
//  * XOR swizzle of the 16-byte k-slots on the source side / fragment-read side as in conv_gemm_dma.h (64-byte rows, four
//    rows per 256-byte bank line): a fragment read touches 32 CONSECUTIVE slab rows whatever the tap, so the sixteen lanes
//    of a read group always hit sixteen distinct (row mod 4, slot xor key) positions;
//  * epilogue: cgd_epilogue (shared with conv_gemm_dma.h).
// Host-checked preconditions (aa_api_impl.h cg_slab_ok): 3x3, stride 1, pad 1, no resize, h_out == h_in, w_out == w_in,
// BM % W == 0, (H*W) % BM == 0, c0 % 64 == 0, c1 % 64 == 0, k_order == 1, no K split.
#pragma once
#include "dev.h"
#include "aa_mi355.h"
#include "conv_gemm_dma.h"

namespace aa {

constexpr int CS_BK = 32;
constexpr int CS_SLAB_ROWS = 416;                 // padded slab rows (multiple of 16): (4+2)*(64+2) = 396 is the largest shape
constexpr int CS_BSTAGES = 3;

__host__ __device__ inline int cs_lds_bytes(int bn) { return 2 * CS_SLAB_ROWS * 64 + CS_BSTAGES * bn * 64 + 1024 + 1024; }

template <typename T, int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(64 * WM * WN, (WM * WN + 3) / 4) conv3x3_slab_kernel(const AaConvGemm p, const int M, const int tiles_n, const int m_begin) {
    constexpr int NW = WM * WN;
    constexpr int MI = BM / WM / 32, NI = BN / WN / 32;
    constexpr int ROWB = 64;                        // bytes per LDS row (32 channels)
    constexpr int GB = BN / 16;                     // DMA groups (16 rows each) of the weight tile
    constexpr int BJ = (GB + NW - 1) / NW;
    constexpr int SI = CS_SLAB_ROWS / 16;           // DMA groups of a slab half
    constexpr int SJ = (SI + NW - 1) / NW;          // slab pieces per wave and unit, one per K step (taps 0 .. SJ-1)
    constexpr unsigned OOB = 0x80000000u;
    static_assert(SJ <= 9 && BM % (32 * WM) == 0 && BN % (32 * WN) == 0, "tile shape");
    char* smem = dyn_smem();
    char* slab = smem;                                                  // [2][CS_SLAB_ROWS][64 B]
    char* wring = smem + 2 * CS_SLAB_ROWS * ROWB;                       // [CS_BSTAGES][BN][64 B]
    char* dummy = wring + CS_BSTAGES * BN * ROWB;
    T* sBias = reinterpret_cast<T*>(dummy + 1024);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = wave_id();
    const int wm = wave / WN, wn = wave % WN;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int tile_m = logical / tiles_n;
    const int tile_n = logical - tile_m * tiles_n;

    const int W = p.w_in, H = p.h_in, WP = W + 2;
    const int m_tile = m_begin + tile_m * BM;                          // first output pixel of the tile
    const int img = m_tile / (H * W);
    const int y0 = (m_tile - img * H * W) / W;                          // first image row of the tile
    const int R = BM / W;
    const int slab_rows = (R + 2) * WP;
    const int nunits = (p.c0 + p.c1) >> 5;                              // 32-channel units
    const int nk = nunits * 9;

    const BufRsrc r_a0 = make_rsrc(p.a0, (unsigned)((int64_t)p.n_img * H * W * p.c0 * 2));
    const BufRsrc r_a1 = make_rsrc(p.a1, p.c1 ? (unsigned)((int64_t)p.n_img * H * W * p.c1 * 2) : 0u);
    const BufRsrc r_w = make_rsrc(p.w, (unsigned)((int64_t)p.n_pad * p.k_pad * 2));

    // ---- slab DMA geometry: piece j of this wave covers slab rows (wave + NW*j)*16 + lane/4, 16-byte slot lane%4 (swizzled) ----
    const int lrow = lane >> 2, lpos = lane & 3;
    int spix[SJ];                                   // source pixel index of the lane's slab row, -1 = halo outside the image / padding row
    unsigned sslot[SJ];                             // byte offset of the lane's (swizzled) k-slot inside a 32-channel unit
#pragma unroll
    for (int j = 0; j < SJ; ++j) {
        const int sr = (wave + NW * j) * 16 + lrow;
        const int yy = sr / WP, xx = sr - yy * WP;
        const int y = y0 - 1 + yy, x = xx - 1;
        const bool ok = sr < slab_rows && y >= 0 && y < H && x >= 0 && x < W;
        spix[j] = ok ? (img * H + y) * W + x : -1;
        sslot[j] = (unsigned)((lpos ^ ((sr >> 2) & 3)) * 16);
    }
    unsigned wb[BJ];                                // weight rows of this lane: byte offset of row start + swizzled slot
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int rr = (wave + NW * j) * 16 + lrow;
        wb[j] = (unsigned)((tile_n * BN + rr) * p.k_pad) * 2u + (unsigned)((lpos ^ ((rr >> 2) & 3)) * 16);
    }
    auto issue_slab_piece = [&](int u, auto j_) {
        constexpr int j = decltype(j_)::value;
        const bool real = (SI % NW == 0) || (wave + NW * j < SI);
        const int cb = u * 32;                                          // channel base of the unit over the concat
        const bool src1 = cb >= p.c0;
        const int cs = src1 ? p.c1 : p.c0;
        const unsigned off = (unsigned)(spix[j] * cs + (src1 ? cb - p.c0 : cb)) * 2u + sslot[j];
        async_copy16_buf(src1 ? r_a1 : r_a0, (real && spix[j] >= 0) ? off : OOB,
                         real ? slab + (u & 1) * (CS_SLAB_ROWS * ROWB) + (wave + NW * j) * 1024 : dummy);
    };
    auto issue_weights = [&](int kk, int u, int t) {
        const unsigned koff = (unsigned)((((u >> 1) * 9 + t) << 6) + ((u & 1) << 5)) * 2u;    // (chunk, tap, half) in the packed K
        char* dst = wring + (kk % CS_BSTAGES) * (BN * ROWB);
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const bool real = (GB % NW == 0) || (wave + NW * j < GB);
            async_copy16_buf(r_w, real ? wb[j] + koff : OOB, real ? dst + (wave + NW * j) * 1024 : dummy);
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // fragment geometry: activation rows = slab rows of the wave's output pixels (tap (0,0)); weight rows permuted (see cgd_epilogue)
    const int frow = lane & 31, fh = lane >> 5;
    int sr0[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int rr = wm * (BM / WM) + i * 32 + frow;                 // tile row = output pixel
        const int ty = rr / W, tx = rr - ty * W;
        sr0[i] = ty * WP + tx;
    }
    const int prow = (frow & 3) + 4 * (frow >> 3) + 16 * ((frow >> 2) & 1);
    int b_off[NI], b_swz[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) { const int rr = wn * (BN / WN) + j * 32 + prow; b_off[j] = rr * ROWB; b_swz[j] = (rr >> 2) & 3; }

    // ---- prologue: slab of unit 0, weight tiles of steps 0 and 1 ----
    {
        auto pieces0 = [&](auto j_) { issue_slab_piece(0, j_); };
        pieces0(IntTag<0>());
        if constexpr (SJ > 1) pieces0(IntTag<1>());
        if constexpr (SJ > 2) pieces0(IntTag<2>());
        if constexpr (SJ > 3) pieces0(IntTag<3>());
        static_assert(SJ <= 4, "slab pieces per wave");
        issue_weights(0, 0, 0);
        if (nk > 1) issue_weights(1, 0, 1);
    }
    if (tid < BN / 8) {                             // bias slice -> LDS behind the first DMA (published by the K-loop barriers)
        const int n = tile_n * BN + tid * 8;
        u32x4 b = u32x4{0u, 0u, 0u, 0u};
        if (p.bias && n + 8 <= p.n_out) b = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.bias) + n);
        *reinterpret_cast<u32x4*>(sBias + tid * 8) = b;
    }
    int u = 0, t = 0, dy = 0, dx = 0;               // unit / tap of step kk
    int u2 = 0, t2 = 2;                             // unit / tap of step kk + 2
    bool piece_prev = false;                        // a slab piece was issued at step kk - 1
    for (int kk = 0; kk < nk; ++kk) {
        // step kk's weight tile (and, transitively, every older DMA incl. this unit's slab) must have landed; younger: the weight
        // tile of step kk+1 and the slab piece of step kk-1
        const bool next_w = kk + 1 < nk;
        if (next_w) { if (piece_prev) dma_wait<BJ + 1>(); else dma_wait<BJ>(); }
        else        { if (piece_prev) dma_wait<1>(); else dma_wait<0>(); }
        block_barrier();
        if (kk + 2 < nk) issue_weights(kk + 2, u2, t2);
        piece_prev = false;
        if (u + 1 < nunits && t < SJ) {             // this wave's piece t of the next unit's slab
            if (t == 0) issue_slab_piece(u + 1, IntTag<0>());
            if constexpr (SJ > 1) { if (t == 1) issue_slab_piece(u + 1, IntTag<1>()); }
            if constexpr (SJ > 2) { if (t == 2) issue_slab_piece(u + 1, IntTag<2>()); }
            if constexpr (SJ > 3) { if (t == 3) issue_slab_piece(u + 1, IntTag<3>()); }
            piece_prev = true;
        }
        {   // multiply: tap (dy, dx) view of slab half u&1 against weight stage kk%3
            const char* sa = slab + (u & 1) * (CS_SLAB_ROWS * ROWB);
            const char* sb = wring + (kk % CS_BSTAGES) * (BN * ROWB);
            const int shift = dy * WP + dx;
            int a_off[MI], a_swz[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) { const int sr = sr0[i] + shift; a_off[i] = sr * ROWB; a_swz[i] = (sr >> 2) & 3; }
#pragma unroll
            for (int ks = 0; ks < CS_BK / 16; ++ks) {
                u32x4 fa[MI], fb[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i) fa[i] = *reinterpret_cast<const u32x4*>(sa + a_off[i] + (((ks * 2 + fh) ^ a_swz[i]) << 4));
#pragma unroll
                for (int j = 0; j < NI; ++j) fb[j] = *reinterpret_cast<const u32x4*>(sb + b_off[j] + (((ks * 2 + fh) ^ b_swz[j]) << 4));
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j] = mfma_32x32x16(T(), fb[j], fa[i], acc[i][j]);
            }
        }
        // advance (selects, no divisions)
        { const bool last = t == 8; t = last ? 0 : t + 1; u = last ? u + 1 : u; const bool wrap = dx == 2; dx = (last || wrap) ? 0 : dx + 1; dy = last ? 0 : (wrap ? dy + 1 : dy); }
        { const bool last = t2 == 8; t2 = last ? 0 : t2 + 1; u2 = last ? u2 + 1 : u2; }
    }
    cgd_epilogue<T, MI, NI>(p, M, acc, m_tile + wm * (BM / WM), tile_n * BN + wn * (BN / WN), sBias + wn * (BN / WN));
}

}  // namespace aa
