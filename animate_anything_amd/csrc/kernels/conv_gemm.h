// Implicit-GEMM convolution / linear kernel for gfx950 (see include/aa_mi355.h: aa_conv_gemm).
//
//   out[M][n_out] = epilogue( A[M][K] * W[n][K]^T ),  A gathered from channels-last activations.
//
// Tiling: one workgroup = 4 wavefronts (2 x 2) owns a 128 x BN output tile and walks K in steps of
// 64.  Each wave owns 64 x BN/2 outputs = 2 x (BN/64) accumulators of v_mfma_f32_32x32x16 (16 fp32
// per lane each).  Operands are staged global -> VGPR -> LDS (the gather, the zero fill of the conv
// halo and the two-source channel concat are per-lane address decisions, so the tile cannot be a
// lane-linear LDS-DMA image); LDS rows are padded to 72 elements (144 B) which makes the
// ds_write_b128 of the staging pass and the ds_read_b128 of the fragment reads conflict free.
// The loop is the "issue-early / write-late" pipeline: tile t+1 is in flight in registers while
// tile t is multiplied out of LDS buffer t&1; one barrier per K step.
#pragma once
#include "dev.h"
#include "aa_mi355.h"

namespace aa {

constexpr int CG_BM = 128;
constexpr int CG_BK = 64;
constexpr int CG_LDS = 72;            // padded LDS row length in elements
constexpr int CG_THREADS = 256;

__host__ __device__ inline int cg_lds_bytes(int bn) { return 2 * (CG_BM + bn) * CG_LDS * 2; }

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// exact (erf) GELU, x * Phi(x), evaluated as x * sigmoid(z(x)) with z an odd quintic fitted (minimax) to
// logit(Phi(x)) on |x| <= 8 and clamped beyond (Phi is 0 / 1 to 1e-15 there): |error| <= 2.6e-5 absolute over the whole
// real line = 1/40 of an fp16 ulp at 1 (1/300 of a bf16 one) - indistinguishable from erf at storage precision.
// 9 VALU (one exp2, one rcp) instead of the 17 of an Abramowitz-Stegun erf: the GEGLU epilogue of a 128x256 tile
// evaluates 16 K of these per workgroup and was VALU-bound on them.  Coefficients carry the -log2(e) of exp -> exp2.
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float xc = fminf(fmaxf(x, -8.0f), 8.0f);
    const float x2 = xc * xc;
    const float pz = __builtin_fmaf(__builtin_fmaf(0.001014264184050262f, x2, -0.10677573084831238f), x2, -2.301121234893799f);
    return x / (1.0f + fast_exp2(xc * pz));
}

// the same function on two values with the packed fp32 instructions (half the multiply / fma issues; exp2 and rcp stay scalar)
__device__ __forceinline__ f32x2 gelu_erf_2(f32x2 x) {
    const f32x2 xc = f32x2{clamp_f(x[0], -8.0f, 8.0f), clamp_f(x[1], -8.0f, 8.0f)};
    const f32x2 x2 = xc * xc;
    const f32x2 c5 = f32x2{0.001014264184050262f, 0.001014264184050262f}, c3 = f32x2{-0.10677573084831238f, -0.10677573084831238f},
                c1 = f32x2{-2.301121234893799f, -2.301121234893799f};
    const f32x2 pz = __builtin_elementwise_fma(__builtin_elementwise_fma(c5, x2, c3), x2, c1);
    const f32x2 t = xc * pz;
    const f32x2 d = f32x2{fast_exp2(t[0]), fast_exp2(t[1])} + f32x2{1.0f, 1.0f};
    return x * f32x2{fast_rcp(d[0]), fast_rcp(d[1])};
}

template <typename T>
__device__ __forceinline__ void store_out(void* out, int out_dtype, int64_t idx, float v) {
    if (out_dtype == AA_F32) reinterpret_cast<float*>(out)[idx] = v;
    else reinterpret_cast<T*>(out)[idx] = (T)v;
}

template <typename T, int BN>
__global__ void __launch_bounds__(CG_THREADS) conv_gemm_kernel(const AaConvGemm p, const int M, const int tiles_n) {
    constexpr int NT = BN / 64;          // 32-column accumulator blocks per wave
    constexpr int BROWS = BN / 32;       // weight rows staged per thread
    T* sA = reinterpret_cast<T*>(dyn_smem());
    T* sB = sA + 2 * CG_BM * CG_LDS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware tile order: hardware round-robins consecutive workgroup ids over the 8 XCDs, so give
    // every XCD a contiguous run of logical tiles (neighbouring tiles share the A rows / W panel in L2).
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int tile_m = logical / tiles_n;
    const int tile_n = logical - tile_m * tiles_n;

    const int ctot = p.c0 + p.c1;
    const int k_total = p.kh * p.kw * ctot;
    const int nk = p.k_pad / CG_BK;

    // ---- staging geometry: thread -> (16-byte k slot, 4 rows of the A tile, BROWS rows of W) ----
    const int slot = tid & 7;
    const int r0 = tid >> 3;
    int row_img[4], row_iy[4], row_ix[4];
    bool row_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = tile_m * CG_BM + r0 + 32 * i;
        row_ok[i] = m < M;
        const int mm = row_ok[i] ? m : 0;
        const int x = mm % p.w_out;
        const int t = mm / p.w_out;
        const int y = t % p.h_out;
        row_img[i] = t / p.h_out;
        row_iy[i] = y * p.stride - p.pad_h;
        row_ix[i] = x * p.stride - p.pad_w;
    }
    // running decomposition of this thread's k = kt*64 + slot*8 into (dy, dx, channel)
    int kc = slot * 8, kdy = 0, kdx = 0, kk = slot * 8;
    while (kc >= ctot) { kc -= ctot; if (++kdx == p.kw) { kdx = 0; ++kdy; } }
    const bool resize = (p.h_virt != p.h_in) || (p.w_virt != p.w_in);
    const T* wbase = reinterpret_cast<const T*>(p.w) + (int64_t)(tile_n * BN + r0) * p.k_pad + slot * 8;

    u32x4 ra[4], rb[BROWS];

    auto fetch = [&](int kt) {
        const bool k_ok = kk < k_total;
        const T* src; int cs, cc;
        if (kc < p.c0) { src = reinterpret_cast<const T*>(p.a0); cs = p.c0; cc = kc; }
        else           { src = reinterpret_cast<const T*>(p.a1); cs = p.c1; cc = kc - p.c0; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int iy = row_iy[i] + kdy, ix = row_ix[i] + kdx;
            const bool ok = row_ok[i] && k_ok && (unsigned)iy < (unsigned)p.h_virt && (unsigned)ix < (unsigned)p.w_virt;
            int sy = iy, sx = ix;
            if (resize) { sy = (iy * p.h_in) / p.h_virt; sx = (ix * p.w_in) / p.w_virt; }
            const int64_t pix = ((int64_t)row_img[i] * p.h_in + sy) * p.w_in + sx;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (ok) v = *reinterpret_cast<const u32x4*>(src + pix * cs + cc);
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < BROWS; ++i)
            rb[i] = *reinterpret_cast<const u32x4*>(wbase + (int64_t)(32 * i) * p.k_pad + (int64_t)kt * CG_BK);
        // advance the k decomposition by one K step
        kk += CG_BK; kc += CG_BK;
        while (kc >= ctot) { kc -= ctot; if (++kdx == p.kw) { kdx = 0; ++kdy; } }
    };
    auto stash = [&](int buf) {
        T* a = sA + buf * CG_BM * CG_LDS;
        T* b = sB + buf * BN * CG_LDS;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(a + (r0 + 32 * i) * CG_LDS + slot * 8) = ra[i];
#pragma unroll
        for (int i = 0; i < BROWS; ++i) *reinterpret_cast<u32x4*>(b + (r0 + 32 * i) * CG_LDS + slot * 8) = rb[i];
    };

    f32x16 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    fetch(0);
    stash(0);
    __syncthreads();

    const int frow = lane & 31, fk = (lane >> 5) * 8;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) fetch(kt + 1);
        const T* a = sA + cur * CG_BM * CG_LDS + (wm * 64 + frow) * CG_LDS + fk;
        const T* b = sB + cur * BN * CG_LDS + (wn * (BN / 2) + frow) * CG_LDS + fk;
#pragma unroll
        for (int ks = 0; ks < CG_BK / 16; ++ks) {
            u32x4 fa[2], fb[NT];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const u32x4*>(a + i * 32 * CG_LDS + ks * 16);
#pragma unroll
            for (int j = 0; j < NT; ++j) fb[j] = *reinterpret_cast<const u32x4*>(b + j * 32 * CG_LDS + ks * 16);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = mfma_32x32x16(T(), fa[i], fb[j], acc[i][j]);
        }
        if (kt + 1 < nk) stash(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue ----
    const T* bias = reinterpret_cast<const T*>(p.bias);
    const T* rowvec = reinterpret_cast<const T*>(p.rowvec);
    const T* resid = reinterpret_cast<const T*>(p.residual);
    const int col_l = lane & 31;
    const int row_l = 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = tile_n * BN + wn * (BN / 2) + j * 32 + col_l;
        const bool c_ok = n < p.n_out;
        const float bcol = (bias && c_ok && !p.bias_per_row) ? (float)bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = tile_m * CG_BM + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + row_l;
                if (m < M && c_ok) {
                    float v = acc[i][j][e] + bcol;
                    if (p.bias_per_row && bias) v += (float)bias[m];
                    if (rowvec) v += (float)rowvec[(int64_t)(m / p.rowvec_div) * (p.rowvec_ld ? p.rowvec_ld : p.n_out) + n];
                    if (p.act == AA_ACT_SILU) v = silu_f(v);
                    if (p.acc_scale != 0.0f) v *= p.acc_scale;
                    if (resid) v += (float)resid[(int64_t)m * p.ldr + n];
                    store_out<T>(p.out, p.out_dtype, (int64_t)m * p.ldo + n, v * p.out_scale);
                }
            }
    }
}

}  // namespace aa
