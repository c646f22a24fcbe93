// Flash-style attention for gfx950, head_dim 64 (see include/aa_mi355.h: aa_attention).
//
// One wavefront owns 32 query rows of one (sequence, head); NW wavefronts of a workgroup share each
// 64-key K/V tile through LDS (double buffered, one barrier per tile, next tile's global loads in flight
// during the current tile's math).  The score tile is computed TRANSPOSED, S^T = K Q^T, so that in the
// 32x32 MFMA result layout (col = lane&31) every lane owns one query row: the online-softmax max and
// sum are register-local (one xor-32 shuffle joins the two half-waves), and the exponentiated
// registers are, unchanged, the B operand of O^T = V^T P^T.
// V is written to LDS transposed, [d][key], with (a) the keys permuted inside each 16-key chunk (quads 1
// and 2 swapped) so the matching A operand is a single ds_read_b128, and (b) the 8-key chunks of row d
// XOR-ed with x(d) = ((d>>3) + 2*(d&7)) & 7, which makes both the 2-byte transposing stores and the
// 16-byte fragment reads bank-conflict free on the unpadded 128-byte rows.
// Softmax runs in base 2: p = exp2(s*c - m*c), c = scale*log2(e), max taken on raw scores.
#pragma once
#include "dev.h"
#include "aa_mi355.h"

namespace aa {

constexpr int AT_KT = 64;       // keys per tile
constexpr int AT_LDK = 72;      // padded K row (elements)
constexpr int AT_LDV = 64;      // V^T row (elements), XOR-swizzled instead of padded
constexpr int AT_BUF = AT_KT * AT_LDK + 64 * AT_LDV;    // elements per (K, V^T) buffer pair

// one (K, V^T) buffer when the sequence fits a single tile (temporal / text attention), else two
__host__ __device__ inline int attn_lds_bytes(int kv_len) { return (kv_len > AT_KT ? 2 : 1) * AT_BUF * 2; }

template <typename T>
__device__ __forceinline__ const T* attn_row(const AaAttnOperand& x, int o, int i, int pos, int head) {
    const int64_t row = (int64_t)(o / x.outer_div) * x.outer_stride + (int64_t)i * x.inner_stride + (int64_t)pos * x.pos_stride;
    return reinterpret_cast<const T*>(x.ptr) + row * x.ld + x.col0 + head * 64;
}

__device__ __forceinline__ int vt_swz(int d) { return ((d >> 3) + 2 * (d & 7)) & 7; }

template <typename T, int NW>
__global__ void __launch_bounds__(64 * NW, NW > 1 ? 3 : 2) attention_kernel(const AaAttention p) {
    constexpr int THREADS = 64 * NW;
    constexpr int SLOTS = (AT_KT * 8) / THREADS;     // 16-byte K (and V) slots staged per thread
    T* lds = reinterpret_cast<T*>(dyn_smem());

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, ql = lane & 31;
    const int head = blockIdx.y;
    const int seq = blockIdx.z;
    const int o = seq / p.n_inner, i = seq - o * p.n_inner;
    const int q0 = (blockIdx.x * NW + wave) * 32;
    const bool wave_active = q0 < p.q_len;
    const float sl2e = p.scale * 1.4426950408889634f;

    // Q fragment: B operand of S^T (col = query, k = d)
    u32x4 qf[4];
    {
        const int q = q0 + ql;
        const bool ok = q < p.q_len;
        const T* src = attn_row<T>(p.q, o, i, ok ? q : 0, head) + 8 * h;
#pragma unroll
        for (int dk = 0; dk < 4; ++dk) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (ok) v = *reinterpret_cast<const u32x4*>(src + 16 * dk);
            qf[dk] = v;
        }
    }

    // per-slot K / V source pointers of tile 0, advanced by one tile (64 keys) per fetch: no 64-bit row
    // arithmetic in the loop
    u32x4 rk[SLOTS], rv[SLOTS];
    const T* kptr[SLOTS];
    const T* vptr[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int sl = tid + s * THREADS;
        kptr[s] = attn_row<T>(p.k, o, i, sl >> 3, head) + (sl & 7) * 8;
        vptr[s] = attn_row<T>(p.v, o, i, sl >> 3, head) + (sl & 7) * 8;
    }
    const int64_t k_step = (int64_t)AT_KT * p.k.pos_stride * p.k.ld, v_step = (int64_t)AT_KT * p.v.pos_stride * p.v.ld;
    auto fetch = [&](int kt) {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int sl = tid + s * THREADS;
            const bool ok = kt * AT_KT + (sl >> 3) < p.kv_len;
            u32x4 a = {0u, 0u, 0u, 0u}, b = {0u, 0u, 0u, 0u};
            if (ok) {
                a = *reinterpret_cast<const u32x4*>(kptr[s]);
                b = *reinterpret_cast<const u32x4*>(vptr[s]);
            }
            rk[s] = a; rv[s] = b;
            kptr[s] += k_step; vptr[s] += v_step;
        }
    };
    auto stash = [&](int buf) {
        T* sK = lds + buf * AT_BUF;
        T* sVt = sK + AT_KT * AT_LDK;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int sl = tid + s * THREADS;
            const int key = sl >> 3, dseg = sl & 7;
            *reinterpret_cast<u32x4*>(sK + key * AT_LDK + dseg * 8) = rk[s];
            const int quad = (key >> 2) & 3;
            const int pos = (key & ~15) | ((((quad & 1) << 1) | (quad >> 1)) << 2) | (key & 3);
            Pack8<T> v; v.raw = rv[s];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int d = dseg * 8 + e;
                sVt[d * AT_LDV + (pos ^ (vt_swz(d) << 3))] = v.e[e];
            }
        }
    };

    f32x16 oacc[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { oacc[0][e] = 0.0f; oacc[1][e] = 0.0f; }
    float m_run = -1.0e30f, l_run = 0.0f;       // running max of RAW scores, running sum of this half-wave's keys

    const int ntiles = (p.kv_len + AT_KT - 1) / AT_KT;
    const bool ragged = (p.kv_len & (AT_KT - 1)) != 0;
    fetch(0);
    stash(0);
    __syncthreads();
    constexpr bool PREFETCH = NW > 1;    // one-wave workgroups stage 8 slots per thread: keep those registers free
    for (int kt = 0; kt < ntiles; ++kt) {
        if (PREFETCH && kt + 1 < ntiles) fetch(kt + 1);
        if (wave_active) {
            const T* sK = lds + (kt & 1) * AT_BUF;
            const T* sVt = sK + AT_KT * AT_LDK;
            f32x16 sacc[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int e = 0; e < 16; ++e) sacc[kb][e] = 0.0f;
#pragma unroll
                for (int dk = 0; dk < 4; ++dk) {
                    const u32x4 kf = *reinterpret_cast<const u32x4*>(sK + (32 * kb + ql) * AT_LDK + 16 * dk + 8 * h);
                    sacc[kb] = mfma_32x32x16(T(), kf, qf[dk], sacc[kb]);
                }
            }
            if (ragged && kt == ntiles - 1) {           // mask the keys past kv_len (last tile only)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int key = kt * AT_KT + 32 * kb + (e & 3) + 8 * (e >> 2) + 4 * h;
                        if (key >= p.kv_len) sacc[kb][e] = -1.0e30f;
                    }
            }
            // row max: four independent chains, then a tree (a single 32-long fmax chain is pure latency)
            float mx[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) mx[c] = fmaxf(sacc[c >> 1][8 * (c & 1)], sacc[c >> 1][8 * (c & 1) + 1]);
#pragma unroll
            for (int e = 2; e < 8; ++e)
#pragma unroll
                for (int c = 0; c < 4; ++c) mx[c] = fmaxf(mx[c], sacc[c >> 1][8 * (c & 1) + e]);
            const float mloc = wave_max_halves(fmaxf(fmaxf(mx[0], mx[1]), fmaxf(mx[2], mx[3])));
            const float m_new = fmaxf(m_run, mloc);
            if (wave_any(m_new > m_run)) {              // some row's max moved: rescale the accumulators
                const float alpha = fast_exp2((m_run - m_new) * sl2e);
                l_run *= alpha;
#pragma unroll
                for (int e = 0; e < 16; ++e) { oacc[0][e] *= alpha; oacc[1][e] *= alpha; }
                m_run = m_new;
            }
            const float nmc = -m_run * sl2e;
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 ps2[4] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}};      // packed partial row sums
            u32x4 pf[4];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    Pack8<T> pk;
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        f32x2 pe;
                        pe[0] = fast_exp2(__builtin_fmaf(sacc[kb][8 * c + e], sl2e, nmc));       // one v_fma + one v_exp per score
                        pe[1] = fast_exp2(__builtin_fmaf(sacc[kb][8 * c + e + 1], sl2e, nmc));
                        ps2[2 * kb + c] += pe;
                        pk.e[e] = (T)pe[0];
                        pk.e[e + 1] = (T)pe[1];
                    }
                    pf[2 * kb + c] = pk.raw;
                }
            const f32x2 pst = (ps2[0] + ps2[1]) + (ps2[2] + ps2[3]);
            const float psum = pst[0] + pst[1];
            l_run += psum;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) {
                    const int d = 32 * db + ql;
                    const u32x4 vf = *reinterpret_cast<const u32x4*>(sVt + d * AT_LDV + (((2 * ch + h) ^ vt_swz(d)) << 3));
                    oacc[db] = mfma_32x32x16(T(), vf, pf[ch], oacc[db]);
                }
        }
        if (kt + 1 < ntiles) {
            if (!PREFETCH) fetch(kt + 1);
            stash((kt + 1) & 1);
        }
        __syncthreads();
    }

    if (wave_active) {
        const float l_tot = wave_sum_halves(l_run);
        const float inv = 1.0f / l_tot;
        const int q = q0 + ql;
        if (q < p.q_len) {
            T* dst = const_cast<T*>(attn_row<T>(p.o, o, i, q, head));
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    union { u32x2 raw; T e[4]; } pk;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pk.e[e] = (T)(oacc[db][4 * g + e] * inv);
                    *reinterpret_cast<u32x2*>(dst + 32 * db + 8 * g + 4 * h) = pk.raw;
                }
        }
    }
}

}  // namespace aa
