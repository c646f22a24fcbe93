// HBM-bound normalisation / elementwise kernels for gfx950 (see include/aa_mi355.h):
// GroupNorm(+SiLU) over channels-last tokens (2-D and clip-wide 3-D statistics, optional two-source
// channel concat), LayerNorm, row softmax, fused CFG + DPM-Solver++ update.
// All global traffic is 16 B per lane (8 half-precision channels), statistics in fp32, cross-lane
// sums by wave64 xor-shuffles, cross-wave sums through LDS.
#pragma once
#include "dev.h"
#include "aa_mi355.h"

namespace aa {

constexpr int GN_THREADS = 256;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += wave_shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, wave_shfl_xor(v, m));
    return v;
}

template <typename T>
__device__ __forceinline__ u32x4 load8(const T* x0, const T* x1, int c0, int c1, int64_t token, int c) {
    const T* src = (c < c0) ? x0 + token * c0 + c : x1 + token * c1 + (c - c0);
    return *reinterpret_cast<const u32x4*>(src);
}
// GroupNorm statistics are accumulated on values CENTRED on a per-(image group, channel group) pivot - the group's first
// element (first token, first channel) - so that var = E[(x-K)^2] - E[x-K]^2 does not cancel when |mean| >> std
// (activations of trained checkpoints are not zero-mean; torch's own kernel uses Welford for the same reason).
template <typename T>
__device__ __forceinline__ float gn_pivot(const T* x0, const T* x1, int c0, int c1, int64_t first_token, int c) {
    return (float)((c < c0) ? x0[first_token * c0 + c] : x1[first_token * c1 + (c - c0)]);
}

// ---- GroupNorm pass 1: per (image group, chunk) partial sums of x and x^2 for every channel group.
// grid = (chunks, n_groups_img).  partial layout: [img_group][chunk][group][2] fp32.
template <typename T>
__global__ void __launch_bounds__(GN_THREADS) groupnorm_stats_kernel(const AaGroupNorm p, float* partial, int chunks) {
    float* s_sum = reinterpret_cast<float*>(dyn_smem());      // [C]
    const int C = p.c0 + p.c1;
    float* s_sq = s_sum + C;
    const int tid = threadIdx.x;
    const int S = C >> 3;                                       // 16-byte slots per token
    const int ig = blockIdx.y, chunk = blockIdx.x;
    const int per = (p.tokens_per_group + chunks - 1) / chunks;
    const int t_begin = chunk * per;
    const int t_end = min(p.tokens_per_group, t_begin + per);
    const int64_t base = (int64_t)ig * p.tokens_per_group;
    const T* x0 = reinterpret_cast<const T*>(p.x0);
    const T* x1 = reinterpret_cast<const T*>(p.x1);

    // Per-thread channel sums are parked in LDS as [row lane][channel] and folded in a fixed order
    // (no floating-point atomics: results are bit-reproducible run to run).
    const int rows_per_pass = S <= GN_THREADS ? GN_THREADS / S : 1;
    float* s_part = s_sq + C;                                   // [rows_per_pass][2][C]
    for (int s0 = 0; s0 < S; s0 += GN_THREADS) {
        const int lin = tid;
        const int slot = s0 + (S <= GN_THREADS ? lin % S : lin);
        const int roff = S <= GN_THREADS ? lin / S : 0;
        const bool active = slot < S && roff < rows_per_pass;
        float a[8], b[8], kp[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { a[e] = 0.0f; b[e] = 0.0f; kp[e] = 0.0f; }
        if (active) {
            const int cg = C / p.num_groups;
#pragma unroll
            for (int e = 0; e < 8; ++e) kp[e] = gn_pivot<T>(x0, x1, p.c0, p.c1, base, ((slot * 8 + e) / cg) * cg);
            int t = t_begin + roff;
            for (; t + 3 * rows_per_pass < t_end; t += 4 * rows_per_pass) {  // four independent rows in flight per thread
                Pack8<T> v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u].raw = load8<T>(x0, x1, p.c0, p.c1, base + t + u * rows_per_pass, slot * 8);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float f = (float)v[u].e[e] - kp[e]; a[e] += f; b[e] += f * f; }
            }
            for (; t + rows_per_pass < t_end; t += 2 * rows_per_pass) {
                Pack8<T> v, w;
                v.raw = load8<T>(x0, x1, p.c0, p.c1, base + t, slot * 8);
                w.raw = load8<T>(x0, x1, p.c0, p.c1, base + t + rows_per_pass, slot * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = (float)v.e[e] - kp[e], g = (float)w.e[e] - kp[e];
                    a[e] += f + g; b[e] += f * f + g * g;
                }
            }
            if (t < t_end) {
                Pack8<T> v; v.raw = load8<T>(x0, x1, p.c0, p.c1, base + t, slot * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float f = (float)v.e[e] - kp[e]; a[e] += f; b[e] += f * f; }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s_part[(roff * 2 + 0) * C + slot * 8 + e] = a[e];
                s_part[(roff * 2 + 1) * C + slot * 8 + e] = b[e];
            }
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += GN_THREADS) {
        float a = 0.0f, b = 0.0f;
        for (int rr = 0; rr < rows_per_pass; ++rr) { a += s_part[(rr * 2 + 0) * C + c]; b += s_part[(rr * 2 + 1) * C + c]; }
        s_sum[c] = a; s_sq[c] = b;
    }
    __syncthreads();
    if (tid < p.num_groups) {
        const int cg = C / p.num_groups;
        float a = 0.0f, b = 0.0f;
        for (int c = tid * cg; c < (tid + 1) * cg; ++c) { a += s_sum[c]; b += s_sq[c]; }
        float* dst = partial + (((int64_t)ig * chunks + chunk) * p.num_groups + tid) * 2;
        dst[0] = a; dst[1] = b;
    }
}

// ---- GroupNorm pass 2: finish the statistics (every block redoes the tiny reduction of the partials
// of its image group), fold them with gamma/beta into per-channel scale/shift in LDS, normalise.
// grid = (apply_chunks, n_groups_img).
template <typename T>
__global__ void __launch_bounds__(GN_THREADS) groupnorm_apply_kernel(const AaGroupNorm p, const float* partial, int chunks, int apply_chunks) {
    const int C = p.c0 + p.c1;
    float* s_scale = reinterpret_cast<float*>(dyn_smem());     // [C]
    float* s_shift = s_scale + C;                               // [C]
    float* s_red = s_shift + C;                                 // [8][num_groups][2] then [num_groups][2]
    const int tid = threadIdx.x;
    const int ig = blockIdx.y;
    const int G = p.num_groups;

    {   // reduce partial[ig][*][g][*]: thread -> (group g, chunk lane j of GN_THREADS/G)
        const int lanes = GN_THREADS / G;
        const int g = tid % G, j = tid / G;
        float a = 0.0f, b = 0.0f;
        if (j < lanes)
            for (int ch = j; ch < chunks; ch += lanes) {
                const float* src = partial + (((int64_t)ig * chunks + ch) * G + g) * 2;
                a += src[0]; b += src[1];
            }
        if (j < lanes) { s_red[(j * G + g) * 2] = a; s_red[(j * G + g) * 2 + 1] = b; }
        __syncthreads();
        if (tid < G) {
            float sa = 0.0f, sb = 0.0f;
            for (int jj = 0; jj < lanes; ++jj) { sa += s_red[(jj * G + tid) * 2]; sb += s_red[(jj * G + tid) * 2 + 1]; }
            const float cnt = (float)p.tokens_per_group * (float)(C / G);
            const float dm = sa / cnt;                                       // mean of the pivot-centred values
            const float var = fmaxf(sb / cnt - dm * dm, 0.0f);
            const float mean = dm + gn_pivot<T>(reinterpret_cast<const T*>(p.x0), reinterpret_cast<const T*>(p.x1), p.c0, p.c1,
                                                (int64_t)ig * p.tokens_per_group, tid * (C / G));
            s_red[(lanes * G + tid) * 2] = mean;
            s_red[(lanes * G + tid) * 2 + 1] = rsqrtf(var + p.eps);
        }
        __syncthreads();
        const T* gamma = reinterpret_cast<const T*>(p.gamma);
        const T* beta = reinterpret_cast<const T*>(p.beta);
        const int cg = C / G;
        for (int c = tid; c < C; c += GN_THREADS) {
            const int gg = c / cg;
            const float mean = s_red[(lanes * G + gg) * 2], rstd = s_red[(lanes * G + gg) * 2 + 1];
            const float sc = rstd * (float)gamma[c];
            s_scale[c] = sc;
            s_shift[c] = (float)beta[c] - mean * sc;
        }
        __syncthreads();
    }

    const int S = C >> 3;
    const int per = (p.tokens_per_group + apply_chunks - 1) / apply_chunks;
    const int t_begin = blockIdx.x * per;
    const int t_end = min(p.tokens_per_group, t_begin + per);
    const int64_t base = (int64_t)ig * p.tokens_per_group;
    const T* x0 = reinterpret_cast<const T*>(p.x0);
    const T* x1 = reinterpret_cast<const T*>(p.x1);
    T* y = reinterpret_cast<T*>(p.y);
    // thread -> fixed 16-byte channel slot (its scale/shift stay in registers), rows strided: no div/mod in
    // the streaming loop, two independent rows in flight per iteration
    const int rows_per_pass = S <= GN_THREADS ? GN_THREADS / S : 1;
    for (int s0 = 0; s0 < S; s0 += GN_THREADS) {
        const int slot = s0 + (S <= GN_THREADS ? tid % S : tid);
        const int roff = S <= GN_THREADS ? tid / S : 0;
        if (slot >= S || roff >= rows_per_pass) continue;
        float sc[8], sh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = s_scale[slot * 8 + e]; sh[e] = s_shift[slot * 8 + e]; }
        const bool silu = p.silu != 0;
        auto norm8 = [&](Pack8<T> v) {
            Pack8<T> o;
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = (float)v.e[e] * sc[e] + sh[e];
            if (silu) {                                   // one uniform branch per eight values, not one per value
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = f[e] / (1.0f + __expf(-f[e]));
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) o.e[e] = (T)f[e];
            return o;
        };
        int t = t_begin + roff;
        for (; t + 3 * rows_per_pass < t_end; t += 4 * rows_per_pass) {          // four independent rows in flight per thread
            Pack8<T> v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u].raw = load8<T>(x0, x1, p.c0, p.c1, base + t + u * rows_per_pass, slot * 8);
#pragma unroll
            for (int u = 0; u < 4; ++u) *reinterpret_cast<u32x4*>(y + (base + t + u * rows_per_pass) * C + slot * 8) = norm8(v[u]).raw;
        }
        for (; t + rows_per_pass < t_end; t += 2 * rows_per_pass) {
            Pack8<T> va, vb;
            va.raw = load8<T>(x0, x1, p.c0, p.c1, base + t, slot * 8);
            vb.raw = load8<T>(x0, x1, p.c0, p.c1, base + t + rows_per_pass, slot * 8);
            *reinterpret_cast<u32x4*>(y + (base + t) * C + slot * 8) = norm8(va).raw;
            *reinterpret_cast<u32x4*>(y + (base + t + rows_per_pass) * C + slot * 8) = norm8(vb).raw;
        }
        if (t < t_end) {
            Pack8<T> va; va.raw = load8<T>(x0, x1, p.c0, p.c1, base + t, slot * 8);
            *reinterpret_cast<u32x4*>(y + (base + t) * C + slot * 8) = norm8(va).raw;
        }
    }
}

// ---- GroupNorm pass 2 WITHOUT the normalisation: the per-channel (scale, shift) of every image group as fp32 [img group][2][C] for a consumer
// that applies them to the rows it holds anyway (aa_linear_rows: `row_affine`) - the arithmetic of groupnorm_apply_kernel's first block.
// grid = n_groups_img.
template <typename T>
__global__ void __launch_bounds__(GN_THREADS) groupnorm_coef_kernel(const AaGroupNorm p, const float* partial, int chunks, float* coef) {
    const int C = p.c0 + p.c1;
    float* s_red = reinterpret_cast<float*>(dyn_smem());      // [lanes][num_groups][2] then [num_groups][2]
    const int tid = threadIdx.x;
    const int ig = blockIdx.x;
    const int G = p.num_groups;
    const int lanes = GN_THREADS / G;
    const int g = tid % G, j = tid / G;
    float a = 0.0f, b = 0.0f;
    if (j < lanes)
        for (int ch = j; ch < chunks; ch += lanes) {
            const float* src = partial + (((int64_t)ig * chunks + ch) * G + g) * 2;
            a += src[0]; b += src[1];
        }
    if (j < lanes) { s_red[(j * G + g) * 2] = a; s_red[(j * G + g) * 2 + 1] = b; }
    __syncthreads();
    if (tid < G) {
        float sa = 0.0f, sb = 0.0f;
        for (int jj = 0; jj < lanes; ++jj) { sa += s_red[(jj * G + tid) * 2]; sb += s_red[(jj * G + tid) * 2 + 1]; }
        const float cnt = (float)p.tokens_per_group * (float)(C / G);
        const float dm = sa / cnt;
        const float var = fmaxf(sb / cnt - dm * dm, 0.0f);
        const float mean = dm + gn_pivot<T>(reinterpret_cast<const T*>(p.x0), reinterpret_cast<const T*>(p.x1), p.c0, p.c1,
                                            (int64_t)ig * p.tokens_per_group, tid * (C / G));
        s_red[(lanes * G + tid) * 2] = mean;
        s_red[(lanes * G + tid) * 2 + 1] = rsqrtf(var + p.eps);
    }
    __syncthreads();
    const T* gamma = reinterpret_cast<const T*>(p.gamma);
    const T* beta = reinterpret_cast<const T*>(p.beta);
    const int cg = C / G;
    for (int c = tid; c < C; c += GN_THREADS) {
        const int gg = c / cg;
        const float mean = s_red[(lanes * G + gg) * 2], rstd = s_red[(lanes * G + gg) * 2 + 1];
        const float sc = rstd * (float)gamma[c];
        coef[((int64_t)ig * 2 + 0) * C + c] = sc;
        coef[((int64_t)ig * 2 + 1) * C + c] = (float)beta[c] - mean * sc;
    }
}

// ---- GroupNorm in ONE pass over HBM: a workgroup owns `GB` channel groups (CW = GB * C/num_groups channels: a row segment of
// CW * 2 bytes) of ALL tokens of one image group (at most rpp * NR) and keeps them in REGISTERS (NR 16-byte pieces per
// thread; a CU's register file holds 512 KB) between the statistics and the normalisation: x is read once and y written
// once, one launch instead of two.
// Only where ONE workgroup can hold every token of its image group (the per-frame norms of the 16x16 / 8x8 levels, some of the
// 32x32 ones): the statistics are then complete inside the workgroup.  (Measured and dropped, r03q: several workgroups per
// image group meeting at a global counter - clip-wide norms, the 64x64 level - cost 30+ us per rendezvous across the XCDs:
// 2-3x slower than the two-kernel pair.)
// grid = (1, num_groups / GB, n_groups_img).  Statistics centred on the group's first element, as above.
constexpr int GNF_THREADS = 512;
__host__ __device__ inline int gnf_lds_bytes(int cw, int gb) { return (GNF_THREADS / (cw / 8)) * 2 * cw * 4 + 4 * cw * 4 + 4 * gb * 4; }

template <typename T, int NR>
__global__ void __launch_bounds__(GNF_THREADS, NR <= 16 ? 4 : 2) groupnorm_fused_kernel(const AaGroupNorm p, const int GB) {
    const int C = p.c0 + p.c1, cg = C / p.num_groups;
    const int CW = GB * cg, SW = CW >> 3, rpp = GNF_THREADS / SW;
    const int tid = threadIdx.x;
    const int slot = tid % SW, roff = tid / SW;
    const bool active = roff < rpp;
    const int gb = blockIdx.y, ig = blockIdx.z;
    const int64_t base = (int64_t)ig * p.tokens_per_group;
    const int cbase = gb * CW + slot * 8;
    const T* x0 = reinterpret_cast<const T*>(p.x0);
    const T* x1 = reinterpret_cast<const T*>(p.x1);
    float* s_part = reinterpret_cast<float*>(dyn_smem());        // [rpp][2][CW]
    float* s_csum = s_part + rpp * 2 * CW;                        // [2][CW]
    float* s_scale = s_csum + 2 * CW;                             // [CW]
    float* s_shift = s_scale + CW;                                // [CW]
    float* s_stat = s_shift + CW;                                 // [GB][2] sums, then mean / rstd

    Pack8<T> v[NR];
    float a[8], b[8], kp[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = 0.0f; b[e] = 0.0f; kp[e] = 0.0f; }
    if (active) {
#pragma unroll
        for (int e = 0; e < 8; ++e) kp[e] = gn_pivot<T>(x0, x1, p.c0, p.c1, base, ((cbase + e) / cg) * cg);
#pragma unroll
        for (int r = 0; r < NR; ++r) {                            // every piece of this thread is in flight before the first use
            const int t = roff + r * rpp;
            v[r].raw = u32x4{0u, 0u, 0u, 0u};
            if (t < p.tokens_per_group) v[r].raw = load8<T>(x0, x1, p.c0, p.c1, base + t, cbase);
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int t = roff + r * rpp;
            if (t < p.tokens_per_group) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float f = (float)v[r].e[e] - kp[e]; a[e] += f; b[e] += f * f; }
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s_part[(roff * 2 + 0) * CW + slot * 8 + e] = a[e];
            s_part[(roff * 2 + 1) * CW + slot * 8 + e] = b[e];
        }
    }
    __syncthreads();
    for (int c = tid; c < 2 * CW; c += GNF_THREADS) {            // fixed summation order: bit-reproducible
        const int which = c / CW, cc = c - which * CW;
        float acc = 0.0f;
        for (int rr = 0; rr < rpp; ++rr) acc += s_part[(rr * 2 + which) * CW + cc];
        s_csum[c] = acc;
    }
    __syncthreads();
    if (tid < GB) {
        float sa = 0.0f, sb = 0.0f;
        for (int c = tid * cg; c < (tid + 1) * cg; ++c) { sa += s_csum[c]; sb += s_csum[CW + c]; }
        s_stat[2 * tid] = sa; s_stat[2 * tid + 1] = sb;
    }
    __syncthreads();
    if (tid < GB) {
        const float cnt = (float)p.tokens_per_group * (float)cg;
        const float dm = s_stat[2 * tid] / cnt;
        const float var = fmaxf(s_stat[2 * tid + 1] / cnt - dm * dm, 0.0f);
        const float mean = dm + gn_pivot<T>(x0, x1, p.c0, p.c1, base, (gb * GB + tid) * cg);
        s_stat[2 * GB + 2 * tid] = mean;
        s_stat[2 * GB + 2 * tid + 1] = rsqrtf(var + p.eps);
    }
    __syncthreads();
    {
        const T* gamma = reinterpret_cast<const T*>(p.gamma);
        const T* beta = reinterpret_cast<const T*>(p.beta);
        for (int c = tid; c < CW; c += GNF_THREADS) {
            const int g = c / cg;
            const float mean = s_stat[2 * GB + 2 * g], rstd = s_stat[2 * GB + 2 * g + 1];
            const float sc = rstd * (float)gamma[gb * CW + c];
            s_scale[c] = sc;
            s_shift[c] = (float)beta[gb * CW + c] - mean * sc;
        }
    }
    __syncthreads();
    if (!active) return;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = s_scale[slot * 8 + e]; sh[e] = s_shift[slot * 8 + e]; }
    const bool silu = p.silu != 0;
    T* y = reinterpret_cast<T*>(p.y);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int t = roff + r * rpp;
        if (t < p.tokens_per_group) {
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = (float)v[r].e[e] * sc[e] + sh[e];
            if (silu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = f[e] / (1.0f + __expf(-f[e]));
            }
            Pack8<T> o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o.e[e] = (T)f[e];
            *reinterpret_cast<u32x4*>(y + (base + t) * C + cbase) = o.raw;
        }
    }
}

// ---- LayerNorm: a wavefront owns LN_ROWS<SJ> rows at a time, the rows live in registers (C <= 2048), two-pass variance.
// LayerNorm fold, between producer and consumer: the partial (sum, sum of squares) pairs a contraction left per output row
// (AaConvGemm.row_stats) -> per row (-mean, sqrt(var + eps), rstd, 0): what the consuming contraction starts its accumulators from
// and scales by (AaConvGemm.ln_stats).  One thread per row; 8 * parts + 16 bytes per row.
template <int UNIT = 0>          // (a template only so that the header may be included by several translation units)
__global__ void __launch_bounds__(256) ln_finalize_kernel(const float* stats, float* coef, int64_t rows, int parts, float inv_c, float eps) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= rows) return;
    const float* st = stats + m * parts * 2;
    float s = 0.0f, q = 0.0f;
    for (int t = 0; t < parts; ++t) { s += st[2 * t]; q += st[2 * t + 1]; }
    const float mean = s * inv_c;
    const float var = fmaxf(q * inv_c - mean * mean, 0.0f);
    const float sd = sqrtf(var + eps);
    *reinterpret_cast<f32x4*>(coef + m * 4) = f32x4{-mean, sd, 1.0f / sd, 0.0f};
}

// SJ = 16-byte slots per lane and row (C <= 512 * SJ); narrow rows are processed several at once so that every wave
// keeps four independent 16-byte loads in flight (a 640-byte row per wave leaves HBM latency exposed).
template <typename T, int SJ>
__global__ void __launch_bounds__(256) layernorm_kernel(const T* x, const T* gamma, const T* beta, T* y,
                                                       int64_t rows, int C, float eps) {
    constexpr int R = 4 / SJ;                               // rows per wave
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    if (row0 >= rows) return;
    const int S = C >> 3;
    Pack8<T> v[R][SJ];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int j = 0; j < SJ; ++j) {
            const int slot = lane + 64 * j;
            v[r][j].raw = u32x4{0u, 0u, 0u, 0u};
            if (slot < S && row0 + r < rows) v[r][j].raw = *reinterpret_cast<const u32x4*>(x + (row0 + r) * C + slot * 8);
        }
    Pack8<T> g[SJ], b[SJ];
#pragma unroll
    for (int j = 0; j < SJ; ++j) {
        const int slot = lane + 64 * j;
        g[j].raw = u32x4{0u, 0u, 0u, 0u}; b[j].raw = u32x4{0u, 0u, 0u, 0u};
        if (slot < S) {
            g[j].raw = *reinterpret_cast<const u32x4*>(gamma + slot * 8);
            b[j].raw = *reinterpret_cast<const u32x4*>(beta + slot * 8);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (row0 + r >= rows) break;
        float sum = 0.0f;
#pragma unroll
        for (int j = 0; j < SJ; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += (float)v[r][j].e[e];          // (slots past the row hold zeros)
        const float mean = wave_sum(sum) / (float)C;
        float sq = 0.0f;
#pragma unroll
        for (int j = 0; j < SJ; ++j)
            if (lane + 64 * j < S) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = (float)v[r][j].e[e] - mean; sq += d * d; }
            }
        const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
#pragma unroll
        for (int j = 0; j < SJ; ++j) {
            const int slot = lane + 64 * j;
            if (slot < S) {
                Pack8<T> o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o.e[e] = (T)(((float)v[r][j].e[e] - mean) * rstd * (float)g[j].e[e] + (float)b[j].e[e]);
                *reinterpret_cast<u32x4*>(y + (row0 + r) * C + slot * 8) = o.raw;
            }
        }
    }
}

// ---- row softmax of fp32 scores (VAE single-head attention), one workgroup per row.
template <typename T>
__global__ void __launch_bounds__(256) softmax_rows_kernel(const float* x, T* y, int cols, int x_ld, int y_ld) {
    float* s_red = reinterpret_cast<float*>(dyn_smem());   // [8]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xr = x + (int64_t)blockIdx.x * x_ld;
    T* yr = y + (int64_t)blockIdx.x * y_ld;
    float mx = -3.0e38f;
    for (int c = tid; c < cols; c += 256) mx = fmaxf(mx, xr[c]);
    mx = wave_max(mx);
    if (lane == 0) s_red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    float sum = 0.0f;
    for (int c = tid; c < cols; c += 256) sum += __expf(xr[c] - mx);
    sum = wave_sum(sum);
    if (lane == 0) s_red[4 + wave] = sum;
    __syncthreads();
    const float inv = 1.0f / (s_red[4] + s_red[5] + s_red[6] + s_red[7]);
    for (int c = tid; c < cols; c += 256) yr[c] = (T)(__expf(xr[c] - mx) * inv);
}

// ---- fused classifier-free guidance + DPM-Solver++(2M) update (elementwise).
template <typename T>
__global__ void __launch_bounds__(256) cfg_dpm_step_kernel(const AaDpmStep p) {
    const T* eu = reinterpret_cast<const T*>(p.eps_uncond);
    const T* et = reinterpret_cast<const T*>(p.eps_text);
    float* x = reinterpret_cast<float*>(p.latents);
    float* x0p = reinterpret_cast<float*>(p.x0_prev);
    T* lp = reinterpret_cast<T*>(p.latents_lp);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.n; i += (int64_t)gridDim.x * 256) {
        const float u = (float)eu[i];
        const float eps = u + p.guidance * ((float)et[i] - u);
        const float xi = x[i];
        const float x0 = (xi - p.sigma_s * eps) / p.alpha_s;
        const float nx = p.c_x * xi - p.c_d0 * x0 - p.c_d1 * (x0 - x0p[i]);
        x[i] = nx;
        x0p[i] = x0;
        lp[i] = (T)nx;
    }
}

}  // namespace aa
