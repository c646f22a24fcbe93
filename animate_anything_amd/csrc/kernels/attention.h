// Flash-style attention for gfx950, head_dim 64 (see include/aa_mi355.h: aa_attention).
//
// One wavefront owns 32 query rows of one (sequence, head); NW wavefronts of a workgroup share each
// 64-key K/V tile through a ring of LDS buffers.  The score tile is computed TRANSPOSED, S^T = K Q^T, so that in
// the 32x32 MFMA result layout (col = lane&31) every lane owns one query row: the online-softmax max and sum are
// register-local (one v_permlane32_swap joins the two half-waves), and the exponentiated registers are,
// unchanged, the B operand of O^T = V^T P^T.
//
// K and V tiles travel HBM/L2 -> LDS with `buffer_load_dwordx4 ... lds` (LDS-DMA: no VGPR staging, no ds_write,
// keys past kv_len are out of range of the descriptor and arrive as zeros).  The DMA deposits lane-linear 16-byte
// pieces, so the layouts are made on the SOURCE side (which piece each lane fetches):
//  * K: row-major [key][64 d], 128-byte rows, the 16-byte d-slots XOR-swizzled with (key>>1)&7 - the A fragment of
//    S^T (row = key, 8 consecutive d) is one conflict-free ds_read_b128;
//  * V: stays ROW-major (no transposing stores): 256-byte blocks [4 keys][32 d], block index (key/4, d/32).  The A
//    fragment of O^T needs, for row d, 8 consecutive keys - a column of V - which gfx950's transposing LDS read
//    delivers directly: one `ds_read_b64_tr_b16` hands lane l of a 16-lane group V[k0..k0+3][d0 + l] from a
//    [4 keys][16 d] block; a 32-lane half reads exactly one 256-byte block = all 64 banks once.
// Softmax runs in base 2 on scores that leave the matrix pipe ready to exponentiate: Q is pre-multiplied by
// c = scale*log2(e) when it is loaded, and the S^T accumulators START at -m (the running row maximum) instead of zero, so
// p = exp2(acc) with no multiply / subtract per score.  The maximum is tracked lazily (defer-max): it is only moved - and
// O, l rescaled - when some row's scores exceed it by more than AT_DEFER (p <= 2^AT_DEFER in between: exact in fp32, and
// well inside fp16 / bf16 range for the P operand); the first tile always centres on its true row maximum.
#pragma once
#include "dev.h"
#include "aa_mi355.h"

namespace aa {

#ifndef AA_ATTN_EAGER_MAX
#define AA_ATTN_EAGER_MAX 0
#endif
#ifndef AA_ATTN_SYNC_FRAGS      // 1: A/B build with compiler-issued fragment reads everywhere (the round-2..4 form)
#define AA_ATTN_SYNC_FRAGS 0
#endif
template <int KT> constexpr bool AT_ASYNC_FRAGS() { return KT == 64 && !AA_ATTN_SYNC_FRAGS; }
constexpr int AT_KT = 64;                       // keys per tile
constexpr float AT_DEFER = 6.0f;                // defer-max threshold, in bits (p stays below 2^6 against a stale maximum)
constexpr int AT_TILE_BYTES = 2 * AT_KT * 128;  // K tile + V tile of one stage

// ring depth: one buffer when the sequence fits a single tile (temporal / text attention), else three
__host__ __device__ inline int attn_stages(int kv_len) { return kv_len > AT_KT ? 3 : 1; }
__host__ __device__ inline int attn_lds_bytes(int kv_len) { return attn_stages(kv_len) * AT_TILE_BYTES; }

// first token row of sequence (o, i), in rows of the operand's matrix
// (seq_mod > 0: the operand is a table of seq_mod sequences and sequence number o * n_inner + i reads entry number % seq_mod)
__device__ __forceinline__ int64_t attn_seq_row(const AaAttnOperand& x, int o, int i, int n_inner) {
    if (x.seq_mod > 0) return (int64_t)((o * n_inner + i) % x.seq_mod) * x.outer_stride;
    return (int64_t)(o / x.outer_div) * x.outer_stride + (int64_t)i * x.inner_stride;
}
template <typename T>
__device__ __forceinline__ const T* attn_row(const AaAttnOperand& x, int o, int i, int n_inner, int pos, int head) {
    return reinterpret_cast<const T*>(x.ptr) + (attn_seq_row(x, o, i, n_inner) + (int64_t)pos * x.pos_stride) * x.ld + x.col0 + head * 64;
}
// bytes of the operand an attention call may touch (descriptor range)
__host__ __device__ inline int64_t attn_extent_bytes(const AaAttnOperand& x, int n_outer, int n_inner, int len) {
    const int64_t last = (x.seq_mod > 0 ? (int64_t)(x.seq_mod - 1) * x.outer_stride
                                        : (int64_t)((n_outer - 1) / x.outer_div) * x.outer_stride + (int64_t)(n_inner - 1) * x.inner_stride) +
                         (int64_t)(len - 1) * x.pos_stride;
    return (last + 1) * x.ld * 2;
}

// KT = keys per tile: 64, or 32 for single-tile sequences of at most 32 keys (the T' = 17 temporal attention: half the LDS,
// half the DMA instructions and half the MFMAs of a 64-key tile whose second half would be masked anyway; one-wave
// workgroups at four per SIMD - that kernel is bound by how many independent sequences a CU keeps in flight).
// G > 1 (single-tile sequences only): a workgroup carries G INDEPENDENT sequences, one per wave, each with its own LDS tile - the
// 17-frame temporal attention of the 64x64 level is 40960 one-wave sequences per call, and one-wave workgroups are launched
// more slowly than they finish.
// QB = 32-query blocks per wave.  QB = 2 (round 6 experiment, AaAttention._pad bit 3, NOT the default - measured slower, see aa_api_impl.h):
// a wave owns 64 queries at two waves per SIMD (256 registers).
// Every K fragment feeds both blocks' S^T MFMAs (half the K reads per MFMA), and the blocks are worked ONE AFTER THE OTHER through
// softmax and O^T: block 1's exponentials are issued behind block 0's O^T MFMAs and run while the matrix pipe executes them - with one
// block per wave the pipe idles through every softmax unless another wave happens to be in its matrix phase (35 % busy at three waves
// per SIMD, profiles/r05zz_pmc_step_sq.json).  Per tile and wave: 32 MFMAs for one barrier / DMA issue / loop pass instead of 16.
template <typename T, int NW, int KT = 64, int G = 1, int QB = 1>
__global__ void __launch_bounds__(64 * NW * G, QB == 2 ? 2 : (NW > 1 ? 3 : (KT == 32 ? 4 : 2))) attention_kernel(const AaAttention p) {
    static_assert(QB == 1 || (QB == 2 && KT == 64 && NW == 4 && G == 1), "two query blocks per wave: the multi-tile kernel only");
    constexpr int PER = (KT / 4) / NW;           // DMA instructions per wave and tile (KT/8 for K + KT/8 for V in total)
    constexpr int KB = KT / 32;                  // 32-key blocks per tile
    constexpr int TILE_BYTES = 2 * KT * 128;
    static_assert(KT == 64 || (KT == 32 && NW == 1), "tile");
    static_assert(G == 1 || NW == 1, "independent sequences per wave");
    constexpr unsigned OOB = 0x80000000u;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = G > 1 ? 0 : wave_id();
    const int h = lane >> 5, ql = lane & 31;
    const int head = blockIdx.y;
    const int seq = G > 1 ? blockIdx.z * G + wave_id() : blockIdx.z;
    if (G > 1 && seq >= p.n_outer * p.n_inner) return;
    char* lds = dyn_smem() + (G > 1 ? wave_id() * TILE_BYTES : 0);
    const int o = seq / p.n_inner, i = seq - o * p.n_inner;
    const int q0 = (blockIdx.x * NW + wave) * 32 * QB;
    const bool wave_active = q0 < p.q_len;
    const float sl2e = p.scale * 1.4426950408889634f;   // folded into Q: scores are in base-2 exponent units

    // Q fragment: B operand of S^T (col = query, k = d)
    u32x4 qf[QB][4];
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        const int q = q0 + 32 * b + ql;
        const bool ok = q < p.q_len;
        const T* src = attn_row<T>(p.q, o, i, p.n_inner, ok ? q : 0, head) + 8 * h;
#pragma unroll
        for (int dk = 0; dk < 4; ++dk) {
            Pack8<T> v;
            v.raw = u32x4{0u, 0u, 0u, 0u};
            if (ok) v.raw = *reinterpret_cast<const u32x4*>(src + 16 * dk);
#pragma unroll
            for (int e = 0; e < 8; ++e) v.e[e] = (T)((float)v.e[e] * sl2e);      // scores come out in units of bits
            qf[b][dk] = v.raw;
        }
    }

    // ---- K / V tile DMA: instruction `n` (0..7) of a tile covers keys 8n .. 8n+7 of K (or V); wave w issues
    // instructions w*PER/2 .. of each.  Per lane: key and byte offset inside the row, fixed for the whole kernel.
    const BufRsrc r_k = make_rsrc(p.k.ptr, (unsigned)attn_extent_bytes(p.k, p.n_outer, p.n_inner, p.kv_len));
    const BufRsrc r_v = make_rsrc(p.v.ptr, (unsigned)attn_extent_bytes(p.v, p.n_outer, p.n_inner, p.kv_len));
    const unsigned k_seq = (unsigned)((attn_seq_row(p.k, o, i, p.n_inner) * p.k.ld + p.k.col0 + head * 64) * 2);
    const unsigned v_seq = (unsigned)((attn_seq_row(p.v, o, i, p.n_inner) * p.v.ld + p.v.col0 + head * 64) * 2);
    const unsigned k_key = (unsigned)(p.k.pos_stride * p.k.ld * 2), v_key = (unsigned)(p.v.pos_stride * p.v.ld * 2);   // bytes per key
    // K piece of this lane: key (lane>>3) of the instruction, d-slot (lane&7) ^ swizzle(key row inside the tile)
    const int kk = lane >> 3;
    // V piece: 256-byte block b = lane>>4 -> (key group b>>1, d half b&1); inside: key%4 = (lane>>2)&3, d-slot lane&3
    const int vk = 4 * (lane >> 5) + ((lane >> 2) & 3), vd = 32 * ((lane >> 4) & 1) + 8 * (lane & 3);
    // per-lane offsets of tile 0, advanced by one tile (64 keys) per issue: no multiplies in the loop
    unsigned koff[PER / 2], voff[PER / 2];
    int klocal[PER / 2], vlocal[PER / 2];
#pragma unroll
    for (int j = 0; j < PER / 2; ++j) {
        const int n = wave * (PER / 2) + j;
        klocal[j] = 8 * n + kk;
        vlocal[j] = 8 * n + vk;
        koff[j] = k_seq + (unsigned)klocal[j] * k_key + (unsigned)(((lane & 7) ^ ((klocal[j] >> 1) & 7)) * 16);
        voff[j] = v_seq + (unsigned)vlocal[j] * v_key + (unsigned)(vd * 2);
    }
    int next_tile_key0 = 0;                      // issue() is called for consecutive tiles
    auto issue = [&](int buf) {
        char* sK = lds + buf * TILE_BYTES;
        char* sV = sK + KT * 128;
        const int rem = p.kv_len - next_tile_key0;                        // keys left from this tile on
        next_tile_key0 += KT;
#pragma unroll
        for (int j = 0; j < PER / 2; ++j) {
            async_copy16_buf(r_k, klocal[j] < rem ? koff[j] : OOB, sK + (wave * (PER / 2) + j) * 1024);
            koff[j] += KT * k_key;
        }
#pragma unroll
        for (int j = 0; j < PER / 2; ++j) {
            async_copy16_buf(r_v, vlocal[j] < rem ? voff[j] : OOB, sV + (wave * (PER / 2) + j) * 1024);
            voff[j] += KT * v_key;
        }
    };

    f32x16 oacc[QB][2];
    float m_run[QB], l_run[QB];                 // running (lazily moved) max of the scaled scores, running sum of this half-wave's keys
    // -m_run in every entry: the C operand that starts a score block.  QB == 2: the accumulators start at zero (an inline constant) and the
    // maximum is subtracted in front of the exponential instead - two of these vectors are 32 registers the 64-query wave does not have
    constexpr bool SUB_MAX = QB == 2;
    f32x16 minus_m[SUB_MAX ? 1 : QB];
    f32x16 zero16;
#pragma unroll
    for (int e = 0; e < 16; ++e) { zero16[e] = 0.0f; minus_m[0][e] = 0.0f; }
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        m_run[b] = 0.0f; l_run[b] = 0.0f;
#pragma unroll
        for (int e = 0; e < 16; ++e) { oacc[b][0][e] = 0.0f; oacc[b][1][e] = 0.0f; }
    }

    const bool prio = (p._pad & 1) != 0;
    constexpr bool eager_max = AA_ATTN_EAGER_MAX != 0;
    const int ntiles = (p.kv_len + KT - 1) / KT;
    const bool ragged = (p.kv_len & (KT - 1)) != 0;
    const int stages = attn_stages(p.kv_len);   // 3 (two tiles in flight ahead of the math) or 1 (single tile)
    issue(0);
    if (ntiles > 1) issue(1);
    // fragment read offsets inside a stage (bytes)
    const int kf_row = ql * 128, kf_swz = (ql >> 1) & 7;                  // + kb*32 rows: (32*kb + ql)>>1 & 7 == (ql>>1)&7
    const int vf_off = (2 * h) * 256 + ((lane & 15) >> 2) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    for (int kt = 0; kt < ntiles; ++kt) {
        if (kt + 1 < ntiles) dma_wait<PER>(); else dma_wait<0>();         // tile kt landed (Q fragments are older still)
        if constexpr (G == 1) block_barrier(); else wave_sync();          // everyone's pieces; buffer (kt-1)%3 is free again
        if (kt + 2 < ntiles) issue((kt + 2) % 3);
        if (wave_active) {
            const char* sK = lds + (stages == 1 ? 0 : (kt % 3)) * TILE_BYTES;
            const char* sV = sK + KT * 128;
            // S^T - m: the accumulators START at minus the running maximum (zero for the first tile) - the first MFMA of every
            // block takes a register set that holds -m in all 16 entries as its C operand, so no accumulator is initialised per tile
            f32x16 sacc[QB][KB];
            auto scores = [&]() __attribute__((always_inline)) {
                if (prio) wave_priority<1>();         // (experiment, AaAttention._pad bit 0: matrix clusters above the co-resident waves' softmax)
                // the two 32-key blocks alternate so that consecutive MFMAs never wait on each other's accumulator
                // The tile's K fragments are read by hand, four reads ahead of the MFMA that consumes them (counted lgkmcnt; LDS returns in
                // order).  Left to hipcc the loop was read -> s_waitcnt lgkmcnt(0) -> MFMA, eight times per tile, through one recycled register
                // quad (the kernel sits at its 168-register cap): every LDS latency exposed.  (All eight ahead spilled 48 registers.)
                if constexpr (AT_ASYNC_FRAGS<KT>()) {       // (the single-tile 32-key kernels run four waves per SIMD on 128 registers: compiler-issued reads there)
                constexpr int NK = 4 * KB, WK = NK < 4 ? NK : 4;
                u32x4 kf[NK];
                auto k_read = [&](auto i_) __attribute__((always_inline)) {
                    constexpr int i = decltype(i_)::value, dk = i / KB, kb = i % KB;
                    lds_read16_async_off<kb * 4096>(kf[i], sK + kf_row + (((2 * dk + h) ^ kf_swz) << 4));
                };
                static_for<WK>(k_read);
                static_for<NK>([&](auto i_) __attribute__((always_inline)) {
                    constexpr int i = decltype(i_)::value, dk = i / KB, kb = i % KB;
                    constexpr int issued = (i + WK < NK) ? i + WK : NK;
                    lds_wait<issued - i - 1>(kf[i]);
#pragma unroll
                    for (int b = 0; b < QB; ++b) sacc[b][kb] = mfma_32x32x16(T(), kf[i], qf[b][dk], dk == 0 ? (SUB_MAX ? zero16 : minus_m[SUB_MAX ? 0 : b]) : sacc[b][kb]);
                    if constexpr (i + WK < NK) k_read(IntTag<i + WK>());
                });
                } else {
#pragma unroll
                    for (int dk = 0; dk < 4; ++dk)
#pragma unroll
                        for (int kb = 0; kb < KB; ++kb) {
                            const u32x4 kf = *reinterpret_cast<const u32x4*>(sK + kb * 4096 + kf_row + (((2 * dk + h) ^ kf_swz) << 4));
#pragma unroll
                            for (int b = 0; b < QB; ++b) sacc[b][kb] = mfma_32x32x16(T(), kf, qf[b][dk], dk == 0 ? (SUB_MAX ? zero16 : minus_m[SUB_MAX ? 0 : b]) : sacc[b][kb]);
                        }
                }
                if (prio) wave_priority<0>();
                if (p.causal && kt * KT + KT - 1 > q0) {     // causal: keys after the query's own position (tiles that reach past the wave's first query)
#pragma unroll
                    for (int b = 0; b < QB; ++b) {
                        const int qpos = q0 + 32 * b + ql;
#pragma unroll
                        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                            for (int e = 0; e < 16; ++e) {
                                const int key = kt * KT + 32 * kb + (e & 3) + 8 * (e >> 2) + 4 * h;
                                if (key > qpos) sacc[b][kb][e] = -1.0e30f;
                            }
                    }
                }
                if (ragged && kt == ntiles - 1) {           // mask the keys past kv_len (last tile only)
#pragma unroll
                    for (int b = 0; b < QB; ++b)
#pragma unroll
                        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                            for (int e = 0; e < 16; ++e) {
                                const int key = kt * KT + 32 * kb + (e & 3) + 8 * (e >> 2) + 4 * h;
                                if (key >= p.kv_len) sacc[b][kb][e] = -1.0e30f;
                            }
                }
            };
            scores();
            // V^T fragments of the tile (A operand of O^T += V^T P^T): hand-issued, four (lo, hi) pairs ahead of the MFMA that consumes them, the
            // first four before the softmax (they do not depend on it: their latency runs under it).  As important: NOT through the builtin - for
            // that hipcc emits s_waitcnt vmcnt(0) in front of the first read (the LDS it reads may alias what an in-flight LDS-DMA deposits), i.e.
            // every wave waited, once per tile, for the K / V tile it had just prefetched.
            constexpr int NP = 4 * KB, WP = QB == 2 ? 2 : (NP < 4 ? NP : 4);              // pairs: (chunk ch = pair / 2, d block db = pair % 2); two ahead is what the 64-query wave has registers for
            u32x2 vlo[NP], vhi[NP];
            const char* const vbase = sV + vf_off;
            auto v_read = [&](auto i_) __attribute__((always_inline)) {
                constexpr int i = decltype(i_)::value, ch = i / 2, db = i % 2;
                lds_read_tr16_b64_async<(2 * ch) * 1024 + db * 256>(vlo[i], vbase);
                lds_read_tr16_b64_async<(2 * ch) * 1024 + db * 256 + 1024>(vhi[i], vbase);
            };
            if constexpr (AT_ASYNC_FRAGS<KT>()) static_for<WP>(v_read);
            // Row maximum of (score - m) over this tile: three-input maxima (v_max3_f32), independent chains per half block, a tree,
            // one v_permlane32_swap for the other half-wave's keys.  Only the FIRST tile pays for it up front (it centres the
            // softmax on its true row maximum); every later tile exponentiates against the running maximum straight away and looks
            // at its maximum only if the sums say it has to (below): ~24 of a tile's ~130 vector instructions - this kernel is bound
            // by them (32 quarter-rate v_exp_f32 + ~100 others per 16 MFMAs), not by the matrix pipe.
            auto row_over = [&](int b_) __attribute__((always_inline)) {
                float mx[2 * KB];
#pragma unroll
                for (int c = 0; c < 2 * KB; ++c) {
                    const f32x16& a = sacc[b_][c >> 1];
                    const int b = 8 * (c & 1);
                    const float m0 = fmaxf(fmaxf(a[b], a[b + 1]), a[b + 2]);
                    const float m1 = fmaxf(fmaxf(a[b + 3], a[b + 4]), a[b + 5]);
                    mx[c] = fmaxf(fmaxf(m0, m1), fmaxf(a[b + 6], a[b + 7]));
                }
                float mall = fmaxf(mx[0], mx[1]);
                if constexpr (KB == 2) mall = fmaxf(fmaxf(mall, mx[2]), mx[3]);
                return wave_max_halves(mall) - (SUB_MAX ? m_run[b_] : 0.0f);          // (SUB_MAX: the accumulators hold the scores themselves)
            };
            // move the maximum by `delta` (>= 0 per row; the first tile: its true maximum): everything still at the old maximum - O,
            // l and this tile's scores - is rescaled exactly once
            auto move_max = [&](int b_, const float delta, const bool first) __attribute__((always_inline)) {
                m_run[b_] += delta;
                if constexpr (!SUB_MAX) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) minus_m[SUB_MAX ? 0 : b_][e] = -m_run[b_];
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                        for (int e = 0; e < 16; ++e) sacc[b_][kb][e] -= delta;
                }
                if (!first) {
                    const float alpha = fast_exp2(-delta);
                    l_run[b_] *= alpha;
#pragma unroll
                    for (int e = 0; e < 16; ++e) { oacc[b_][0][e] *= alpha; oacc[b_][1][e] *= alpha; }
                }
            };
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            u32x4 pf[QB][2 * KB];
            // p = 2^(score - m) for this lane's keys -> the P^T operand registers; returns their sum
            auto exponentiate = [&](int b_) __attribute__((always_inline)) {
                f32x2 ps2[2 * KB];                                                          // packed partial row sums
#pragma unroll
                for (int c = 0; c < 2 * KB; ++c) ps2[c] = f32x2{0.0f, 0.0f};
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        Pack8<T> pk;
#pragma unroll
                        for (int e = 0; e < 8; e += 2) {
                            f32x2 pe;
                            pe[0] = fast_exp2(sacc[b_][kb][8 * c + e] - (SUB_MAX ? m_run[b_] : 0.0f));                  // one v_exp per score
                            pe[1] = fast_exp2(sacc[b_][kb][8 * c + e + 1] - (SUB_MAX ? m_run[b_] : 0.0f));
                            ps2[2 * kb + c] += pe;
                            pk.e[e] = (T)pe[0];
                            pk.e[e + 1] = (T)pe[1];
                        }
                        pf[b_][2 * kb + c] = pk.raw;
                    }
                f32x2 pst = ps2[0] + ps2[1];
                if constexpr (KB == 2) pst += ps2[2] + ps2[3];
                return pst[0] + pst[1];
            };
            if constexpr (eager_max) {                  // (-DAA_ATTN_EAGER_MAX=1 build, A/B only: the round-2..4 form - every tile computes its maximum.  A RUN-time
#pragma unroll
                for (int b = 0; b < QB; ++b) {          //  switch made hipcc spill 25-37 registers in every instance of this kernel: build.py's audit rejects that)
                    const float over = row_over(b);
                    if (kt == 0 || wave_any(over > AT_DEFER)) move_max(b, kt == 0 ? over : fmaxf(over, 0.0f), kt == 0);
                }
            } else if (kt == 0) {
#pragma unroll
                for (int b = 0; b < QB; ++b) move_max(b, row_over(b), true);
            }
            // O^T += V^T P^T for one query block: chunk ch = 16 keys; this half-wave's 8 k-slots are keys 16ch + 4h + {0..3} and
            // 16ch + 8 + 4h + {0..3} (the order P^T's registers came out of the S^T accumulator layout)
            auto pv = [&](auto b__, auto first_) __attribute__((always_inline)) {
                constexpr int b_ = decltype(b__)::value;
                constexpr bool reads_issued = decltype(first_)::value;      // (block 0: the first WP pairs were issued in front of the softmax)
                if (prio) wave_priority<1>();
                if constexpr (AT_ASYNC_FRAGS<KT>()) {
                if constexpr (!reads_issued) static_for<WP>(v_read);
                static_for<NP>([&](auto i_) __attribute__((always_inline)) {
                    constexpr int i = decltype(i_)::value, ch = i / 2, db = i % 2;
                    constexpr int issued = (i + WP < NP) ? i + WP : NP;
                    lds_wait2<2 * (issued - i - 1)>(vlo[i], vhi[i]);                 // this pair has landed; the younger ones stay in flight
                    const u32x4 vf = {vlo[i][0], vlo[i][1], vhi[i][0], vhi[i][1]};
                    oacc[b_][db] = mfma_32x32x16(T(), vf, pf[b_][ch], oacc[b_][db]);
                    if constexpr (i + WP < NP) v_read(IntTag<i + WP>());
                });
                } else {
#pragma unroll
                    for (int ch = 0; ch < 2 * KB; ++ch)
#pragma unroll
                        for (int db = 0; db < 2; ++db) {
                            const char* base = sV + (2 * ch) * 1024 + db * 256 + vf_off;
                            const u32x2 lo = lds_read_tr16_b64(base), hi = lds_read_tr16_b64(base + 1024);
                            const u32x4 vf = {lo[0], lo[1], hi[0], hi[1]};
                            oacc[b_][db] = mfma_32x32x16(T(), vf, pf[b_][ch], oacc[b_][db]);
                        }
                }
                if (prio) wave_priority<0>();
            };
            // Lazy maximum (defer-max, round 5 form): the running maximum has to move only when some p would leave the range the
            // deferral allows (p <= 2^AT_DEFER).  A lane's p's are non-negative, so "one of them exceeds 2^AT_DEFER" implies "their
            // sum does" (and an overflowed p makes the sum inf, a NaN fails the comparison): the sum - needed anyway - is the whole
            // check.  When it fires the tile is multiplied again and its scores are looked at after all; rows that are more than
            // a bit above their maximum move it (so a flat run of p ~ 2..64 cannot fire tile after tile) and the tile is exponentiated
            // again.  Rare: after the first tile has centred a row, later keys seldom beat it by 6 bits.
            float psum[QB];
#pragma unroll
            for (int b = 0; b < QB; ++b) psum[b] = 0.0f;
            psum[0] = exponentiate(0);
            bool redo = kt != 0 && !eager_max && wave_any(!(psum[0] <= 64.0f));
            if constexpr (QB == 2) {
                // One block after the other; block 1's exponentials are issued behind block 0's O^T MFMAs and run while the pipe executes them.
                // The rare path keeps the block's scores (they are live up to here anyway, and at 256 registers there is room): nothing is
                // multiplied again.
                static_assert(AT_DEFER == 6.0f, "the sum test above is 2^AT_DEFER");
                if (redo) {
                    const float over = row_over(0);
                    if (wave_any(over > 1.0f)) move_max(0, fmaxf(over, 0.0f), false);
                    psum[0] = exponentiate(0);
                }
                l_run[0] += psum[0];
                pv(IntTag<0>(), BoolTag<true>());
                psum[1] = exponentiate(1);
                if (kt != 0 && !eager_max && wave_any(!(psum[1] <= 64.0f))) {
                    const float over = row_over(1);
                    if (wave_any(over > 1.0f)) move_max(1, fmaxf(over, 0.0f), false);
                    psum[1] = exponentiate(1);
                }
                l_run[1] += psum[1];
                pv(IntTag<1>(), BoolTag<false>());
            } else {
                if (redo) {
                    static_assert(AT_DEFER == 6.0f, "the sum test above is 2^AT_DEFER");
                    // (the scores are NOT kept alive across the test - 32 more live registers at the kernel's 168-register cap spilled
                    //  into the hot loop and cost the 4096-key kernel 55 %, r05e: the rare path multiplies the tile again, K is still in its slot)
                    scores();
                    const float over = row_over(0);
                    if (wave_any(over > 1.0f)) move_max(0, fmaxf(over, 0.0f), false);
                    psum[0] = exponentiate(0);
                }
                l_run[0] += psum[0];
                pv(IntTag<0>(), BoolTag<true>());
            }
        }
    }

    if (wave_active) {
#pragma unroll
        for (int b = 0; b < QB; ++b) {
            const float l_tot = wave_sum_halves(l_run[b]);
            const float inv = 1.0f / l_tot;
            const int q = q0 + 32 * b + ql;
            if (q < p.q_len) {
                T* dst = const_cast<T*>(attn_row<T>(p.o, o, i, p.n_inner, q, head));
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        union { u32x2 raw; T e[4]; } pk;
#pragma unroll
                        for (int e = 0; e < 4; ++e) pk.e[e] = (T)(oacc[b][db][4 * g + e] * inv);
                        *reinterpret_cast<u32x2*>(dst + 32 * db + 8 * g + 4 * h) = pk.raw;
                    }
            }
        }
    }
}


// ---- Short key sequences, many queries (the text cross-attention: 77 keys against 4096 / 1024 / 256 queries per image) ----------
// The general kernel above spends a workgroup per 128 queries: K / V staged anew for each, two 64-key tiles (the second holds 13
// keys), two barriers, an online-softmax rescale - 64.6 us for 178 MB at the 64x64 level (2.8 TB/s), a chain of latencies.  Here
// the K and V^T fragments of up to 96 keys (3 x 4 + 6 x 2 registers-of-4: 96 registers) are read ONCE per wave from one staged tile
// pair and STAY in registers while the wave walks `qblocks` 32-query blocks: per block 12 + 12 MFMAs, one single-pass softmax over the
// lane's 48 scores (every key is present: no running maximum, no rescale), no LDS traffic and no barrier; the next block's Q
// fragment is fetched while the current one is multiplied.  Same layouts as above (S^T = K Q^T, P^T registers = B operand of
// O^T = V^T P^T, V^T through ds_read_b64_tr_b16), same operand addressing.  grid = (ceil(q blocks / (4 * qblocks)), heads, sequences).
constexpr int ATS_KEYS = 96;
template <typename T>
__global__ void __launch_bounds__(256, 2) attention_shortkv_kernel(const AaAttention p, const int qblocks) {
    constexpr int NW = 4, KT = 64, PER = (KT / 4) / NW, TILE_BYTES = 2 * KT * 128;
    constexpr unsigned OOB = 0x80000000u;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = wave_id();
    const int h = lane >> 5, ql = lane & 31;
    const int head = blockIdx.y, seq = blockIdx.z;
    char* lds = dyn_smem();
    const int o = seq / p.n_inner, i = seq - o * p.n_inner;
    const float sl2e = p.scale * 1.4426950408889634f;

    // ---- stage keys 0..127 as two 64-key tiles (layouts of attention_kernel: K rows XOR-swizzled, V in [4 keys][32 d] blocks)
    const BufRsrc r_k = make_rsrc(p.k.ptr, (unsigned)attn_extent_bytes(p.k, p.n_outer, p.n_inner, p.kv_len));
    const BufRsrc r_v = make_rsrc(p.v.ptr, (unsigned)attn_extent_bytes(p.v, p.n_outer, p.n_inner, p.kv_len));
    const unsigned k_seq = (unsigned)((attn_seq_row(p.k, o, i, p.n_inner) * p.k.ld + p.k.col0 + head * 64) * 2);
    const unsigned v_seq = (unsigned)((attn_seq_row(p.v, o, i, p.n_inner) * p.v.ld + p.v.col0 + head * 64) * 2);
    const unsigned k_key = (unsigned)(p.k.pos_stride * p.k.ld * 2), v_key = (unsigned)(p.v.pos_stride * p.v.ld * 2);
    const int kk = lane >> 3;
    const int vk = 4 * (lane >> 5) + ((lane >> 2) & 3), vd = 32 * ((lane >> 4) & 1) + 8 * (lane & 3);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        char* sK = lds + t * TILE_BYTES;
        char* sV = sK + KT * 128;
#pragma unroll
        for (int j = 0; j < PER / 2; ++j) {
            const int n = wave * (PER / 2) + j;
            const int kl = t * KT + 8 * n + kk, vl = t * KT + 8 * n + vk;
            async_copy16_buf(r_k, kl < p.kv_len ? k_seq + (unsigned)kl * k_key + (unsigned)(((lane & 7) ^ ((kl >> 1) & 7)) * 16) : OOB, sK + n * 1024);
            async_copy16_buf(r_v, vl < p.kv_len ? v_seq + (unsigned)vl * v_key + (unsigned)(vd * 2) : OOB, sV + n * 1024);
        }
    }
    // the first Q fragment travels while the keys land
    const int qb0 = (blockIdx.x * NW + wave) * qblocks;                  // this wave's first 32-query block
    auto load_q = [&](int qb, u32x4 (&qf)[4]) __attribute__((always_inline)) {
        const int q = qb * 32 + ql;
        const bool ok = q < p.q_len;
        const T* src = attn_row<T>(p.q, o, i, p.n_inner, ok ? q : 0, head) + 8 * h;
#pragma unroll
        for (int dk = 0; dk < 4; ++dk) {
            qf[dk] = u32x4{0u, 0u, 0u, 0u};
            if (ok) qf[dk] = *reinterpret_cast<const u32x4*>(src + 16 * dk);
        }
    };
    u32x4 q_next[4];
    load_q(qb0, q_next);
    dma_wait<0>();
    block_barrier();
    // ---- K fragments (A operand of S^T: row = key, 8 consecutive d) and V^T fragments (A operand of O^T) into registers
    u32x4 kf[3][4], vf[6][2];
    {
        const int kf_row = ql * 128, kf_swz = (ql >> 1) & 7;
        const int vf_off = (2 * h) * 256 + ((lane & 15) >> 2) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
#pragma unroll
        for (int kb = 0; kb < 3; ++kb)
#pragma unroll
            for (int dk = 0; dk < 4; ++dk)
                kf[kb][dk] = *reinterpret_cast<const u32x4*>(lds + (kb >> 1) * TILE_BYTES + (kb & 1) * 4096 + kf_row + (((2 * dk + h) ^ kf_swz) << 4));
#pragma unroll
        for (int ch = 0; ch < 6; ++ch)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const char* base = lds + (ch >> 2) * TILE_BYTES + KT * 128 + (2 * (ch & 3)) * 1024 + db * 256 + vf_off;
                const u32x2 lo = lds_read_tr16_b64(base), hi = lds_read_tr16_b64(base + 1024);
                vf[ch][db] = u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
    }
    f32x16 zero16;
#pragma unroll
    for (int e = 0; e < 16; ++e) zero16[e] = 0.0f;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    for (int it = 0; it < qblocks; ++it) {
        const int qb = qb0 + it;
        if (qb * 32 >= p.q_len) break;                                   // (wave-uniform)
        u32x4 qf[4];
#pragma unroll
        for (int dk = 0; dk < 4; ++dk) {                                  // scores come out in units of bits: Q * scale * log2(e)
            Pack8<T> v;
            v.raw = q_next[dk];
#pragma unroll
            for (int e = 0; e < 8; ++e) v.e[e] = (T)((float)v.e[e] * sl2e);
            qf[dk] = v.raw;
        }
        if (it + 1 < qblocks) load_q(qb + 1, q_next);                     // (rows past q_len load zeros)
        f32x16 sacc[3];
#pragma unroll
        for (int dk = 0; dk < 4; ++dk)
#pragma unroll
            for (int kb = 0; kb < 3; ++kb) sacc[kb] = mfma_32x32x16(T(), kf[kb][dk], qf[dk], dk == 0 ? zero16 : sacc[kb]);
        if (p.kv_len < ATS_KEYS) {                                       // keys past kv_len (their K rows are zeros: score 0, not -inf)
#pragma unroll
            for (int kb = 0; kb < 3; ++kb)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = 32 * kb + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (key >= p.kv_len) sacc[kb][e] = -1.0e30f;
                }
        }
        float mx[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const f32x16& a = sacc[c >> 1];
            const int b = 8 * (c & 1);
            const float m0 = fmaxf(fmaxf(a[b], a[b + 1]), a[b + 2]);
            const float m1 = fmaxf(fmaxf(a[b + 3], a[b + 4]), a[b + 5]);
            mx[c] = fmaxf(fmaxf(m0, m1), fmaxf(a[b + 6], a[b + 7]));
        }
        const float m = wave_max_halves(fmaxf(fmaxf(fmaxf(mx[0], mx[1]), fmaxf(mx[2], mx[3])), fmaxf(mx[4], mx[5])));
        f32x2 ps = f32x2{0.0f, 0.0f};
        u32x4 pf[6];
#pragma unroll
        for (int kb = 0; kb < 3; ++kb)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                Pack8<T> pk;
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    f32x2 pe;
                    pe[0] = fast_exp2(sacc[kb][8 * c + e] - m);
                    pe[1] = fast_exp2(sacc[kb][8 * c + e + 1] - m);
                    ps += pe;
                    pk.e[e] = (T)pe[0];
                    pk.e[e + 1] = (T)pe[1];
                }
                pf[2 * kb + c] = pk.raw;
            }
        const float l_tot = wave_sum_halves(ps[0] + ps[1]);
        f32x16 oacc[2];
#pragma unroll
        for (int ch = 0; ch < 6; ++ch)
#pragma unroll
            for (int db = 0; db < 2; ++db) oacc[db] = mfma_32x32x16(T(), vf[ch][db], pf[ch], ch == 0 ? zero16 : oacc[db]);
        const float inv = 1.0f / l_tot;
        const int q = qb * 32 + ql;
        if (q < p.q_len) {
            T* dst = const_cast<T*>(attn_row<T>(p.o, o, i, p.n_inner, q, head));
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


// ---- attention on the vector ALUs for the head sizes the matrix-core kernel does not cover: head_dim 8 (the layerdiffuse
// alpha decoder UNet384: 32 heads of 8 channels at 256 channels; reference models/layerdiffuse_VAE.py:58 - two MFMA k-slices
// would be 3/4 padding at d = 8) and head_dim 80 (the CLIP ViT-H/14 vision tower of the SVD path, 16 heads at 1280 channels,
// 257 tokens, once per clip).  A thread owns one query (q, the running max / sum and the D outputs in registers), a workgroup
// of 256 queries shares 256-key K / V tiles through LDS (every lane reads the SAME key row: an LDS broadcast).  Same operand
// addressing as the head_dim-64 kernel.
template <typename T, int D>
__global__ void __launch_bounds__(256) attention_small_kernel(const AaAttention p) {
    constexpr int KT = 256, DC = D / 8;
    static_assert(D % 8 == 0, "head_dim");
    T* sK = reinterpret_cast<T*>(dyn_smem());                   // [KT][D]
    T* sV = sK + KT * D;
    const int tid = threadIdx.x;
    const int head = blockIdx.y, seq = blockIdx.z;
    const int o = seq / p.n_inner, i = seq - o * p.n_inner;
    const int q = blockIdx.x * 256 + tid;
    const bool active = q < p.q_len;
    const float sl2e = p.scale * 1.4426950408889634f;
    float qv[D], acc[D];
    {
        const T* qrow = reinterpret_cast<const T*>(p.q.ptr) +
                        (attn_seq_row(p.q, o, i, p.n_inner) + (int64_t)(active ? q : 0) * p.q.pos_stride) * p.q.ld + p.q.col0 + head * D;
#pragma unroll
        for (int c8 = 0; c8 < DC; ++c8) {
            Pack8<T> raw;
            raw.raw = u32x4{0u, 0u, 0u, 0u};
            if (active) raw.raw = *reinterpret_cast<const u32x4*>(qrow + c8 * 8);
#pragma unroll
            for (int d = 0; d < 8; ++d) { qv[c8 * 8 + d] = (float)raw.e[d] * sl2e; acc[c8 * 8 + d] = 0.0f; }
        }
    }
    float m_run = -1.0e30f, l_run = 0.0f;
    for (int k0 = 0; k0 < p.kv_len; k0 += KT) {
        __syncthreads();                                          // the previous tile has been consumed
        {
            const int key = k0 + tid;
            const bool ok = key < p.kv_len;
            const T* krow = reinterpret_cast<const T*>(p.k.ptr) +
                            (attn_seq_row(p.k, o, i, p.n_inner) + (int64_t)(ok ? key : 0) * p.k.pos_stride) * p.k.ld + p.k.col0 + head * D;
            const T* vrow = reinterpret_cast<const T*>(p.v.ptr) +
                            (attn_seq_row(p.v, o, i, p.n_inner) + (int64_t)(ok ? key : 0) * p.v.pos_stride) * p.v.ld + p.v.col0 + head * D;
#pragma unroll
            for (int c8 = 0; c8 < DC; ++c8) {
                u32x4 kr = {0u, 0u, 0u, 0u}, vr = {0u, 0u, 0u, 0u};
                if (ok) { kr = *reinterpret_cast<const u32x4*>(krow + c8 * 8); vr = *reinterpret_cast<const u32x4*>(vrow + c8 * 8); }
                *reinterpret_cast<u32x4*>(sK + tid * D + c8 * 8) = kr;
                *reinterpret_cast<u32x4*>(sV + tid * D + c8 * 8) = vr;
            }
        }
        __syncthreads();
        const int n = min(KT, p.kv_len - k0);
        for (int j = 0; j < n; j += 8) {                          // chunks of 8 keys: one max / rescale decision per chunk
            float s[8];
            float cmax = -1.0e30f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float d = 0.0f;
#pragma unroll
                for (int c8 = 0; c8 < DC; ++c8) {
                    Pack8<T> kr; kr.raw = *reinterpret_cast<const u32x4*>(sK + (j + e) * D + c8 * 8);
#pragma unroll
                    for (int c = 0; c < 8; ++c) d = __builtin_fmaf(qv[c8 * 8 + c], (float)kr.e[c], d);
                }
                s[e] = (j + e < n) ? d : -1.0e30f;
                cmax = fmaxf(cmax, s[e]);
            }
            if (cmax > m_run) {
                const float alpha = fast_exp2(m_run - cmax);
                l_run *= alpha;
#pragma unroll
                for (int c = 0; c < D; ++c) acc[c] *= alpha;
                m_run = cmax;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float pe = fast_exp2(s[e] - m_run);
                l_run += pe;
#pragma unroll
                for (int c8 = 0; c8 < DC; ++c8) {
                    Pack8<T> vr; vr.raw = *reinterpret_cast<const u32x4*>(sV + (j + e) * D + c8 * 8);
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc[c8 * 8 + c] = __builtin_fmaf(pe, (float)vr.e[c], acc[c8 * 8 + c]);
                }
            }
        }
    }
    if (active) {
        const float inv = 1.0f / l_run;
        T* orow = reinterpret_cast<T*>(const_cast<void*>(p.o.ptr)) +
                  (attn_seq_row(p.o, o, i, p.n_inner) + (int64_t)q * p.o.pos_stride) * p.o.ld + p.o.col0 + head * D;
#pragma unroll
        for (int c8 = 0; c8 < DC; ++c8) {
            Pack8<T> out;
#pragma unroll
            for (int c = 0; c < 8; ++c) out.e[c] = (T)(acc[c8 * 8 + c] * inv);
            *reinterpret_cast<u32x4*>(orow + c8 * 8) = out.raw;
        }
    }
}

}  // namespace aa
