// gfx950 device primitives used by every kernel: dynamic LDS base, 32x32x16 MFMA wrappers
// (f16 / bf16 in, f32 accumulate), wave64 cross-lane exchange.  tests/emu/dev.h supplies
// host-side stand-ins with the same names so kernel bodies can be exercised on the CPU.
#pragma once
#include <hip/hip_runtime.h>
#include "../types.h"

__device__ __forceinline__ char* dyn_smem() {
    extern __shared__ __attribute__((aligned(16))) char aa_lds_[];
    return aa_lds_;
}

// D = A(32x16) * B(16x32) + C on one wavefront.  Lane l supplies row (l&31) of A and column
// (l&31) of B, eight consecutive k each (half-wave l>>5 picks which eight); D comes back as
// col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5) for register r (cdna_hip_programming.md section 3).
__device__ __forceinline__ f32x16 mfma_32x32x16(f16_t, u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_32x32x16(bf16_t, u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ float wave_shfl_xor(float v, int mask) { return __shfl_xor(v, mask, 64); }
__device__ __forceinline__ float wave_shfl(float v, int src) { return __shfl(v, src, 64); }
// bare v_exp_f32 (2^x): no denormal-range fix-up sequence around it - callers only pass x <= 0 and are happy with
// results that underflow to 0.
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float clamp_f(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }   // one v_med3_f32
__device__ __forceinline__ bool wave_any(bool pred) { return __any((int)pred) != 0; }
// Combine a value with the one held by the lane 32 positions away (the other half-wave) with ONE
// v_permlane32_swap (VALU, no LDS crossbar): after swapping (x, x) every lane holds {own, other}.
__device__ __forceinline__ float wave_max_halves(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}
__device__ __forceinline__ float wave_sum_halves(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}

// Shader-clock timestamp (s_memtime) for the phase probe (AaConvGemm.debug bit 8).
__device__ __forceinline__ long long clock_now() { return (long long)__builtin_amdgcn_s_memtime(); }
__device__ __forceinline__ long long wall_now() { return (long long)__builtin_amdgcn_s_memrealtime(); }      // 100 MHz, chip-wide
// HW_ID (wave / SIMD / CU / SH / SE of this wave) in the low word, XCC_ID in the high word
__device__ __forceinline__ long long hw_id() {
    return (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
}

// busy-wait for `ticks` of the 100 MHz wall clock (s_sleep between the reads: the wave leaves its issue slots to the others)
__device__ __forceinline__ void spin_wall_ticks(int ticks) {
    const long long t0 = wall_now();
    while (wall_now() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

// LDS-DMA through a buffer descriptor: every lane fetches 16 bytes, the wave's 64 pieces land lane-linear at
// lds_wave_base + lane*16 (wave-uniform base) without passing through VGPRs; operand base + range sit in SGPRs, each lane
// supplies one 32-bit byte offset, and a lane whose offset is out of range deposits 16 zero bytes
// (buffer_load_dwordx4 ... offen lds).  Completion is tracked by vmcnt.
struct BufRsrc { __amdgpu_buffer_rsrc_t v; };
__device__ __forceinline__ BufRsrc make_rsrc(const void* base, unsigned bytes) {
    return BufRsrc{__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000)};
}
__device__ __forceinline__ void async_copy16_buf(const BufRsrc& r, unsigned byte_offset, void* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r.v, (__attribute__((address_space(3))) void*)lds_wave_base, 16, byte_offset, 0, 0, 0);
}
// 16 bytes per lane through a buffer descriptor: a lane whose byte offset (+16) is out of range loads zeros / stores nothing,
// so row tails and column tails of a tile cost a v_cndmask on the offset instead of a branch around the access
__device__ __forceinline__ u32x4 buf_load16(const BufRsrc& r, unsigned byte_offset) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r.v, byte_offset, 0, 0));
}
// the same with the non-temporal hint (aux bit 1: nt): streamed once, not worth an L2 line that a re-read operand (weights) could keep
__device__ __forceinline__ u32x4 buf_load16_nt(const BufRsrc& r, unsigned byte_offset) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r.v, byte_offset, 0, 2));
}
__device__ __forceinline__ void buf_store16_nt(const BufRsrc& r, unsigned byte_offset, u32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), r.v, byte_offset, 0, 2);
}
__device__ __forceinline__ void buf_store16(const BufRsrc& r, unsigned byte_offset, u32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), r.v, byte_offset, 0, 0);
}
// Transposing LDS read (ds_read_b64_tr_b16): every lane passes the address of 4 consecutive 16-bit elements; inside
// each 16-lane group, lane l = 4a + b receives element b of lanes a, 4 + a, 8 + a, 12 + a.  With lane s pointing at
// row (s>>2), columns 4(s&3).. of a row-major [4][16] block, lane l gets column l of the block, rows 0..3.
__device__ __forceinline__ u32x2 lds_read_tr16_b64(const void* lds_ptr) {
    typedef short s16x4_ __attribute__((ext_vector_type(4)));
    const auto v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_*)(lds_ptr));
    return __builtin_bit_cast(u32x2, v);
}
// Index of this wavefront inside the workgroup, as a scalar.
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// Counted wait on this wave's outstanding vector-memory operations (LDS-DMA included): returns once at
// most N are still in flight, i.e. everything issued before the N youngest has landed.
template <int N>
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// everything this wave has issued - LDS-DMA, global loads / stores, LDS reads / writes - is complete (in front of a raw s_barrier that
// publishes plain LDS stores: the compiler does not know the asm statement is a barrier and would not wait for them)
__device__ __forceinline__ void mem_wait_all() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
// the compiler's scheduler moves nothing across this point (no instruction emitted)
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// Workgroup barrier WITHOUT the implicit vmcnt(0) drain of __syncthreads(); the "memory" clobber keeps the
// compiler from moving LDS / DMA accesses across it.
__device__ __forceinline__ void block_barrier() { asm volatile("s_barrier" ::: "memory"); }
// the lanes of a wavefront run in lock step: nothing to wait for (the host emulator, which runs lanes one after another, does)
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_wave_barrier(); }

// Hand-issued LDS fragment read: `dst` is written asynchronously (lgkmcnt); hipcc neither counts it nor waits
// for it, so every consumer must sit behind lds_wait<N>() / lds_pin() naming the register (cdna guide 5.7).
// Lets a k sub-step's ds_reads stay in flight under the previous sub-step's MFMAs with a COUNTED wait -
// compiled C++ loads are re-materialised into one register set and waited with lgkmcnt(0) every sub-step.
__device__ __forceinline__ void lds_read16_async(u32x4& dst, const void* lds_ptr) {
    const unsigned addr = (unsigned)(unsigned long long)lds_ptr;        // low 32 bits of a flat LDS pointer = LDS offset
    asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr));
}
// Hand-issued LDS store of 16 bytes per lane (LDS operations of one wave execute in order: a later hand-issued read of the same wave sees it).
// Unlike a plain C++ store it does not make hipcc drain the LDS-DMA queue (s_waitcnt vmcnt(0)) in front of the next read.
__device__ __forceinline__ void lds_write16_async(void* lds_ptr, const u32x4& v) {
    const unsigned addr = (unsigned)(unsigned long long)lds_ptr;
    asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
// Wait until at most N LDS operations are outstanding; `x` becomes available to consumers here.
template <int N>
__device__ __forceinline__ void lds_wait(u32x4& x) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x) : "n"(N)); }
// Order the consumers of `x` behind the preceding lds_wait (no instruction emitted).
__device__ __forceinline__ void lds_pin(u32x4& x) { asm volatile("" : "+v"(x)); }
// The same for the 8-byte transposing read (ds_read_b64_tr_b16, see lds_read_tr16_b64), with a compile-time byte offset in the
// instruction's offset field (one address register for all pieces of a tile).  Besides overlapping the reads with matrix work, the
// hand-issued form keeps hipcc from draining the LDS-DMA queue in front of them: for the builtin it emits s_waitcnt vmcnt(0) (the read
// may alias what an in-flight `buffer_load ... lds` deposits), which stalls every wave on the K / V tiles it has just prefetched.
template <int OFF>
__device__ __forceinline__ void lds_read_tr16_b64_async(u32x2& dst, const void* lds_ptr) {
    const unsigned addr = (unsigned)(unsigned long long)lds_ptr;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void lds_read16_async_off(u32x4& dst, const void* lds_ptr) {
    const unsigned addr = (unsigned)(unsigned long long)lds_ptr;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void lds_wait2(u32x2& x, u32x2& y) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(x), "+v"(y) : "n"(N)); }

// ---- accumulator file of the hand-scheduled contraction kernel (conv_gemm_x.h) --------------------------------------
// 32x32 accumulator block b (16 fp32 per lane) lives under a LITERAL name in the accumulation half of the unified register
// file, a[16 b : 16 b + 15], for b < 16 (a one-wave-per-SIMD kernel then has all 256 architectural VGPRs for fragments, DMA
// offsets and the epilogue); blocks 16..19 (the 256 x 320 tile) sit in compiler-allocated VGPRs.  The MFMAs are
// `asm volatile` statements: program order = issue order.  hipcc does not allocate the literal registers: every statement
// that writes a block lists its 16 registers as clobbers, so hipcc (a) counts them in the kernel descriptor and (b) knows
// nothing of its own survives there (without the clobbers it spills INTO accumulation registers it believes free as soon
// as it runs short of VGPRs - seen in the first build of the 320-column tile).  Letting hipcc allocate the blocks through
// "+a" operands was tried as well: it then shuffles accumulators between the two halves of the file around the loop
// tails (864 v_accvgpr moves + 284 scratch accesses in one K step).  tests/test_abi.py audits the disassembly of these
// kernels: no scratch, no compiler-made v_accvgpr_* (cdna guide 5.7 item 4).
constexpr int ACC_BLOCKS = 20;
struct AccFile { f32x16 v[ACC_BLOCKS - 16]; };
// -DAA_X_ABLATE=bits builds a timing-only variant of the hand-scheduled kernels (scripts/x_ablate.py; results are garbage):
// 1 = no operand DMA, 2 = no MFMA, 4 = no fragment reads
#ifndef AA_X_ABLATE
#define AA_X_ABLATE 0
#endif
#define AA_ACC_LITERAL_BLOCKS(X) \
    X(0, "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15") \
    X(1, "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31") \
    X(2, "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47") \
    X(3, "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63") \
    X(4, "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79") \
    X(5, "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95") \
    X(6, "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111") \
    X(7, "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127") \
    X(8, "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143") \
    X(9, "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159") \
    X(10, "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175") \
    X(11, "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191") \
    X(12, "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207") \
    X(13, "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223") \
    X(14, "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239") \
    X(15, "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255")
template <int B>
__device__ __forceinline__ void acc_zero(AccFile& af) {
    if constexpr (B < 16) {
#define AA_X(b, ...) if constexpr (B == b) asm volatile( \
            "v_accvgpr_write_b32 a[%c0+0], 0\n\tv_accvgpr_write_b32 a[%c0+1], 0\n\tv_accvgpr_write_b32 a[%c0+2], 0\n\tv_accvgpr_write_b32 a[%c0+3], 0\n\t" \
            "v_accvgpr_write_b32 a[%c0+4], 0\n\tv_accvgpr_write_b32 a[%c0+5], 0\n\tv_accvgpr_write_b32 a[%c0+6], 0\n\tv_accvgpr_write_b32 a[%c0+7], 0\n\t" \
            "v_accvgpr_write_b32 a[%c0+8], 0\n\tv_accvgpr_write_b32 a[%c0+9], 0\n\tv_accvgpr_write_b32 a[%c0+10], 0\n\tv_accvgpr_write_b32 a[%c0+11], 0\n\t" \
            "v_accvgpr_write_b32 a[%c0+12], 0\n\tv_accvgpr_write_b32 a[%c0+13], 0\n\tv_accvgpr_write_b32 a[%c0+14], 0\n\tv_accvgpr_write_b32 a[%c0+15], 0\n\t" \
            "s_nop 4" ::"i"(16 * b) : __VA_ARGS__);
        AA_ACC_LITERAL_BLOCKS(AA_X)
#undef AA_X
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) af.v[B - 16][e] = 0.0f;
        asm volatile("s_nop 4" : "+v"(af.v[B - 16]));
    }
}
// block B = v (the lane's 16 columns of the bias: the accumulation starts from it, the epilogue adds nothing)
template <int B>
__device__ __forceinline__ void acc_init(AccFile& af, const f32x16& v) {
    if constexpr (B < 16) {
#define AA_X(b, ...) if constexpr (B == b) asm volatile( \
            "v_accvgpr_write_b32 a[%c0+0], %1\n\tv_accvgpr_write_b32 a[%c0+1], %2\n\tv_accvgpr_write_b32 a[%c0+2], %3\n\tv_accvgpr_write_b32 a[%c0+3], %4\n\t" \
            "v_accvgpr_write_b32 a[%c0+4], %5\n\tv_accvgpr_write_b32 a[%c0+5], %6\n\tv_accvgpr_write_b32 a[%c0+6], %7\n\tv_accvgpr_write_b32 a[%c0+7], %8\n\t" \
            "v_accvgpr_write_b32 a[%c0+8], %9\n\tv_accvgpr_write_b32 a[%c0+9], %10\n\tv_accvgpr_write_b32 a[%c0+10], %11\n\tv_accvgpr_write_b32 a[%c0+11], %12\n\t" \
            "v_accvgpr_write_b32 a[%c0+12], %13\n\tv_accvgpr_write_b32 a[%c0+13], %14\n\tv_accvgpr_write_b32 a[%c0+14], %15\n\tv_accvgpr_write_b32 a[%c0+15], %16\n\t" \
            "s_nop 4" ::"i"(16 * b), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), \
                        "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]) : __VA_ARGS__);
        AA_ACC_LITERAL_BLOCKS(AA_X)
#undef AA_X
    } else {
        af.v[B - 16] = v;
        asm volatile("s_nop 4" : "+v"(af.v[B - 16]));
    }
}
// block B += W(32 x 16, MFMA "A" operand: rows -> accumulator registers) * A(16 x 32, "B" operand: columns -> lanes)
template <int B>
__device__ __forceinline__ void acc_mfma(AccFile& af, f16_t, const u32x4& w, const u32x4& a) {
    if constexpr ((AA_X_ABLATE & 2) != 0) { asm volatile("" :: "v"(w), "v"(a)); return; }
    if constexpr (B < 16) {
#define AA_X(b, ...) if constexpr (B == b) asm volatile("v_mfma_f32_32x32x16_f16 a[%c0:%c1], %2, %3, a[%c0:%c1]" ::"i"(16 * b), "i"(16 * b + 15), "v"(w), "v"(a) : __VA_ARGS__);
        AA_ACC_LITERAL_BLOCKS(AA_X)
#undef AA_X
    } else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(af.v[B - 16]) : "v"(w), "v"(a));
}
template <int B>
__device__ __forceinline__ void acc_mfma(AccFile& af, bf16_t, const u32x4& w, const u32x4& a) {
    if constexpr (B < 16) {
#define AA_X(b, ...) if constexpr (B == b) asm volatile("v_mfma_f32_32x32x16_bf16 a[%c0:%c1], %2, %3, a[%c0:%c1]" ::"i"(16 * b), "i"(16 * b + 15), "v"(w), "v"(a) : __VA_ARGS__);
        AA_ACC_LITERAL_BLOCKS(AA_X)
#undef AA_X
    } else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(af.v[B - 16]) : "v"(w), "v"(a));
}
// all MFMAs issued so far have written their accumulators (8-pass XDL -> VALU read needs 11 wait states; hipcc does not
// know the statements above are MFMAs, so the states are spent by hand: 32)
__device__ __forceinline__ void acc_settle() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }
template <int B>
__device__ __forceinline__ f32x16 acc_get(AccFile& af) {
    if constexpr (B < 16) {
        f32x16 r;
#define AA_ACC_RD(e) asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(r[e]) : "i"(16 * B + e));
        AA_ACC_RD(0) AA_ACC_RD(1) AA_ACC_RD(2) AA_ACC_RD(3) AA_ACC_RD(4) AA_ACC_RD(5) AA_ACC_RD(6) AA_ACC_RD(7)
        AA_ACC_RD(8) AA_ACC_RD(9) AA_ACC_RD(10) AA_ACC_RD(11) AA_ACC_RD(12) AA_ACC_RD(13) AA_ACC_RD(14) AA_ACC_RD(15)
#undef AA_ACC_RD
        return r;
    } else {
        return af.v[B - 16];
    }
}
// c += a.x * b.x + a.y * b.y on one packed pair of 16-bit values (v_dot2c_f32_f16 / v_dot2c_f32_bf16): products of 16-bit values
// are exact in fp32.  The row statistics of a contraction's epilogue are built from these (two scalars per lane).  Round 4 first
// used matrix-core products for them: inside the hand-scheduled kernels the result of the MFMA builtin landed in accumulation
// registers hipcc believed free - the literal accumulator blocks that had not been read out yet - and for some (tile, epilogue form)
// pairs the output was wrong by a few per cent of its range (tile 47 without a residual; found by tile fuzzing,
// scripts/debug/fuzz_tiles_fullsize.py); pinning those accumulators to VGPRs made hipcc spill elsewhere into the same registers.
__device__ __forceinline__ float dot2_f32(f16_t, unsigned a, unsigned b, float c) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, a), __builtin_bit_cast(h2, b), c, false);
}
__device__ __forceinline__ float dot2_f32(bf16_t, unsigned a, unsigned b, float c) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2, a), __builtin_bit_cast(b2, b), c, false);
}
__device__ __forceinline__ unsigned ones_pair(f16_t) { return 0x3C003C00u; }
__device__ __forceinline__ unsigned ones_pair(bf16_t) { return 0x3F803F80u; }

// hand-issued 16-byte LDS read from (address ^ X): the XOR sits inside the statement (hipcc would otherwise keep one
// precomputed address register per sub-step and fragment)
template <int X>
__device__ __forceinline__ void lds_read16_xor(u32x4& dst, const void* lds_ptr, IntTag<X>) {
    if constexpr ((AA_X_ABLATE & 4) != 0) { asm volatile("" : "=v"(dst) : "v"(lds_ptr)); return; }
    const unsigned addr = (unsigned)(unsigned long long)lds_ptr;
    if constexpr (X == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr));
    else { unsigned t; asm volatile("v_xor_b32 %1, %3, %2\n\tds_read_b128 %0, %1" : "=v"(dst), "=&v"(t) : "v"(addr), "i"(X)); }
}
// issue priority of this wave among the waves of its SIMD (0..3; the K loop of a workgroup runs above the epilogue of the
// workgroup it shares the CU with, so the sparse MFMA / DMA stream never queues behind the other's VALU-dense epilogue)
template <int P>
__device__ __forceinline__ void wave_priority() { asm volatile("s_setprio %0" ::"n"(P)); }
// every hand-issued fragment read has landed
__device__ __forceinline__ void lds_wait_all() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// LDS-DMA piece with a wave-uniform byte offset on top of the per-lane one (the range check covers the per-lane part only)
__device__ __forceinline__ void async_copy16_buf_s(const BufRsrc& r, unsigned lane_offset, unsigned uniform_offset, void* lds_wave_base) {
    if constexpr ((AA_X_ABLATE & 1) != 0) { asm volatile("" :: "v"(lane_offset), "s"(uniform_offset)); return; }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r.v, (__attribute__((address_space(3))) void*)lds_wave_base, 16, lane_offset, uniform_offset, 0, 0);
}

// Source byte offset of one fed activation row for the prepared (tap, source) of an implicit-GEMM K step, branch-free:
//   pb = pix * c2 + t          pix = pixel under tap (0, 0) (< 2^24), c2 = bytes per pixel, t = 16-byte slot + tap offset
//   pb |= bit 31               when bit (31 - sh) of `invalid` is set (the tap reads padding / the row is behind the tile):
//                              the buffer range check then deposits zeros
// One asm statement = three VALU instructions where the source puts them (in an MFMA gap of conv_gemm_x.h).
__device__ __forceinline__ void im2col_offset(unsigned& pb, unsigned pix, unsigned c2, unsigned t, unsigned sh, unsigned invalid) {
    unsigned f;
    asm volatile("v_mad_u32_u24 %0, %2, %3, %4\n\tv_lshlrev_b32 %1, %5, %6\n\tv_and_or_b32 %0, %1, %7, %0"
                 : "=&v"(pb), "=&v"(f) : "v"(pix), "s"(c2), "v"(t), "s"(sh), "v"(invalid), "s"(0x80000000u));
}
