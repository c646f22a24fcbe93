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
// Workgroup barrier WITHOUT the implicit vmcnt(0) drain of __syncthreads(); the "memory" clobber keeps the
// compiler from moving LDS / DMA accesses across it.
__device__ __forceinline__ void block_barrier() { asm volatile("s_barrier" ::: "memory"); }

// Hand-issued LDS fragment read: `dst` is written asynchronously (lgkmcnt); hipcc neither counts it nor waits
// for it, so every consumer must sit behind lds_wait<N>() / lds_pin() naming the register (cdna guide 5.7).
// Lets a k sub-step's ds_reads stay in flight under the previous sub-step's MFMAs with a COUNTED wait -
// compiled C++ loads are re-materialised into one register set and waited with lgkmcnt(0) every sub-step.
__device__ __forceinline__ void lds_read16_async(u32x4& dst, const void* lds_ptr) {
    const unsigned addr = (unsigned)(unsigned long long)lds_ptr;        // low 32 bits of a flat LDS pointer = LDS offset
    asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr));
}
// Wait until at most N LDS operations are outstanding; `x` becomes available to consumers here.
template <int N>
__device__ __forceinline__ void lds_wait(u32x4& x) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x) : "n"(N)); }
// Order the consumers of `x` behind the preceding lds_wait (no instruction emitted).
__device__ __forceinline__ void lds_pin(u32x4& x) { asm volatile("" : "+v"(x)); }
