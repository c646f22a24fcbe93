// Host-side SIMT emulator standing in for csrc/kernels/device/dev.h.  TEST INFRASTRUCTURE ONLY.
//
// Lets the kernel bodies in animate_anything_amd/csrc/kernels/*.h run on the CPU (this container has
// no GPU) so their index arithmetic, LDS choreography and MFMA fragment bookkeeping can be checked
// against numpy before a GPU-minute is spent.  Every thread of a workgroup is a ucontext fiber;
// __syncthreads() and the wave64 collectives (MFMA, shuffles) are rendezvous points.  The 32x32x16
// MFMA is modelled with the gfx950 register layout documented in cdna_hip_programming.md section 3.
// It models semantics only - no timing, no bank conflicts, no memory-ordering hazards.
#pragma once
#include <ucontext.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <functional>
#include <memory>
#include <thread>
#include <vector>
#include "types.h"

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace emu {
struct Fiber {
    ucontext_t ctx;
    std::unique_ptr<char[]> stack;      // allocated once per worker thread and launch, reused block after block
    bool done = false;
    dim3 tid;
};
constexpr size_t kStackBytes = 256 * 1024;
struct Block {
    std::vector<Fiber> fibers;
    ucontext_t sched;
    int cur = 0;
    int alive = 0;
    int bar_count = 0;
    unsigned bar_gen = 0;
    std::vector<int> wave_count;
    std::vector<unsigned> wave_gen;
    std::vector<std::vector<const void*>> wave_slots;   // [wave][lane]
    std::vector<char> lds;
    std::function<void()> body;
};
inline Block*& blk() { static thread_local Block* b = nullptr; return b; }
}  // namespace emu

// workgroups are independent: a launch spreads them over OS threads (AA_EMU_THREADS, default = hardware threads); the
// per-workgroup state (current block, the running fiber's indices) is thread-local
inline thread_local dim3 threadIdx, blockIdx;
inline dim3 blockDim, gridDim;

namespace emu {
inline void yield() {
    Block* b = blk();
    Fiber& f = b->fibers[b->cur];
    swapcontext(&f.ctx, &b->sched);
}
inline int linear_tid() { return threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z); }
inline void block_barrier() {
    Block* b = blk();
    const unsigned g = b->bar_gen;
    if (++b->bar_count == b->alive) { b->bar_count = 0; ++b->bar_gen; }
    else while (b->bar_gen == g) yield();
}
inline void wave_barrier() {
    Block* b = blk();
    const int w = linear_tid() >> 6;
    const int lanes = std::min<int>(64, (int)b->fibers.size() - 64 * w);
    const unsigned g = b->wave_gen[w];
    if (++b->wave_count[w] == lanes) { b->wave_count[w] = 0; ++b->wave_gen[w]; }
    else while (b->wave_gen[w] == g) yield();
}
// publish a pointer to this lane's operand, wait for the whole wave, let `use` read any lane's
// operand, wait again so no lane's stack storage disappears while others still read it.
template <typename F>
inline void wave_collective(const void* mine, F&& use) {
    Block* b = blk();
    const int t = linear_tid();
    b->wave_slots[t >> 6][t & 63] = mine;
    wave_barrier();
    use(b->wave_slots[t >> 6]);
    wave_barrier();
}
inline void fiber_entry() {
    Block* b = blk();
    b->body();
    b->fibers[b->cur].done = true;
    --b->alive;
    if (b->alive > 0 && b->bar_count == b->alive) { b->bar_count = 0; ++b->bar_gen; }
    swapcontext(&b->fibers[b->cur].ctx, &b->sched);
}
// Run `body` once per thread of a grid x block launch with `lds_bytes` of dynamic LDS per block.
inline void run_blocks(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& body, std::atomic<long>& next) {
    const int nthreads = block.x * block.y * block.z;
    const long total = (long)grid.x * grid.y * grid.z;
    const int nw = (nthreads + 63) / 64;
    Block b;
    b.body = body;
    b.fibers.resize(nthreads);
    for (int t = 0; t < nthreads; ++t) {
        b.fibers[t].stack.reset(new char[kStackBytes]);
        b.fibers[t].tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
    }
    blk() = &b;
    for (long idx = next.fetch_add(1); idx < total; idx = next.fetch_add(1)) {
        b.lds.assign(lds_bytes + 64, (char)0xAB);
        b.wave_count.assign(nw, 0); b.wave_gen.assign(nw, 0);
        b.wave_slots.assign(nw, std::vector<const void*>(64, nullptr));
        b.alive = nthreads; b.bar_count = 0; b.bar_gen = 0; b.cur = 0;
        for (int t = 0; t < nthreads; ++t) {
            Fiber& f = b.fibers[t];
            f.done = false;
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = f.stack.get();
            f.ctx.uc_stack.ss_size = kStackBytes;
            f.ctx.uc_link = nullptr;
            makecontext(&f.ctx, (void (*)())fiber_entry, 0);
        }
        blockIdx = dim3((unsigned)(idx % grid.x), (unsigned)((idx / grid.x) % grid.y), (unsigned)(idx / ((long)grid.x * grid.y)));
        while (b.alive > 0)
            for (int t = 0; t < nthreads; ++t) {
                if (b.fibers[t].done) continue;
                b.cur = t;
                threadIdx = b.fibers[t].tid;
                swapcontext(&b.sched, &b.fibers[t].ctx);
            }
    }
    blk() = nullptr;
}
inline int worker_count() {
    static const int n = [] {
        const char* e = getenv("AA_EMU_THREADS");
        int v = e ? atoi(e) : (int)std::thread::hardware_concurrency();
        return v < 1 ? 1 : (v > 64 ? 64 : v);
    }();
    return n;
}
inline void launch(dim3 grid, dim3 block, size_t lds_bytes, std::function<void()> body) {
    gridDim = grid; blockDim = block;
    const long total = (long)grid.x * grid.y * grid.z;
    std::atomic<long> next{0};
    const int workers = (int)std::min<long>(worker_count(), total);
    if (workers <= 1) { run_blocks(grid, block, lds_bytes, body, next); return; }
    std::vector<std::thread> pool;
    for (int w = 0; w < workers; ++w) pool.emplace_back([&] { run_blocks(grid, block, lds_bytes, body, next); });
    for (auto& t : pool) t.join();
}
}  // namespace emu

inline void __syncthreads() { emu::block_barrier(); }
inline char* dyn_smem() {
    char* p = emu::blk()->lds.data();
    return p + ((16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15);
}
inline float __expf(float x) { return std::exp(x); }
inline float rsqrtf(float x) { return 1.0f / std::sqrt(x); }
using std::min;
using std::max;
inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
// the workgroups of a launch run on concurrent host threads: tickets (AaConvGemm.tickets) need the real thing
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

inline float wave_shfl_xor(float v, int mask) {
    float out = 0.0f;
    emu::wave_collective(&v, [&](const std::vector<const void*>& s) {
        out = *static_cast<const float*>(s[(emu::linear_tid() & 63) ^ mask]);
    });
    return out;
}
inline float wave_shfl(float v, int src) {
    float out = 0.0f;
    emu::wave_collective(&v, [&](const std::vector<const void*>& s) { out = *static_cast<const float*>(s[src]); });
    return out;
}

inline float fast_exp2(float x) { return std::exp2(x); }
inline float fast_rcp(float x) { return 1.0f / x; }
inline float clamp_f(float x, float lo, float hi) { return std::fmin(std::fmax(x, lo), hi); }
inline float wave_max_halves(float x) { return std::fmax(x, wave_shfl_xor(x, 32)); }
inline float wave_sum_halves(float x) { return x + wave_shfl_xor(x, 32); }

inline bool wave_any(bool pred) {
    bool out = false;
    emu::wave_collective(&pred, [&](const std::vector<const void*>& s) {
        const int lanes = std::min<int>(64, (int)emu::blk()->fibers.size() - 64 * (emu::linear_tid() >> 6));
        for (int l = 0; l < lanes; ++l) out = out || *static_cast<const bool*>(s[l]);
    });
    return out;
}

template <typename T>
inline f32x16 emu_mfma_32x32x16(u32x4 a, u32x4 b, f32x16 c) {
    struct Ops { u32x4 a, b; } mine{a, b};
    f32x16 d = c;
    emu::wave_collective(&mine, [&](const std::vector<const void*>& s) {
        const int lane = emu::linear_tid() & 63;
        const int j = lane & 31;
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            float acc = c[r];
            for (int half = 0; half < 2; ++half) {
                Pack8<T> pa, pb;
                pa.raw = static_cast<const Ops*>(s[i + 32 * half])->a;
                pb.raw = static_cast<const Ops*>(s[j + 32 * half])->b;
                for (int e = 0; e < 8; ++e) acc += (float)pa.e[e] * (float)pb.e[e];
            }
            d[r] = acc;
        }
    });
    return d;
}
inline f32x16 mfma_32x32x16(f16_t, u32x4 a, u32x4 b, f32x16 c) { return emu_mfma_32x32x16<f16_t>(a, b, c); }
inline f32x16 mfma_32x32x16(bf16_t, u32x4 a, u32x4 b, f32x16 c) { return emu_mfma_32x32x16<bf16_t>(a, b, c); }
// (device/dev.h: v_dot2c_f32_f16 / v_dot2c_f32_bf16 on one packed pair)
template <typename T>
inline float emu_dot2(unsigned a, unsigned b, float c) {
    T x[2], y[2];
    std::memcpy(x, &a, 4); std::memcpy(y, &b, 4);
    return c + (float)x[0] * (float)y[0] + (float)x[1] * (float)y[1];
}
inline float dot2_f32(f16_t, unsigned a, unsigned b, float c) { return emu_dot2<f16_t>(a, b, c); }
inline float dot2_f32(bf16_t, unsigned a, unsigned b, float c) { return emu_dot2<bf16_t>(a, b, c); }
inline unsigned ones_pair(f16_t) { return 0x3C003C00u; }
inline unsigned ones_pair(bf16_t) { return 0x3F803F80u; }

inline long long clock_now() { static thread_local long long t = 0; return t += 64; }
inline long long wall_now() { return clock_now(); }
inline long long hw_id() { return 0; }
inline void spin_wall_ticks(int) {}
struct BufRsrc { const char* base; unsigned bytes; };
inline BufRsrc make_rsrc(const void* base, unsigned bytes) { return BufRsrc{static_cast<const char*>(base), bytes}; }
inline void async_copy16_buf(const BufRsrc& r, unsigned byte_offset, void* lds_wave_base) {
    char* dst = static_cast<char*>(lds_wave_base) + (emu::linear_tid() & 63) * 16;
    if ((unsigned long long)byte_offset + 16 <= r.bytes) std::memcpy(dst, r.base + byte_offset, 16);
    else std::memset(dst, 0, 16);
}
inline u32x4 buf_load16(const BufRsrc& r, unsigned byte_offset) {
    u32x4 v = u32x4{0u, 0u, 0u, 0u};
    if ((unsigned long long)byte_offset + 16 <= r.bytes) std::memcpy(&v, r.base + byte_offset, 16);
    return v;
}
inline u32x4 buf_load16_nt(const BufRsrc& r, unsigned byte_offset) { return buf_load16(r, byte_offset); }
inline void buf_store16(const BufRsrc& r, unsigned byte_offset, u32x4 v);
inline void buf_store16_nt(const BufRsrc& r, unsigned byte_offset, u32x4 v) { buf_store16(r, byte_offset, v); }
inline void buf_store16(const BufRsrc& r, unsigned byte_offset, u32x4 v) {
    if ((unsigned long long)byte_offset + 16 <= r.bytes) std::memcpy(const_cast<char*>(r.base) + byte_offset, &v, 16);
}
inline int wave_id() { return emu::linear_tid() >> 6; }
inline u32x2 lds_read_tr16_b64(const void* lds_ptr) {
    unsigned short r[4] = {0, 0, 0, 0};
    emu::wave_collective(&lds_ptr, [&](const std::vector<const void*>& s) {
        const int l = emu::linear_tid() & 63, g = l & ~15, a = (l & 15) >> 2, b = l & 3;
        for (int j = 0; j < 4; ++j) {
            const void* src = *static_cast<const void* const*>(s[g + 4 * j + a]);
            r[j] = static_cast<const unsigned short*>(src)[b];
        }
    });
    u32x2 out;
    out[0] = (unsigned)r[0] | ((unsigned)r[1] << 16);
    out[1] = (unsigned)r[2] | ((unsigned)r[3] << 16);
    return out;
}

template <int N>
inline void dma_wait() {}                       // the emulator's DMA is synchronous
inline void mem_wait_all() {}
inline void sched_fence() {}
inline void block_barrier() { emu::block_barrier(); }
inline void wave_sync() { emu::wave_barrier(); }

inline void lds_read16_async(u32x4& dst, const void* lds_ptr) { dst = *reinterpret_cast<const u32x4*>(lds_ptr); }
inline void lds_write16_async(void* lds_ptr, const u32x4& v) { *reinterpret_cast<u32x4*>(lds_ptr) = v; }
template <int N>
inline void lds_wait(u32x4&) {}
inline void lds_pin(u32x4&) {}
template <int OFF> inline void lds_read_tr16_b64_async(u32x2& dst, const void* lds_ptr) { dst = lds_read_tr16_b64(static_cast<const char*>(lds_ptr) + OFF); }
template <int OFF> inline void lds_read16_async_off(u32x4& dst, const void* lds_ptr) { dst = *reinterpret_cast<const u32x4*>(static_cast<const char*>(lds_ptr) + OFF); }
template <int N> inline void lds_wait2(u32x2&, u32x2&) {}

using std::fabs;
inline float fabsf_(float x) { return std::fabs(x); }

#define AA_X_ABLATE 0
// accumulator file of conv_gemm_x.h: plain storage here
constexpr int ACC_BLOCKS = 20;
struct AccFile { f32x16 blk[ACC_BLOCKS]; };
template <int B> inline void acc_zero(AccFile& af) { for (int e = 0; e < 16; ++e) af.blk[B][e] = 0.0f; }
template <int B> inline void acc_init(AccFile& af, const f32x16& v) { af.blk[B] = v; }
template <int B> inline void acc_mfma(AccFile& af, f16_t, const u32x4& w, const u32x4& a) { af.blk[B] = emu_mfma_32x32x16<f16_t>(w, a, af.blk[B]); }
template <int B> inline void acc_mfma(AccFile& af, bf16_t, const u32x4& w, const u32x4& a) { af.blk[B] = emu_mfma_32x32x16<bf16_t>(w, a, af.blk[B]); }
inline void acc_settle() {}
template <int B> inline f32x16 acc_get(AccFile& af) { return af.blk[B]; }
inline void lds_wait_all() {}
template <int P> inline void wave_priority() {}
template <int X>
inline void lds_read16_xor(u32x4& dst, const void* lds_ptr, IntTag<X>) {
    // (address ^ X) relative to the LDS base: the emulator's LDS buffer is 16-byte aligned only, so XOR the offset
    char* base = dyn_smem();
    const uintptr_t off = (uintptr_t)(static_cast<const char*>(lds_ptr) - base);
    dst = *reinterpret_cast<const u32x4*>(base + (off ^ (uintptr_t)X));
}
inline void async_copy16_buf_s(const BufRsrc& r, unsigned lane_offset, unsigned uniform_offset, void* lds_wave_base) {
    char* dst = static_cast<char*>(lds_wave_base) + (emu::linear_tid() & 63) * 16;
    if ((unsigned long long)lane_offset + 16 <= r.bytes) std::memcpy(dst, r.base + uniform_offset + lane_offset, 16);
    else std::memset(dst, 0, 16);
}

inline void im2col_offset(unsigned& pb, unsigned pix, unsigned c2, unsigned t, unsigned sh, unsigned invalid) {
    pb = ((pix & 0xffffffu) * (c2 & 0xffffffu) + t) | ((invalid << sh) & 0x80000000u);
}
