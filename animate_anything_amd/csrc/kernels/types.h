// Shared plain types for the gfx950 kernels (no HIP runtime dependency: this header is also
// read by the host-side SIMT emulator used in tests/emu).
#pragma once
#include <stdint.h>

typedef _Float16 f16_t;
typedef __bf16 bf16_t;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));     // packed fp32 arithmetic (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // one 16-byte global/LDS transaction
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// 8 half-precision values viewed either as a raw 16-byte word or as elements.
template <typename T>
union Pack8 {
    u32x4 raw;
    T e[8];
};

// compile-time integer carried as a type (lets a generic lambda take a constant)
template <int N>
struct IntTag { static constexpr int value = N; };
template <bool B>
struct BoolTag { static constexpr bool value = B; };
// f(IntTag<0>()), f(IntTag<1>()), ... f(IntTag<N-1>()): a loop whose index is a constant expression inside the body
#if defined(__HIPCC__)
#define AA_HD __host__ __device__
#else
#define AA_HD
#endif
template <int N, int I = 0, typename F>
AA_HD inline __attribute__((always_inline)) void static_for(F&& f) {
    if constexpr (I < N) { f(IntTag<I>()); static_for<N, I + 1>(f); }
}
