// Shared helpers for the gfx950 kernels and the C ABI (no torch types anywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <type_traits>

#include "../../include/mi355_nnunet.h"

namespace mi355 {

void set_error(const char *fmt, ...);

// One process per GPU: the library binds to the HIP device that is current at its first compute call (weights, the
// activation arena, the scratch buffers below and the raised dynamic-LDS limits all live on / apply to that device)
// and every later entry point fails with MI355_ERR_INVALID if another device is current.  Also fails with
// MI355_ERR_NO_DEVICE when no gfx950 device is visible.
int bind_device();

// Process-wide scratch on the bound device, one buffer per slot, grown on demand (synchronise, free, allocate) under a
// mutex and never freed per call; `zeroed` buffers are cleared when (re)allocated.  Work that uses a slot is ordered
// by the caller's stream: one stream at a time per process ("Stream semantics").
enum ScratchSlot { SCR_ARENA = 0, SCR_SW_AGG, SCR_SPLITK_F32, SCR_SPLITK_F16, SCR_ZEROS, SCR_ZERO_BIAS, SCR_SMALL, SCR_TOPK,
                   SCR_CROP, SCR_ZSCORE, SCR_COUNT };
int device_scratch(int slot, size_t bytes, void **out, bool zeroed = false);

#define MI355_HIP(expr)                                                                    \
    do {                                                                                   \
        hipError_t e__ = (expr);                                                           \
        if (e__ != hipSuccess) {                                                           \
            ::mi355::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),     \
                               __FILE__, __LINE__);                                        \
            return MI355_ERR_HIP;                                                          \
        }                                                                                  \
    } while (0)

#define MI355_TRY(expr)            \
    do {                           \
        int rc__ = (expr);         \
        if (rc__ != MI355_OK)      \
            return rc__;           \
    } while (0)

#define MI355_REQUIRE(cond, ...)              \
    do {                                      \
        if (!(cond)) {                        \
            ::mi355::set_error(__VA_ARGS__);  \
            return MI355_ERR_INVALID;         \
        }                                     \
    } while (0)

// Exact unsigned division by a runtime constant for n < 2^31 (round-up method).
struct FastDiv {
    uint32_t d, m, s;
};
inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    uint32_t s = 0;
    while ((1u << s) < d)
        ++s;
    f.s = s;
    f.m = (uint32_t)(((uint64_t(1) << 32) * ((uint64_t(1) << s) - d)) / d + 1);
    return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv &f) {
    return (__umulhi(n, f.m) + n) >> f.s;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// Blocks b and b+8 share an XCD (round-robin dispatch): give each XCD one contiguous
// range of the logical tile list so halo re-reads hit that XCD's L2.  Bijective for any nwg.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Compile-time loop: f(std::integral_constant<int, I>{}) for I in [I0, N).  Indices are constant EXPRESSIONS inside the body
// (inline-asm immediates, register-array subscripts), whatever the optimiser decides about a "#pragma unroll" loop.
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

inline int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v)
        ++l;
    return l;
}
inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

}  // namespace mi355
