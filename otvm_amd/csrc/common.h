// Shared helpers for the OTVM gfx950 kernels (internal; the public ABI is include/otvm_hip.h).
#pragma once
#include <type_traits>
#include <utility>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/otvm_hip.h"

void otvm_set_error(const char* fmt, ...);

#define OTVM_CHECK_LAUNCH(name)                                                        \
    do {                                                                               \
        hipError_t e__ = hipGetLastError();                                            \
        if (e__ != hipSuccess) {                                                       \
            otvm_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
            return 2;                                                                  \
        }                                                                              \
    } while (0)

#define OTVM_REQUIRE(cond, ...)                                                        \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            otvm_set_error(__VA_ARGS__);                                               \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

static inline int otvm_ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}).  `#pragma unroll` is a
// request the unroller may decline for a large body; an accumulator array indexed by a loop that stayed rolled lives in
// scratch memory (found on the 4x4-tile kernel: 1 KB of scratch, 10x slower)
template <class F, int... I>
__device__ __forceinline__ void otvm_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void otvm_static_for(F&& f) {
    otvm_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ float otvm_act(float v, int act) {
    if (act == OTVM_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == OTVM_ACT_LEAKY) return v > 0.f ? v : 0.01f * v;
    return v;
}
