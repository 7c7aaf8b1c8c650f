// Shared helpers for the OTVM gfx950 kernels (internal; the public ABI is include/otvm_hip.h).
#pragma once
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

__device__ __forceinline__ float otvm_act(float v, int act) {
    if (act == OTVM_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == OTVM_ACT_LEAKY) return v > 0.f ? v : 0.01f * v;
    return v;
}

// ---- activation storage formats (include/otvm_hip.h OTVM_FMT_*) -----------------------------------------------------
// F32 : plain fp32, element e of a view at base[e].
// HL8 : "pre-split" fp32 for the f16x3 matrix-core kernels: every group of 8 consecutive elements (32 bytes, same
//       footprint as fp32) holds 8 fp16 "hi" halves (16 B) then 8 fp16 "lo" halves (16 B), x ~= hi + lo with
//       hi = fp16(x) rounded TOWARD ZERO and lo = fp16(x - hi) (22 significant bits; lo has the sign of x, so
//       relu(x) = (max(hi,0), max(lo,0))).  A conv kernel stages such a tensor into its LDS operand tiles with 16-byte
//       copies -- no fp32 -> (hi, lo) conversion per tap and per N tile.  Views need ld % 8 == 0 and a 32-byte aligned
//       origin; the element index arithmetic (pixel * ld + channel) is the same as for fp32.
typedef float otvm_f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 otvm_f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 otvm_f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void otvm_split4(const otvm_f32x4 v, otvm_f16x4& hi, otvm_f16x4& lo) {
    typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
    const fp16x2 p01 = __builtin_amdgcn_cvt_pkrtz(v.x, v.y);
    const fp16x2 p23 = __builtin_amdgcn_cvt_pkrtz(v.z, v.w);
    const otvm_f16x2 h01 = __builtin_bit_cast(otvm_f16x2, p01);
    const otvm_f16x2 h23 = __builtin_bit_cast(otvm_f16x2, p23);
    hi = otvm_f16x4{h01.x, h01.y, h23.x, h23.y};
    lo = otvm_f16x4{(_Float16)(v.x - (float)h01.x), (_Float16)(v.y - (float)h01.y), (_Float16)(v.z - (float)h23.x),
                    (_Float16)(v.w - (float)h23.y)};
}

// relu of a 16-byte piece of an HL8 tensor (8 fp16 halves, all hi or all lo): hi is rounded toward zero, so lo carries
// the sign of x (or is zero) and relu(x) = (max(hi,0), max(lo,0)).  On the bit patterns that is a signed 16-bit integer
// max with 0 (negative floats, incl. -0, have the sign bit set): one v_pk_max_i16 per two halves, no canonicalisation.
__device__ __forceinline__ otvm_f32x4 otvm_relu_hl8(const otvm_f32x4 v) {
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    s16x8 s = __builtin_bit_cast(s16x8, v);
    s = __builtin_elementwise_max(s, z);
    return __builtin_bit_cast(otvm_f32x4, s);
}

// byte offset of element e's hi half inside an HL8 view (the lo half is 16 bytes further)
__device__ __forceinline__ int64_t otvm_hl8_off(int64_t e) { return ((e >> 3) << 5) + ((e & 7) << 1); }

// four consecutive elements e .. e+3 (e % 4 == 0) of a view in format fmt
__device__ __forceinline__ otvm_f32x4 otvm_ld4(const float* base, int fmt, int64_t e) {
    if (fmt == OTVM_FMT_F32) return *reinterpret_cast<const otvm_f32x4*>(base + e);
    const char* p = reinterpret_cast<const char*>(base) + otvm_hl8_off(e);
    const otvm_f16x4 hi = *reinterpret_cast<const otvm_f16x4*>(p);
    const otvm_f16x4 lo = *reinterpret_cast<const otvm_f16x4*>(p + 16);
    return otvm_f32x4{(float)hi.x + (float)lo.x, (float)hi.y + (float)lo.y, (float)hi.z + (float)lo.z, (float)hi.w + (float)lo.w};
}

__device__ __forceinline__ void otvm_st4(float* base, int fmt, int64_t e, const otvm_f32x4 v) {
    if (fmt == OTVM_FMT_F32) {
        *reinterpret_cast<otvm_f32x4*>(base + e) = v;
        return;
    }
    otvm_f16x4 hi, lo;
    otvm_split4(v, hi, lo);
    char* p = reinterpret_cast<char*>(base) + otvm_hl8_off(e);
    *reinterpret_cast<otvm_f16x4*>(p) = hi;
    *reinterpret_cast<otvm_f16x4*>(p + 16) = lo;
}

// scalar access (tails, tests)
__device__ __forceinline__ float otvm_ld1(const float* base, int fmt, int64_t e) {
    if (fmt == OTVM_FMT_F32) return base[e];
    const char* p = reinterpret_cast<const char*>(base) + otvm_hl8_off(e);
    return (float)*reinterpret_cast<const _Float16*>(p) + (float)*reinterpret_cast<const _Float16*>(p + 16);
}

__device__ __forceinline__ void otvm_st1(float* base, int fmt, int64_t e, float v) {
    if (fmt == OTVM_FMT_F32) {
        base[e] = v;
        return;
    }
    typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
    const otvm_f16x2 h = __builtin_bit_cast(otvm_f16x2, (fp16x2)__builtin_amdgcn_cvt_pkrtz(v, 0.f));
    char* p = reinterpret_cast<char*>(base) + otvm_hl8_off(e);
    *reinterpret_cast<_Float16*>(p) = h.x;
    *reinterpret_cast<_Float16*>(p + 16) = (_Float16)(v - (float)h.x);
}

// GEN = false: the view is known to be fp32 (the common case keeps its plain 16-byte accesses, no format branch in the
// unrolled epilogues); GEN = true: dispatch on fmt at run time
template <bool GEN> __device__ __forceinline__ otvm_f32x4 otvm_ldq(const float* b, int fmt, int64_t e) {
    if constexpr (GEN) return otvm_ld4(b, fmt, e);
    else return *reinterpret_cast<const otvm_f32x4*>(b + e);
}
template <bool GEN> __device__ __forceinline__ void otvm_stq(float* b, int fmt, int64_t e, const otvm_f32x4 v) {
    if constexpr (GEN) otvm_st4(b, fmt, e, v);
    else *reinterpret_cast<otvm_f32x4*>(b + e) = v;
}
template <bool GEN> __device__ __forceinline__ float otvm_lds(const float* b, int fmt, int64_t e) {
    if constexpr (GEN) return otvm_ld1(b, fmt, e);
    else return b[e];
}
template <bool GEN> __device__ __forceinline__ void otvm_sts(float* b, int fmt, int64_t e, float v) {
    if constexpr (GEN) otvm_st1(b, fmt, e, v);
    else b[e] = v;
}

// host-side view check shared by the entry points
static inline bool otvm_view_ok(const void* p, int ld, int fmt) {
    if (fmt == OTVM_FMT_F32) return true;
    return fmt == OTVM_FMT_HL8 && (ld & 7) == 0 && (reinterpret_cast<uintptr_t>(p) & 31) == 0;
}
