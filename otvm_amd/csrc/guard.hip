// Range guard of the recurrent state + a stream-ordered clear.
//
// f16x3 represents every fp32 operand as fp16 hi + lo: a value beyond fp16's range (|x| >= 65504) loses its low bits and,
// from 131008 on, turns into inf.  What must never happen silently is such a value (or a NaN / inf) entering the state that
// lives on from frame to frame -- the key / value maps memorised into the bank (STM.py:201-228), the 16-channel hidden
// state and the propagated trimap logits (alpha/model.py:432-471).  otvm_finite_guard scans such a tensor with one
// streaming pass (HBM-bound: 4 B per element read, nothing written) and records the smallest `tag` (the caller passes the
// frame number) at which an element failed |x| < limit in *flag (initialised to INT32_MAX by the caller); the host looks at
// the flag whenever it synchronises anyway.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__global__ __launch_bounds__(256) void finite_guard_kernel(const float* __restrict__ x, int64_t P, int C, int ld, float limit,
                                                           int tag, int* __restrict__ flag) {
    const int Q = C >> 2;                                            // float4 per pixel (C % 4 == 0, ld % 4 == 0: host check)
    const int64_t total = P * Q;
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / Q;
        const int c = (int)(i - m * Q) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + m * ld + c);
        // !(|v| < limit) is true for NaN as well
        bad = bad || !(fabsf(v.x) < limit) || !(fabsf(v.y) < limit) || !(fabsf(v.z) < limit) || !(fabsf(v.w) < limit);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicMin(flag, tag);
}

}  // namespace

extern "C" int otvm_finite_guard(const float* x, int64_t P, int C, int ld, float limit, int tag, int* flag, void* stream) {
    OTVM_REQUIRE(x && flag && P > 0 && C > 0, "otvm_finite_guard: bad arguments");
    OTVM_REQUIRE(C % 4 == 0 && ld % 4 == 0 && ((uintptr_t)x & 15) == 0, "otvm_finite_guard: view must be 16-byte aligned, C and ld multiples of 4");
    int64_t nb = (P * (C / 4) + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(finite_guard_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, x, P, C, ld, limit, tag, flag);
    OTVM_CHECK_LAUNCH("otvm_finite_guard");
    return 0;
}

extern "C" int otvm_clear(void* p, int64_t bytes, void* stream) {
    OTVM_REQUIRE(p && bytes >= 0, "otvm_clear: bad arguments");
    if (bytes == 0) return 0;
    const hipError_t e = hipMemsetAsync(p, 0, (size_t)bytes, (hipStream_t)stream);
    if (e != hipSuccess) {
        otvm_set_error("otvm_clear: hipMemsetAsync failed: %s", hipGetErrorString(e));
        return 2;
    }
    return 0;
}
