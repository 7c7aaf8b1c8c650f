// Matting metrics on device (reference utils/tmp/metric.py:177-189,252-264): SAD, MSE and dtSSD partial sums of
// one frame against its ground truth, accumulated over a clip.  Inputs are the 8-bit alphas the path produces
// (eval.py:209) -- all per-pixel terms are integers, so the fp64 sums are exact and order independent:
//   acc[0] += sum |p - t| * m                (SAD  = acc[0] / 255 / 1000)
//   acc[1] += sum (p - t)^2 * m              (MSE  = acc[1] / 255^2 / (acc[2] + 1))
//   acc[2] += sum m
//   acc[3] += sum ((p - p_prev) - (t - t_prev))^2 * m_prev      (dtSSD error^2 * 255^2 of the pair (prev, cur))
//   acc[4] += sum m_prev
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void metrics_kernel(const uint8_t* __restrict__ p, const uint8_t* __restrict__ t,
                                                      const uint8_t* __restrict__ m, const uint8_t* __restrict__ pp,
                                                      const uint8_t* __restrict__ tp, const uint8_t* __restrict__ mp, int64_t N,
                                                      double* __restrict__ acc) {
    unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        const int pv = p[i], tv = t[i], mv = m ? (m[i] != 0) : 1;
        const int d = pv - tv;
        s0 += (unsigned)(mv * (d < 0 ? -d : d));
        s1 += (unsigned)(mv * d * d);
        s2 += (unsigned)mv;
        if (pp) {
            const int mpv = mp ? (mp[i] != 0) : 1;
            const int e = (pv - (int)pp[i]) - (tv - (int)tp[i]);
            s3 += (unsigned)(mpv * e * e);
            s4 += (unsigned)mpv;
        }
    }
    __shared__ unsigned long long red[5][256];
    red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1; red[2][threadIdx.x] = s2; red[3][threadIdx.x] = s3; red[4][threadIdx.x] = s4;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s)
            for (int k = 0; k < 5; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x < 5) atomicAdd(&acc[threadIdx.x], (double)red[threadIdx.x][0]);
}

}  // namespace

extern "C" int otvm_matting_metrics(const uint8_t* pred, const uint8_t* target, const uint8_t* mask, const uint8_t* prev_pred,
                                    const uint8_t* prev_target, const uint8_t* prev_mask, int64_t n, double* acc, void* stream) {
    OTVM_REQUIRE(pred && target && acc && n > 0, "otvm_matting_metrics: bad arguments");
    OTVM_REQUIRE((prev_pred == nullptr) == (prev_target == nullptr), "otvm_matting_metrics: prev_pred and prev_target go together");
    int64_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(metrics_kernel, dim3((int)(nb > 1024 ? 1024 : nb)), dim3(256), 0, (hipStream_t)stream, pred, target, mask,
                       prev_pred, prev_target, prev_mask, n, acc);
    OTVM_CHECK_LAUNCH("otvm_matting_metrics");
    return 0;
}
