// Loss terms of the reference's TRAINING forward (models/alpha/model.py:100-187,189-312; utils/loss_func.py), forward only.
// Every kernel streams planar fp32 tensors once (HBM-bound) and leaves partial SUMS in fp64 accumulators (block reduction in
// LDS, one atomicAdd per block); the host divides by the element counts the reference's torch.mean / mse_loss use and adds
// the terms up (otvm_amd/train.py).  Images are planar: a tensor [N, H, W] holds N single-channel images (N = batch x
// frames x channels); `group = n % groups` selects the accumulator row where the reference keeps a statistic per frame.
#include "common.h"

namespace {

__device__ __forceinline__ void block_add(double v, double* dst, double* red) {
    // red: 256 doubles of LDS
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0 && red[0] != 0.0) atomicAdd(dst, red[0]);
    __syncthreads();
}

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

// scaled = in.flip(channel) * (1/255)  (model.py:59-60); in / out planar [N, 3, P]
__global__ __launch_bounds__(256) void scale_flip_kernel(const float* __restrict__ in, int64_t N, int64_t P, float s,
                                                         float* __restrict__ out) {
    const int64_t total = N * 3 * P;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / (3 * P), r = i - n * 3 * P;
        const int c = (int)(r / P);
        out[i] = in[n * 3 * P + (int64_t)(2 - c) * P + (r - (int64_t)c * P)] * s;
    }
}

// trimask = (argmax over [bg, un, fg] == 1) (model.py:42-44) + the class map for the cross-entropy; tri planar [N, 3, P]
__global__ __launch_bounds__(256) void trimask_kernel(const float* __restrict__ tri, int64_t N, int64_t P, float* __restrict__ mask,
                                                      unsigned char* __restrict__ cls, const float* __restrict__ gts,
                                                      float* __restrict__ vis) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N * P; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / P, p = i - n * P;
        const float a = tri[n * 3 * P + p], b = tri[n * 3 * P + P + p], c = tri[n * 3 * P + 2 * P + p];
        int k = 0;
        float m = a;
        if (b > m) { m = b; k = 1; }
        if (c > m) { k = 2; }
        mask[i] = k == 1 ? 1.f : 0.f;
        cls[i] = (unsigned char)k;
        if (vis) vis[i] = k == 1 ? 128.f * (1.f / 255.f) : gts[i];      // tris_vis (model.py:296-300)
    }
}

// fba_single_image_loss, the per-pixel part (model.py:117-150), over all (batch, frame) images n:
//   cF = where(trimask & gt > 0, predF, fgs), cB = where(trimask, predB, bgs), comp = cF a + cB (1 - a)
//   acc[0] += |a - gt|, acc[1] += |cF gt + cB (1 - gt) - img|, acc[2] += |fgs a + bgs (1 - a) - img|, acc[3] += |cF - fgs|,
//   acc[4] += |cB - bgs|
__global__ __launch_bounds__(256) void fba_comp_kernel(const float* __restrict__ pred7, const float* __restrict__ gt,
                                                       const float* __restrict__ tm, const float* __restrict__ fgs,
                                                       const float* __restrict__ bgs, const float* __restrict__ img, int64_t N,
                                                       int64_t P, float* __restrict__ cF, float* __restrict__ cB,
                                                       float* __restrict__ comp, float* __restrict__ alpha_out,
                                                       double* __restrict__ acc) {
    __shared__ double red[256];
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N * P; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / P, p = i - n * P;
        const float a = pred7[n * 7 * P + p], g = gt[i], m = tm[i];
        s0 += fabsf(a - g);
        alpha_out[i] = a;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int64_t j = n * 3 * P + (int64_t)c * P + p;
            const float f0 = fgs[j], b0 = bgs[j], im = img[j];
            const float f = (m != 0.f && g > 0.f) ? pred7[n * 7 * P + (int64_t)(1 + c) * P + p] : f0;
            const float b = (m != 0.f) ? pred7[n * 7 * P + (int64_t)(4 + c) * P + p] : b0;
            cF[j] = f;
            cB[j] = b;
            comp[j] = f * a + b * (1.f - a);
            s1 += fabsf(f * g + b * (1.f - g) - im);
            s2 += fabsf(f0 * a + b0 * (1.f - a) - im);
            s3 += fabsf(f - f0);
            s4 += fabsf(b - b0);
        }
    }
    block_add(s0, acc + 0, red); block_add(s1, acc + 1, red); block_add(s2, acc + 2, red);
    block_add(s3, acc + 3, red); block_add(s4, acc + 4, red);
}

__device__ __forceinline__ void grad_at(const float* __restrict__ im, int y, int x, int H, int W, float& gx, float& gy) {
    const float v = im[(int64_t)y * W + x];
    gx = x + 1 < W ? im[(int64_t)y * W + x + 1] - v : 0.f;        // loss_func.py:35-42
    gy = y + 1 < H ? im[(int64_t)(y + 1) * W + x] - v : 0.f;
}

// L1_grad (loss_func.py:44-51): acc[0] += | sqrt(gx^2 + gy^2 + eps)(x) - same(y) |; x, y planar [N, H, W]
__global__ __launch_bounds__(256) void grad_l1_kernel(const float* __restrict__ x, const float* __restrict__ y, int64_t N, int H, int W,
                                                      float eps, double* __restrict__ acc) {
    __shared__ double red[256];
    double s = 0;
    const int64_t P = (int64_t)H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N * P; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / P, p = i - n * P;
        const int yy = (int)(p / W), xx = (int)(p - (int64_t)yy * W);
        float ax, ay, bx, by;
        grad_at(x + n * P, yy, xx, H, W, ax, ay);
        grad_at(y + n * P, yy, xx, H, W, bx, by);
        s += fabsf(sqrtf(ax * ax + ay * ay + eps) - sqrtf(bx * bx + by * by + eps));
    }
    block_add(s, acc, red);
}

// exclusion_loss (loss_func.py:56-82), one pyramid level.  Images n = (b * S + c) * 3 + channel; frame c = (n / 3) % S.
// pass 1: acc1[c][0..3] += |gx1|, |gy1|, |gx2|, |gy2|   (the reference's global means per frame, over the batch)
__global__ __launch_bounds__(256) void excl_pass1_kernel(const float* __restrict__ i1, const float* __restrict__ i2, int64_t N, int H,
                                                         int W, int S, double* __restrict__ acc1) {
    __shared__ double red[256];
    const int64_t P = (int64_t)H * W;
    const int n = blockIdx.y;
    const int c = (n / 3) % S;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (int64_t)gridDim.x * blockDim.x) {
        const int yy = (int)(p / W), xx = (int)(p - (int64_t)yy * W);
        float ax, ay, bx, by;
        grad_at(i1 + n * P, yy, xx, H, W, ax, ay);
        grad_at(i2 + n * P, yy, xx, H, W, bx, by);
        s0 += fabsf(ax); s1 += fabsf(ay); s2 += fabsf(bx); s3 += fabsf(by);
    }
    block_add(s0, acc1 + c * 4 + 0, red); block_add(s1, acc1 + c * 4 + 1, red);
    block_add(s2, acc1 + c * 4 + 2, red); block_add(s3, acc1 + c * 4 + 3, red);
}

// pass 2: alphax = 2 mean|gx1| / (mean|gx2| + eps) (per frame), acc2[b * S + c][0..1] += (2 sig(gx1) - 1)^2 (2 sig(gx2 alphax) - 1)^2, y likewise
__global__ __launch_bounds__(256) void excl_pass2_kernel(const float* __restrict__ i1, const float* __restrict__ i2, int64_t N, int H,
                                                         int W, int S, int B, float eps, const double* __restrict__ acc1,
                                                         double* __restrict__ acc2) {
    __shared__ double red[256];
    const int64_t P = (int64_t)H * W;
    const int n = blockIdx.y;
    const int c = (n / 3) % S, bs = n / 3;
    const double cnt = (double)B * 3.0 * (double)P;                  // elements of a frame's [B, 3, H, W] gradient tensor
    const float ax_ = (float)(2.0 * (float)(acc1[c * 4 + 0] / cnt) / ((float)(acc1[c * 4 + 2] / cnt) + eps));
    const float ay_ = (float)(2.0 * (float)(acc1[c * 4 + 1] / cnt) / ((float)(acc1[c * 4 + 3] / cnt) + eps));
    double sx = 0, sy = 0;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (int64_t)gridDim.x * blockDim.x) {
        const int yy = (int)(p / W), xx = (int)(p - (int64_t)yy * W);
        float ax, ay, bx, by;
        grad_at(i1 + n * P, yy, xx, H, W, ax, ay);
        grad_at(i2 + n * P, yy, xx, H, W, bx, by);
        const float x1 = sigm(ax) * 2.f - 1.f, y1 = sigm(ay) * 2.f - 1.f;
        const float x2 = sigm(bx * ax_) * 2.f - 1.f, y2 = sigm(by * ay_) * 2.f - 1.f;
        sx += (x1 * x1) * (x2 * x2);
        sy += (y1 * y1) * (y2 * y2);
    }
    block_add(sx, acc2 + bs * 2 + 0, red);
    block_add(sy, acc2 + bs * 2 + 1, red);
}

// F.avg_pool2d(x, 2, 2) on planar [N, H, W] -> [N, H/2, W/2]
__global__ __launch_bounds__(256) void avgpool2_kernel(const float* __restrict__ x, int64_t N, int H, int W, float* __restrict__ y) {
    const int h2 = H / 2, w2 = W / 2;
    const int64_t P2 = (int64_t)h2 * w2, P = (int64_t)H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N * P2; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / P2, p = i - n * P2;
        const int yy = (int)(p / w2), xx = (int)(p - (int64_t)yy * w2);
        const float* s = x + n * P + (int64_t)(2 * yy) * W + 2 * xx;
        y[i] = (s[0] + s[1] + s[W] + s[W + 1]) * 0.25f;
    }
}

__device__ __forceinline__ int reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }
__constant__ float GK[5] = {1.f, 4.f, 6.f, 4.f, 1.f};

// LapLoss level, step 1 (loss_func.py:128-139): down = conv_gauss(cur)[::2, ::2] (reflect padding, kernel / 256)
__global__ __launch_bounds__(256) void lap_down_kernel(const float* __restrict__ cur, int64_t N, int H, int W, float* __restrict__ down) {
    const int h2 = H / 2, w2 = W / 2;
    const int64_t P2 = (int64_t)h2 * w2, P = (int64_t)H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N * P2; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / P2, p = i - n * P2;
        const int yy = 2 * (int)(p / w2), xx = 2 * (int)(p % w2);
        const float* s = cur + n * P;
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < 5; ++ky) {
            const int ry = reflect(yy + ky - 2, H);
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) acc += (GK[ky] * GK[kx] * (1.f / 256.f)) * s[(int64_t)ry * W + reflect(xx + kx - 2, W)];
        }
        down[i] = acc;
    }
}

// step 2: up = conv_gauss(zero-interleaved down, 4 K) (loss_func.py:111-121), diff = cur - up for image and target;
// acc[0] += weight * | diff_img - diff_tgt |
__global__ __launch_bounds__(256) void lap_diff_kernel(const float* __restrict__ cur_i, const float* __restrict__ down_i,
                                                       const float* __restrict__ cur_t, const float* __restrict__ down_t, int64_t N,
                                                       int H, int W, double weight, double* __restrict__ acc) {
    __shared__ double red[256];
    const int h2 = H / 2, w2 = W / 2;
    const int64_t P2 = (int64_t)h2 * w2, P = (int64_t)H * W;
    double s = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N * P; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / P, p = i - n * P;
        const int yy = (int)(p / W), xx = (int)(p - (int64_t)yy * W);
        float ui = 0.f, ut = 0.f;
#pragma unroll
        for (int ky = 0; ky < 5; ++ky) {
            const int ry = reflect(yy + ky - 2, H);
            if (ry & 1) continue;                                   // interleaved zeros
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) {
                const int rx = reflect(xx + kx - 2, W);
                if (rx & 1) continue;
                const float k = GK[ky] * GK[kx] * (4.f / 256.f);
                const int64_t j = n * P2 + (int64_t)(ry >> 1) * w2 + (rx >> 1);
                ui += k * down_i[j];
                ut += k * down_t[j];
            }
        }
        s += fabsf((cur_i[i] - ui) - (cur_t[i] - ut));
    }
    block_add(s * weight, acc, red);
}

// temporal term (model.py:177-182): acc[0] += ((x[b,t+1] - x[b,t]) - (y[b,t+1] - y[b,t]))^2 over b, t < S-1; planar [B, S, CP]
__global__ __launch_bounds__(256) void temporal_kernel(const float* __restrict__ x, const float* __restrict__ y, int B, int S, int64_t CP,
                                                       double* __restrict__ acc) {
    __shared__ double red[256];
    double s = 0;
    const int64_t total = (int64_t)B * (S - 1) * CP;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t bt = i / CP, q = i - bt * CP;
        const int64_t b = bt / (S - 1), t = bt - b * (S - 1);
        const int64_t j = (b * S + t) * CP + q;
        const float d = (x[j + CP] - x[j]) - (y[j + CP] - y[j]);
        s += (double)d * d;
    }
    block_add(s, acc, red);
}

// nn.CrossEntropyLoss over 3 classes (model.py:286-290): acc[0] += -log_softmax(logits)[cls]; logits planar [N, 3, P]
__global__ __launch_bounds__(256) void ce3_kernel(const float* __restrict__ lg, const unsigned char* __restrict__ cls, int64_t N, int64_t P,
                                                  double* __restrict__ acc) {
    __shared__ double red[256];
    double s = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N * P; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / P, p = i - n * P;
        const float a = lg[n * 3 * P + p], b = lg[n * 3 * P + P + p], c = lg[n * 3 * P + 2 * P + p];
        const float m = fmaxf(a, fmaxf(b, c));
        const float lse = m + logf(expf(a - m) + expf(b - m) + expf(c - m));
        const int k = cls[i];
        s += lse - (k == 0 ? a : (k == 1 ? b : c));
    }
    block_add(s, acc, red);
}

inline int lgrid(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

}  // namespace

#define LS (hipStream_t) stream

extern "C" int otvm_scale_flip3(const float* in, int64_t N, int64_t P, float s, float* out, void* stream) {
    OTVM_REQUIRE(in && out, "otvm_scale_flip3: null pointer");
    hipLaunchKernelGGL(scale_flip_kernel, dim3(lgrid(N * 3 * P)), dim3(256), 0, LS, in, N, P, s, out);
    OTVM_CHECK_LAUNCH("otvm_scale_flip3");
    return 0;
}

extern "C" int otvm_trimask(const float* tri, int64_t N, int64_t P, float* mask, unsigned char* cls, const float* gts, float* vis,
                            void* stream) {
    OTVM_REQUIRE(tri && mask && cls && (!vis || gts), "otvm_trimask: null pointer");
    hipLaunchKernelGGL(trimask_kernel, dim3(lgrid(N * P)), dim3(256), 0, LS, tri, N, P, mask, cls, gts, vis);
    OTVM_CHECK_LAUNCH("otvm_trimask");
    return 0;
}

extern "C" int otvm_loss_fba_comp(const float* pred7, const float* gt, const float* trimask, const float* fgs, const float* bgs,
                                  const float* img, int64_t N, int64_t P, float* cF, float* cB, float* comp, float* alpha_out,
                                  double* acc5, void* stream) {
    OTVM_REQUIRE(pred7 && gt && trimask && fgs && bgs && img && cF && cB && comp && alpha_out && acc5, "otvm_loss_fba_comp: null pointer");
    hipLaunchKernelGGL(fba_comp_kernel, dim3(lgrid(N * P)), dim3(256), 0, LS, pred7, gt, trimask, fgs, bgs, img, N, P, cF, cB, comp,
                       alpha_out, acc5);
    OTVM_CHECK_LAUNCH("otvm_loss_fba_comp");
    return 0;
}

extern "C" int otvm_loss_grad_l1(const float* x, const float* y, int64_t N, int H, int W, float eps, double* acc, void* stream) {
    OTVM_REQUIRE(x && y && acc, "otvm_loss_grad_l1: null pointer");
    hipLaunchKernelGGL(grad_l1_kernel, dim3(lgrid(N * H * W)), dim3(256), 0, LS, x, y, N, H, W, eps, acc);
    OTVM_CHECK_LAUNCH("otvm_loss_grad_l1");
    return 0;
}

extern "C" int otvm_loss_exclusion_level(const float* img1, const float* img2, int B, int S, int H, int W, float eps, double* acc1,
                                         double* acc2, void* stream) {
    OTVM_REQUIRE(img1 && img2 && acc1 && acc2 && B >= 1 && S >= 1, "otvm_loss_exclusion_level: bad arguments");
    const int64_t N = (int64_t)B * S * 3;
    OTVM_REQUIRE(N <= 65535, "otvm_loss_exclusion_level: too many images");
    int bx = (int)(((int64_t)H * W + 255) / 256);
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(excl_pass1_kernel, dim3(bx, (int)N), dim3(256), 0, LS, img1, img2, N, H, W, S, acc1);
    hipLaunchKernelGGL(excl_pass2_kernel, dim3(bx, (int)N), dim3(256), 0, LS, img1, img2, N, H, W, S, B, eps, (const double*)acc1, acc2);
    OTVM_CHECK_LAUNCH("otvm_loss_exclusion_level");
    return 0;
}

extern "C" int otvm_avgpool2(const float* x, int64_t N, int H, int W, float* y, void* stream) {
    OTVM_REQUIRE(x && y && H % 2 == 0 && W % 2 == 0, "otvm_avgpool2: even sizes only");
    hipLaunchKernelGGL(avgpool2_kernel, dim3(lgrid(N * (H / 2) * (W / 2))), dim3(256), 0, LS, x, N, H, W, y);
    OTVM_CHECK_LAUNCH("otvm_avgpool2");
    return 0;
}

extern "C" int otvm_loss_lap_level(const float* cur_img, const float* cur_tgt, int64_t N, int H, int W, double weight, float* down_img,
                                   float* down_tgt, double* acc, void* stream) {
    OTVM_REQUIRE(cur_img && cur_tgt && down_img && down_tgt && acc && H % 2 == 0 && W % 2 == 0 && H >= 4 && W >= 4,
                 "otvm_loss_lap_level: even sizes >= 4 only");
    hipLaunchKernelGGL(lap_down_kernel, dim3(lgrid(N * (H / 2) * (W / 2))), dim3(256), 0, LS, cur_img, N, H, W, down_img);
    hipLaunchKernelGGL(lap_down_kernel, dim3(lgrid(N * (H / 2) * (W / 2))), dim3(256), 0, LS, cur_tgt, N, H, W, down_tgt);
    hipLaunchKernelGGL(lap_diff_kernel, dim3(lgrid(N * H * W)), dim3(256), 0, LS, cur_img, (const float*)down_img, cur_tgt,
                       (const float*)down_tgt, N, H, W, weight, acc);
    OTVM_CHECK_LAUNCH("otvm_loss_lap_level");
    return 0;
}

extern "C" int otvm_loss_temporal(const float* x, const float* y, int B, int S, int64_t CP, double* acc, void* stream) {
    OTVM_REQUIRE(x && y && acc && S >= 2, "otvm_loss_temporal: needs at least two frames");
    hipLaunchKernelGGL(temporal_kernel, dim3(lgrid((int64_t)B * (S - 1) * CP)), dim3(256), 0, LS, x, y, B, S, CP, acc);
    OTVM_CHECK_LAUNCH("otvm_loss_temporal");
    return 0;
}

extern "C" int otvm_loss_ce3(const float* logits, const unsigned char* cls, int64_t N, int64_t P, double* acc, void* stream) {
    OTVM_REQUIRE(logits && cls && acc, "otvm_loss_ce3: null pointer");
    hipLaunchKernelGGL(ce3_kernel, dim3(lgrid(N * P)), dim3(256), 0, LS, logits, cls, N, P, acc);
    OTVM_CHECK_LAUNCH("otvm_loss_ce3");
    return 0;
}
