// GroupNorm(32 groups, eps 1e-5, affine) on NHWC fp32 activations -- HBM-bound streaming kernels.
//   stats : one pass, 16-byte loads, per-thread fp32 partials over a short pixel run, promoted to
//           fp64 for the block (LDS) and device (global atomic) reductions, so the result does not
//           depend on the reduction order beyond fp64 rounding.
//   apply : y = act((x - mean)*rstd*gamma + beta [+ residual]); per-channel scale/shift are built
//           once per block in LDS from the fp64 sums.
#include "common.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int G = 32;

__global__ void gn_stats_kernel(const float* __restrict__ x, int64_t P, int C, int ld, double* __restrict__ stats,
                                int64_t x_bs, int stats_bs) {
    x += blockIdx.y * x_bs;                       // image blockIdx.y
    stats += blockIdx.y * stats_bs;
    __shared__ double red[G * 2];
    const int Q = C >> 2;                         // float4 columns per pixel
    const int rows = blockDim.x / Q;              // pixels covered per block iteration
    const int q = threadIdx.x % Q, r = threadIdx.x / Q;
    if (threadIdx.x < G * 2) red[threadIdx.x] = 0.0;
    __syncthreads();
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, ss = {0.f, 0.f, 0.f, 0.f};
    if (r < rows) {
        for (int64_t pix = (int64_t)blockIdx.x * rows + r; pix < P; pix += (int64_t)gridDim.x * rows) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + pix * ld + q * 4);
            s += v;
            ss += v * v;
        }
    }
    const int cg = C / G;                         // channels per group (>= 2)
    if (r < rows) {
        if ((cg & 3) == 0) {                      // a float4 lies inside one group
            const int g = (q * 4) / cg;
            atomicAdd(&red[g * 2], (double)s.x + (double)s.y + (double)s.z + (double)s.w);
            atomicAdd(&red[g * 2 + 1], (double)ss.x + (double)ss.y + (double)ss.z + (double)ss.w);
        } else {                                  // cg = 2, 6, 10, ...: a float4 straddles groups -> per channel
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int g = (q * 4 + j) / cg;
                atomicAdd(&red[g * 2], (double)s[j]);
                atomicAdd(&red[g * 2 + 1], (double)ss[j]);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < G * 2) atomicAdd(&stats[threadIdx.x], red[threadIdx.x]);
}

// y = act((x - mean_g) * rstd_g * gamma_c + beta_c [+ residual]).  The grid stride (gridDim.x * 256 float4 items) is a
// multiple of the C/4 quads of a pixel (host side), so a thread keeps ONE channel quad for the whole launch: its
// scale/shift live in registers and the pixel index advances by a constant.  mean / rstd are derived from the fp64
// sums once per group per block (32 fp64 divisions and square roots, not C of them as in the first version).
// res_scale / res_shift (optional): the residual is itself a raw GroupNorm input whose apply pass was skipped (its
// consumers normalise on the fly): residual' = res_act(residual * res_scale[c] + res_shift[c]), table from otvm_gn_table.
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, int64_t P, int C, int ld,
                                                       const double* __restrict__ stats, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ residual,
                                                       int res_ld, const float* __restrict__ res_scale,
                                                       const float* __restrict__ res_shift, int res_act, int act,
                                                       float* __restrict__ out, int out_ld, int64_t x_bs, int64_t res_bs,
                                                       int64_t out_bs, int stats_bs, int norm_bs) {
    {   // image blockIdx.y
        const int zb = blockIdx.y;
        x += zb * x_bs;
        out += zb * out_bs;
        stats += zb * stats_bs;
        if (residual) residual += zb * res_bs;
        if (res_scale) { res_scale += zb * norm_bs; res_shift += zb * norm_bs; }
    }
    __shared__ float mean_s[G], rstd_s[G];
    const int cg = C / G;
    if (threadIdx.x < G) {
        const double cnt = (double)P * cg;
        const double mean = stats[threadIdx.x * 2] / cnt;
        double var = stats[threadIdx.x * 2 + 1] / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        mean_s[threadIdx.x] = (float)mean;
        rstd_s[threadIdx.x] = (float)(1.0 / sqrt(var + 1e-5));
    }
    __syncthreads();
    const int Q = C >> 2;
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;            // multiple of Q
    int64_t pix = i0 / Q;
    const int c = (int)(i0 - pix * Q) * 4;
    const int64_t dpix = stride / Q;
    f32x4 a, b;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int g = (c + j) / cg;
        a[j] = rstd_s[g] * gamma[c + j];
        b[j] = beta[c + j] - mean_s[g] * a[j];
    }
    // (measured and rejected, round 2: four independent pixel loads in flight per thread -- 52.3 vs 45.7 us per launch;
    //  round 3: the loop software-pipelined so that the loads of pixel i + 1 go out before the store of pixel i -- no wait
    //  on a store any more, vmcnt(2) instead of vmcnt(0) in the ISA -- with activation / residual mode as template
    //  parameters: 42.95 -> 42.79 frames/s at 1080p.  With 32 waves per CU the store acknowledgement is already hidden.)
    f32x4 ra = {1.f, 1.f, 1.f, 1.f}, rb = {0.f, 0.f, 0.f, 0.f};
    if (res_scale) {
        ra = *reinterpret_cast<const f32x4*>(res_scale + c);
        rb = *reinterpret_cast<const f32x4*>(res_shift + c);
    }
    for (; pix < P; pix += dpix) {
        f32x4 v = *reinterpret_cast<const f32x4*>(x + pix * ld + c);
        v = v * a + b;
        if (residual) {
            f32x4 r = *reinterpret_cast<const f32x4*>(residual + pix * res_ld + c);
            if (res_scale) {
                r = r * ra + rb;
                r.x = otvm_act(r.x, res_act); r.y = otvm_act(r.y, res_act); r.z = otvm_act(r.z, res_act); r.w = otvm_act(r.w, res_act);
            }
            v += r;
        }
        v.x = otvm_act(v.x, act); v.y = otvm_act(v.y, act); v.z = otvm_act(v.z, act); v.w = otvm_act(v.w, act);
        *reinterpret_cast<f32x4*>(out + pix * out_ld + c) = v;
    }
}

__global__ void gn_table_kernel(const double* __restrict__ stats, int64_t P, int C, const float* __restrict__ gamma,
                                const float* __restrict__ beta, float* __restrict__ scale, float* __restrict__ shift,
                                int stats_bs, int norm_bs) {
    stats += blockIdx.x * stats_bs;               // image blockIdx.x
    scale += blockIdx.x * norm_bs;
    shift += blockIdx.x * norm_bs;
    __shared__ float mean_s[G], rstd_s[G];
    const int cg = C / G;
    if (threadIdx.x < G) {
        const double cnt = (double)P * cg;
        const double mean = stats[threadIdx.x * 2] / cnt;
        double var = stats[threadIdx.x * 2 + 1] / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        mean_s[threadIdx.x] = (float)mean;
        rstd_s[threadIdx.x] = (float)(1.0 / sqrt(var + 1e-5));
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int g = c / cg;
        const float a = rstd_s[g] * gamma[c];
        scale[c] = a;
        shift[c] = beta[c] - mean_s[g] * a;
    }
}

}  // namespace

extern "C" int otvm_gn_table_b(const double* stats, int64_t P, int C, const float* gamma, const float* beta, float* scale,
                               float* shift, int batch, int stats_bs, int norm_bs, void* stream) {
    OTVM_REQUIRE(stats && gamma && beta && scale && shift && C % 32 == 0 && C <= 4096 && batch >= 1,
                 "otvm_gn_table: bad arguments (C=%d)", C);
    hipLaunchKernelGGL(gn_table_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, stats, P, C, gamma, beta, scale, shift,
                       stats_bs, norm_bs);
    OTVM_CHECK_LAUNCH("otvm_gn_table");
    return 0;
}

extern "C" int otvm_gn_table(const double* stats, int64_t P, int C, const float* gamma, const float* beta, float* scale,
                             float* shift, void* stream) {
    return otvm_gn_table_b(stats, P, C, gamma, beta, scale, shift, 1, 0, 0, stream);
}

extern "C" int otvm_gn_stats_b(const float* x, int64_t P, int C, int ld, double* stats, int batch, int64_t x_bs, int stats_bs,
                               void* stream) {
    OTVM_REQUIRE(C % 64 == 0 && C <= 2048, "otvm_gn_stats: C=%d unsupported (need multiple of 64, <= 2048)", C);
    OTVM_REQUIRE(ld % 4 == 0 && ((uintptr_t)x & 15) == 0 && x_bs % 4 == 0 && batch >= 1, "otvm_gn_stats: unaligned view");
    const int Q = C / 4;
    const int threads = Q <= 256 ? 256 : 512;
    const int rows = threads / Q;
    int64_t blocks = (P + rows - 1) / rows;
    // every block ends with 64 fp64 atomics on the same 64 addresses: 512 blocks instead of 2048 (480p, where this pass
    // follows the split-K layers three times per frame: 146.3 -> 148.6 frames/s; 1080p unchanged)
    static const int cap = otvm_probe_int("OTVM_GN_STATS_BLOCKS", 512);
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(gn_stats_kernel, dim3((int)blocks, batch), dim3(threads), 0, (hipStream_t)stream, x, P, C, ld, stats, x_bs,
                       stats_bs);
    OTVM_CHECK_LAUNCH("otvm_gn_stats");
    return 0;
}

extern "C" int otvm_gn_stats(const float* x, int64_t P, int C, int ld, double* stats, void* stream) {
    return otvm_gn_stats_b(x, P, C, ld, stats, 1, 0, 0, stream);
}

extern "C" int otvm_gn_apply_b(const otvm_gn_apply_params* q, void* stream) {
    OTVM_REQUIRE(q && q->x && q->out && q->stats && q->gamma && q->beta, "otvm_gn_apply: null pointer");
    const int C = q->C;
    OTVM_REQUIRE(C % 64 == 0 && C <= 2048, "otvm_gn_apply: C=%d unsupported", C);
    OTVM_REQUIRE(!q->res_scale == !q->res_shift && (!q->res_scale || q->residual),
                 "otvm_gn_apply: res_scale / res_shift go together, with a residual");
    OTVM_REQUIRE(q->ld % 4 == 0 && q->out_ld % 4 == 0 && (!q->residual || q->res_ld % 4 == 0), "otvm_gn_apply: unaligned view");
    const int batch = q->batch > 1 ? q->batch : 1;
    OTVM_REQUIRE(batch == 1 || (q->x_bs % 4 == 0 && q->out_bs % 4 == 0 && q->res_bs % 4 == 0 && q->norm_bs % 4 == 0),
                 "otvm_gn_apply: batch strides must be multiples of 4 elements");
    const int Q = C / 4;
    const int64_t total = q->P * Q;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    // the grid stride must be a multiple of Q (the kernel keeps one channel quad per thread): Q | blocks * 256
    int gcd = 256, r = Q;
    while (r) { const int t = gcd % r; gcd = r; r = t; }
    const int m = Q / gcd;                                   // smallest m with Q | 256 m
    blocks = (blocks + m - 1) / m * m;
    const bool b = batch > 1;
    hipLaunchKernelGGL(gn_apply_kernel, dim3((int)blocks, batch), dim3(256), 0, (hipStream_t)stream, q->x, q->P, C, q->ld, q->stats,
                       q->gamma, q->beta, q->residual, q->res_ld, q->res_scale, q->res_shift, q->res_act, q->act, q->out, q->out_ld,
                       b ? q->x_bs : 0, b ? q->res_bs : 0, b ? q->out_bs : 0, b ? q->stats_bs : 0, b ? q->norm_bs : 0);
    OTVM_CHECK_LAUNCH("otvm_gn_apply");
    return 0;
}

extern "C" int otvm_gn_apply(const float* x, int64_t P, int C, int ld, const double* stats, const float* gamma,
                             const float* beta, const float* residual, int res_ld, const float* res_scale,
                             const float* res_shift, int res_act, int act, float* out, int out_ld, void* stream) {
    otvm_gn_apply_params q;
    q.x = x; q.P = P; q.C = C; q.ld = ld; q.stats = stats; q.gamma = gamma; q.beta = beta; q.residual = residual; q.res_ld = res_ld;
    q.res_scale = res_scale; q.res_shift = res_shift; q.res_act = res_act; q.act = act; q.out = out; q.out_ld = out_ld;
    q.batch = 1; q.x_bs = q.res_bs = q.out_bs = 0; q.stats_bs = q.norm_bs = 0;
    return otvm_gn_apply_b(&q, stream);
}
