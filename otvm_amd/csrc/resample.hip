// Pooling / resampling kernels on NHWC fp32 (HBM-bound, 16-byte accesses along channels).
#include "common.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ f32x4 vmax(f32x4 a, f32x4 b) {
    f32x4 r = {fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)};
    return r;
}

// F.max_pool2d(kernel 3, stride 2, padding 1): padding acts as -inf
__global__ void maxpool3x3s2_kernel(const float* __restrict__ in, int H, int W, int C, int ld, float* __restrict__ out,
                                    int Ho, int Wo, int out_ld, int64_t in_bs, int64_t out_bs) {
    in += blockIdx.y * in_bs;                     // image blockIdx.y
    out += blockIdx.y * out_bs;
    const int Q = C >> 2;
    const int64_t total = (int64_t)Ho * Wo * Q;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i / Q;
        const int c = (int)(i - pix * Q) * 4;
        const int oy = (int)(pix / Wo), ox = (int)(pix - (int64_t)oy * Wo);
        const float ninf = -__builtin_huge_valf();
        f32x4 m = {ninf, ninf, ninf, ninf};
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int iy = oy * 2 + dy;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int ix = ox * 2 + dx;
                if ((unsigned)ix >= (unsigned)W) continue;
                m = vmax(m, *reinterpret_cast<const f32x4*>(in + ((int64_t)iy * W + ix) * ld + c));
            }
        }
        *reinterpret_cast<f32x4*>(out + pix * out_ld + c) = m;
    }
}

// F.interpolate(bilinear, align_corners=False): src = (dst + 0.5) * (in/out) - 0.5, clamped at 0;
// the upper neighbour index is clamped to the last row/col (PyTorch area_pixel_compute_source_index).
// One output row per blockIdx.y: the row terms (y0, y1, ly) are uniform, the column index is a 32-bit division by
// the channel-quad count (the flat 64-bit index of the first version cost two int64 divisions per float4).
// in_scale / in_shift (optional): the input is a raw GroupNorm input whose apply pass is folded in here -- every source
// pixel is normalised in' = in_act(in * in_scale[c] + in_shift[c]) (table from otvm_gn_table) before it is interpolated.
__global__ __launch_bounds__(256) void upsample_bilinear_kernel(const float* __restrict__ in, int Hi, int Wi, int C, int in_ld,
                                                                const float* __restrict__ in_scale,
                                                                const float* __restrict__ in_shift, int in_act,
                                                                const float* __restrict__ add, int add_ld,
                                                                float* __restrict__ out, int Ho, int Wo, int out_ld, float sy,
                                                                float sx, int64_t in_bs, int64_t add_bs, int64_t out_bs,
                                                                int norm_bs) {
    {   // image blockIdx.z
        const int zb = blockIdx.z;
        in += zb * in_bs;
        out += zb * out_bs;
        if (add) add += zb * add_bs;
        if (in_scale) { in_scale += zb * norm_bs; in_shift += zb * norm_bs; }
    }
    const int Q = C >> 2;
    const int oy = blockIdx.y;
    float fy = ((float)oy + 0.5f) * sy - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    const int y0 = (int)fy;
    const int y1 = y0 + (y0 < Hi - 1 ? 1 : 0);
    const float ly = fy - (float)y0, hy = 1.f - ly;
    const float* r0 = in + (int64_t)y0 * Wi * in_ld;
    const float* r1 = in + (int64_t)y1 * Wi * in_ld;
    const unsigned total = (unsigned)Wo * (unsigned)Q;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned ox = i / (unsigned)Q;
        const int c = (int)(i - ox * (unsigned)Q) * 4;
        float fx = ((float)ox + 0.5f) * sx - 0.5f;
        fx = fx < 0.f ? 0.f : fx;
        const int x0 = (int)fx;
        const int x1 = x0 + (x0 < Wi - 1 ? 1 : 0);
        const float lx = fx - (float)x0, hx = 1.f - lx;
        f32x4 v00 = *reinterpret_cast<const f32x4*>(r0 + (int64_t)x0 * in_ld + c);
        f32x4 v01 = *reinterpret_cast<const f32x4*>(r0 + (int64_t)x1 * in_ld + c);
        f32x4 v10 = *reinterpret_cast<const f32x4*>(r1 + (int64_t)x0 * in_ld + c);
        f32x4 v11 = *reinterpret_cast<const f32x4*>(r1 + (int64_t)x1 * in_ld + c);
        if (in_scale) {
            const f32x4 sa = *reinterpret_cast<const f32x4*>(in_scale + c), sb = *reinterpret_cast<const f32x4*>(in_shift + c);
            v00 = v00 * sa + sb; v01 = v01 * sa + sb; v10 = v10 * sa + sb; v11 = v11 * sa + sb;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v00[j] = otvm_act(v00[j], in_act); v01[j] = otvm_act(v01[j], in_act);
                v10[j] = otvm_act(v10[j], in_act); v11[j] = otvm_act(v11[j], in_act);
            }
        }
        f32x4 v = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
        const int64_t pix = (int64_t)oy * Wo + ox;
        if (add) v += *reinterpret_cast<const f32x4*>(add + pix * add_ld + c);
        *reinterpret_cast<f32x4*>(out + pix * out_ld + c) = v;
    }
}

// Exact x2 case (every decoder upsampling): output rows 2y+1 and 2y+2 read the same two input rows y, y+1 (weights
// 0.75/0.25 and 0.25/0.75), likewise the columns, so a thread that loads the 2x2 input cell (y..y+1, x..x+1) of a channel
// quad ONCE (and normalises it once) produces the four outputs it feeds (+ output row / column 0 from the first cell):
// 4 gathers and 4 normalisations per 4 outputs instead of 16 -- the general kernel above moves four times the output
// bytes through the vector-memory path (2.1 GB for the 544x960x256 map).  Same index / weight expressions and the same
// blend expression per output as the general kernel: bit-identical results.
__global__ __launch_bounds__(256) void upsample2x_bilinear_kernel(const float* __restrict__ in, int Hi, int Wi, int C, int in_ld,
                                                                  const float* __restrict__ in_scale,
                                                                  const float* __restrict__ in_shift, int in_act,
                                                                  const float* __restrict__ add, int add_ld,
                                                                  float* __restrict__ out, int out_ld, int64_t in_bs,
                                                                  int64_t add_bs, int64_t out_bs, int norm_bs) {
    {   // image blockIdx.z
        const int zb = blockIdx.z;
        in += zb * in_bs;
        out += zb * out_bs;
        if (add) add += zb * add_bs;
        if (in_scale) { in_scale += zb * norm_bs; in_shift += zb * norm_bs; }
    }
    const int Q = C >> 2, Ho = 2 * Hi, Wo = 2 * Wi;
    const int y = blockIdx.y;
    const int y1 = y + (y < Hi - 1 ? 1 : 0);
    const float* r0 = in + (int64_t)y * Wi * in_ld;
    const float* r1 = in + (int64_t)y1 * Wi * in_ld;
    const unsigned total = (unsigned)Wi * (unsigned)Q;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int x = (int)(i / (unsigned)Q);
        const int c = (int)(i - (unsigned)x * (unsigned)Q) * 4;
        const int x1 = x + (x < Wi - 1 ? 1 : 0);
        f32x4 v00 = *reinterpret_cast<const f32x4*>(r0 + (int64_t)x * in_ld + c);
        f32x4 v01 = *reinterpret_cast<const f32x4*>(r0 + (int64_t)x1 * in_ld + c);
        f32x4 v10 = *reinterpret_cast<const f32x4*>(r1 + (int64_t)x * in_ld + c);
        f32x4 v11 = *reinterpret_cast<const f32x4*>(r1 + (int64_t)x1 * in_ld + c);
        if (in_scale) {
            const f32x4 sa = *reinterpret_cast<const f32x4*>(in_scale + c), sb = *reinterpret_cast<const f32x4*>(in_shift + c);
            v00 = v00 * sa + sb; v01 = v01 * sa + sb; v10 = v10 * sa + sb; v11 = v11 * sa + sb;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v00[j] = otvm_act(v00[j], in_act); v01[j] = otvm_act(v01[j], in_act);
                v10[j] = otvm_act(v10[j], in_act); v11[j] = otvm_act(v11[j], in_act);
            }
        }
        // output rows fed by this cell: 2y+1, 2y+2 (and 0 from the first cell), columns likewise
#pragma unroll
        for (int ry = 0; ry < 3; ++ry) {
            const int oy = ry == 2 ? 0 : 2 * y + 1 + ry;
            if ((ry == 2 && y != 0) || oy >= Ho) continue;
            float fy = ((float)oy + 0.5f) * 0.5f - 0.5f;
            fy = fy < 0.f ? 0.f : fy;
            const float ly = fy - (float)(int)fy, hy = 1.f - ly;
#pragma unroll
            for (int rx = 0; rx < 3; ++rx) {
                const int ox = rx == 2 ? 0 : 2 * x + 1 + rx;
                if ((rx == 2 && x != 0) || ox >= Wo) continue;
                float fx = ((float)ox + 0.5f) * 0.5f - 0.5f;
                fx = fx < 0.f ? 0.f : fx;
                const float lx = fx - (float)(int)fx, hx = 1.f - lx;
                f32x4 v = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
                const int64_t pix = (int64_t)oy * Wo + ox;
                if (add) v += *reinterpret_cast<const f32x4*>(add + pix * add_ld + c);
                *reinterpret_cast<f32x4*>(out + pix * out_ld + c) = v;
            }
        }
    }
}

// AdaptiveAvgPool2d(s), s in {1,2,3,6}: bin i covers [floor(i*N/s), ceil((i+1)*N/s)) (neighbouring bins may share a
// row / column).  ONE pass over the map: a workgroup owns an image row and a 256-channel slab and produces the row's
// sums over the 12 column bins (1 + 2 + 3 + 6); a second, tiny kernel adds the rows of every bin in a fixed order.
// (The first version read the 2048-channel map once per scale: 4 x 267 MB, 0.33 ms at 1080p.)
constexpr int PPM_XBINS = 12;
constexpr int PPM_MAXSEG = 24;

__device__ __forceinline__ void ppm_scale(int bin, int& s, int& base, int& xbase) {
    if (bin < 1) { s = 1; base = 0; xbase = 0; }
    else if (bin < 5) { s = 2; base = 1; xbase = 1; }
    else if (bin < 14) { s = 3; base = 5; xbase = 3; }
    else { s = 6; base = 14; xbase = 6; }
}

// Round 4: the 12 column bins of a row overlap and nest, so the row is cut at every bin boundary into elementary segments
// (<= 23; at W = 240: six of 40 columns) -- a thread adds the pixels of a segment into ONE accumulator, without a
// predicate per bin (the first version tested every pixel against all 12 bins: ~130 VALU operations per 16-byte load, 99 us
// for a 267 MB read), and the segment's sum then goes to the bins that contain it (mask, wave-uniform).
struct PpmSegs { int nseg; int b[PPM_MAXSEG + 1]; unsigned short mask[PPM_MAXSEG]; };

__global__ __launch_bounds__(256) void ppm_pool_rows_kernel(const float* __restrict__ in, int H, int W, int C, int ld,
                                                            float* __restrict__ rowsum, const PpmSegs sg) {
    const int y = blockIdx.x;
    const int q = threadIdx.x & 63, lanep = threadIdx.x >> 6;
    const int c = blockIdx.y * 256 + q * 4;
    f32x4 acc[PPM_XBINS];
#pragma unroll
    for (int j = 0; j < PPM_XBINS; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        const float* row = in + (int64_t)y * W * ld + c;
        for (int i = 0; i < sg.nseg; ++i) {
            const int x1 = sg.b[i + 1];
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
            int x = sg.b[i] + lanep;
            for (; x + 12 < x1; x += 16) {                  // four independent loads in flight per thread
                a0 += *reinterpret_cast<const f32x4*>(row + (int64_t)x * ld);
                a1 += *reinterpret_cast<const f32x4*>(row + (int64_t)(x + 4) * ld);
                a2 += *reinterpret_cast<const f32x4*>(row + (int64_t)(x + 8) * ld);
                a3 += *reinterpret_cast<const f32x4*>(row + (int64_t)(x + 12) * ld);
            }
            for (; x < x1; x += 4) a0 += *reinterpret_cast<const f32x4*>(row + (int64_t)x * ld);
            const f32x4 t = (a0 + a1) + (a2 + a3);
            const unsigned m = sg.mask[i];
#pragma unroll
            for (int j = 0; j < PPM_XBINS; ++j)
                if ((m >> j) & 1u) acc[j] += t;
        }
    }
    __shared__ f32x4 red[PPM_XBINS][256];
#pragma unroll
    for (int j = 0; j < PPM_XBINS; ++j) red[j][threadIdx.x] = acc[j];
    __syncthreads();
    if (c < C) {
        for (int j = lanep; j < PPM_XBINS; j += 4) {
            const f32x4 t = (red[j][q] + red[j][q + 64]) + (red[j][q + 128] + red[j][q + 192]);
            *reinterpret_cast<f32x4*>(rowsum + ((int64_t)y * PPM_XBINS + j) * C + c) = t;
        }
    }
}

// rows of every bin added in a fixed order: four row phases per workgroup (independent loads in flight), then (r0 + r1) + (r2 + r3)
__global__ __launch_bounds__(256) void ppm_pool_final_kernel(const float* __restrict__ rowsum, int H, int W, int C, float* __restrict__ out) {
    const int bin = blockIdx.x;
    int s, base, xbase;
    ppm_scale(bin, s, base, xbase);
    const int b = bin - base, by = b / s, bx = b - by * s;
    const int y0 = (by * H) / s, y1 = ((by + 1) * H + s - 1) / s;
    const int x0 = (bx * W) / s, x1 = ((bx + 1) * W + s - 1) / s;
    const float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
    const int q = threadIdx.x & 63, rp = threadIdx.x >> 6;
    const int c = blockIdx.y * 256 + q * 4;
    __shared__ f32x4 red[256];
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    if (c < C)
        for (int y = y0 + rp; y < y1; y += 4) t += *reinterpret_cast<const f32x4*>(rowsum + ((int64_t)y * PPM_XBINS + xbase + bx) * C + c);
    red[threadIdx.x] = t;
    __syncthreads();
    if (rp == 0 && c < C)
        *reinterpret_cast<f32x4*>(out + (int64_t)bin * C + c) = ((red[q] + red[q + 64]) + (red[q + 128] + red[q + 192])) * inv;
}

}  // namespace

static int grid_for(int64_t total) {
    int64_t b = (total + 255) / 256;
    return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

extern "C" int otvm_maxpool3x3s2_b(const float* in, int H, int W, int C, int ld, float* out, int out_ld, int batch, int64_t in_bs,
                                   int64_t out_bs, void* stream) {
    OTVM_REQUIRE(C % 4 == 0 && ld % 4 == 0 && out_ld % 4 == 0, "otvm_maxpool3x3s2: channels must be multiples of 4");
    OTVM_REQUIRE(batch >= 1 && in_bs % 4 == 0 && out_bs % 4 == 0, "otvm_maxpool3x3s2: bad batch arguments");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid_for((int64_t)Ho * Wo * (C / 4)), batch), dim3(256), 0, (hipStream_t)stream,
                       in, H, W, C, ld, out, Ho, Wo, out_ld, in_bs, out_bs);
    OTVM_CHECK_LAUNCH("otvm_maxpool3x3s2");
    return 0;
}

extern "C" int otvm_maxpool3x3s2(const float* in, int H, int W, int C, int ld, float* out, int out_ld, void* stream) {
    return otvm_maxpool3x3s2_b(in, H, W, C, ld, out, out_ld, 1, 0, 0, stream);
}

extern "C" int otvm_upsample_bilinear_b(const float* in, int Hi, int Wi, int C, int in_ld, const float* in_scale,
                                        const float* in_shift, int in_act, const float* add, int add_ld,
                                        float* out, int Ho, int Wo, int out_ld, int batch, int64_t in_bs, int64_t add_bs,
                                        int64_t out_bs, int norm_bs, void* stream) {
    OTVM_REQUIRE(!in_scale == !in_shift, "otvm_upsample_bilinear: in_scale and in_shift go together");
    OTVM_REQUIRE(C % 4 == 0 && in_ld % 4 == 0 && out_ld % 4 == 0 && (!add || add_ld % 4 == 0),
                 "otvm_upsample_bilinear: channels must be multiples of 4");
    OTVM_REQUIRE(batch >= 1 && in_bs % 4 == 0 && add_bs % 4 == 0 && out_bs % 4 == 0 && norm_bs % 4 == 0,
                 "otvm_upsample_bilinear: bad batch arguments");
    const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
    OTVM_REQUIRE(Ho <= 65535 && (int64_t)Wo * (C / 4) < (1ll << 31), "otvm_upsample_bilinear: output %dx%d too large", Ho, Wo);
    static const int x2_on = otvm_probe_int("OTVM_UPSAMPLE2X", 1);
    if (x2_on && Ho == 2 * Hi && Wo == 2 * Wi && Hi >= 2 && Wi >= 2) {
        int bx2 = otvm_ceil_div(Wi * (C / 4), 256);
        if (bx2 > 64) bx2 = 64;
        hipLaunchKernelGGL(upsample2x_bilinear_kernel, dim3(bx2, Hi, batch), dim3(256), 0, (hipStream_t)stream, in, Hi, Wi, C, in_ld,
                           in_scale, in_shift, in_act, add, add_ld, out, out_ld, in_bs, add_bs, out_bs, norm_bs);
        OTVM_CHECK_LAUNCH("otvm_upsample_bilinear(x2)");
        return 0;
    }
    int bx = otvm_ceil_div(Wo * (C / 4), 256);
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(upsample_bilinear_kernel, dim3(bx, Ho, batch), dim3(256), 0, (hipStream_t)stream, in, Hi, Wi, C, in_ld,
                       in_scale, in_shift, in_act, add, add_ld, out, Ho, Wo, out_ld, sy, sx, in_bs, add_bs, out_bs, norm_bs);
    OTVM_CHECK_LAUNCH("otvm_upsample_bilinear");
    return 0;
}

extern "C" int otvm_upsample_bilinear(const float* in, int Hi, int Wi, int C, int in_ld, const float* in_scale,
                                      const float* in_shift, int in_act, const float* add, int add_ld,
                                      float* out, int Ho, int Wo, int out_ld, void* stream) {
    return otvm_upsample_bilinear_b(in, Hi, Wi, C, in_ld, in_scale, in_shift, in_act, add, add_ld, out, Ho, Wo, out_ld, 1, 0, 0, 0,
                                    0, stream);
}

extern "C" int64_t otvm_ppm_pool_ws_bytes(int H, int C) { return (int64_t)H * PPM_XBINS * C * sizeof(float); }

extern "C" int otvm_ppm_pool(const float* in, int H, int W, int C, int ld, float* out, void* ws, void* stream) {
    OTVM_REQUIRE(C % 4 == 0 && ld % 4 == 0 && ws, "otvm_ppm_pool: channels must be multiples of 4, ws required");
    // elementary segments of a row: cut at every boundary of the 12 column bins
    PpmSegs sg;
    int x0[PPM_XBINS], x1[PPM_XBINS], cuts[2 * PPM_XBINS], ncut = 0;
    {
        const int sc[4] = {1, 2, 3, 6};
        int j = 0;
        for (int k = 0; k < 4; ++k)
            for (int bx = 0; bx < sc[k]; ++bx, ++j) {
                x0[j] = (bx * W) / sc[k];
                x1[j] = ((bx + 1) * W + sc[k] - 1) / sc[k];
                cuts[ncut++] = x0[j];
                cuts[ncut++] = x1[j];
            }
    }
    for (int i = 1; i < ncut; ++i)                                  // insertion sort, then unique
        for (int k = i; k > 0 && cuts[k] < cuts[k - 1]; --k) { const int t = cuts[k]; cuts[k] = cuts[k - 1]; cuts[k - 1] = t; }
    int nu = 0;
    for (int i = 0; i < ncut; ++i)
        if (nu == 0 || cuts[i] != cuts[nu - 1]) cuts[nu++] = cuts[i];
    sg.nseg = nu - 1;
    OTVM_REQUIRE(sg.nseg >= 1 && sg.nseg <= PPM_MAXSEG, "otvm_ppm_pool: %d row segments", sg.nseg);
    for (int i = 0; i <= sg.nseg; ++i) sg.b[i] = cuts[i];
    for (int i = 0; i < sg.nseg; ++i) {
        unsigned m = 0;
        for (int j = 0; j < PPM_XBINS; ++j)
            if (x0[j] <= cuts[i] && cuts[i + 1] <= x1[j]) m |= 1u << j;
        sg.mask[i] = (unsigned short)m;
    }
    hipLaunchKernelGGL(ppm_pool_rows_kernel, dim3(H, otvm_ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, in, H, W, C,
                       ld, (float*)ws, sg);
    hipLaunchKernelGGL(ppm_pool_final_kernel, dim3(50, otvm_ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)ws, H, W, C, out);
    OTVM_CHECK_LAUNCH("otvm_ppm_pool");
    return 0;
}

// ---- the four PPM heads in ONE library call (FBA/models.py:298-307, 357-361): for each pooled map (1x1, 2x2, 3x3, 6x6 = 50
// pixels of 2048 channels) a 1x1 convolution to 256 channels (+bias), GroupNorm(32) over the map, LeakyReLU.  As separate
// conv / statistics / table launches that was 16 launches of a few microseconds of work each on a serial chain.
// Round 2 made it one launch of 128 workgroups (a workgroup = one GroupNorm group of one map: 70 us, the 6x6 map's 32
// workgroups walking 36 pixels with six lane shuffles per (pixel, channel)).  Round 4: the convolution is spread over the
// chip -- a workgroup = up to four pooled pixels of one branch x 16 output channels, a wave = four of those channels: every
// weight row is read once per workgroup with coalesced 16-byte loads (lane = four consecutive k) and used for the
// workgroup's pixels, whose 2048 values sit in registers; one wave reduction per (pixel, channel) -- and the normalisation
// (288 values per group at most) follows as a second, tiny kernel.  fp32 FMA arithmetic (exact products, fp32 accumulate):
// the same class as the matrix-core paths.
namespace {

struct PpmHeadArgs {
    const float* pooled; int K_pad;
    const float* w[4]; const float* bias[4]; const float* gamma[4]; const float* beta[4]; float* out[4];
    int out_ld, act;
};

constexpr int PPMH_GROUPS = 14;       // pixel groups: 1x1 -> {1}; 2x2 -> {4}; 3x3 -> {3,3,3}; 6x6 -> 9 x {4}
constexpr int PPMH_CB = 16;           // output channels per workgroup

__global__ __launch_bounds__(256) void ppm_head_conv_kernel(const PpmHeadArgs p) {
    constexpr int C = 2048, KJ = C / 256;                       // float4 per lane and row
    const int pg = blockIdx.x, cb = blockIdx.y;
    int br, px0, npx;
    if (pg < 1) { br = 0; px0 = 0; npx = 1; }
    else if (pg < 2) { br = 1; px0 = 0; npx = 4; }
    else if (pg < 5) { br = 2; px0 = (pg - 2) * 3; npx = 3; }
    else { br = 3; px0 = (pg - 5) * 4; npx = 4; }
    const int base = br == 0 ? 0 : (br == 1 ? 1 : (br == 2 ? 5 : 14));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 xv[4][KJ];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int px = q < npx ? px0 + q : px0;                  // tail: re-read the first pixel, results dropped
        const float* x = p.pooled + (int64_t)(base + px) * C + lane * 4;
#pragma unroll
        for (int j = 0; j < KJ; ++j) xv[q][j] = *reinterpret_cast<const f32x4*>(x + 256 * j);
    }
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
        const int c = cb * PPMH_CB + wave * 4 + ch;
        const float* wr = p.w[br] + (int64_t)c * p.K_pad + lane * 4;
        f32x4 wv[KJ];
#pragma unroll
        for (int j = 0; j < KJ; ++j) wv[j] = *reinterpret_cast<const f32x4*>(wr + 256 * j);
        float a[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < KJ; ++j) {
                t.x = fmaf(wv[j].x, xv[q][j].x, t.x); t.y = fmaf(wv[j].y, xv[q][j].y, t.y);
                t.z = fmaf(wv[j].z, xv[q][j].z, t.z); t.w = fmaf(wv[j].w, xv[q][j].w, t.w);
            }
            a[q] = (t.x + t.y) + (t.z + t.w);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] += __shfl_xor(a[q], off);
        if (lane < npx) {
            const float v = (lane == 0 ? a[0] : (lane == 1 ? a[1] : (lane == 2 ? a[2] : a[3]))) + (p.bias[br] ? p.bias[br][c] : 0.f);
            p.out[br][(int64_t)(px0 + lane) * p.out_ld + c] = v;
        }
    }
}

// GroupNorm(32) + activation of the four maps in place: a workgroup (one wave) = one group (8 channels) of one map
__global__ __launch_bounds__(64) void ppm_head_norm_kernel(const PpmHeadArgs p) {
    const int g = blockIdx.x, br = blockIdx.y;
    const int s = br == 0 ? 1 : (br == 1 ? 2 : (br == 2 ? 3 : 6));
    const int n = s * s * 8, lane = threadIdx.x;
    float v[5];                                                  // 288 values at most
    double sm = 0.0, sq = 0.0;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int i = lane + 64 * k;
        v[k] = i < n ? p.out[br][(int64_t)(i >> 3) * p.out_ld + g * 8 + (i & 7)] : 0.f;
        sm += (double)v[k];
        sq += (double)v[k] * (double)v[k];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sm += __shfl_xor(sm, off);
        sq += __shfl_xor(sq, off);
    }
    const double cnt = (double)n, mean = sm / cnt;
    double var = sq / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const float mean_s = (float)mean, rstd_s = (float)(1.0 / sqrt(var + 1e-5));
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int i = lane + 64 * k;
        if (i < n) {
            const int cc = g * 8 + (i & 7);
            const float a = rstd_s * p.gamma[br][cc];
            const float b = p.beta[br][cc] - mean_s * a;
            p.out[br][(int64_t)(i >> 3) * p.out_ld + cc] = otvm_act(v[k] * a + b, p.act);
        }
    }
}

}  // namespace

extern "C" int otvm_ppm_head(const otvm_ppm_head_params* q, void* stream) {
    OTVM_REQUIRE(q && q->pooled && q->C == 2048 && q->K_pad >= 2048 && q->Cout == 256,
                 "otvm_ppm_head: built for 2048 -> 256 channels (got %d -> %d)", q ? q->C : 0, q ? q->Cout : 0);
    OTVM_REQUIRE(q->K_pad % 4 == 0 && ((uintptr_t)q->pooled & 15) == 0, "otvm_ppm_head: pooled / weights must be 16-byte aligned");
    PpmHeadArgs a;
    a.pooled = q->pooled; a.K_pad = q->K_pad; a.out_ld = q->out_ld; a.act = q->act;
    for (int i = 0; i < 4; ++i) {
        OTVM_REQUIRE(q->w[i] && q->gamma[i] && q->beta[i] && q->out[i], "otvm_ppm_head: null pointer (branch %d)", i);
        OTVM_REQUIRE(((uintptr_t)q->w[i] & 15) == 0, "otvm_ppm_head: weights must be 16-byte aligned");
        a.w[i] = q->w[i]; a.bias[i] = q->bias[i]; a.gamma[i] = q->gamma[i]; a.beta[i] = q->beta[i]; a.out[i] = q->out[i];
    }
    hipLaunchKernelGGL(ppm_head_conv_kernel, dim3(PPMH_GROUPS, 256 / PPMH_CB), dim3(256), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(ppm_head_norm_kernel, dim3(32, 4), dim3(64), 0, (hipStream_t)stream, a);
    OTVM_CHECK_LAUNCH("otvm_ppm_head");
    return 0;
}


// ---------------------------------------------------------------------------------------------------------------------
// Round 3: the PPM branches never materialise at 1/8 resolution.
//
// FBA/models.py:358-365 upsamples the four pooled-and-projected maps (1x1, 2x2, 3x3, 6x6 pixels x 256 channels) to the
// layer-4 resolution, concatenates them with layer 4 (2048 + 4 x 256 = 3072 channels) and convolves the result with
// conv_up1.0 (3x3, 3072 -> 256): a third of that layer's 462 GFLOP at 1080p multiplies weights with bilinear interpolations
// of 50 pixels.  Convolution and interpolation are both linear, so the contribution of the PPM channels to output pixel p is
//      sum_tap [p + tap inside the image]  sum_scale  up_scale( Z[tap][scale] )(p + tap),
//      Z[tap][scale][j][o] = sum_c W[o][2048 + 256 scale + c][tap] * y_scale[j][c]        (j = one of the 50 pooled pixels)
// -- a 9 x 50 x 256 table of 256-long dot products (59 MFLOP instead of 154 GFLOP) followed by a gather of 9 taps x 4 scales
// x 4 bilinear neighbours per output value.  The zero padding of the convolution (taps outside the image contribute
// nothing) and PyTorch's bilinear index / weight arithmetic (align_corners = False) are reproduced exactly; only the order
// of the fp32 additions differs from the materialised form.  conv_up1.0 then runs on the 2048 layer-4 channels alone.
constexpr int PPMZ_BINS = 50, PPMZ_TAPS = 9;

// grid (50 pooled pixels, 9 taps), 256 threads = output channels; w: [4 scales][9 taps][256 c][256 o] fp32
__global__ __launch_bounds__(256) void ppm_z_kernel(const float* __restrict__ y0, const float* __restrict__ y1,
                                                    const float* __restrict__ y2, const float* __restrict__ y3, int y_ld,
                                                    const float* __restrict__ w, float* __restrict__ Z) {
    const int j = blockIdx.x, tap = blockIdx.y, o = threadIdx.x;
    const int sc = j < 1 ? 0 : (j < 5 ? 1 : (j < 14 ? 2 : 3));
    const int base = sc == 0 ? 0 : (sc == 1 ? 1 : (sc == 2 ? 5 : 14));
    const float* y = (sc == 0 ? y0 : (sc == 1 ? y1 : (sc == 2 ? y2 : y3))) + (int64_t)(j - base) * y_ld;
    __shared__ float ys[256];
    ys[o] = y[o];
    __syncthreads();
    const float* wp = w + ((int64_t)(sc * PPMZ_TAPS + tap) * 256) * 256 + o;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 4
    for (int c = 0; c < 256; c += 4) {
        a0 = fmaf(wp[(int64_t)(c + 0) * 256], ys[c + 0], a0);
        a1 = fmaf(wp[(int64_t)(c + 1) * 256], ys[c + 1], a1);
        a2 = fmaf(wp[(int64_t)(c + 2) * 256], ys[c + 2], a2);
        a3 = fmaf(wp[(int64_t)(c + 3) * 256], ys[c + 3], a3);
    }
    Z[((int64_t)tap * PPMZ_BINS + j) * 256 + o] = (a0 + a1) + (a2 + a3);
}

// out[p][o] += contribution(p)[o].
// Round 3's gather kept a 9 x 50 x 64 slice of Z in LDS and evaluated 9 taps x 4 scales x 4 bilinear neighbours per output value
// (144 LDS reads; 191 us at 136x240, 0.05 of HBM speed).  Bilinear interpolation is separable: for a fixed output ROW y the sum
// over the three filter rows and the two interpolation rows collapses the table to
//     R[kx][scale][j][o] = sum_ky [y + ky - 1 inside] sum_{i in {y0, y1}} wy_i * Z[(ky, kx)][scale][i][j][o]      (3 x 12 x channels)
// and a pixel of that row needs 3 filter columns x 4 scales x 2 interpolation columns = 24 reads.  A workgroup owns an image
// row and 128 channels: it builds its R (18 KB of LDS; Z comes from L2) and streams the row -- read, add, write, GroupNorm sums.
constexpr int PPMA_CH = 128;
__global__ __launch_bounds__(256) void ppm_add_kernel(const float* __restrict__ Z, int H, int W, float* __restrict__ out, int out_ld,
                                                      double* __restrict__ gn_stats) {
    __shared__ float R[3 * 12 * PPMA_CH];                       // [kx][column j of the 12 (scale, j) pairs][channel]
    __shared__ double gred[2 * (PPMA_CH / 8)];
    const int y = blockIdx.x, c0 = blockIdx.y * PPMA_CH, tid = threadIdx.x;
    // ---- the row table (18 entries per thread, three at a time: 18 independent L2 loads in flight)
#pragma unroll 3
    for (int i = tid; i < 3 * 12 * PPMA_CH; i += 256) {
        const int ch = i % PPMA_CH, kj = i / PPMA_CH, kx = kj / 12, jc = kj - kx * 12;
        const int sc = jc < 1 ? 0 : (jc < 3 ? 1 : (jc < 6 ? 2 : 3));
        const int s = sc == 0 ? 1 : (sc == 1 ? 2 : (sc == 2 ? 3 : 6));
        const int base = sc == 0 ? 0 : (sc == 1 ? 1 : (sc == 2 ? 5 : 14));
        const int j = jc - (sc == 0 ? 0 : (sc == 1 ? 1 : (sc == 2 ? 3 : 6)));
        const float sy = (float)s / (float)H;
        float za[3], zb[3], wa[3], wb[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            int qy = y + ky - 1;
            const bool in = (unsigned)qy < (unsigned)H;           // zero padding of the convolution
            qy = in ? qy : y;
            float fy = ((float)qy + 0.5f) * sy - 0.5f;           // F.interpolate(bilinear, align_corners=False)
            fy = fy < 0.f ? 0.f : fy;
            const int y0 = (int)fy, y1 = y0 + (y0 < s - 1 ? 1 : 0);
            const float ly = fy - (float)y0;
            const float* zt = Z + ((int64_t)(ky * 3 + kx) * PPMZ_BINS + base + j) * 256 + c0 + ch;
            za[ky] = zt[(int64_t)y0 * s * 256];
            zb[ky] = zt[(int64_t)y1 * s * 256];
            wa[ky] = in ? 1.f - ly : 0.f;
            wb[ky] = in ? ly : 0.f;
        }
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) acc += wa[ky] * za[ky] + wb[ky] * zb[ky];
        R[i] = acc;
    }
    if (tid < 2 * (PPMA_CH / 8)) gred[tid] = 0.0;
    // ---- the column terms of every source column qx = -1 .. W of the four scales, once per workgroup: (x0 | x1 << 16, lx);
    // a column outside the image (the convolution's zero padding) gets weight 0 on both neighbours
    extern __shared__ int2 xt[];                                 // [4][W + 2]
    for (int i = tid; i < 4 * (W + 2); i += 256) {
        const int sc = i / (W + 2), qx = i - sc * (W + 2) - 1;
        const int s = sc == 0 ? 1 : (sc == 1 ? 2 : (sc == 2 ? 3 : 6));
        float fx = ((float)qx + 0.5f) * ((float)s / (float)W) - 0.5f;   // F.interpolate(bilinear, align_corners=False)
        fx = fx < 0.f ? 0.f : fx;
        const int x0 = (int)fx, x1 = x0 + (x0 < s - 1 ? 1 : 0);
        const bool in = (unsigned)qx < (unsigned)W;
        xt[i] = int2{in ? (x0 | (x1 << 16)) : 0, in ? __float_as_int(fx - (float)x0) : (int)0x7fc00000};
    }
    __syncthreads();
    // ---- the row segment of this workgroup: thread = (channel, x phase), four consecutive pixels per step -- their four
    // loads go out together (one pixel per step was one exposed memory round trip per pixel: 120 us for a 67 MB pass)
    const int ch = tid % PPMA_CH, xp = tid / PPMA_CH;
    const int wseg = (W + gridDim.z - 1) / gridDim.z;
    const int xbeg = blockIdx.z * wseg, xend = xbeg + wseg < W ? xbeg + wseg : W;
    float gs = 0.f, gss = 0.f;
    float* orow = out + (int64_t)y * W * out_ld + c0 + ch;
    for (int xv = xbeg + xp * 4; xv < xend; xv += 4 * (256 / PPMA_CH)) {
        const int xb = __builtin_amdgcn_readfirstlane(xv);       // (the same for all lanes of a wave: scalar index arithmetic)
        float prev[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int x = xb + u < xend ? xb + u : xend - 1;
            prev[u] = orow[(int64_t)x * out_ld];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int x = xb + u;
            if (x >= xend) break;
            float acc = 0.f;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
                for (int sc = 0; sc < 4; ++sc) {
                    const int jb = sc == 0 ? 0 : (sc == 1 ? 1 : (sc == 2 ? 3 : 6));
                    const int2 e = xt[sc * (W + 2) + x + kx];    // source column qx = x + kx - 1
                    const int x0 = e.x & 0xffff, x1 = e.x >> 16;
                    const bool in = e.y != (int)0x7fc00000;
                    const float lx = in ? __int_as_float(e.y) : 0.f, hx = in ? 1.f - lx : 0.f;
                    const float* r = R + (kx * 12 + jb) * PPMA_CH + ch;
                    acc += hx * r[x0 * PPMA_CH] + lx * r[x1 * PPMA_CH];
                }
            }
            const float v = prev[u] + acc;
            orow[(int64_t)x * out_ld] = v;
            gs += v;
            gss += v * v;
        }
    }
    if (gn_stats) {
        // 256 channels in 32 groups of 8: lanes 8k .. 8k+7 share a group; per-thread partial sums in fp32 over <= W / 2 pixels,
        // promoted to fp64 for the workgroup (LDS) and device (atomic) reductions
        for (int off = 1; off < 8; off <<= 1) {
            gs += __shfl_xor(gs, off);
            gss += __shfl_xor(gss, off);
        }
        if ((tid & 7) == 0) {
            atomicAdd(&gred[2 * (ch >> 3)], (double)gs);
            atomicAdd(&gred[2 * (ch >> 3) + 1], (double)gss);
        }
        __syncthreads();
        if (tid < 2 * (PPMA_CH / 8)) atomicAdd(&gn_stats[2 * (c0 / 8 + (tid >> 1)) + (tid & 1)], gred[tid]);
    }
}

extern "C" int otvm_ppm_conv_z(const float* const* y, int y_ld, const float* w_ppm, float* Z, void* stream) {
    OTVM_REQUIRE(y && y[0] && y[1] && y[2] && y[3] && w_ppm && Z, "otvm_ppm_conv_z: null pointer");
    hipLaunchKernelGGL(ppm_z_kernel, dim3(PPMZ_BINS, PPMZ_TAPS), dim3(256), 0, (hipStream_t)stream, y[0], y[1], y[2], y[3], y_ld,
                       w_ppm, Z);
    OTVM_CHECK_LAUNCH("otvm_ppm_conv_z");
    return 0;
}

extern "C" int otvm_ppm_conv_add(const float* Z, int H, int W, float* out, int out_ld, double* gn_stats, void* stream) {
    OTVM_REQUIRE(Z && out && H > 0 && W > 0, "otvm_ppm_conv_add: bad arguments");
    OTVM_REQUIRE(W + 2 <= 4096, "otvm_ppm_conv_add: map too wide (%d)", W);
    hipLaunchKernelGGL(ppm_add_kernel, dim3(H, 256 / PPMA_CH, W >= 64 ? 2 : 1), dim3(256), 4 * (W + 2) * sizeof(int2),
                       (hipStream_t)stream, Z, H, W, out, out_ld, gn_stats);
    OTVM_CHECK_LAUNCH("otvm_ppm_conv_add");
    return 0;
}
