// Implicit-GEMM convolution for gfx950 on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// GEMM view:  M = Ho*Wo output pixels, N = Cout, K = kh*kw*Cin (NHWC: the Cin run of one tap is
// contiguous).  A[M,K] is gathered on the fly from the NHWC input (im2col never materialised),
// B[N,K] is the pre-packed weight (otvm_pack_conv_weight).  A 256-thread workgroup (4 wave64) owns a
// BM x BN output tile, K is walked in chunks of 32 through LDS; each wave owns TM x TN accumulator
// tiles of 32x32 (16 fp32 AGPRs each).  The fp32 MFMA is an exact fp32 fma chain (guide section 3),
// which is what lets this path meet the reference's 1e-3 fp32 contract.
//
// LDS rows are K-contiguous with a 4-float pad (144-byte stride): the 16 lanes of one ds_read_b128
// group then hit 16 distinct 16-byte slots (9*row mod 16 is a permutation) -> conflict free.
// K ordering inside a 8-wide group is permuted (lane half h supplies k = 8j+4h+i at MFMA step i) --
// identically for A and B, so the contraction is unchanged while every LDS read is a b128.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

struct ConvArgs {
    const float* in; const float* w; const float* bias; const float* residual; float* out; double* gn_stats;
    int H, W, Cin, in_ld, K_pad, res_ld, Ho, Wo, Cout, out_ld;
    int kh, kw, stride, pad, dil, in_relu, act;
    int M, taps, nchunks, tiles_m, tiles_n;
    int64_t in_bs, out_bs, res_bs; int gn_bs, batch;     // batch: image blockIdx.y lives *_bs elements behind image 0
};

constexpr int BK = 32;
constexpr int LDK = 36;

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_f32_kernel(const ConvArgs pa) {
    ConvArgs p = pa;
    {
        const int zb = blockIdx.y;
        p.in += zb * p.in_bs;
        p.out += zb * p.out_bs;
        if (p.residual) p.residual += zb * p.res_bs;
        if (p.gn_stats) p.gn_stats += zb * p.gn_bs;
    }
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_LD = BM / 32, B_LD = BN / 32;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(TM >= 1 && TN >= 1, "tile too small");
    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDK];
    float* As = smem;
    float* Bs = smem + BM * LDK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware, bijective remap: block b runs on XCD b%8; give each XCD a contiguous tile range so
    // neighbouring tiles (same weights, overlapping input halo) share one L2.
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tile_n = wgid % p.tiles_n, tile_m = wgid / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int lrow = tid >> 3, lk = (tid & 7) * 4;
    int iy0[A_LD], ix0[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        const int m = m0 + lrow + 32 * i;
        if (m < p.M) {
            const int oy = m / p.Wo, ox = m - oy * p.Wo;
            iy0[i] = oy * p.stride - p.pad;
            ix0[i] = ox * p.stride - p.pad;
        } else {
            iy0[i] = -(1 << 28);
            ix0[i] = -(1 << 28);
        }
    }
    const float* wrow0 = p.w + (int64_t)(n0 + lrow) * p.K_pad + lk;
    const int64_t wstep = (int64_t)32 * p.K_pad;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    f32x4 ra[A_LD], rb[B_LD];
    auto load_chunk = [&](int c) __attribute__((always_inline)) {
        const int kk = c * BK + lk;
        const int tap = kk / p.Cin;
        const int ci = kk - tap * p.Cin;
        const int ky = tap / p.kw, kx = tap - ky * p.kw;
        const int dy = ky * p.dil, dx = kx * p.dil;
        const bool tap_ok = tap < p.taps;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int iy = iy0[i] + dy, ix = ix0[i] + dx;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (tap_ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
                v = *reinterpret_cast<const f32x4*>(p.in + ((int64_t)iy * p.W + ix) * p.in_ld + ci);
                if (p.in_relu) {
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) rb[i] = *reinterpret_cast<const f32x4*>(wrow0 + i * wstep + c * BK);
    };
    auto store_chunk = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) *reinterpret_cast<f32x4*>(&As[(lrow + 32 * i) * LDK + lk]) = ra[i];
#pragma unroll
        for (int i = 0; i < B_LD; ++i) *reinterpret_cast<f32x4*>(&Bs[(lrow + 32 * i) * LDK + lk]) = rb[i];
    };

    load_chunk(0);
    const int frag_row = lane & 31, frag_k = (lane >> 5) * 4;
    for (int c = 0; c < p.nchunks; ++c) {
        __syncthreads();              // previous chunk's LDS reads are done
        store_chunk();
        __syncthreads();
        if (c + 1 < p.nchunks) load_chunk(c + 1);   // global loads in flight under the MFMAs
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a)
                af[a] = *reinterpret_cast<const f32x4*>(&As[((wm * TM + a) * 32 + frag_row) * LDK + 8 * j + frag_k]);
#pragma unroll
            for (int b = 0; b < TN; ++b)
                bf[b] = *reinterpret_cast<const f32x4*>(&Bs[((wn * TN + b) * 32 + frag_row) * LDK + 8 * j + frag_k]);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].x, bf[b].x, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].y, bf[b].y, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].z, bf[b].z, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].w, bf[b].w, acc[a][b], 0, 0, 0);
                }
        }
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
    const int col = lane & 31, rbase = (lane >> 5) * 4;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int n = n0 + (wn * TN + b) * 32 + col;
        if (n >= p.Cout) continue;
        const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int a = 0; a < TM; ++a) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + (wm * TM + a) * 32 + (e & 3) + 8 * (e >> 2) + rbase;
                if (m < p.M) {
                    float v = acc[a][b][e] + bias;
                    if (p.residual) v += p.residual[(int64_t)m * p.res_ld + n];
                    p.out[(int64_t)m * p.out_ld + n] = otvm_act(v, p.act);
                }
            }
        }
    }

    // ---- fused GroupNorm statistics of the tile just written (sum / sum of squares per group, fp64 atomics)
    if (p.gn_stats) {
        __shared__ double gred[2 * BN];                     // at most BN/2 groups per tile, (sum, sumsq) each
        const int cg = p.Cout >> 5;                         // channels per group (>= 2)
        const int seg = cg < 32 ? cg : 32;                  // lanes of one 32-column tile that share a group
        for (int i = threadIdx.x; i < 2 * BN; i += blockDim.x) gred[i] = 0.0;
        __syncthreads();
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int nl = (wn * TN + b) * 32 + col;        // column inside the tile
            const int n = n0 + nl;
            float s = 0.f, ss = 0.f;
            if (n < p.Cout) {
                const float sc_ = 1.f;
                const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int m = m0 + (wm * TM + a) * 32 + (e & 3) + 8 * (e >> 2) + rbase;
                        if (m < p.M) {
                            const float v = acc[a][b][e] * sc_ + bias;
                            s += v;
                            ss += v * v;
                        }
                    }
            }
            s += __shfl_xor(s, 32);
            ss += __shfl_xor(ss, 32);
            for (int off = 1; off < seg; off <<= 1) {
                s += __shfl_xor(s, off);
                ss += __shfl_xor(ss, off);
            }
            if (lane < 32 && (lane & (seg - 1)) == 0 && n < p.Cout) {
                const int gl = nl / cg;                     // group index local to the tile
                atomicAdd(&gred[2 * gl], (double)s);
                atomicAdd(&gred[2 * gl + 1], (double)ss);
            }
        }
        __syncthreads();
        const int ng = (BN + cg - 1) / cg;                  // groups touched by this tile (cg >= 32: BN/cg, else more)
        for (int i = threadIdx.x; i < 2 * ng; i += blockDim.x) {
            const int g = n0 / cg + (i >> 1);
            if (g < 32 && gred[i] != 0.0) atomicAdd(&p.gn_stats[2 * g + (i & 1)], gred[i]);
        }
    }
}

template <int BM, int BN, int WM, int WN>
int launch(ConvArgs& a, hipStream_t s) {
    a.tiles_m = otvm_ceil_div(a.M, BM);
    a.tiles_n = otvm_ceil_div(a.Cout, BN);
    const int grid = a.tiles_m * a.tiles_n;
    hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WM, WN>), dim3(grid, a.batch), dim3(256), 0, s, a);
    OTVM_CHECK_LAUNCH("otvm_conv2d");
    return 0;
}

// ---- weight packing (+ weight standardisation, + per-filter scale) -------------------------------
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, int O, int I, int kh, int kw,
                                                          int ws, const float* __restrict__ scale, float* out,
                                                          int I_pad, int K_pad) {
    const int o = blockIdx.x;
    float* row = out + (int64_t)o * K_pad;
    for (int k = threadIdx.x; k < K_pad; k += 256) row[k] = 0.f;
    if (o >= O) return;
    const int n = I * kh * kw;
    const float* src = w + (int64_t)o * n;
    __shared__ double red[256];
    __shared__ double s_mean, s_div;
    double mean = 0.0, div = 1.0;
    if (ws) {
        double acc = 0.0;
        for (int i = threadIdx.x; i < n; i += 256) acc += src[i];
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) s_mean = red[0] / n;
        __syncthreads();
        mean = s_mean;
        acc = 0.0;
        for (int i = threadIdx.x; i < n; i += 256) {
            const double d = (double)(float)(src[i] - (float)mean);     // centred weight, fp32 like the reference
            acc += d * d;
        }
        __syncthreads();
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        // the centred weights have (numerically) zero mean: unbiased var = sum d^2 / (n-1)
        if (threadIdx.x == 0) s_div = (double)(sqrtf((float)(red[0] / (n - 1)) + 1e-12f) + 1e-5f);
        __syncthreads();
        div = s_div;
    }
    __syncthreads();
    const float sc = scale ? scale[o] : 1.f;
    const int taps = kh * kw;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int c = i / taps, t = i - c * taps;        // OIHW: i = c*taps + t
        float v = src[i];
        if (ws) v = (v - (float)mean) / (float)div;
        row[t * I_pad + c] = v * sc;
    }
}

__global__ void fold_bn_kernel(const float* g, const float* b, const float* m, const float* v, float eps, int n, float* scale,
                               float* bias) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float s = g[i] / sqrtf(v[i] + eps);
        scale[i] = s;
        bias[i] = b[i] - m[i] * s;
    }
}

}  // namespace

extern "C" int otvm_fold_bn(const float* gamma, const float* beta, const float* mean, const float* var, float eps, int n,
                            float* scale, float* bias, void* stream) {
    hipLaunchKernelGGL(fold_bn_kernel, dim3(otvm_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta, mean, var,
                       eps, n, scale, bias);
    OTVM_CHECK_LAUNCH("otvm_fold_bn");
    return 0;
}

extern "C" int otvm_pack_conv_weight(const float* w_oihw, int O, int I, int kh, int kw, int ws, const float* scale,
                                     float* w_packed, int O_pad, int I_pad, int K_pad, void* stream) {
    OTVM_REQUIRE(I_pad >= I && O_pad >= O && K_pad >= kh * kw * I_pad && K_pad % 32 == 0,
                 "otvm_pack_conv_weight: bad padding (O=%d I=%d O_pad=%d I_pad=%d K_pad=%d)", O, I, O_pad, I_pad, K_pad);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(O_pad), dim3(256), 0, (hipStream_t)stream, w_oihw, O, I, kh, kw, ws,
                       scale, w_packed, I_pad, K_pad);
    OTVM_CHECK_LAUNCH("otvm_pack_conv_weight");
    return 0;
}

int otvm_conv2d_f16x3_impl(const otvm_conv_params* p, void* stream);   // conv_f16x3.hip

extern "C" int otvm_conv2d(const otvm_conv_params* p, void* stream) {
    OTVM_REQUIRE(p && p->in && p->out && (p->w || otvm_prec_is_split(p->precision)), "otvm_conv2d: null pointer");
    OTVM_REQUIRE(p->Cin % 4 == 0 && p->in_ld % 4 == 0, "otvm_conv2d: Cin (%d) and in_ld (%d) must be multiples of 4",
                 p->Cin, p->in_ld);
    OTVM_REQUIRE(((uintptr_t)p->in & 15) == 0 && ((uintptr_t)p->w & 15) == 0, "otvm_conv2d: in/w must be 16-byte aligned");
    OTVM_REQUIRE(p->precision == OTVM_PREC_F32 || otvm_prec_is_split(p->precision), "otvm_conv2d: unknown precision %d",
                 p->precision);
    // the in-tile group reduction works on power-of-two runs of channels: 32 groups of 2, 4, 8, ... channels
    OTVM_REQUIRE(!p->gn_stats || (p->Cout % 64 == 0 && ((p->Cout / 32) & (p->Cout / 32 - 1)) == 0 &&
                                  p->act == OTVM_ACT_NONE && !p->residual),
                 "otvm_conv2d: fused GroupNorm statistics need Cout = 64, 128, 256, ... (got %d), no activation, no residual",
                 p->Cout);
    OTVM_REQUIRE(p->K_pad % 32 == 0 && p->K_pad >= p->kh * p->kw * p->Cin, "otvm_conv2d: K_pad %d too small", p->K_pad);
    const int Ho = (p->H + 2 * p->pad - p->dil * (p->kh - 1) - 1) / p->stride + 1;
    const int Wo = (p->W + 2 * p->pad - p->dil * (p->kw - 1) - 1) / p->stride + 1;
    OTVM_REQUIRE(Ho == p->Ho && Wo == p->Wo, "otvm_conv2d: output size mismatch (%dx%d expected %dx%d)", p->Ho, p->Wo, Ho, Wo);
    OTVM_REQUIRE(!p->in_scale == !p->in_shift, "otvm_conv2d: in_scale and in_shift go together");
    OTVM_REQUIRE(!p->in_scale || otvm_conv2d_accepts_input_norm(p),
                 "otvm_conv2d: fused input normalisation requested for a layer otvm_conv2d_accepts_input_norm() rejects");
    OTVM_REQUIRE(!p->gn_scale_out || (p->gn_stats && p->gn_shift_out && p->gn_gamma && p->gn_beta && p->gn_counter),
                 "otvm_conv2d: gn_scale_out needs gn_stats, gn_shift_out, gn_gamma, gn_beta and gn_counter");
    if (otvm_prec_is_split(p->precision)) return otvm_conv2d_f16x3_impl(p, stream);
    if (p->gn_scale_out) {            // exact fp32: the kernels below leave the table to one more launch
        otvm_conv_params q = *p;
        q.gn_scale_out = nullptr;
        const int rc = otvm_conv2d(&q, stream);
        if (rc) return rc;
        const int nb = p->batch > 1 ? p->batch : 1;
        return otvm_gn_table_b(p->gn_stats, (int64_t)p->Ho * p->Wo, p->Cout, p->gn_gamma, p->gn_beta, p->gn_scale_out, p->gn_shift_out,
                               nb, nb > 1 ? p->gn_bs : 0, nb > 1 ? p->gn_tab_bs : 0, stream);
    }
    ConvArgs a;
    a.in = p->in; a.w = p->w; a.bias = p->bias; a.residual = p->residual; a.out = p->out; a.gn_stats = p->gn_stats;
    a.H = p->H; a.W = p->W; a.Cin = p->Cin; a.in_ld = p->in_ld; a.K_pad = p->K_pad; a.res_ld = p->res_ld;
    a.Ho = p->Ho; a.Wo = p->Wo; a.Cout = p->Cout; a.out_ld = p->out_ld;
    a.kh = p->kh; a.kw = p->kw; a.stride = p->stride; a.pad = p->pad; a.dil = p->dil;
    a.in_relu = p->in_relu; a.act = p->act;
    a.M = p->Ho * p->Wo; a.taps = p->kh * p->kw; a.nchunks = p->K_pad / 32;
    a.batch = p->batch > 1 ? p->batch : 1;
    a.in_bs = a.batch > 1 ? p->in_bs : 0; a.out_bs = a.batch > 1 ? p->out_bs : 0; a.res_bs = a.batch > 1 ? p->res_bs : 0;
    a.gn_bs = a.batch > 1 ? p->gn_bs : 0;
    hipStream_t s = (hipStream_t)stream;
    // Tile choice: weights are padded to 128 output rows, so any BN <= 128 is legal.  Prefer the big
    // tile; fall back to smaller ones when the launch would not fill 256 CUs or Cout is narrow.
    const int64_t M = a.M;
    if (p->Cout <= 32) return launch<256, 32, 4, 1>(a, s);
    if (p->Cout <= 64) return (M >= 256 * 128) ? launch<128, 64, 2, 2>(a, s) : launch<64, 64, 2, 2>(a, s);
    const int64_t big = (int64_t)otvm_ceil_div(M, 128) * otvm_ceil_div(p->Cout, 128);
    if (big >= 384) return launch<128, 128, 2, 2>(a, s);
    const int64_t mid = (int64_t)otvm_ceil_div(M, 128) * otvm_ceil_div(p->Cout, 64);
    if (mid >= 384) return launch<128, 64, 2, 2>(a, s);
    return launch<64, 64, 2, 2>(a, s);
}
