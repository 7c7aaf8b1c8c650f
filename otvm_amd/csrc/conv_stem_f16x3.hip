// 7x7 stride-2 "stem" convolutions (pad 3) with few input channels as a patch kernel on the f16 MFMA with split-fp16
// operands (f16x3): FBA encoder conv1 (11 -> 64 channels, FBA/models.py:66-85), STM Encoder_Q conv1 (3 -> 64) and the
// merged Encoder_M stem (22 -> 64, STM.py:56-66).
//
// The implicit-GEMM kernel (conv_f16x3.hip) runs these layers on its generic K decode (Cin % 32 != 0): per 32-wide
// K chunk every thread decodes (tap, channel) with integer divisions, gathers 4-channel quads, and re-splits every input
// element once per tap that touches it (49 / 4 = 12 times) -- 80..160 TFLOP/s.  Here a workgroup owns an 8 x 32 block
// of OUTPUT pixels (stride 2: a 21 x 69 input patch).  Per group of 8 input channels the patch is loaded and split ONCE
// into LDS (16 bytes of hi halves and 16 bytes of lo halves per pixel); the K dimension of the MFMA is (tap, channel):
// one 16-wide k-step covers two taps x 8 channels, lanes 0-31 of a wave read tap 2s, lanes 32-63 tap 2s+1 (49 taps ->
// 25 k-steps, the 50th tap carries zero weights).  Weights are packed at load time in MFMA B-fragment order
// (otvm_pack_stem_weight_f16x3: [channel group][k-step][n/32][hi|lo][lane][8 halfs], 1-KiB blocks) and go from L2
// straight into registers, one k-step ahead.  Epilogue as conv_patch_f16x3.hip (scale, bias, activation, optional fused
// GroupNorm statistics).
#include "common.h"
#include <type_traits>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

namespace {

struct StemArgs {
    const float* in; const _Float16* wf; const float* wscale; const float* bias; float* out; double* gn_stats;
    int H, W, Cin, in_ld, Ho, Wo, Cout, out_ld, act, groups;
    int tiles_x, tiles_y; OtvmTileWalk walk;
    int64_t in_bs, out_bs; int gn_bs;     // batch: image blockIdx.y lives *_bs elements behind image 0
    OtvmGnTail tail;                      // ABI 16: the output's GroupNorm table, written by the last workgroup (common.h)
};

constexpr int TH = 8, TW = 32, NW = 4, NT = NW * 64;
constexpr int TM = TH / NW, TN = 2;                        // 2 output rows x 64 channels per wave
constexpr int PH = 2 * TH + 5, PW = 2 * TW + 5, NPIX = PH * PW;          // 21 x 69 input pixels
constexpr int KSTEPS = 25;                                                // 49 taps, two per k-step

__device__ __forceinline__ void split4s(const f32x4 v, f16x4& hi, f16x4& lo) {
    typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
    const fp16x2 p01 = __builtin_amdgcn_cvt_pkrtz(v.x, v.y);
    const fp16x2 p23 = __builtin_amdgcn_cvt_pkrtz(v.z, v.w);
    const f16x2 h01 = __builtin_bit_cast(f16x2, p01);
    const f16x2 h23 = __builtin_bit_cast(f16x2, p23);
    hi = f16x4{h01.x, h01.y, h23.x, h23.y};
    lo = f16x4{(_Float16)(v.x - (float)h01.x), (_Float16)(v.y - (float)h01.y), (_Float16)(v.z - (float)h23.x),
               (_Float16)(v.w - (float)h23.y)};
}

__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_stem_f16x3_kernel(const StemArgs pa) {
    StemArgs p = pa;
    {
        const int zb = blockIdx.y;
        p.in += zb * p.in_bs;
        p.out += zb * p.out_bs;
        if (p.gn_stats) p.gn_stats += zb * p.gn_bs;
    }
    constexpr int PATCH_HALFS = 2 * NPIX * 8;                            // hi[NPIX][8] + lo[NPIX][8]
    constexpr int EPI_HALFS = NW * 32 * 36 * 2;
    constexpr int SM_HALFS = PATCH_HALFS > EPI_HALFS ? PATCH_HALFS : EPI_HALFS;
    __shared__ __attribute__((aligned(16))) _Float16 smem[SM_HALFS];
    _Float16* Ph = smem;
    _Float16* Pl = smem + NPIX * 8;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int tile_n_, tile_x, tile_y;
    otvm_tile_decode(p.walk, blockIdx.x, gridDim.x, 1, p.tiles_x, p.tiles_y, tile_n_, tile_x, tile_y);
    const int ty0 = tile_y * TH, tx0 = tile_x * TW;
    const int iy00 = 2 * ty0 - 3, ix00 = 2 * tx0 - 3;                    // input pixel of patch position (0, 0)

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    const int frow = lane & 31, fh = lane >> 5;
    // this lane's two taps per k-step are 2s + fh: their patch offsets advance irregularly (7 taps per row), so the
    // (ky, kx) walk is kept incrementally: tap -> tap + 2
    constexpr int NP = (NPIX * 2 + NT - 1) / NT;                         // (pixel, quad) items per thread per group
    for (int g = 0; g < p.groups; ++g) {
        __syncthreads();                                                 // the previous group's fragment reads are done
        // ---- stage the patch of channels 8g .. 8g+7: load, split, 8-byte LDS writes
#pragma unroll 4
        for (int k = 0; k < NP; ++k) {
            const int idx = tid + k * NT;
            if (idx < NPIX * 2) {
                const int pix = idx >> 1, q = idx & 1;
                const int py = pix / PW, px = pix - py * PW;
                const int iy = iy00 + py, ix = ix00 + px;
                const int c = g * 8 + q * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && c < p.Cin)
                    v = *reinterpret_cast<const f32x4*>(p.in + ((int64_t)iy * p.W + ix) * p.in_ld + c);
                f16x4 hi, lo;
                split4s(v, hi, lo);
                *reinterpret_cast<f16x4*>(&Ph[pix * 8 + q * 4]) = hi;
                *reinterpret_cast<f16x4*>(&Pl[pix * 8 + q * 4]) = lo;
            }
        }
        __syncthreads();
        // ---- 25 k-steps in 5 blocks of 5.  The weights of a whole block (20 x 1 KiB per wave, L2-resident) are fetched
        // into registers one block ahead: with a single k-step of lookahead every k-step (12 MFMAs, ~0.2 us) waited for
        // an L2 round trip (~1 us): 0.318 ms for the 24-channel stem at 1088x1920, 30 % of the MFMA rate.
        const _Float16* wg = p.wf + (int64_t)g * KSTEPS * TN * 2 * 512;
        constexpr int KB = 5;
        f16x8 wcur[KB][TN][2], wnxt[KB][TN][2];
#pragma unroll
        for (int u = 0; u < KB; ++u)
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                wcur[u][b][0] = *reinterpret_cast<const f16x8*>(wg + ((u * TN + b) * 2 + 0) * 512 + lane * 8);
                wcur[u][b][1] = *reinterpret_cast<const f16x8*>(wg + ((u * TN + b) * 2 + 1) * 512 + lane * 8);
            }
        int ky = 0, kx = fh;                                             // tap fh of k-step 0
#pragma unroll 1
        for (int blk = 0; blk < KSTEPS / KB; ++blk) {
            if (blk + 1 < KSTEPS / KB) {
#pragma unroll
                for (int u = 0; u < KB; ++u)
#pragma unroll
                    for (int b = 0; b < TN; ++b) {
                        const int sn = (blk + 1) * KB + u;
                        wnxt[u][b][0] = *reinterpret_cast<const f16x8*>(wg + ((sn * TN + b) * 2 + 0) * 512 + lane * 8);
                        wnxt[u][b][1] = *reinterpret_cast<const f16x8*>(wg + ((sn * TN + b) * 2 + 1) * 512 + lane * 8);
                    }
            }
#pragma unroll
            for (int u = 0; u < KB; ++u) {
                // tap 49 (k-step 24, upper half-wave) has zero weights: read tap 48's pixel, any finite value will do
                const int kyc = ky > 6 ? 6 : ky, kxc = ky > 6 ? 6 : kx;
                f16x8 ah[TM], al[TM];
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    const int o = ((2 * (wave * TM + a) + kyc) * PW + 2 * frow + kxc) * 8;
                    ah[a] = *reinterpret_cast<const f16x8*>(&Ph[o]);
                    al[a] = *reinterpret_cast<const f16x8*>(&Pl[o]);
                }
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int a = 0; a < TM; ++a)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[a], wcur[u][b][0], acc[a][b], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int a = 0; a < TM; ++a)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], wcur[u][b][1], acc[a][b], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int a = 0; a < TM; ++a)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], wcur[u][b][0], acc[a][b], 0, 0, 0);
                kx += 2;
                if (kx >= 7) { kx -= 7; ++ky; }
            }
#pragma unroll
            for (int u = 0; u < KB; ++u)
#pragma unroll
                for (int b = 0; b < TN; ++b) { wcur[u][b][0] = wnxt[u][b][0]; wcur[u][b][1] = wnxt[u][b][1]; }
        }
    }

    // ---- epilogue: accumulator tile -> wave-private LDS patch -> 16-byte row-major stores (see conv_f16x3.hip)
    const int col = lane & 31, rbase = (lane >> 5) * 4;
    __syncthreads();
    const bool vec_ok = ((p.out_ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.out) & 15) == 0);
    // interior blocks: epilogue copy without per-row predicates (no vmcnt(0) in front of every store, see conv_f16x3.hip)
    const bool interior = vec_ok && ty0 + TH <= p.Ho && tx0 + 32 <= p.Wo && TN * 32 <= p.Cout;    // workgroup-uniform
    // interior path specialised on the activation (no uniform branches between the stores), scale / bias fetched up front
    auto epilogue_full = [&](auto act_c) __attribute__((always_inline)) {
        constexpr int ACT = decltype(act_c)::value;
        float* patch = reinterpret_cast<float*>(smem) + wave * (32 * 36);
        const int prow = lane >> 3, pc = (lane & 7) * 4;
        f32x4 sc4[TN], bi4[TN];
#pragma unroll
        for (int b = 0; b < TN; ++b) {
#pragma unroll
            for (int j = 0; j < 4; ++j) sc4[b][j] = p.wscale[b * 32 + pc + j];
            bi4[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (p.bias) {
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j) bi4[b][j] = p.bias[b * 32 + pc + j];
        }
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int64_t m0r = (int64_t)(ty0 + wave * TM + a) * p.Wo + tx0 + prow;
#pragma unroll
                for (int e = 0; e < 16; ++e) patch[((e & 3) + 8 * (e >> 2) + rbase) * 36 + col] = acc[a][b][e];
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(&patch[(r4 * 8 + prow) * 36 + pc]);
                    v = v * sc4[b] + bi4[b];
                    v.x = otvm_act(v.x, ACT); v.y = otvm_act(v.y, ACT); v.z = otvm_act(v.z, ACT); v.w = otvm_act(v.w, ACT);
                    *reinterpret_cast<f32x4*>(p.out + (m0r + r4 * 8) * p.out_ld + b * 32 + pc) = v;
                }
            }
    };
    auto epilogue = [&](auto full_c) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_c)::value;
        float* patch = reinterpret_cast<float*>(smem) + wave * (32 * 36);
        const int prow = lane >> 3, pc = (lane & 7) * 4;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int n4 = b * 32 + pc;
            f32x4 sc4 = {0.f, 0.f, 0.f, 0.f}, bi4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (FULL || n4 + j < p.Cout) {
                    sc4[j] = p.wscale[n4 + j];
                    bi4[j] = p.bias ? p.bias[n4 + j] : 0.f;
                }
            }
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int y = ty0 + wave * TM + a;
#pragma unroll
                for (int e = 0; e < 16; ++e) patch[((e & 3) + 8 * (e >> 2) + rbase) * 36 + col] = acc[a][b][e];
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int xi = r4 * 8 + prow;
                    const int x = tx0 + xi;
                    f32x4 v = *reinterpret_cast<const f32x4*>(&patch[xi * 36 + pc]);
                    v = v * sc4 + bi4;
                    if (FULL || (y < p.Ho && x < p.Wo)) {
                        const int64_t m = (int64_t)y * p.Wo + x;
                        if (FULL || (vec_ok && n4 + 3 < p.Cout)) {
                            v.x = otvm_act(v.x, p.act); v.y = otvm_act(v.y, p.act);
                            v.z = otvm_act(v.z, p.act); v.w = otvm_act(v.w, p.act);
                            *reinterpret_cast<f32x4*>(p.out + m * p.out_ld + n4) = v;
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (n4 + j < p.Cout) p.out[m * p.out_ld + n4 + j] = otvm_act(v[j], p.act);
                        }
                    }
                }
            }
        }
    };
    if (interior) {
        if (p.act == OTVM_ACT_RELU) epilogue_full(std::integral_constant<int, OTVM_ACT_RELU>{});
        else if (p.act == OTVM_ACT_LEAKY) epilogue_full(std::integral_constant<int, OTVM_ACT_LEAKY>{});
        else epilogue_full(std::integral_constant<int, OTVM_ACT_NONE>{});
    } else {
        epilogue(std::false_type{});
    }
    // ---- fused GroupNorm statistics (sum / sum of squares per group, fp64 atomics), as conv_patch_f16x3.hip
    if (p.gn_stats) {
        __shared__ double gred[2 * 64];
        const int cg = p.Cout >> 5;
        const int seg = cg < 32 ? cg : 32;
        for (int i = tid; i < 2 * 64; i += NT) gred[i] = 0.0;
        __syncthreads();
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int n = b * 32 + col;
            float s = 0.f, ss = 0.f;
            if (n < p.Cout) {
                const float sc_ = p.wscale[n];
                const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    const int y = ty0 + wave * TM + a;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int x = tx0 + (e & 3) + 8 * (e >> 2) + rbase;
                        if (y < p.Ho && x < p.Wo) {
                            const float v = acc[a][b][e] * sc_ + bias;
                            s += v;
                            ss += v * v;
                        }
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
                const int gl = n / cg;
                atomicAdd(&gred[2 * gl], (double)s);
                atomicAdd(&gred[2 * gl + 1], (double)ss);
            }
        }
        __syncthreads();
        const int ng = (64 + cg - 1) / cg;
        for (int i = tid; i < 2 * ng; i += NT) {
            const int g = i >> 1;
            if (g < 32 && gred[i] != 0.0) atomicAdd(&p.gn_stats[2 * g + (i & 1)], gred[i]);
        }
        __syncthreads();
        otvm_gn_table_tail(p.gn_stats, (int64_t)p.Ho * p.Wo, p.Cout, p.tail, blockIdx.y, gridDim.x, reinterpret_cast<float*>(gred));
    }
}

// packed fp32 weight [O_pad][K_pad] (k = tap*I_pad + c, 49 taps) -> fragment-major split fp16 for the stem kernel
__global__ __launch_bounds__(256) void pack_stem_weight_kernel(const float* __restrict__ w, int O, int K_pad, int I_pad, int groups,
                                                               _Float16* __restrict__ wf, float* __restrict__ wscale) {
    const int n = blockIdx.x;                                            // output filter 0 .. 63
    __shared__ float red[256];
    const bool real = n < O;
    const float* row = w + (int64_t)n * K_pad;
    float mx = 0.f;
    if (real)
        for (int k = threadIdx.x; k < 49 * I_pad; k += 256) mx = fmaxf(mx, fabsf(row[k]));
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    int e = 0;
    if (red[0] > 0.f) frexpf(red[0], &e);
    const float inv = ldexpf(1.f, -e);
    if (threadIdx.x == 0 && real) wscale[n] = ldexpf(1.f, e);
    const int b = n >> 5, nl = n & 31;
    // every (group, k-step, half, channel-in-group) slot of this filter, zeros where there is no tap / channel
    for (int i = threadIdx.x; i < groups * KSTEPS * 2 * 8; i += 256) {
        const int j = i & 7, h = (i >> 3) & 1, s = (i >> 4) % KSTEPS, g = (i >> 4) / KSTEPS;
        const int tap = 2 * s + h, c = g * 8 + j;
        const float v = (real && tap < 49 && c < I_pad) ? row[tap * I_pad + c] * inv : 0.f;
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        const int l = nl + 32 * h;
        const int64_t blk = (((int64_t)g * KSTEPS + s) * TN + b) * 2;
        wf[(blk + 0) * 512 + l * 8 + j] = hi;
        wf[(blk + 1) * 512 + l * 8 + j] = lo;
    }
}

}  // namespace

extern "C" int64_t otvm_stem_weight_bytes_f16x3(int I_pad) {
    return (int64_t)((I_pad + 7) / 8) * KSTEPS * TN * 2 * 512 * sizeof(_Float16);
}

extern "C" int otvm_pack_stem_weight_f16x3(const float* w_packed, int O, int K_pad, int I_pad, void* w_frag, float* w_scale,
                                           void* stream) {
    OTVM_REQUIRE(w_packed && w_frag && w_scale && O <= 64 && I_pad % 4 == 0 && K_pad >= 49 * I_pad,
                 "otvm_pack_stem_weight_f16x3: needs O <= 64, I_pad %% 4 == 0 and a 7x7 packed weight");
    hipLaunchKernelGGL(pack_stem_weight_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, w_packed, O, K_pad, I_pad,
                       (I_pad + 7) / 8, (_Float16*)w_frag, w_scale);
    OTVM_CHECK_LAUNCH("otvm_pack_stem_weight_f16x3");
    return 0;
}

// shape-wise eligibility: 7x7, stride 2, pad 3, <= 64 output channels, fragment weights present, plain epilogue
int otvm_conv2d_stem_eligible(const otvm_conv_params* p) {
    return p->w_frag && p->kh == 7 && p->kw == 7 && p->stride == 2 && p->pad == 3 && p->dil == 1 && p->Cout <= 64 &&
                   p->Cin <= 64 && !p->residual && !p->in_relu && !p->in_scale ? 1 : 0;
}

int otvm_conv2d_stem_f16x3(const otvm_conv_params* p, void* stream) {
    StemArgs a;
    a.in = p->in; a.wf = (const _Float16*)p->w_frag; a.wscale = p->w_scale; a.bias = p->bias; a.out = p->out;
    a.gn_stats = p->gn_stats;
    a.tail = otvm_gn_tail_of(p);
    a.H = p->H; a.W = p->W; a.Cin = p->Cin; a.in_ld = p->in_ld; a.Ho = p->Ho; a.Wo = p->Wo; a.Cout = p->Cout;
    a.out_ld = p->out_ld; a.act = p->act; a.groups = (p->Cin + 7) / 8;
    a.tiles_x = otvm_ceil_div(p->Wo, TW);
    a.tiles_y = otvm_ceil_div(p->Ho, TH);
    a.walk = otvm_tile_walk_of(2);
    const int batch = p->batch > 1 ? p->batch : 1;
    a.in_bs = batch > 1 ? p->in_bs : 0; a.out_bs = batch > 1 ? p->out_bs : 0; a.gn_bs = batch > 1 ? p->gn_bs : 0;
    hipLaunchKernelGGL(conv_stem_f16x3_kernel, dim3(a.tiles_x * a.tiles_y, batch), dim3(NT), 0, (hipStream_t)stream, a);
    OTVM_CHECK_LAUNCH("otvm_conv2d(stem f16x3)");
    return 0;
}
