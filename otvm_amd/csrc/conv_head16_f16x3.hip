// The last 3x3 convolution of the FBA decoder / refinement -- 32 -> 16 channels, LeakyReLU (FBA/models.py:383-388, 425-432:
// conv_up4.2, pred.2) -- with the head behind it (conv_up4.4 / pred.4 + fba_fusion + softmax, head_math.h) on the 16-wide
// matrix-core tile (round 4).
//
// On the 32-wide tiles of conv_patch_f16x3.hip half of every MFMA of these two layers multiplies zero padding (16 real output
// channels in a 32-column tile): 0.18 ms each at 0.12 of the MFMA peak and 0.27 of HBM speed -- bound by neither.
// v_mfma_f32_16x16x32_f16 fits the layer exactly: M = 16 pixels, N = the 16 output channels, K = ALL 32 input channels of one
// filter tap, at the same FLOP rate as the 32x32x16 form -- 12 MFMAs of ~16 cycles per tap and wave instead of 12 of 32.
//   * a workgroup owns an 8 x 32 block of output pixels (4 waves x 2 image rows x two 16-pixel tiles); the 10 x 34 x 32-channel
//     input patch is loaded ONCE, split into fp16 hi / lo (f16x3: hi*hi + hi*lo + lo*hi) and kept in LDS with 80-byte pixel
//     rows; the nine taps read shifted windows of it (lane l: pixel l & 15 of its tile, channels 8 (l >> 4) .. + 7);
//   * the weights (9 taps x hi / lo x 1 KiB in fragment order, otvm_pack_head16_weight_f16x3) live in 72 registers per lane for
//     the whole kernel: no weight stage in LDS, no barrier between taps;
//   * epilogue as the HEAD variant of the patch kernel: accumulators -> [pixel][channel] rows in LDS, one lane per pixel: filter
//     scale, bias, LeakyReLU, the hidden state (optional), then the head on 16 registers.
#include "common.h"
#include "head_math.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

namespace {

struct Head16Args {
    const float* in; const _Float16* wf; const float* wscale; const float* bias; float* out;
    int H, W, in_ld, out_ld, act;
    unsigned in_bytes;                   // size of one image's input view (buffer-resource range)
    int tiles_x, tiles_y; OtvmTileWalk walk;
    int64_t in_bs, out_bs;
    OtvmHeadArgs head; int64_t head_img_bs, head_alpha_bs, head_tri_bs, head_sm_bs;
};

constexpr int TH = 8, TW = 32, PH = TH + 2, PW = TW + 2, NPIX = PH * PW;
constexpr int LDP = 40;                  // halfs per patch pixel: 32 channels + 8 pad = 80 bytes
constexpr int EPL = 20;                  // floats per epilogue row: 16 channels + 4 pad

__device__ __forceinline__ void split4h(const f32x4 v, f16x4& hi, f16x4& lo) {
    typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
    const fp16x2 p01 = __builtin_amdgcn_cvt_pkrtz(v.x, v.y);
    const fp16x2 p23 = __builtin_amdgcn_cvt_pkrtz(v.z, v.w);
    const f16x2 h01 = __builtin_bit_cast(f16x2, p01);
    const f16x2 h23 = __builtin_bit_cast(f16x2, p23);
    hi = f16x4{h01.x, h01.y, h23.x, h23.y};
    lo = f16x4{(_Float16)(v.x - (float)h01.x), (_Float16)(v.y - (float)h01.y), (_Float16)(v.z - (float)h23.x),
               (_Float16)(v.w - (float)h23.y)};
}

// timing probes for tools/build_variant.sh (the results are WRONG with any of them set): which part of the kernel costs what
#ifndef OTVM_H16_NOPATCH
#define OTVM_H16_NOPATCH 0     // no global loads of the input patch (LDS is filled from constants)
#endif
#ifndef OTVM_H16_NOW
#define OTVM_H16_NOW 0         // no weight loads
#endif
#ifndef OTVM_H16_NOMFMA
#define OTVM_H16_NOMFMA 0      // no fragment reads, no MFMAs
#endif
#ifndef OTVM_H16_NOHEAD
#define OTVM_H16_NOHEAD 0      // no per-pixel head: alpha = the first hidden value
#endif
#ifndef OTVM_H16_NOSTORE
#define OTVM_H16_NOSTORE 0     // the head is evaluated, nothing is stored but alpha
#endif
#ifndef OTVM_HEAD16_WGS
#define OTVM_HEAD16_WGS 3                // workgroups per CU the register budget is set for (LDS: 3 x 54 400 B fit 160 KiB)
#endif
// N_OUT = 7 / 10: the head's width (its loops unrolled without run-time tests); FAST = the layer as the frame issues it --
// LeakyReLU and a bias -- with the epilogue's constants (filter scales, bias) read as whole vectors; the generic form keeps the
// run-time activation switch.
// Round 5 (tools/head_bench.py, profiles/r05_head16_ablation.txt): the per-pixel head cost 67 of the launch's 174 us, and not for
// its arithmetic -- its 170 weights were fetched by (wave-uniform) GLOBAL loads inside the epilogue, each batch waited for with
// vmcnt(0) right where it was issued, the pixel's RGB likewise, and the sixteen filter scales / biases one s_load + wait + branch
// each (the run-time `act` switch and `if (bias)` per element kept the compiler from batching them).  Now: every thread fetches
// ONE of the head's weights and its pixel's RGB at the top of the kernel (the latency passes under the patch load and the tap
// loop), the weights go through 680 bytes of LDS behind the epilogue rows and are read as broadcast ds_read_b128.
template <int N_OUT, bool FAST>
__global__ __launch_bounds__(256, OTVM_HEAD16_WGS) void conv_head16_f16x3_kernel(const Head16Args pa) {
    Head16Args p = pa;
    {
        const int zb = blockIdx.y;
        p.in += zb * p.in_bs;
        if (p.out) p.out += zb * p.out_bs;
        p.head.img += zb * p.head_img_bs;
        if (p.head.alpha_out) p.head.alpha_out += zb * p.head_alpha_bs;
        if (p.head.tri_out) p.head.tri_out += zb * p.head_tri_bs;
        if (p.head.sm) p.head.sm += zb * p.head_sm_bs;
    }
    constexpr int PATCH_HALFS = 2 * NPIX * LDP;                  // hi + lo
    constexpr int EPI_HALFS = 4 * 2 * 32 * EPL * 2;              // four waves x two rows x 32 pixels (fp32, in halfs)
    __shared__ __attribute__((aligned(16))) _Float16 smem[PATCH_HALFS > EPI_HALFS ? PATCH_HALFS : EPI_HALFS];
    _Float16* Ph = smem;
    _Float16* Pl = smem + NPIX * LDP;
    static_assert(EPI_HALFS + 2 * (N_OUT * 17 + 3) <= PATCH_HALFS, "the head's weights live behind the epilogue rows");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int tile_n_, tile_x, tile_y;
    otvm_tile_decode(p.walk, blockIdx.x, gridDim.x, 1, p.tiles_x, p.tiles_y, tile_n_, tile_x, tile_y);
    const int ty0 = tile_y * TH, tx0 = tile_x * TW;
    // ---- what the epilogue needs from global memory, requested first: one value of the head's [N_OUT][16] weights + [N_OUT] bias
    // per thread, and the RGB of the pixel this lane finishes (lanes 0-31: the wave's first image row, 32-63: the second)
    const int ea = lane >> 5, epx = lane & 31;
    const int ey = ty0 + wave * 2 + ea, ex = tx0 + epx;
    const bool e_in = ey < p.H && ex < p.W;
    const int64_t em = e_in ? (int64_t)ey * p.W + ex : 0;
    float hwv = 0.f;
    if (tid < N_OUT * 17) hwv = tid < N_OUT * 16 ? p.head.w[tid] : p.head.b[tid - N_OUT * 16];
    float eim[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) eim[c] = p.head.img[em * p.head.img_ld + c];

    // ---- the input patch: 10 x 34 pixels x 32 channels, loaded and split once (zero outside the image: the conv's padding)
    constexpr int NP = (NPIX * 8 + 255) / 256;
    // (round 5: buffer loads without a branch -- a pixel outside the image carries an out-of-range offset and reads zeros)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    f32x4 rp[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int idx = tid + k * 256;
        const int pix = idx >> 3, c4 = (idx & 7) * 4;
        const int py = pix / PW, px = pix - py * PW;
        const int iy = ty0 - 1 + py, ix = tx0 - 1 + px;
        const unsigned oob = (unsigned)((int)((unsigned)iy < (unsigned)p.H) & (int)((unsigned)ix < (unsigned)p.W) & (int)(idx < NPIX * 8)) - 1u;
#if OTVM_H16_NOPATCH
        rp[k] = f32x4{(float)(idx & 7), 0.5f, -0.25f, (float)oob};
#else
        rp[k] = __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(
            in_rsrc, (((unsigned)(iy * p.W + ix) * (unsigned)p.in_ld + c4) << 2) | oob, 0, 0));
#endif
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int idx = tid + k * 256;
        if (idx < NPIX * 8) {
            f16x4 hi, lo;
            split4h(rp[k], hi, lo);
            const int pix = idx >> 3, c4 = (idx & 7) * 4;
            *reinterpret_cast<f16x4*>(&Ph[pix * LDP + c4]) = hi;
            *reinterpret_cast<f16x4*>(&Pl[pix * LDP + c4]) = lo;
        }
    }
    // ---- the weights of all nine taps: 18 coalesced 16-byte loads per lane (L2-resident), in registers for the tap loop; issued
    // behind the patch so that their 72 registers and the patch's 44 are not live together
    f16x8 wh[9], wl[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#if OTVM_H16_NOW
        { const _Float16 c1 = (_Float16)(float)(lane + t); wh[t] = f16x8{c1, c1, c1, c1, c1, c1, c1, c1}; wl[t] = wh[t] * (_Float16)0.001f; }
#else
        wh[t] = *reinterpret_cast<const f16x8*>(p.wf + (t * 2 + 0) * 512 + lane * 8);
        wl[t] = *reinterpret_cast<const f16x8*>(p.wf + (t * 2 + 1) * 512 + lane * 8);
#endif
    }
    __syncthreads();

    // ---- nine taps: wave = two image rows x two 16-pixel tiles; K = 32 channels per MFMA
    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[a][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fpx = lane & 15, fk = (lane >> 4) * 8;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int t = ky * 3 + kx;
#if OTVM_H16_NOMFMA
            acc[0][0][0] += (float)wh[t][0] + (float)wl[t][1]; if (t) continue;
#endif
            f16x8 ah[2][2], al[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int o = ((wave * 2 + a + ky) * PW + kx + m * 16 + fpx) * LDP + fk;
                    ah[a][m] = *reinterpret_cast<const f16x8*>(&Ph[o]);
                    al[a][m] = *reinterpret_cast<const f16x8*>(&Pl[o]);
                }
            // three passes over the four accumulators: consecutive MFMAs never share one
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[a][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[a][m], wh[t], acc[a][m], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[a][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[a][m], wl[t], acc[a][m], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[a][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[a][m], wh[t], acc[a][m], 0, 0, 0);
        }

    // ---- epilogue: C / D of the 16x16 tile: column (channel) = lane & 15, row (pixel) = 4 (lane >> 4) + register
    __syncthreads();                                             // every wave is done with the patch
    float* ep = reinterpret_cast<float*>(smem) + wave * (2 * 32 * EPL);
    typedef __attribute__((address_space(3))) float lds_float;
    float* hw_gen = reinterpret_cast<float*>(smem) + 4 * (2 * 32 * EPL);      // behind the four waves' rows
    if (tid < N_OUT * 17) hw_gen[tid] = hwv;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) ep[(a * 32 + m * 16 + 4 * (lane >> 4) + r) * EPL + (lane & 15)] = acc[a][m][r];
    __syncthreads();                                             // the head's weights are in place (the rows are wave-private)
    float h[16];
    if constexpr (FAST) {
        // filter scales and bias: two wave-uniform 64-byte vectors (scalar loads, batched by the compiler), LeakyReLU inline
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f32x4 v = *reinterpret_cast<const f32x4*>(&ep[(ea * 32 + epx) * EPL + 4 * k]);
            const f32x4 sc4 = *reinterpret_cast<const f32x4*>(p.wscale + 4 * k);
            const f32x4 bi4 = *reinterpret_cast<const f32x4*>(p.bias + 4 * k);
            v = v * sc4 + bi4;                                    // (the arithmetic of the convolutions' epilogues)
            h[4 * k] = v.x > 0.f ? v.x : 0.01f * v.x; h[4 * k + 1] = v.y > 0.f ? v.y : 0.01f * v.y;
            h[4 * k + 2] = v.z > 0.f ? v.z : 0.01f * v.z; h[4 * k + 3] = v.w > 0.f ? v.w : 0.01f * v.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f32x4 v = *reinterpret_cast<const f32x4*>(&ep[(ea * 32 + epx) * EPL + 4 * k]);
            f32x4 sc4, bi4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sc4[j] = p.wscale[4 * k + j];
                if (p.bias) bi4[j] = p.bias[4 * k + j];
            }
            v = v * sc4 + bi4;
            h[4 * k] = otvm_act(v.x, p.act); h[4 * k + 1] = otvm_act(v.y, p.act);
            h[4 * k + 2] = otvm_act(v.z, p.act); h[4 * k + 3] = otvm_act(v.w, p.act);
        }
    }
    if (e_in) {
        if (p.out) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                *reinterpret_cast<f32x4*>(p.out + em * p.out_ld + 4 * k) = f32x4{h[4 * k], h[4 * k + 1], h[4 * k + 2], h[4 * k + 3]};
        }
        const lds_float* hw = (const lds_float*)hw_gen;
#if OTVM_H16_NOHEAD
        p.head.alpha_out[em * p.head.alpha_stride] = h[0] + eim[0] + hw[3];
#elif OTVM_H16_NOSTORE
        { OtvmHeadArgs q = p.head; q.tri_out = nullptr; q.sm = nullptr; otvm_head_pixel_w<N_OUT, const lds_float*>(h, q, em, hw, hw + N_OUT * 16, eim); }
#else
        otvm_head_pixel_w<N_OUT, const lds_float*>(h, p.head, em, hw, hw + N_OUT * 16, eim);
#endif
    }
}

// packed fp32 weight [O_pad][K_pad] (k = tap * I_pad + c, I_pad = 32, 16 filters) -> B fragments of the 16x16x32 MFMA:
// [tap][hi | lo][lane = n + 16 (c / 8)][c % 8], each filter scaled by the power of two the split weights use (w_scale)
__global__ __launch_bounds__(256) void pack_head16_weight_kernel(const float* __restrict__ w, int K_pad, const float* __restrict__ wscale,
                                                                 _Float16* __restrict__ wf) {
    for (int i = threadIdx.x; i < 9 * 16 * 32; i += 256) {
        const int t = i / 512, r = i - t * 512, n = r / 32, c = r - n * 32;
        const float v = w[(int64_t)n * K_pad + t * 32 + c] / wscale[n];      // exact (power of two)
        const _Float16 hi = (_Float16)v;
        const int l = n + 16 * (c >> 3), j = c & 7;
        wf[(t * 2 + 0) * 512 + l * 8 + j] = hi;
        wf[(t * 2 + 1) * 512 + l * 8 + j] = (_Float16)(v - (float)hi);
    }
}

}  // namespace

extern "C" int64_t otvm_head16_weight_bytes_f16x3(void) { return 9 * 2 * 512 * (int64_t)sizeof(_Float16); }

extern "C" int otvm_pack_head16_weight_f16x3(const float* w_packed, int O, int K_pad, int I_pad, const float* w_scale, void* w16,
                                             void* stream) {
    OTVM_REQUIRE(w_packed && w_scale && w16 && O == 16 && I_pad == 32 && K_pad >= 9 * 32,
                 "otvm_pack_head16_weight_f16x3: a 3x3 layer with 32 input and 16 output channels (got %d -> %d)", I_pad, O);
    hipLaunchKernelGGL(pack_head16_weight_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, w_packed, K_pad, w_scale, (_Float16*)w16);
    OTVM_CHECK_LAUNCH("otvm_pack_head16_weight_f16x3");
    return 0;
}

// called by otvm_conv2d_head (conv_patch_f16x3.hip) when the head's 16-wide weights are given; p validated there
int otvm_conv2d_head16_impl(const otvm_conv_params* p, const otvm_head_params* hd, void* stream) {
    OTVM_REQUIRE(p->Cin == 32 && p->Cout == 16 && (p->in_ld & 3) == 0 && ((uintptr_t)p->in & 15) == 0 && !p->in_scale && !p->in_relu,
                 "otvm_conv2d_head (16-wide tile): 32 input channels, plain input");
    Head16Args a;
    a.in = p->in; a.wf = (const _Float16*)hd->w16; a.wscale = p->w_scale; a.bias = p->bias; a.out = p->out;
    a.H = p->H; a.W = p->W; a.in_ld = p->in_ld; a.out_ld = p->out_ld; a.act = p->act;
    OTVM_REQUIRE((int64_t)p->H * p->W * p->in_ld * 4 < 0xFFFFFFF0ll, "otvm_conv2d_head (16-wide tile): input view beyond 32-bit offsets");
    a.in_bytes = (unsigned)((int64_t)p->H * p->W * p->in_ld * 4);
    a.tiles_x = otvm_ceil_div(p->W, TW); a.tiles_y = otvm_ceil_div(p->H, TH);
    a.walk = otvm_tile_walk_of(4);
    const int batch = p->batch > 1 ? p->batch : 1;
    a.in_bs = batch > 1 ? p->in_bs : 0; a.out_bs = batch > 1 ? p->out_bs : 0;
    a.head.w = hd->w; a.head.b = hd->b; a.head.n_out = hd->n_out; a.head.img = hd->img; a.head.img_ld = hd->img_ld;
    a.head.P = hd->P; a.head.alpha_out = hd->alpha_out; a.head.alpha_stride = hd->alpha_stride; a.head.tri_out = hd->tri_out;
    a.head.sm = hd->sm; a.head.sm_ld = hd->sm_ld; a.head.out7 = nullptr; a.head.logits_out = nullptr;
    a.head_img_bs = batch > 1 ? hd->img_bs : 0; a.head_alpha_bs = batch > 1 ? hd->alpha_bs : 0;
    a.head_tri_bs = batch > 1 ? hd->tri_bs : 0; a.head_sm_bs = batch > 1 ? hd->sm_bs : 0;
    OTVM_REQUIRE(hd->n_out == 7 || hd->n_out == 10, "otvm_conv2d_head (16-wide tile): a head of 7 or 10 outputs (got %d)", hd->n_out);
    const dim3 grid(a.tiles_x * a.tiles_y, batch);
    hipStream_t s = (hipStream_t)stream;
    const bool fast = p->act == OTVM_ACT_LEAKY && p->bias && ((uintptr_t)p->bias & 15) == 0 && ((uintptr_t)p->w_scale & 15) == 0;
    if (hd->n_out == 7) {
        if (fast) hipLaunchKernelGGL((conv_head16_f16x3_kernel<7, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv_head16_f16x3_kernel<7, false>), grid, dim3(256), 0, s, a);
    } else {
        if (fast) hipLaunchKernelGGL((conv_head16_f16x3_kernel<10, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv_head16_f16x3_kernel<10, false>), grid, dim3(256), 0, s, a);
    }
    OTVM_CHECK_LAUNCH("otvm_conv2d_head (16-wide tile)");
    return 0;
}
