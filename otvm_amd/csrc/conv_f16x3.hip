// Implicit-GEMM convolution on the 16-bit matrix cores with fp32-class accuracy ("f16x3").
//
// gfx950 has no TF32/xf32 MFMA; its exact-fp32 MFMA runs at the fp32 vector rate (157 TFLOP/s), 1/16
// of the f16/bf16 rate (2.5 PFLOP/s).  This kernel keeps the reference's fp32 contract on the fast
// pipe by splitting every fp32 operand into two halves, x = hi + lo with hi = fp16(x) and
// lo = fp16(x - hi) (22 significant bits), and accumulating  hi*hi + hi*lo + lo*hi  in the fp32 MFMA
// accumulator (v_mfma_f32_32x32x16_f16; fp16 products are exact in fp32).  The dropped lo*lo term is
// 2^-22 relative: measured error vs fp64 is within ~2x of a plain fp32 dot product.  Three
// MFMA passes at the f16 rate = 833 TFLOP/s fp32-equivalent peak, 5.3x the fp32-MFMA ceiling.
//
//   activations: fp32 NHWC in HBM, split on the fly while staging A tiles into LDS (3 VALU/element);
//   weights    : split once at load time (otvm_pack_conv_weight_f16x3), each filter scaled by a power
//                of two so the lo halves stay in the fp16 normal range; the scale is undone in the
//                epilogue (exact).
// Tiling as conv_igemm.hip: BM x BN output tile per workgroup, K walked in chunks of 32 through LDS
// (80-byte padded rows: conflict-free ds_read_b128), NW wave64 each owning TM x TN 32x32 accumulators.
#include "common.h"
#include <type_traits>
#include <hip/hip_fp16.h>
#include <stdlib.h>

#include "conv_f16x3_kernel.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// Round 3: the "wave tile" -- a 64x64 output tile computed by ONE wavefront, no LDS and no barrier in the K loop.
//
// Why: on the small maps (OS8 / OS16 layers, everything at 480p) the 4-wave 64x64 / 128x64 tiles above are LDS-bound, not
// MFMA-bound: with one 32x32 accumulator per wave every MFMA triple needs a fresh A and a fresh B fragment pair from LDS
// (4 KB per 96 MFMA cycles and wave, 170 B/clk for the four waves of a CU against the 128 B/clk LDS delivers), plus two
// barriers per 32-deep chunk: MFMA busy 12-15 % (profiles/r03_mfma_busy_*).  Register reuse needs a 2x2 block of
// accumulators per wave; with four such waves a workgroup covers 128x128 and an 8160-pixel map has too few workgroups.
// So the workgroup IS one wave here: it owns a 64x64 tile (2x2 accumulators, 12 MFMAs per 16-deep k-step against 4 + 4
// fragment loads), and both operands go from global memory / L2 straight into MFMA operand registers:
//   A: lane l needs pixel row (l & 31) and 8 consecutive channels 8 (l >> 5) .. of the k-step -- 32 contiguous bytes of the
//      NHWC tensor per lane (two float4 loads), split into fp16 hi / lo in registers; the two k-steps of a 32-channel chunk
//      consume one 128-byte line per row, so every fetched byte is used;
//   B: the split weights packed ONCE in fragment order (otvm_pack_wave_weight_f16x3:
//      [n/32][chunk][k-step][hi|lo][64 lanes][8 halfs] = 1-KiB blocks), one fully coalesced load per fragment.
// No staging, no conversion in LDS, no __syncthreads; latency is covered by a ring of register sets (PF chunks of loads in
// flight; branch-free so that the compiler waits with exact vmcnt counts, see above).  K order, tap walk, split-K and the
// epilogue (LDS patch for 16-byte stores, residual, activation, fused GroupNorm sums) as in the kernel above.
#ifndef OTVM_WAVE_PF
#define OTVM_WAVE_PF 3
#endif

template <bool RELU_IN>
__global__ __launch_bounds__(64) void conv_wave_f16x3_kernel(const Conv3Args pa) {
    Conv3Args p = pa;
    {
        const int zb = blockIdx.z;
        p.in += zb * p.in_bs;
        p.out += zb * p.out_bs;
        if (p.residual) p.residual += zb * p.res_bs;
        if (p.gn_stats) p.gn_stats += zb * p.gn_bs;
        p.wscale += zb * p.ws_bs;
        if (p.bias) p.bias += zb * p.ws_bs;
    }
    constexpr int BM = 64, BN = 64, TM = 2, TN = 2, PF = OTVM_WAVE_PF;
    __shared__ __attribute__((aligned(16))) float patch[32 * 36];
    __shared__ double gred[2 * BN];
    const int lane = threadIdx.x;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tile_n = wgid % p.tiles_n, tile_m = wgid / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int frow = lane & 31, fk = (lane >> 5) * 8;

    int rowoff[TM];
    unsigned tapmask[TM];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        const int m = m0 + a * 32 + frow;
        int iy0 = -(1 << 28), ix0 = -(1 << 28);
        if (m < p.M) {
            const int oy = m / p.Wo, ox = m - oy * p.Wo;
            iy0 = oy * p.stride - p.pad;
            ix0 = ox * p.stride - p.pad;
        }
        rowoff[a] = (iy0 * p.W + ix0) * p.in_ld + fk;
        unsigned mk = 0;
        for (int t = 0; t < p.taps; ++t) {
            const int ky = t / p.kw, kx = t - ky * p.kw;
            const int iy = iy0 + ky * p.dil, ix = ix0 + kx * p.dil;
            if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) mk |= 1u << t;
        }
        tapmask[a] = mk;
    }
    const int c_begin = (int)(((int64_t)blockIdx.y * p.nchunks) / gridDim.y);
    const int c_end = (int)(((int64_t)(blockIdx.y + 1) * p.nchunks) / gridDim.y);
    float* const outp = p.out + (int64_t)blockIdx.y * p.split_stride;
    int u_cb = c_begin / p.taps, u_tap = c_begin - u_cb * p.taps;
    int u_ky = u_tap / p.kw, u_kx = u_tap - u_ky * p.kw;
    // fragment blocks of this tile's two 32-column n-tiles: [n-tile][chunk][k-step][hi|lo][lane][8]
    const _Float16* const wfl = p.wf + (int64_t)(n0 >> 5) * p.nchunks * 2048 + lane * 8;
    const int64_t wnt = (int64_t)p.nchunks * 2048;            // halfs between consecutive n-tiles

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    struct RegSet {
        f32x4 ra[TM][2][2];          // [m-tile][k-step][first / second quad of the lane's 8 channels]
        f16x8 rb[TN][2][2];          // [n-tile][k-step][hi|lo]
        unsigned okmask;
    };
    RegSet rs[PF];
    auto load_chunk = [&](int c, RegSet& R, const bool valid) __attribute__((always_inline)) {
        const int delta = (u_ky * p.dil * p.W + u_kx * p.dil) * p.in_ld + (u_cb << 5);   // scalar
        const unsigned bit = 1u << u_tap;
        unsigned okm = 0;
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const bool ok = valid && (tapmask[a] & bit) != 0;
            const float* src = p.in + (int64_t)(ok ? rowoff[a] + delta : 0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                R.ra[a][ks][0] = *reinterpret_cast<const f32x4*>(src + ks * 16);
                R.ra[a][ks][1] = *reinterpret_cast<const f32x4*>(src + ks * 16 + 4);
            }
            okm |= ok ? (1u << a) : 0u;
        }
        R.okmask = okm;
        ++u_tap;
        if (++u_kx == p.kw) { u_kx = 0; ++u_ky; }
        if (u_tap == p.taps) { u_tap = 0; u_kx = 0; u_ky = 0; ++u_cb; }
        const int cw = valid ? c : c_end - 1;
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl)
                    R.rb[b][ks][hl] = *reinterpret_cast<const f16x8*>(wfl + b * wnt + ((int64_t)(cw * 2 + ks) * 2 + hl) * 512);
    };
    auto compute = [&](RegSet& R) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 ah[TM], al[TM];
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                f32x4 v0 = R.ra[a][ks][0], v1 = R.ra[a][ks][1];
                if (RELU_IN) {
                    v0.x = fmaxf(v0.x, 0.f); v0.y = fmaxf(v0.y, 0.f); v0.z = fmaxf(v0.z, 0.f); v0.w = fmaxf(v0.w, 0.f);
                    v1.x = fmaxf(v1.x, 0.f); v1.y = fmaxf(v1.y, 0.f); v1.z = fmaxf(v1.z, 0.f); v1.w = fmaxf(v1.w, 0.f);
                }
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                const bool ok = (R.okmask >> a) & 1u;
                v0 = ok ? v0 : z;
                v1 = ok ? v1 : z;
                f16x4 h0, l0, h1, l1;
                split4(v0, h0, l0);
                split4(v1, h1, l1);
                ah[a] = f16x8{h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                al[a] = f16x8{l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
            }
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[a], R.rb[b][ks][0], acc[a][b], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], R.rb[b][ks][1], acc[a][b], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], R.rb[b][ks][0], acc[a][b], 0, 0, 0);
        }
    };
#pragma unroll
    for (int j = 0; j < PF; ++j) load_chunk(c_begin + j, rs[j], c_begin + j < c_end);
    for (int c = c_begin; c < c_end; c += PF) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            if (j > 0 && c + j >= c_end) break;
            RegSet cur = rs[j];                                // (registers: the set is reloaded below, its values used after)
            load_chunk(c + j + PF, rs[j], c + j + PF < c_end);
            compute(cur);
        }
    }

    // ---- epilogue (as conv_igemm_f16x3_kernel with one wave: wm = wn = 0)
    const int col = lane & 31, rbase = (lane >> 5) * 4;
    {
        const int prow = lane >> 3, pc = (lane & 7) * 4;
        const bool vec_ok = ((p.out_ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(outp) & 15) == 0) &&
                            (!p.residual || (((p.res_ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.residual) & 15) == 0)));
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int nb = n0 + b * 32;
            if (nb >= p.Cout) continue;
            const int n4 = nb + pc;
            f32x4 sc4 = {0.f, 0.f, 0.f, 0.f}, bi4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (n4 + j < p.Cout) {
                    sc4[j] = p.wscale[n4 + j];
                    bi4[j] = p.bias ? p.bias[n4 + j] : 0.f;
                }
            }
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int mb = m0 + a * 32;
                f32x4 rres[4];
                const bool res_vec = p.residual && vec_ok && n4 + 3 < p.Cout;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int m = mb + r4 * 8 + prow;
                    rres[r4] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (res_vec && m < p.M) rres[r4] = *reinterpret_cast<const f32x4*>(p.residual + (int64_t)m * p.res_ld + n4);
                }
                __syncthreads();                               // (one wave: orders the LDS patch between its two uses)
#pragma unroll
                for (int e = 0; e < 16; ++e) patch[((e & 3) + 8 * (e >> 2) + rbase) * 36 + col] = acc[a][b][e];
                __syncthreads();
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int row = r4 * 8 + prow;
                    const int m = mb + row;
                    f32x4 v = *reinterpret_cast<const f32x4*>(&patch[row * 36 + pc]);
                    v = v * sc4 + bi4;
                    if (m < p.M) {
                        if (vec_ok && n4 + 3 < p.Cout) {
                            v += rres[r4];
                            v.x = otvm_act(v.x, p.act); v.y = otvm_act(v.y, p.act);
                            v.z = otvm_act(v.z, p.act); v.w = otvm_act(v.w, p.act);
                            *reinterpret_cast<f32x4*>(outp + (int64_t)m * p.out_ld + n4) = v;
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                if (n4 + j < p.Cout) {
                                    float x = v[j];
                                    if (p.residual) x += p.residual[(int64_t)m * p.res_ld + n4 + j];
                                    outp[(int64_t)m * p.out_ld + n4 + j] = otvm_act(x, p.act);
                                }
                            }
                        }
                    }
                }
            }
        }
    }
    if (p.gn_stats) {
        const int cg = p.Cout >> 5;
        const int seg = cg < 32 ? cg : 32;
        for (int i = threadIdx.x; i < 2 * BN; i += blockDim.x) gred[i] = 0.0;
        __syncthreads();
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int nl = b * 32 + col;
            const int n = n0 + nl;
            float s = 0.f, ss = 0.f;
            if (n < p.Cout) {
                const float sc_ = p.wscale[n];
                const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int m = m0 + a * 32 + (e & 3) + 8 * (e >> 2) + rbase;
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
                const int gl = nl / cg;
                atomicAdd(&gred[2 * gl], (double)s);
                atomicAdd(&gred[2 * gl + 1], (double)ss);
            }
        }
        __syncthreads();
        const int ng = (BN + cg - 1) / cg;
        for (int i = threadIdx.x; i < 2 * ng; i += blockDim.x) {
            const int g = n0 / cg + (i >> 1);
            if (g < 32 && gred[i] != 0.0) atomicAdd(&p.gn_stats[2 * g + (i & 1)], gred[i]);
        }
        __syncthreads();
        otvm_gn_table_tail(p.gn_stats, p.M, p.Cout, p.tail, blockIdx.z, gridDim.x * gridDim.y, reinterpret_cast<float*>(gred));
    }
}

// [O_pad][K_pad] split weights (K already in the kernel's chunk order) -> fragment blocks [n/32][chunk][k-step][hi|lo][lane][8]
__global__ __launch_bounds__(256) void pack_wave_weight_kernel(const _Float16* __restrict__ wh, const _Float16* __restrict__ wl,
                                                               int K_pad, _Float16* __restrict__ wf) {
    const int nt = blockIdx.x, nchunks = K_pad >> 5;
    for (int i = threadIdx.x; i < nchunks * 2 * 2 * 64; i += 256) {
        const int l = i & 63, hl = (i >> 6) & 1, ks = (i >> 7) & 1, c = i >> 8;
        const _Float16* src = (hl ? wl : wh) + (int64_t)(nt * 32 + (l & 31)) * K_pad + c * 32 + ks * 16 + 8 * (l >> 5);
        _Float16* dst = wf + ((((int64_t)nt * nchunks + c) * 2 + ks) * 2 + hl) * 512 + l * 8;
        *reinterpret_cast<f16x8*>(dst) = *reinterpret_cast<const f16x8*>(src);
    }
}

// split-K epilogue: out = act(sum_z part[z] + bias + residual), partials added in a fixed order (deterministic)
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ part, int S, int64_t stride, int64_t M,
                                                            int Cout, int ldp, const float* __restrict__ bias,
                                                            const float* __restrict__ residual, int res_ld, int act,
                                                            float* __restrict__ out, int out_ld, int64_t out_bs, int64_t res_bs,
                                                            int ws_bs, const float* __restrict__ res_scale, int rs_bs) {
    part += (int64_t)blockIdx.y * S * stride;                   // image blockIdx.y
    out += blockIdx.y * out_bs;
    if (residual) residual += blockIdx.y * res_bs;
    if (bias) bias += blockIdx.y * ws_bs;
    if (res_scale) res_scale += blockIdx.y * rs_bs;
    const unsigned Q = (unsigned)ldp >> 2;
    const unsigned total = (unsigned)M * Q;                     // split layers are small: M * ldp / 4 < 2^32 (host check)
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int64_t m = i / Q;
        const int c = (int)(i - (unsigned)m * Q) * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(part + m * ldp + c);
        for (int z = 1; z < S; ++z) v += *reinterpret_cast<const f32x4*>(part + z * stride + m * ldp + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (c + j < Cout) {
                float x = v[j] + (bias ? bias[c + j] : 0.f);
                if (residual) x += residual[m * res_ld + c + j] * (res_scale ? res_scale[c + j] : 1.f);
                out[m * out_ld + c + j] = otvm_act(x, act);
            }
        }
    }
}

// split a packed fp32 weight row into power-of-two-scaled fp16 hi/lo halves
__global__ __launch_bounds__(256) void split_weight_kernel(const float* __restrict__ w, int O, int K_pad, int taps, int I_pad,
                                                           int reorder, _Float16* __restrict__ wh,
                                                           _Float16* __restrict__ wl, float* __restrict__ wscale) {
    const int o = blockIdx.x;
    const float* row = w + (int64_t)o * K_pad;
    __shared__ float red[256];
    float mx = 0.f;
    for (int k = threadIdx.x; k < K_pad; k += 256) mx = fmaxf(mx, fabsf(row[k]));
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    mx = red[0];
    int e = 0;
    if (mx > 0.f) frexpf(mx, &e);                 // mx = f * 2^e, f in [0.5, 1)
    const float sc = ldexpf(1.f, e);              // |w| / sc <= 1
    const float inv = ldexpf(1.f, -e);
    if (threadIdx.x == 0 && o < O) wscale[o] = sc;
    for (int k = threadIdx.x; k < K_pad; k += 256) {
        const float v = row[k] * inv;             // exact (power of two)
        const _Float16 hi = (_Float16)v;
        int kn = k;
        if (reorder && k < taps * I_pad) {        // [tap][c] -> [c/32][tap][c%32]
            const int t = k / I_pad, c = k - t * I_pad;
            kn = ((c >> 5) * taps + t) * 32 + (c & 31);
        }
        wh[(int64_t)o * K_pad + kn] = hi;
        wl[(int64_t)o * K_pad + kn] = (_Float16)(v - (float)hi);
    }
}

}  // namespace

extern "C" int otvm_split_conv_weight_f16x3(const float* w_packed, int O, int O_pad, int K_pad, int taps, int I_pad, void* w_hi,
                                            void* w_lo, float* w_scale, void* stream) {
    OTVM_REQUIRE(w_packed && w_hi && w_lo && w_scale, "otvm_split_conv_weight_f16x3: null pointer");
    hipLaunchKernelGGL(split_weight_kernel, dim3(O_pad), dim3(256), 0, (hipStream_t)stream, w_packed, O, K_pad, taps, I_pad,
                       f16x3_fast_layout(taps, I_pad) ? 1 : 0, (_Float16*)w_hi, (_Float16*)w_lo, w_scale);
    OTVM_CHECK_LAUNCH("otvm_split_conv_weight_f16x3");
    return 0;
}

extern "C" int64_t otvm_wave_weight_bytes_f16x3(int O_pad, int K_pad) { return (int64_t)O_pad * K_pad * 2 * sizeof(_Float16); }

extern "C" int otvm_pack_wave_weight_f16x3(const void* w_hi, const void* w_lo, int O_pad, int K_pad, void* w_wfrag, void* stream) {
    OTVM_REQUIRE(w_hi && w_lo && w_wfrag && O_pad % 32 == 0 && K_pad % 32 == 0, "otvm_pack_wave_weight_f16x3: bad arguments");
    hipLaunchKernelGGL(pack_wave_weight_kernel, dim3(O_pad / 32), dim3(256), 0, (hipStream_t)stream, (const _Float16*)w_hi,
                       (const _Float16*)w_lo, K_pad, (_Float16*)w_wfrag);
    OTVM_CHECK_LAUNCH("otvm_pack_wave_weight_f16x3");
    return 0;
}

int otvm_conv2d_patch_f16x3_impl(const otvm_conv_params* p, void* stream);    // conv_patch_f16x3.hip, -1 = not eligible

// (measured and rejected, round 2: an "activation-stationary" kernel for the expanding 1x1 layers of the bottlenecks --
// the 64/128-pixel x Cin tile split once into LDS, every wave streaming its own weight fragments from L2, two or three
// workgroups per CU -- 64->256 at 272x480 49 vs 51 us, 128->512 at 136x240 35 vs 35, 256->1024 108 vs 89, 512->2048
// 246 vs 266, and 2x slower with fused GroupNorm sums (four times as many fp64 atomics per group as the 256-row tiles):
// these layers move 2.4-3.9 TB/s of output + input + residual in either kernel, i.e. they sit within 1.6-2.5x of the HBM
// bound, not of the MFMA bound.  A deeper register prefetch ring (OTVM_PF_DEPTH) on the 256x128 tile: +-2 %.)
// ---- dispatch.  A configuration is (tile, S): one of the implicit-GEMM tiles below with the K chunks of every output
// tile shared by S workgroups (S > 1: partial tiles through the caller's workspace, added in a fixed order by
// splitk_finish_kernel), or the 3x3 patch kernel.  otvm_conv_params.tune forces one (the host's plan-time autotuner,
// otvm_amd/engine.py, times the candidates of otvm_conv2d_candidates on the device); 0 = the heuristic below.
enum { T256x256 = 0, T256x128, T128x128, T128x64, T64x64, T256x64, T256x32, T256x128W4, T128x256W4, T64x64W1, T64x64D, T128x64D,
       T_STEM = 12, T256x256W4 = 13, T_PATCH = 14, T_COUNT = 15,
       // round 5: tile t with LDS-DMA weight stages = T_GLDS + t (conv_f16x3_glds.hip; see the kernel's GLDS comment)
       T_GLDS = 32,
       // ... and T_M16 + t: the LDS-DMA tile t multiplying with v_mfma_f32_16x16x32_f16 (conv_f16x3_m16.hip; the kernel's M16 comment)
       T_M16 = 64 };
static inline int tune_code(int tile, int S) { return (tile + 1) * 16 + S; }
static inline bool is_m16_tile(int t) { return t >= T_M16 && t < T_M16 + T_STEM; }
static inline bool is_glds_tile(int t) {                    // (either matrix-core form)
    const int b = t - (is_m16_tile(t) ? T_M16 : T_GLDS);
    return b == T256x256 || b == T256x128 || b == T128x128 || b == T128x64 || b == T64x64 || b == T256x64 || b == T256x32 ||
           b == T256x128W4 || b == T128x256W4 || b == T64x64D || b == T128x64D;
}
static inline bool is_gemm_tile(int t) { return (t >= 0 && t < T_STEM) || t == T256x256W4 || is_glds_tile(t); }
static inline int base_tile(int t) { return is_glds_tile(t) ? t - (is_m16_tile(t) ? T_M16 : T_GLDS) : t; }
static const int TILE_BM_[T_COUNT] = {256, 256, 128, 128, 64, 256, 256, 256, 128, 64, 64, 128, 0, 256, 0};
static const int TILE_BN_[T_COUNT] = {256, 128, 128, 64, 64, 64, 32, 128, 256, 64, 64, 64, 0, 256, 0};
static inline int TILE_BM(int t) { return TILE_BM_[base_tile(t)]; }
static inline int TILE_BN(int t) { return TILE_BN_[base_tile(t)]; }

static int launch_wave(Conv3Args& a, hipStream_t s, int ksplit) {
    a.tiles_m = otvm_ceil_div(a.M, 64);
    a.tiles_n = otvm_ceil_div(a.Cout, 64);
    const dim3 grid(a.tiles_m * a.tiles_n, ksplit, a.batch), block(64);
    if ((int64_t)a.H * a.W * a.in_ld >= (1ll << 31) - (1 << 20)) {
        otvm_set_error("otvm_conv2d(f16x3 wave tile): input view too large for 32-bit offsets");
        return 1;
    }
    if (a.in_relu) hipLaunchKernelGGL((conv_wave_f16x3_kernel<true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((conv_wave_f16x3_kernel<false>), grid, block, 0, s, a);
    OTVM_CHECK_LAUNCH("otvm_conv2d(f16x3 wave tile)");
    return 0;
}

static int launch_tile(int tile, Conv3Args& a, hipStream_t s, int S) {
    if (a.npass == 1) return otvm_launch_tile_p1(tile, a, s, S);       // precision "f16": conv_f16x3_p1.hip / _p1g.hip
    switch (tile) {
        case T256x256: return launch3<256, 256, 4, 2>(a, s, S);
        case T256x128: return launch3<256, 128, 4, 2>(a, s, S);
        case T128x128: return launch3<128, 128, 2, 2>(a, s, S);
        case T128x64: return launch3<128, 64, 2, 2>(a, s, S);
        case T64x64: return launch3<64, 64, 2, 2>(a, s, S);
        case T256x64: return launch3<256, 64, 4, 1>(a, s, S);
        case T256x32: return launch3<256, 32, 4, 1>(a, s, S);
        // 4-wave workgroups with a single LDS stage (61 KB): two per CU, so one workgroup's epilogue (output stores,
        // GroupNorm sums) overlaps the other's main loop -- candidates for the short-K, output-heavy 1x1 layers
        case T256x128W4: return launch3<256, 128, 2, 2, false, true>(a, s, S);
        case T128x256W4: return launch3<128, 256, 2, 2, false, true>(a, s, S);
        // one-wave workgroups, operands straight from L2 into MFMA registers (small maps)
        case T64x64W1: return launch_wave(a, s, S);
        // pipelined small tiles: two LDS stages, one barrier per chunk, the next chunk converted under the MFMAs
        case T64x64D: return launch3<64, 64, 2, 2, true>(a, s, S);
        case T128x64D: return launch3<128, 64, 2, 2, true>(a, s, S);
        // 256x256 with FOUR waves: every wave owns 128x128 (4x4 accumulator tiles, 256 registers), one wave per SIMD.  Per
        // 16-deep k-step a wave reads 16 fragments for 48 MFMAs, the 8-wave tile 12 for 24: a third less LDS traffic per
        // MFMA (tools/probes/lds_probe.hip: ~13 clocks per wave-wide b128 access with four waves issuing, so the 8-wave
        // tile's fragment reads take about as long as its MFMAs).  Measured: 256->256 3x3 at 272x480 0.462 vs 0.479 ms, but
        // 512->512 0.445 vs 0.408, 2048->256 1.45 vs 1.15, the 1x1 layers 20-40 % slower -- a single wave per SIMD has
        // nothing to overlap its own fragment reads with.  Kept as a forced configuration (tune code 225), not a candidate.
        case T256x256W4: return launch3<256, 256, 2, 2, false, true>(a, s, S);
    }
    if (is_glds_tile(tile)) return is_m16_tile(tile) ? otvm_launch_m16_tile(tile - T_M16, a, s, S) : otvm_launch_glds_tile(tile - T_GLDS, a, s, S);
    otvm_set_error("otvm_conv2d(f16x3): unknown tile %d", tile);
    return 1;
}

// is (tile, S) a legal configuration of this layer?
static bool config_ok(const otvm_conv_params* p, int tile, int S) {
    if (!is_gemm_tile(tile) || S < 1 || S > 8) return false;
    const int64_t M = (int64_t)p->Ho * p->Wo;
    // the weight arrays hold O_pad = Cout rounded up to 128 rows (include/otvm_hip.h): a 256-wide N tile may only be
    // used when that is a multiple of 256, or its last tile would read rows past the allocation
    if (TILE_BN(tile) == 256 && !(p->Cout >= 256 && (otvm_ceil_div(p->Cout, 128) & 1) == 0)) return false;
    // fused input normalisation: the whole-chunk implicit-GEMM kernels (and the patch kernel), not the one-wave tile
    if (p->in_scale && !(f16x3_fast_layout(p->kh * p->kw, p->Cin) && !p->in_relu && tile != T64x64W1)) return false;
    // the 4-wave big tiles hold 128 accumulator registers per lane: only their wave-uniform-tap-walk variants fit two
    // waves per SIMD without spilling
    const int bt = base_tile(tile);
    if ((bt == T256x128W4 || bt == T128x256W4 || bt == T256x256W4) && !f16x3_fast_layout(p->kh * p->kw, p->Cin)) return false;
    // the wave tile reads fragment-major weights and walks whole 32-channel blocks: fast layout + w_wfrag only
    if ((bt == T64x64D || bt == T128x64D) && !f16x3_fast_layout(p->kh * p->kw, p->Cin)) return false;
    if (tile == T64x64W1 && !(p->w_wfrag && f16x3_fast_layout(p->kh * p->kw, p->Cin) && (p->in_ld & 3) == 0)) return false;
    if (tile == T64x64W1 && p->res_scale) return false;           // (the one-wave tile's epilogue has no residual scale)
    if (p->precision == OTVM_PREC_F16 && (tile == T64x64W1 || tile == T256x256W4 || is_m16_tile(tile))) return false;   // (no single-pass form)
    // LDS-DMA weight stages: fragment-major weights, whole chunks, the input view inside a 2-GiB buffer resource
    if (is_glds_tile(tile) && !(p->w_wfrag && f16x3_fast_layout(p->kh * p->kw, p->Cin) && (p->in_ld & 3) == 0 &&
                                (int64_t)p->H * p->W * p->in_ld * 4 < (1ll << 31))) return false;
    if (S > 1) {
        const int nchunks = p->K_pad / 32;
        const int ldp = (p->Cout + 3) & ~3;
        if (!p->splitk_ws || nchunks / S < 4) return false;
        const int64_t nb = p->batch > 1 ? p->batch : 1;
        if (nb * S * M * ldp * (int64_t)sizeof(float) > p->splitk_ws_bytes || M * (ldp / 4) >= (1ll << 32)) return false;
    }
    return true;
}

static int run_config(const otvm_conv_params* p, Conv3Args& a, int tile, int S, hipStream_t s) {
    if (S <= 1) return launch_tile(tile, a, s, 1);
    const int64_t M = a.M;
    const int ldp = (p->Cout + 3) & ~3;
    Conv3Args b = a;
    b.out = (float*)p->splitk_ws; b.out_ld = ldp; b.split_stride = M * ldp; b.out_bs = (int64_t)S * M * ldp;
    b.bias = nullptr; b.residual = nullptr; b.res_scale = nullptr; b.act = OTVM_ACT_NONE; b.gn_stats = nullptr; b.tail.scale = nullptr;
    const int rc = launch_tile(tile, b, s, S);
    if (rc) return rc;
    int64_t blocks = (M * (ldp / 4) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    const int g_batch = a.batch;
    hipLaunchKernelGGL(splitk_finish_kernel, dim3((int)blocks, g_batch), dim3(256), 0, s, (const float*)p->splitk_ws, S,
                       (int64_t)M * ldp, M, p->Cout, ldp, p->bias, p->residual, p->res_ld, p->act, p->out,
                       p->out_ld, a.out_bs, a.res_bs, a.ws_bs, a.res_scale, a.rs_bs);
    OTVM_CHECK_LAUNCH("otvm_conv2d(split-K finish)");
    if (p->gn_stats) {
        const int rc2 = otvm_gn_stats_b(p->out, M, p->Cout, p->out_ld, p->gn_stats, g_batch, a.out_bs, a.gn_bs, (void*)s);
        if (rc2 || !a.tail.scale) return rc2;
        // the statistics were finished by a pass of their own: the table is one more (tiny) launch here, not the caller's
        return otvm_gn_table_b(p->gn_stats, M, p->Cout, p->gn_gamma, p->gn_beta, p->gn_scale_out, p->gn_shift_out, g_batch,
                               a.gn_bs, a.tail.tab_bs, (void*)s);
    }
    return 0;
}

int otvm_conv2d_patch_eligible(const otvm_conv_params* p);                  // conv_patch_f16x3.hip: shape-wise eligibility
int otvm_conv2d_patch_f16x3_forced(const otvm_conv_params* p, void* stream);
int otvm_conv2d_stem_eligible(const otvm_conv_params* p);                   // conv_stem_f16x3.hip: 7x7 stride-2 stems
int otvm_conv2d_stem_f16x3(const otvm_conv_params* p, void* stream);

static int glds_mode() {
    static const int m = otvm_probe_int("OTVM_IGEMM_GLDS", 1);
    return m;
}
// the 16x16x32 form of the LDS-DMA tiles: 2 (default) = instead of the 32x32x16 form everywhere; 1 = offered to the tuner next to
// the 32x32x16 form and taken by the heuristic on maps of >= OTVM_IGEMM_M16_MIN_PIXELS output pixels; 0 = never (A/B runs).
// Whole frame, one box, alternating (profiles/r05_igemm_mfma16_ab.txt): 1080p 47.51 (2) / 47.36 (1) / 46.51 (0) frames/s,
// 832x480 153.2 / 153.2 / 153.3
static int m16_mode() {
    static const int m = otvm_probe_int("OTVM_IGEMM_M16", 2);
    return m;
}
static int64_t m16_min_pixels() {
    static const int64_t v = otvm_probe_int("OTVM_IGEMM_M16_MIN_PIXELS", 16384);
    return v;
}

extern "C" int otvm_conv2d_candidates(const otvm_conv_params* p, int* out, int max_n) {
    int n = 0;
    if (!p || !out || !otvm_prec_is_split(p->precision)) return 0;
    auto add = [&](int code) { if (n < max_n) out[n++] = code; };
    if (otvm_conv2d_stem_eligible(p)) add(tune_code(T_STEM, 1));
    if (otvm_conv2d_patch_eligible(p)) add(tune_code(T_PATCH, 1));
    if (p->in_res) return n;                                       // (the patch kernel only)
    const int64_t M = (int64_t)p->Ho * p->Wo;
    const int nchunks = p->K_pad / 32;
    // (T256x256W4 is legal when forced but not offered: it won one of seven large layers by 3.5 % and lost the others by
    //  6-40 %, profiles/r03_lds_and_wait_counters.md)
    static const int tiles_wide[] = {T256x256, T256x128, T128x128, T128x64, T64x64, T256x128W4, T128x256W4, T64x64W1, T64x64D, T128x64D};
    static const int tiles_64[] = {T256x64, T128x64, T64x64, T64x64W1, T64x64D, T128x64D};
    static const int tiles_32[] = {T256x32, T64x64};
    const int* tl = p->Cout <= 32 ? tiles_32 : (p->Cout <= 64 ? tiles_64 : tiles_wide);
    const int ntl = p->Cout <= 32 ? 2 : (p->Cout <= 64 ? 6 : 10);
    static const int splits[] = {1, 2, 3, 4, 6, 8};
    for (int i = 0; i < ntl; ++i) {
        const int t = tl[i];
        const int64_t wgs = (int64_t)otvm_ceil_div(M, TILE_BM(t)) * otvm_ceil_div(p->Cout, TILE_BN(t));
        if (wgs > 16384 && TILE_BM(t) < 256) continue;              // huge maps: only the 256-row tiles are worth timing
        if (t == T64x64W1 && wgs > 4096) continue;                  // the wave tile is for maps that cannot fill the chip otherwise
        for (int j = 0; j < 6; ++j) {
            const int S = splits[j];
            if (S > 1 && (wgs * S > (t == T64x64W1 ? 4096 : 1536) || nchunks < 16)) continue;   // splitting K only helps launches that cannot fill the chip
            // the tile's LDS-DMA form, where it exists and the layer qualifies (fragment-major weights, whole chunks), REPLACES the
            // register-staged form in the list (OTVM_IGEMM_GLDS=2 offers both, 0 the staged form only: A/B runs)
            const bool g_ok = glds_mode() != 0 && is_glds_tile(T_GLDS + t) && config_ok(p, T_GLDS + t, S);
            const bool m_ok = g_ok && m16_mode() != 0 && config_ok(p, T_M16 + t, S);
            if (g_ok && !(m_ok && m16_mode() == 2)) add(tune_code(T_GLDS + t, S));
            if (m_ok) add(tune_code(T_M16 + t, S));
            if ((!g_ok || glds_mode() == 2) && config_ok(p, t, S)) add(tune_code(t, S));
        }
    }
    return n;
}

// the whole-chunk implicit-GEMM kernels fold the producer's GroupNorm apply into their A staging (NORM_IN)
int otvm_conv2d_igemm_accepts_input_norm(const otvm_conv_params* p) {
    return p && otvm_prec_is_split(p->precision) && f16x3_fast_layout(p->kh * p->kw, p->Cin) && !p->in_relu && p->w_hi ? 1 : 0;
}

int otvm_conv2d_f16x3_impl(const otvm_conv_params* p, void* stream) {
    OTVM_REQUIRE(p->w_hi && p->w_lo && p->w_scale, "otvm_conv2d: precision f16x3 needs w_hi / w_lo / w_scale");
    const int forced_tile = p->tune ? p->tune / 16 - 1 : -1, forced_S = p->tune & 15;
    OTVM_REQUIRE(!p->in_res || (otvm_conv2d_patch_eligible(p) && (!p->tune || forced_tile == T_PATCH)),
                 "otvm_conv2d: in_res (identity inside the fused input normalisation) is a patch-kernel feature");
    const bool gemm_only = (p->batch > 1 && p->ws_bs) || (p->residual && p->res_scale);   // ABI 17 fields: implicit GEMM only
    OTVM_REQUIRE(!gemm_only || (p->kh == 1 && p->kw == 1), "otvm_conv2d: ws_bs / res_scale are for 1x1 layers (implicit-GEMM route)");
    if (forced_tile == T_STEM || (!p->tune && otvm_conv2d_stem_eligible(p))) {
        OTVM_REQUIRE(otvm_conv2d_stem_eligible(p), "otvm_conv2d: tune asks for the stem kernel on a layer it cannot take");
        return otvm_conv2d_stem_f16x3(p, stream);
    }
    if (p->tune && forced_tile == T_PATCH) {
        OTVM_REQUIRE(otvm_conv2d_patch_eligible(p), "otvm_conv2d: tune asks for the patch kernel on a layer it cannot take");
        return otvm_conv2d_patch_f16x3_forced(p, stream);
    }
    if (!p->tune) {
        const int rc = otvm_conv2d_patch_f16x3_impl(p, stream);
        if (rc != -1) return rc;
    }
    Conv3Args a;
    a.in = p->in; a.wh = (const _Float16*)p->w_hi; a.wl = (const _Float16*)p->w_lo; a.wscale = p->w_scale;
    a.bias = p->bias; a.residual = p->residual; a.out = p->out; a.gn_stats = p->gn_stats;
    a.H = p->H; a.W = p->W; a.Cin = p->Cin; a.in_ld = p->in_ld; a.K_pad = p->K_pad; a.res_ld = p->res_ld;
    a.Ho = p->Ho; a.Wo = p->Wo; a.Cout = p->Cout; a.out_ld = p->out_ld;
    a.kh = p->kh; a.kw = p->kw; a.stride = p->stride; a.pad = p->pad; a.dil = p->dil;
    a.in_relu = p->in_relu; a.act = p->act;
    a.M = p->Ho * p->Wo; a.taps = p->kh * p->kw; a.nchunks = p->K_pad / 32;
    a.split_stride = 0;
    a.wf = (const _Float16*)p->w_wfrag;
    a.npass = p->precision == OTVM_PREC_F16 ? 1 : 3;
    a.tail = otvm_gn_tail_of(p);
    a.in_scale = p->in_scale; a.in_shift = p->in_shift;
    a.in_slope = p->in_act == OTVM_ACT_RELU ? 0.f : (p->in_act == OTVM_ACT_LEAKY ? 0.01f : 1.f);
    a.norm_bs = p->batch > 1 ? p->norm_bs : 0;
    OTVM_REQUIRE(!p->in_scale || otvm_conv2d_igemm_accepts_input_norm(p),
                 "otvm_conv2d: fused input normalisation on the implicit-GEMM path needs Cin %% 32 == 0 and no in_relu");
    const int g_batch = a.batch = p->batch > 1 ? p->batch : 1;
    a.in_bs = g_batch > 1 ? p->in_bs : 0; a.out_bs = g_batch > 1 ? p->out_bs : 0; a.res_bs = g_batch > 1 ? p->res_bs : 0;
    a.gn_bs = g_batch > 1 ? p->gn_bs : 0;
    a.ws_bs = g_batch > 1 ? p->ws_bs : 0;
    a.res_scale = p->residual ? p->res_scale : nullptr; a.rs_bs = g_batch > 1 ? p->res_scale_bs : 0;
    OTVM_REQUIRE(!a.res_scale || (((uintptr_t)a.res_scale & 15) == 0 && (a.rs_bs & 3) == 0 && p->Cout % 4 == 0),
                 "otvm_conv2d: res_scale must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int64_t M = a.M;
    if (p->tune) {
        OTVM_REQUIRE(config_ok(p, forced_tile, forced_S), "otvm_conv2d: tune code %d is not a legal configuration of this layer",
                     p->tune);
        return run_config(p, a, forced_tile, forced_S, s);
    }
    // ---- heuristic.  Split-K for layers that cannot fill the chip with output tiles alone (OS16/OS32 maps, the whole
    // 480p frame): 128-row tiles, the K chunks of a tile shared by up to 8 workgroups
    static const int splitk = otvm_probe_int("OTVM_SPLITK", 1);
    static const int min_total = otvm_probe_int("OTVM_SPLITK_MINTOTAL", 32);
    if (splitk && p->splitk_ws && a.nchunks >= min_total) {
        const bool wide = p->Cout > 64;
        const int64_t tiles = (int64_t)otvm_ceil_div(M, 128) * otvm_ceil_div(p->Cout, wide ? 128 : 64);
        // at least 8 chunks per workgroup (16 -> 8 and 64 -> 32 total: 480p 124.9 -> 126.3 fps, 1080p unchanged); the reduction pass (and, with fused GroupNorm sums, a statistics pass over
        // the output) costs two small launches, so moderately deep layers keep the single-pass kernel (measured per
        // layer at 480p / 1080p: 1024->128 3x3 at OS16 0.186 -> 0.058 ms, 1024->256 1x1 + GN 0.035 -> 0.074 ms)
        int S = (int)(384 / (tiles > 0 ? tiles : 1));
        if (S > 8) S = 8;
        static const int min_chunks = otvm_probe_int("OTVM_SPLITK_MINCHUNKS", 8);
        if (S > a.nchunks / min_chunks) S = a.nchunks / min_chunks;
        if (p->gn_stats && a.nchunks < 256 && tiles > 8) S = 1;
        const int tk = wide ? T128x128 : T128x64;
        while (S >= 2 && !config_ok(p, tk, S)) --S;
        if (tiles < 192 && S >= 2) return run_config(p, a, glds_mode() && config_ok(p, T_GLDS + tk, S) ? (m16_mode() == 2 && config_ok(p, T_M16 + tk, S) ? T_M16 : T_GLDS) + tk : tk, S, s);
    }
    // round 5: every tile in its LDS-DMA form wherever that is legal (OTVM_IGEMM_GLDS=0: the register-staged forms, for A/B runs)
    const int use_glds = glds_mode();
    const bool use_m16 = m16_mode() == 2 || (m16_mode() == 1 && M >= m16_min_pixels());
    auto pick = [&](int t) { return use_glds && config_ok(p, T_GLDS + t, 1) ? (use_m16 && config_ok(p, T_M16 + t, 1) ? T_M16 : T_GLDS) + t : t; };
    if (p->Cout <= 32) return launch_tile(pick(T256x32), a, s, 1);
    if (p->Cout <= 64) return launch_tile(pick((M >= 256 * 128) ? T256x64 : T64x64), a, s, 1);
    // Tile choice by workgroup count (thresholds tuned on the whole 1080p frame after the 256-row tiles got their
    // second LDS stage: 256x256 from 256 workgroups (was 480), 256x128 from 128 (was 480): 36.8 -> 37.8 frames/s;
    // OTVM_T_* override them for sweeps).
    if (config_ok(p, T256x256, 1)) {
        const int64_t huge = (int64_t)otvm_ceil_div(M, 256) * otvm_ceil_div(p->Cout, 256);
        static const int t_huge = otvm_probe_int("OTVM_T_HUGE", 256);
        if (huge >= t_huge) return launch_tile(pick(T256x256), a, s, 1);
    }
    const int64_t big = (int64_t)otvm_ceil_div(M, 256) * otvm_ceil_div(p->Cout, 128);
    static const int t_big = otvm_probe_int("OTVM_T_BIG", 128);
    if (big >= t_big) return launch_tile(pick(T256x128), a, s, 1);
    const int64_t mid = (int64_t)otvm_ceil_div(M, 128) * otvm_ceil_div(p->Cout, 128);
    static const int t_mid = otvm_probe_int("OTVM_T_MID", 384);
    if (mid >= t_mid) return launch_tile(pick(T128x128), a, s, 1);
    const int64_t sm = (int64_t)otvm_ceil_div(M, 128) * otvm_ceil_div(p->Cout, 64);
    static const int t_sm = otvm_probe_int("OTVM_T_SM", 384);
    if (sm >= t_sm) return launch_tile(pick(T128x64), a, s, 1);
    return launch_tile(pick(T64x64), a, s, 1);
}
