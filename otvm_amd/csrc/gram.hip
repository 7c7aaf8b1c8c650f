// GroupNorm statistics of a 1x1 convolution's output PREDICTED from its input (round 4).
//
// The FBA bottleneck ends with  out = relu(GroupNorm(conv3(x')) + identity)  (resnet_GN_WS.py:66-86), x' = relu(GroupNorm(conv2 ..))
// being conv3's (1x1, bias-free, weight-standardised) input.  The GroupNorm needs the mean and variance of conv3's WHOLE
// output before a single value can be normalised, so until round 3 conv3 wrote its raw output (with the statistics
// accumulated in its epilogue) and a separate pass read it back, normalised it, added the identity and wrote the block
// output: 16 such passes per frame, 1.25 ms at 1080p at HBM speed.  But y = W x' is linear, so per group g of cg output channels
//     sum_p sum_{c in g} y_pc    = v_g . s            s  = sum_p x'_p                      (channel sums of the input)
//     sum_p sum_{c in g} y_pc^2  = < G , M_g >        G  = sum_p x'_p x'_p^T               (the input's Gram matrix)
// with v_g = sum_{c in g} w_c and M_g = sum_{c in g} w_c w_c^T computed ONCE per checkpoint (otvm_amd/engine.py).  The Gram
// matrix is a [planes x P] x [P x planes] product -- a quarter of conv3's FLOPs, half of that by symmetry -- on the input that
// conv3 reads anyway.  With mean and rstd known BEFORE conv3 runs, its epilogue normalises (a per-channel scale and shift: the
// epilogue's filter-scale / bias slots), adds the identity, applies the ReLU and writes the block output once.
//
// Precision.  The statistics are sums over 10^5..10^7 values, so unbiased rounding noise on the Gram matrix's operands
// averages out: with x' rounded to fp16 (round-to-nearest, relative error 2^-12 rms 1.7e-4) the error of sum y^2 is
//     2 sum_pc y_pc (w_c . e_p)  [zero mean, relative std ~ 3.4e-4 / sqrt(P cg) ~ 3e-7]  +  sum_pc (w_c . e_p)^2  [bias ~ 3e-8],
// i.e. at the level of the fp32 rounding of mean / rstd themselves; one MFMA pass (PASSES = 1).  PASSES = 3 keeps the
// f16x3 operand split of the convolutions (hi*hi + hi*lo + lo*hi) for comparison (tests/test_gpu_kernels.py measures both
// against the statistics the convolution's own epilogue accumulates).  The channel sums s are taken from the fp32 values.
// Accumulation: products of two fp16 values are exact in the fp32 MFMA accumulator; a workgroup adds 256 pixels in the
// MFMA accumulator, then folds it into a second fp32 accumulator (<= 16 folds), and the per-workgroup partials -- one per
// (block of the matrix, chunk of <= ~1000 pixels) -- are added in fp64 by gn_predict_kernel, which also contracts with M_g.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {

struct GramArgs {
    const float* x; int64_t P; int C, ld;
    const float* in_scale; const float* in_shift; float in_slope;
    float* gpart; float* spart;
    int nk, pch, nblk, nb;
    int64_t x_bs, g_bs, s_bs; int norm_bs;
};

constexpr int LDT = 40;                 // halfs per LDS row: 32 pixels + 8 pad = 80 bytes (conflict-free ds_read_b128, as conv_f16x3.hip)
constexpr int SUBN = 8;                 // chunks of 32 pixels added in the MFMA accumulator before it is folded

// Workgroup = one BS x BS block (bi <= bj) of the Gram matrix over one chunk of pixels; 4 waves as 2 x 2, each (BS/2) x (BS/2).
// K of the MFMA = pixels: both operands are [channel][pixel] with 8 consecutive pixels per lane, i.e. the TRANSPOSE of the
// NHWC tensor.  Staging: a thread loads the same channel quad of four consecutive pixels (lanes along pixel groups, then
// channel quads: 128-byte lines), normalises, and writes four 8-byte pieces [channel][4 pixels] -- conflict-free with the
// 80-byte rows (8 lanes fill 64 bytes of a row, the next 8 lanes sit four rows = 16 banks further).
template <int BS, int PASSES>
__global__ __launch_bounds__(256, 2) void gram_f16_kernel(const GramArgs pa) {
    GramArgs p = pa;
    {
        const int zb = blockIdx.z;
        p.x += zb * p.x_bs;
        p.gpart += zb * p.g_bs;
        p.spart += zb * p.s_bs;
        if (p.in_scale) { p.in_scale += zb * p.norm_bs; p.in_shift += zb * p.norm_bs; }
    }
    constexpr int T = BS / 64;                                   // 32x32 tiles per wave and side
    constexpr int ROWS = 2 * BS;
    constexpr int STAGE = ROWS * LDT * (PASSES == 3 ? 2 : 1);    // halfs per stage
    __shared__ __attribute__((aligned(16))) _Float16 smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    int bi = 0, bj = 0;
    {
        int b = blockIdx.x;
        while (b >= p.nb - bi) { b -= p.nb - bi; ++bi; }
        bj = bi + b;
    }
    const bool diag = bi == bj;
    const int boff = diag ? 0 : BS;                              // first LDS row of the bj channels
    const int kc = blockIdx.y;
    const int64_t p0 = (int64_t)kc * p.pch;
    const int64_t p1 = p0 + p.pch < p.P ? p0 + p.pch : p.P;
    const int nchunks = (int)((p1 - p0 + 31) >> 5);

    // staging items of this thread: (pixel group pg of 4 pixels, channel quad cq); NQ quads in all
    constexpr int MAXIT = (2 * BS / 4) * 8 / 256 > 0 ? (2 * BS / 4) * 8 / 256 : 1;
    const int NQ = (diag ? BS : 2 * BS) / 4;
    const int pg = tid & 7;
    int row0[MAXIT], chan[MAXIT];
    bool live[MAXIT];
    f32x4 sc[MAXIT], sh[MAXIT], ssum[MAXIT];
#pragma unroll
    for (int k = 0; k < MAXIT; ++k) {
        const int cq = (tid >> 3) + 32 * k;
        live[k] = cq < NQ;
        row0[k] = cq * 4;
        chan[k] = live[k] ? (cq * 4 < BS ? bi * BS + cq * 4 : bj * BS + cq * 4 - BS) : 0;
        sc[k] = f32x4{1.f, 1.f, 1.f, 1.f};
        sh[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.in_scale && live[k]) {
            sc[k] = *reinterpret_cast<const f32x4*>(p.in_scale + chan[k]);
            sh[k] = *reinterpret_cast<const f32x4*>(p.in_shift + chan[k]);
        }
        ssum[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 rg[MAXIT][4];
    auto load_chunk = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < MAXIT; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int64_t px = p0 + (int64_t)c * 32 + pg * 4 + i;
                px = px < p1 ? px : p1 - 1;                      // (clamped: a branch around a load hides it from the waitcnt pass)
                rg[k][i] = *reinterpret_cast<const f32x4*>(p.x + px * p.ld + chan[k]);
            }
    };
    auto store_chunk = [&](int c, int buf) __attribute__((always_inline)) {
        _Float16* Th = smem + buf * STAGE;
        _Float16* Tl = Th + ROWS * LDT;
#pragma unroll
        for (int k = 0; k < MAXIT; ++k) {
            if (!live[k]) continue;
            f32x4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = p0 + (int64_t)c * 32 + pg * 4 + i < p1;
                f32x4 t = rg[k][i] * sc[k] + sh[k];
                t.x = t.x > 0.f ? t.x : t.x * p.in_slope; t.y = t.y > 0.f ? t.y : t.y * p.in_slope;
                t.z = t.z > 0.f ? t.z : t.z * p.in_slope; t.w = t.w > 0.f ? t.w : t.w * p.in_slope;
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                v[i] = ok ? t : z;
                ssum[k] += v[i];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f16x4 hi = {(_Float16)v[0][j], (_Float16)v[1][j], (_Float16)v[2][j], (_Float16)v[3][j]};   // round to nearest
                *reinterpret_cast<f16x4*>(&Th[(row0[k] + j) * LDT + pg * 4]) = hi;
                if (PASSES == 3) {
                    const f16x4 lo = {(_Float16)(v[0][j] - (float)hi.x), (_Float16)(v[1][j] - (float)hi.y),
                                      (_Float16)(v[2][j] - (float)hi.z), (_Float16)(v[3][j] - (float)hi.w)};
                    *reinterpret_cast<f16x4*>(&Tl[(row0[k] + j) * LDT + pg * 4]) = lo;
                }
            }
        }
    };
    f32x16 acc[T][T], acc2[T][T];
#pragma unroll
    for (int a = 0; a < T; ++a)
#pragma unroll
        for (int b = 0; b < T; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[a][b][e] = 0.f; acc2[a][b][e] = 0.f; }
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const _Float16* Th = smem + buf * STAGE;
        const _Float16* Tl = Th + ROWS * LDT;
        const int frow = lane & 31, fk = (lane >> 5) * 8;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 ah[T], bh[T], al[T], bl[T];
#pragma unroll
            for (int a = 0; a < T; ++a) {
                const int o = ((wm * T + a) * 32 + frow) * LDT + 16 * ks + fk;
                ah[a] = *reinterpret_cast<const f16x8*>(&Th[o]);
                if (PASSES == 3) al[a] = *reinterpret_cast<const f16x8*>(&Tl[o]);
            }
#pragma unroll
            for (int b = 0; b < T; ++b) {
                const int o = (boff + (wn * T + b) * 32 + frow) * LDT + 16 * ks + fk;
                bh[b] = *reinterpret_cast<const f16x8*>(&Th[o]);
                if (PASSES == 3) bl[b] = *reinterpret_cast<const f16x8*>(&Tl[o]);
            }
            if (PASSES == 3) {
#pragma unroll
                for (int a = 0; a < T; ++a)
#pragma unroll
                    for (int b = 0; b < T; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[a], bh[b], acc[a][b], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < T; ++a)
#pragma unroll
                    for (int b = 0; b < T; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bl[b], acc[a][b], 0, 0, 0);
            }
#pragma unroll
            for (int a = 0; a < T; ++a)
#pragma unroll
                for (int b = 0; b < T; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bh[b], acc[a][b], 0, 0, 0);
        }
    };
    auto fold = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < T; ++a)
#pragma unroll
            for (int b = 0; b < T; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) { acc2[a][b][e] += acc[a][b][e]; acc[a][b][e] = 0.f; }
    };

    if (nchunks > 0) {
        load_chunk(0);
        store_chunk(0, 0);
        __syncthreads();
        for (int c = 0; c < nchunks; ++c) {
            const bool more = c + 1 < nchunks;
            if (more) load_chunk(c + 1);
            compute(c & 1);
            if ((c & (SUBN - 1)) == SUBN - 1) fold();
            if (more) store_chunk(c + 1, (c + 1) & 1);
            __syncthreads();
        }
    }
    fold();
    // ---- partial block
    float* gp = p.gpart + ((int64_t)kc * p.nblk + blockIdx.x) * (BS * BS);
    const int col = lane & 31, rbase = (lane >> 5) * 4;
#pragma unroll
    for (int a = 0; a < T; ++a)
#pragma unroll
        for (int b = 0; b < T; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = (wm * T + a) * 32 + (e & 3) + 8 * (e >> 2) + rbase, n = (wn * T + b) * 32 + col;
                gp[m * BS + n] = acc2[a][b][e];
            }
    // ---- partial channel sums (diagonal blocks cover every channel once)
    if (diag) {
#pragma unroll
        for (int k = 0; k < MAXIT; ++k) {
            f32x4 s = ssum[k];
#pragma unroll
            for (int off = 1; off < 8; off <<= 1) {
                s.x += __shfl_xor(s.x, off); s.y += __shfl_xor(s.y, off); s.z += __shfl_xor(s.z, off); s.w += __shfl_xor(s.w, off);
            }
            if (live[k] && pg == 0) *reinterpret_cast<f32x4*>(p.spart + (int64_t)kc * p.C + chan[k]) = s;
        }
    }
}

struct PredArgs {
    const float* gpart; const float* spart; int nk, E, C;
    const float* Mp; const double* v;
    double* sums; unsigned* counter;
    int64_t P; int Cout;
    const float* wscale; const float* gamma; const float* beta; const float* res_shift;
    float* scale_eff; float* bias_eff; float* stat_out;
    int64_t g_bs, s_bs; int sums_bs, tab_bs, rs_bs;
};

// partial Gram blocks and channel sums -> (sum y, sum y^2) per group -> (mean, rstd) -> the per-channel scale / shift conv3's
// epilogue applies.  Two kinds of workgroups in one launch:
//   E workgroups: PRED_EW entries of the (block upper-triangular) Gram matrix each.  Thread (entry, k phase) adds every
//     PRED_KP-th partial of its entry in fp64 (four independent chains: the first version walked all nk partials of an entry
//     in one dependent chain per thread -- 224 us per launch), the phases meet in LDS, then thread (group, entry phase)
//     multiplies with M_g (stored entry-major: the 32 groups of an entry are one 128-byte line; off-diagonal blocks carry
//     the factor 2 of the symmetric half) and the workgroup adds its 32 partial <G, M_g> to sums[g][1];
//   S workgroups: 16 channels each: the channel sums s_c (partials added the same way) times v_g[c] into sums[g][0].
// fp64 atomics; the last workgroup (ticket) writes the tables and re-arms sums / counter.
constexpr int PRED_EW = 32, PRED_KP = 256 / PRED_EW, PRED_SC = 16, PRED_SKP = 256 / PRED_SC;
__global__ __launch_bounds__(256) void gn_predict_kernel(const PredArgs pa) {
    PredArgs p = pa;
    const int zb = blockIdx.y;
    p.gpart += zb * p.g_bs;
    p.spart += zb * p.s_bs;
    p.sums += zb * p.sums_bs;
    const int tid = threadIdx.x;
    const int n_e = (p.E + PRED_EW - 1) / PRED_EW;
    __shared__ double red[256];
    __shared__ double tot[PRED_EW];
    __shared__ float t_mean[32], t_rstd[32];
    __shared__ unsigned t_last;
    if ((int)blockIdx.x < n_e) {
        const int el = tid % PRED_EW, kp = tid / PRED_EW;
        const int e = blockIdx.x * PRED_EW + el;
        double g0 = 0.0, g1 = 0.0, g2 = 0.0, g3 = 0.0;
        if (e < p.E) {
            const float* gp = p.gpart + e;
            int k = kp;
            for (; k + 3 * PRED_KP < p.nk; k += 4 * PRED_KP) {
                g0 += (double)gp[(int64_t)k * p.E];
                g1 += (double)gp[(int64_t)(k + PRED_KP) * p.E];
                g2 += (double)gp[(int64_t)(k + 2 * PRED_KP) * p.E];
                g3 += (double)gp[(int64_t)(k + 3 * PRED_KP) * p.E];
            }
            for (; k < p.nk; k += PRED_KP) g0 += (double)gp[(int64_t)k * p.E];
        }
        red[tid] = (g0 + g1) + (g2 + g3);
        __syncthreads();
        if (tid < PRED_EW) {
            double t = 0.0;
#pragma unroll
            for (int j = 0; j < PRED_KP; ++j) t += red[j * PRED_EW + tid];
            tot[tid] = t;
        }
        __syncthreads();
        const int g = tid & 31, es = tid >> 5;                       // 8 entry phases
        double a = 0.0;
#pragma unroll
        for (int j = 0; j < PRED_EW / 8; ++j) {
            const int ee = blockIdx.x * PRED_EW + es + 8 * j;
            if (ee < p.E) a += tot[es + 8 * j] * (double)p.Mp[(int64_t)ee * 32 + g];
        }
        __syncthreads();
        red[tid] = a;
        __syncthreads();
        if (tid < 32) {
            double t = 0.0;
#pragma unroll
            for (int j = 0; j < 8; ++j) t += red[j * 32 + tid];
            atomicAdd(&p.sums[2 * tid + 1], t);
        }
    } else {
        const int c0 = ((int)blockIdx.x - n_e) * PRED_SC;
        const int cl = tid % PRED_SC, kp = tid / PRED_SC, c = c0 + cl;
        double s0 = 0.0, s1 = 0.0;
        if (c < p.C) {
            const float* sp = p.spart + c;
            int k = kp;
            for (; k + PRED_SKP < p.nk; k += 2 * PRED_SKP) {
                s0 += (double)sp[(int64_t)k * p.C];
                s1 += (double)sp[(int64_t)(k + PRED_SKP) * p.C];
            }
            for (; k < p.nk; k += PRED_SKP) s0 += (double)sp[(int64_t)k * p.C];
        }
        red[tid] = s0 + s1;
        __syncthreads();
        if (tid < PRED_SC) {
            double t = 0.0;
#pragma unroll
            for (int j = 0; j < PRED_SKP; ++j) t += red[j * PRED_SC + tid];
            tot[tid] = t;
        }
        __syncthreads();
        if (tid < 32) {
            double t = 0.0;
            for (int j = 0; j < PRED_SC; ++j)
                if (c0 + j < p.C) t += tot[j] * p.v[(int64_t)tid * p.C + c0 + j];
            atomicAdd(&p.sums[2 * tid], t);
        }
    }
    // ---- ticket: the last workgroup of this image turns the sums into the tables (common.h::otvm_gn_table_tail has the
    // ordering argument: device-scope atomics, acknowledged before the ticket is taken)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0) t_last = atomicAdd(p.counter + zb, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!t_last) return;
    const int cg = p.Cout / 32;
    if (tid < 32) {
        const double cnt = (double)p.P * cg;
        const double sum = atomicAdd(&p.sums[tid * 2], 0.0), sq = atomicAdd(&p.sums[tid * 2 + 1], 0.0);   // coherent reads
        const double mean = sum / cnt;
        double var = sq / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        t_mean[tid] = (float)mean;
        t_rstd[tid] = (float)(1.0 / sqrt(var + 1e-5));
        if (p.stat_out) { p.stat_out[zb * 64 + 2 * tid] = t_mean[tid]; p.stat_out[zb * 64 + 2 * tid + 1] = t_rstd[tid]; }
    }
    __syncthreads();
    float* se = p.scale_eff + (int64_t)zb * p.tab_bs;
    float* be = p.bias_eff + (int64_t)zb * p.tab_bs;
    const float* rs = p.res_shift ? p.res_shift + (int64_t)zb * p.rs_bs : nullptr;
    for (int c = tid; c < p.Cout; c += 256) {
        const int g = c / cg;
        const float a = t_rstd[g] * p.gamma[c];                   // the arithmetic of gn_apply / gn_table ...
        se[c] = p.wscale[c] * a;                                   // ... times the filter's power-of-two scale (exact)
        float b = p.beta[c] - t_mean[g] * a;
        if (rs) b += rs[c];
        be[c] = b;
    }
    __syncthreads();
    if (tid < 64) p.sums[tid] = 0.0;                               // everyone is done: re-armed for the next launch
    if (tid == 0) __hip_atomic_store(p.counter + zb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace

extern "C" int otvm_gram_block(int C) { return C % 128 == 0 ? 128 : 64; }

extern "C" int64_t otvm_gram_entries(int C) {
    const int bs = otvm_gram_block(C), nb = C / bs;
    return (int64_t)(nb * (nb + 1) / 2) * bs * bs;
}

// pixel chunks of the Gram kernel for a [P][C] input (about 1.5 workgroups per CU in all, >= 256 pixels each)
extern "C" int otvm_gram_chunks(int64_t P, int C, int* pch_out) {
    const int bs = otvm_gram_block(C), nb = C / bs, nblk = nb * (nb + 1) / 2;
    int64_t nk = 384 / nblk;
    if (nk > P / 256) nk = P / 256;
    if (nk < 1) nk = 1;
    int64_t pch = (P + nk - 1) / nk;
    pch = (pch + 31) / 32 * 32;
    nk = (P + pch - 1) / pch;
    if (pch_out) *pch_out = (int)pch;
    return (int)nk;
}

extern "C" int otvm_gram_f16(const otvm_gram_params* q, void* stream) {
    OTVM_REQUIRE(q && q->x && q->gpart && q->spart, "otvm_gram_f16: null pointer");
    OTVM_REQUIRE(q->C % 64 == 0 && q->C >= 64 && q->ld % 4 == 0 && q->ld >= q->C && ((uintptr_t)q->x & 15) == 0,
                 "otvm_gram_f16: C must be a multiple of 64 in a 16-byte aligned view (C %d, ld %d)", q->C, q->ld);
    OTVM_REQUIRE(!q->in_scale == !q->in_shift, "otvm_gram_f16: in_scale and in_shift go together");
    OTVM_REQUIRE(q->passes == 1 || q->passes == 3, "otvm_gram_f16: passes must be 1 (fp16) or 3 (f16x3)");
    GramArgs a;
    a.x = q->x; a.P = q->P; a.C = q->C; a.ld = q->ld;
    a.in_scale = q->in_scale; a.in_shift = q->in_shift;
    a.in_slope = q->in_act == OTVM_ACT_RELU ? 0.f : (q->in_act == OTVM_ACT_LEAKY ? 0.01f : 1.f);
    a.gpart = q->gpart; a.spart = q->spart;
    const int bs = otvm_gram_block(q->C);
    a.nb = q->C / bs; a.nblk = a.nb * (a.nb + 1) / 2;
    int pch = 0;
    a.nk = otvm_gram_chunks(q->P, q->C, &pch);
    a.pch = pch;
    const int batch = q->batch > 1 ? q->batch : 1;
    a.x_bs = batch > 1 ? q->x_bs : 0; a.norm_bs = batch > 1 ? q->norm_bs : 0;
    a.g_bs = (int64_t)a.nk * otvm_gram_entries(q->C); a.s_bs = (int64_t)a.nk * q->C;
    const dim3 grid(a.nblk, a.nk, batch), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (bs == 128) {
        if (q->passes == 3) hipLaunchKernelGGL((gram_f16_kernel<128, 3>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((gram_f16_kernel<128, 1>), grid, block, 0, s, a);
    } else {
        if (q->passes == 3) hipLaunchKernelGGL((gram_f16_kernel<64, 3>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((gram_f16_kernel<64, 1>), grid, block, 0, s, a);
    }
    OTVM_CHECK_LAUNCH("otvm_gram_f16");
    return 0;
}

extern "C" int otvm_gn_predict(const otvm_gn_predict_params* q, void* stream) {
    OTVM_REQUIRE(q && q->gpart && q->spart && q->Mp && q->v && q->sums && q->counter && q->wscale && q->gamma && q->beta &&
                     q->scale_eff && q->bias_eff, "otvm_gn_predict: null pointer");
    OTVM_REQUIRE(q->Cout % 32 == 0 && q->C % 64 == 0, "otvm_gn_predict: Cout must be a multiple of 32 (got %d)", q->Cout);
    PredArgs a;
    a.gpart = q->gpart; a.spart = q->spart; a.C = q->C;
    a.E = (int)otvm_gram_entries(q->C);
    a.nk = otvm_gram_chunks(q->P, q->C, nullptr);
    a.Mp = q->Mp; a.v = q->v; a.sums = q->sums; a.counter = q->counter;
    a.P = q->P; a.Cout = q->Cout;
    a.wscale = q->wscale; a.gamma = q->gamma; a.beta = q->beta; a.res_shift = q->res_shift;
    a.scale_eff = q->scale_eff; a.bias_eff = q->bias_eff; a.stat_out = q->stat_out;
    const int batch = q->batch > 1 ? q->batch : 1;
    a.g_bs = (int64_t)a.nk * a.E; a.s_bs = (int64_t)a.nk * q->C;
    a.sums_bs = batch > 1 ? q->sums_bs : 0; a.tab_bs = batch > 1 ? q->tab_bs : 0; a.rs_bs = batch > 1 ? q->rs_bs : 0;
    hipLaunchKernelGGL(gn_predict_kernel, dim3(otvm_ceil_div(a.E, PRED_EW) + otvm_ceil_div(a.C, PRED_SC), batch), dim3(256), 0,
                       (hipStream_t)stream, a);
    OTVM_CHECK_LAUNCH("otvm_gn_predict");
    return 0;
}
