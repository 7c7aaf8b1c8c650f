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
// Accumulation: products of two fp16 values are exact in the fp32 MFMA accumulator; a workgroup adds its chunk of <= ~1000
// pixels there (~1e-6 of unbiased noise per entry, far below the operands' own), and the per-workgroup partials -- one per
// (block of the matrix, pixel chunk) -- are added in fp64 by gn_predict_kernel, which also contracts them with M_g.
#include "common.h"
#include <stdlib.h>

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
    unsigned* diag;                 // ABI 18, optional: diag[1] |= 2 when an operand had to be clamped to fp16's range
};

constexpr int LDT = 40;                 // halfs per LDS row: 32 pixels + 8 pad = 80 bytes (conflict-free ds_read_b128, as conv_f16x3.hip)

// Workgroup = one BS x BS block (bi <= bj) of the Gram matrix over one chunk of <= 1024 pixels.  BS = 256: 8 waves as 4 x 2
// (each 64 x 128), BS = 128 / 64: 4 waves as 2 x 2.  K of the MFMA = pixels: both operands are [channel][pixel] with 8
// consecutive pixels per lane, i.e. the TRANSPOSE of the NHWC tensor.  Staging: a thread loads the same channel quad of four
// consecutive pixels (lanes along pixel groups, then channel quads: 128-byte lines), normalises, and writes four 8-byte
// pieces [channel][4 pixels] -- conflict-free with the 80-byte rows (8 lanes fill 64 bytes of a row, the next 8 lanes sit
// four rows = 16 banks further).  One fp16 pass costs 8 MFMAs per 32-pixel chunk and wave: the kernel is bound by the loads
// (a block pair stages 2 BS channels: with BS = 128 a 512-channel tensor is read five times, with BS = 256 twice), so the
// loads of two chunks are in flight while a third is multiplied (two LDS stages, one barrier per chunk).
// fp32 accumulation over the <= 1024 pixels of a workgroup: ~1e-6 of unbiased noise per entry, far below the operands' own.
template <int BS, int PASSES>
__global__ __launch_bounds__(BS == 256 ? 512 : 256, BS == 256 ? 1 : 2) void gram_f16_kernel(const GramArgs pa) {
    GramArgs p = pa;
    {
        const int zb = blockIdx.y;
        p.x += zb * p.x_bs;
        p.gpart += zb * p.g_bs;
        p.spart += zb * p.s_bs;
        if (p.in_scale) { p.in_scale += zb * p.norm_bs; p.in_shift += zb * p.norm_bs; }
    }
    // XCD-aware decode: workgroups are dealt to the 8 XCDs round-robin (id & 7), each XCD has its own L2.  All blocks of one
    // pixel chunk go to ONE XCD, back to back: the chunk's pixels come over the fabric once and its nblk block pairs re-read
    // them from that L2 (with the plain (block, chunk) grid a 512-channel tensor crossed the fabric five times: 334 MB, 64-74 us)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int kc = xcd + 8 * (slot / p.nblk), blk = slot % p.nblk;
    if (kc >= p.nk) return;                                      // (the grid is padded to a multiple of 8 chunks)
    constexpr int NT = BS == 256 ? 512 : 256;
    constexpr int WM = BS == 256 ? 4 : 2, WN = 2;
    constexpr int TM = BS / (32 * WM), TN = BS / (32 * WN);      // 32x32 tiles per wave
    constexpr int ROWS = 2 * BS;
    constexpr int STAGE = ROWS * LDT * (PASSES == 3 ? 2 : 1);    // halfs per stage
    __shared__ __attribute__((aligned(16))) _Float16 smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WN, wn = wave % WN;
    int bi = 0, bj = 0;
    {
        int b = blk;
        while (b >= p.nb - bi) { b -= p.nb - bi; ++bi; }
        bj = bi + b;
    }
    const bool diag = bi == bj;
    const int boff = diag ? 0 : BS;                              // first LDS row of the bj channels
    const int64_t p0 = (int64_t)kc * p.pch;
    const int64_t p1 = p0 + p.pch < p.P ? p0 + p.pch : p.P;
    const int nchunks = (int)((p1 - p0 + 31) >> 5);

    // staging items of this thread: (pixel group pg of 4 pixels, channel quad cq); NQ quads in all
    constexpr int MAXIT = ((2 * BS / 4) * 8 + NT - 1) / NT;
    const int NQ = (diag ? BS : 2 * BS) / 4;
    const int pg = tid & 7;
    int row0[MAXIT], chan[MAXIT];
    bool live[MAXIT];
    f32x4 sc[MAXIT], sh[MAXIT], ssum[MAXIT];
#pragma unroll
    for (int k = 0; k < MAXIT; ++k) {
        const int cq = (tid >> 3) + (NT / 8) * k;
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
    constexpr int PF = 2;                                        // chunks of global loads in flight
    f32x4 rg[PF][MAXIT][4];
    auto load_chunk = [&](int c, f32x4 (&r)[MAXIT][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < MAXIT; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int64_t px = p0 + (int64_t)c * 32 + pg * 4 + i;
                px = px < p1 ? px : p1 - 1;                      // (clamped -- also for chunks past the end: no branch around a load)
                r[k][i] = *reinterpret_cast<const f32x4*>(p.x + px * p.ld + chan[k]);
            }
    };
    bool sat = false;
    auto store_chunk = [&](int c, int buf, f32x4 (&r)[MAXIT][4]) __attribute__((always_inline)) {
        _Float16* Th = smem + buf * STAGE;
        _Float16* Tl = Th + ROWS * LDT;
#pragma unroll
        for (int k = 0; k < MAXIT; ++k) {
            if (!live[k]) continue;
            f32x4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = p0 + (int64_t)c * 32 + pg * 4 + i < p1;
                f32x4 t = r[k][i] * sc[k] + sh[k];
                t.x = t.x > 0.f ? t.x : t.x * p.in_slope; t.y = t.y > 0.f ? t.y : t.y * p.in_slope;
                t.z = t.z > 0.f ? t.z : t.z * p.in_slope; t.w = t.w > 0.f ? t.w : t.w * p.in_slope;
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                t = ok ? t : z;
                // the fp16 operands saturate instead of becoming inf (an inf would turn every statistic of the layer into NaN);
                // the event is recorded -- the statistics are wrong then, the host falls back (engine.py, ABI 18)
                const f32x4 c4 = {__builtin_amdgcn_fmed3f(t.x, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(t.y, -65504.f, 65504.f),
                                  __builtin_amdgcn_fmed3f(t.z, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(t.w, -65504.f, 65504.f)};
                sat |= (c4.x != t.x) | (c4.y != t.y) | (c4.z != t.z) | (c4.w != t.w);      // (NaN: med3 returns a bound, NaN != x)
                v[i] = c4;
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
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const _Float16* Th = smem + buf * STAGE;
        const _Float16* Tl = Th + ROWS * LDT;
        const int frow = lane & 31, fk = (lane >> 5) * 8;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 ah[TM], bh[TN], al[TM], bl[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int o = ((wm * TM + a) * 32 + frow) * LDT + 16 * ks + fk;
                ah[a] = *reinterpret_cast<const f16x8*>(&Th[o]);
                if (PASSES == 3) al[a] = *reinterpret_cast<const f16x8*>(&Tl[o]);
            }
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int o = (boff + (wn * TN + b) * 32 + frow) * LDT + 16 * ks + fk;
                bh[b] = *reinterpret_cast<const f16x8*>(&Th[o]);
                if (PASSES == 3) bl[b] = *reinterpret_cast<const f16x8*>(&Tl[o]);
            }
            if (PASSES == 3) {
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[a], bh[b], acc[a][b], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bl[b], acc[a][b], 0, 0, 0);
            }
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bh[b], acc[a][b], 0, 0, 0);
        }
    };

    // chunk c sits in register set c % PF until it is converted into LDS stage c & 1; every path issues the same loads
    // (chunks past the end re-read the last pixel), so the compiler waits with exact vmcnt counts (conv_f16x3.hip, PFS)
    if (nchunks > 0) {
        load_chunk(0, rg[0]);
        load_chunk(1, rg[1]);
        store_chunk(0, 0, rg[0]);
        load_chunk(2, rg[0]);
        __syncthreads();
        for (int c = 0; c < nchunks; c += PF) {
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                if (c + j >= nchunks) break;
                compute((c + j) & 1);
                if (c + j + 1 < nchunks) store_chunk(c + j + 1, (c + j + 1) & 1, rg[(j + 1) % PF]);
                load_chunk(c + j + 3, rg[(j + 1) % PF]);
                __syncthreads();
            }
        }
    }
    if (sat && p.diag) atomicOr(p.diag + 1, 2u);
    // ---- partial block
    float* gp = p.gpart + ((int64_t)kc * p.nblk + blk) * (BS * BS);
    const int col = lane & 31, rbase = (lane >> 5) * 4;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = (wm * TM + a) * 32 + (e & 3) + 8 * (e >> 2) + rbase, n = (wn * TN + b) * 32 + col;
                gp[m * BS + n] = acc[a][b][e];
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
    double* ws; unsigned* counter;                                 // ws: [nwg][64] partial (sum y, sum y^2) per group
    int64_t P; int Cout;
    const float* wscale; const float* gamma; const float* beta; const float* res_shift;
    float* scale_eff; float* bias_eff; float* stat_out;
    int64_t g_bs, s_bs; int ws_bs, tab_bs, rs_bs;
    unsigned* diag;                 // ABI 18, optional: [0] = max over groups / images / launches of the bits of kappa = mean^2 / var
                                    // (non-negative floats order like their bit patterns), [1] |= 1 for a non-finite statistic
};

// partial Gram blocks and channel sums -> (sum y, sum y^2) per group -> (mean, rstd) -> the per-channel scale / shift conv3's
// epilogue applies.  A streaming reduction: thread (entry quad, k phase) adds every KP-th partial of its four consecutive
// entries of the (block upper-triangular) Gram matrix in fp64 -- 16-byte loads, consecutive threads on consecutive entries, two
// independent chains -- and contracts its partial sum with M_g right away (the contraction is linear, so the k phases need
// not meet first): for each of the 32 groups one 16-byte load of M (group-major [32][entries]: coalesced; off-diagonal blocks
// carry the factor 2 of the symmetric half) and four fp64 FMAs.  The 32 sums are reduced over the wave (shuffles) and the
// workgroup (LDS); a workgroup leaves ONE [32][2] partial in the workspace (device-scope atomic stores) and the last
// workgroup (ticket) adds the partials in a fixed order -- deterministic statistics -- and writes the tables.  The first
// workgroups also contract the channel sums s with v_g.
// (Earlier versions: one dependent chain over all partials per thread: 224 us per launch; 5000 workgroups x 32 fp64 atomics on
// the same 32 addresses: 53 us; 8-entry rounds per wave with 32-byte segments per k row: 14-80 us, 0.8 TB/s.)
constexpr int PRED_WG = 1024;
constexpr int PRED_KP = 4;                                         // k phases = waves of a workgroup; a wave = 64 entry quads
__global__ __launch_bounds__(256) void gn_predict_kernel(const PredArgs pa) {
    PredArgs p = pa;
    const int zb = blockIdx.y;
    p.gpart += zb * p.g_bs;
    p.spart += zb * p.s_bs;
    p.ws += (int64_t)zb * p.ws_bs;
    const int tid = threadIdx.x, lane = tid & 63, kp = tid >> 6;
    __shared__ double part[PRED_KP][64][4];
    __shared__ double red[4][64];
    __shared__ float t_mean[32], t_rstd[32];
    __shared__ unsigned t_last;
    const int nq = p.E >> 2;                                       // (E is a multiple of 4096)
    double a2[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};       // groups 8 kp .. 8 kp + 7
    for (int q0 = blockIdx.x * 64; q0 < nq; q0 += gridDim.x * 64) {   // (uniform trip count: nq is a multiple of 64)
        const int q = q0 + lane;
        // ---- this phase's partials of the entry quad: batches of eight 16-byte loads in flight
        const float* gp = p.gpart + 4 * (int64_t)q;
        double s4[4] = {0.0, 0.0, 0.0, 0.0};
        for (int k0 = kp; k0 < p.nk; k0 += 8 * PRED_KP) {
            f32x4 u[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kk = k0 + j * PRED_KP;
                u[j] = *reinterpret_cast<const f32x4*>(gp + (int64_t)(kk < p.nk ? kk : p.nk - 1) * p.E);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (k0 + j * PRED_KP < p.nk) {
                    s4[0] += (double)u[j].x; s4[1] += (double)u[j].y; s4[2] += (double)u[j].z; s4[3] += (double)u[j].w;
                }
            }
        }
        __syncthreads();                                           // (the previous round's readers are done)
#pragma unroll
        for (int j = 0; j < 4; ++j) part[kp][lane][j] = s4[j];
        __syncthreads();
        double g4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) g4[j] = (part[0][lane][j] + part[1][lane][j]) + (part[2][lane][j] + part[3][lane][j]);
        // ---- contraction with M_g for this wave's eight groups (group-major M: consecutive lanes, consecutive entries)
        f32x4 m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = *reinterpret_cast<const f32x4*>(p.Mp + (int64_t)(kp * 8 + j) * p.E + 4 * (int64_t)q);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            a2[j] += (g4[0] * (double)m[j].x + g4[1] * (double)m[j].y) + (g4[2] * (double)m[j].z + g4[3] * (double)m[j].w);
    }
    // ---- channel sums: workgroup w takes channels 64 w .. 64 w + 63 (the first C / 64 workgroups), the same roles
    double a1[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if ((int)blockIdx.x * 64 < p.C) {                              // (uniform)
        const int c = blockIdx.x * 64 + lane;
        double sv = 0.0;
        if (c < p.C)
            for (int k = kp; k < p.nk; k += PRED_KP) sv += (double)p.spart[(int64_t)k * p.C + c];
        __syncthreads();
        part[kp][lane][0] = sv;
        __syncthreads();
        const double st = (part[0][lane][0] + part[1][lane][0]) + (part[2][lane][0] + part[3][lane][0]);
        if (c < p.C) {
#pragma unroll
            for (int j = 0; j < 8; ++j) a1[j] = st * p.v[(int64_t)(kp * 8 + j) * p.C + c];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        double x = a2[j], y = a1[j];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            x += __shfl_xor(x, off);
            y += __shfl_xor(y, off);
        }
        if (lane == 0) { red[0][2 * (kp * 8 + j)] = y; red[0][2 * (kp * 8 + j) + 1] = x; }
    }
    __syncthreads();
    if (tid < 64) __hip_atomic_store(p.ws + (int64_t)blockIdx.x * 64 + tid, red[0][tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ---- ticket (common.h::otvm_gn_table_tail has the ordering argument: device-scope accesses, acknowledged -- vmcnt(0) --
    // before the workgroup takes its ticket)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0) t_last = atomicAdd(p.counter + zb, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!t_last) return;
    {
        // 64 values x nwg partials: thread (value, phase) adds every 4th partial in a fixed order, eight (device-scope, cache-
        // bypassing) loads in flight at a time -- one at a time this tail was most of the kernel (640 partials: ~50 us)
        const int vi = tid & 63, ph = tid >> 6, n = (int)gridDim.x;
        double t = 0.0;
        for (int w0 = ph; w0 < n; w0 += 32) {
            double u[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int w = w0 + 4 * j;
                u[j] = __hip_atomic_load(p.ws + (int64_t)(w < n ? w : n - 1) * 64 + vi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (w0 + 4 * j < n) t += u[j];
        }
        red[ph][vi] = t;
    }
    __syncthreads();
    const int cg = p.Cout / 32;
    if (tid < 32) {
        const double cnt = (double)p.P * cg;
        const double sum = (red[0][2 * tid] + red[1][2 * tid]) + (red[2][2 * tid] + red[3][2 * tid]);
        const double sq = (red[0][2 * tid + 1] + red[1][2 * tid + 1]) + (red[2][2 * tid + 1] + red[3][2 * tid + 1]);
        const double mean = sum / cnt;
        double var = sq / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        t_mean[tid] = (float)mean;
        t_rstd[tid] = (float)(1.0 / sqrt(var + 1e-5));
        if (p.diag) {
            // conditioning of var = E[y^2] - mean^2: an error delta on E[y^2] (the fp16 operands' rounding, ~3e-7 relative after
            // averaging) reaches var amplified by 1 + kappa.  The host reads the running maximum and switches the layer to the
            // f16x3 Gram, or off, when it is large (engine.py)
            const double kappa = mean * mean / (var + 1e-12);
            const bool bad = !(sum == sum) || !(sq == sq) || !(fabs(sum) < 1e300) || !(fabs(sq) < 1e300);
            if (bad) atomicOr(p.diag + 1, 1u);
            else atomicMax(p.diag, __float_as_uint((float)(kappa < 3e38 ? kappa : 3e38)));
        }
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
    if (tid == 0) __hip_atomic_store(p.counter + zb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed
}

}  // namespace

// block size: the largest of 256 / 128 / 64 that divides C (OTVM_GRAM_BS caps it: experiments)
extern "C" int otvm_gram_block(int C) {
    static const int cap = otvm_probe_int("OTVM_GRAM_BS", 128);   // (256: measured slower, one 512-thread workgroup per CU)
    if (C % 256 == 0 && cap >= 256) return 256;
    return C % 128 == 0 && cap >= 128 ? 128 : 64;
}

extern "C" int64_t otvm_gram_entries(int C) {
    const int bs = otvm_gram_block(C), nb = C / bs;
    return (int64_t)(nb * (nb + 1) / 2) * bs * bs;
}

// pixel chunks of the Gram kernel for a [P][C] input: enough workgroups for the chip (about two per CU for the 256-thread
// kernels, one for the 512-thread one), 256 .. 1024 pixels each, at most 128 chunks
extern "C" int otvm_gram_chunks(int64_t P, int C, int* pch_out) {
    const int bs = otvm_gram_block(C), nb = C / bs, nblk = nb * (nb + 1) / 2;
    static const int wgs = otvm_probe_int("OTVM_GRAM_WGS", 0);
    int64_t nk = (wgs > 0 ? wgs : (bs == 256 ? 192 : 384)) / nblk;
    if (nk > 128) nk = 128;
    if (nk > P / 256) nk = P / 256;
    if (nk < (P + 1023) / 1024) nk = (P + 1023) / 1024;
    if (nk < 1) nk = 1;
    int64_t pch = (P + nk - 1) / nk;
    pch = (pch + 31) / 32 * 32;
    nk = (P + pch - 1) / pch;
    if (pch_out) *pch_out = (int)pch;
    return (int)nk;
}

extern "C" int64_t otvm_gn_predict_ws_bytes(void) { return (int64_t)PRED_WG * 64 * sizeof(double); }

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
    a.diag = q->diag;
    const dim3 grid(a.nblk * ((a.nk + 7) / 8) * 8, batch);
    hipStream_t s = (hipStream_t)stream;
    const bool p3 = q->passes == 3;
    if (bs == 256) {
        if (p3) hipLaunchKernelGGL((gram_f16_kernel<256, 3>), grid, dim3(512), 0, s, a);
        else hipLaunchKernelGGL((gram_f16_kernel<256, 1>), grid, dim3(512), 0, s, a);
    } else if (bs == 128) {
        if (p3) hipLaunchKernelGGL((gram_f16_kernel<128, 3>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((gram_f16_kernel<128, 1>), grid, dim3(256), 0, s, a);
    } else {
        if (p3) hipLaunchKernelGGL((gram_f16_kernel<64, 3>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((gram_f16_kernel<64, 1>), grid, dim3(256), 0, s, a);
    }
    OTVM_CHECK_LAUNCH("otvm_gram_f16");
    return 0;
}

extern "C" int otvm_gn_predict(const otvm_gn_predict_params* q, void* stream) {
    OTVM_REQUIRE(q && q->gpart && q->spart && q->Mp && q->v && q->ws && q->counter && q->wscale && q->gamma && q->beta &&
                     q->scale_eff && q->bias_eff, "otvm_gn_predict: null pointer");
    OTVM_REQUIRE(q->Cout % 32 == 0 && q->C % 64 == 0, "otvm_gn_predict: Cout must be a multiple of 32 (got %d)", q->Cout);
    OTVM_REQUIRE(((uintptr_t)q->Mp & 15) == 0 && ((uintptr_t)q->gpart & 15) == 0, "otvm_gn_predict: Mp / gpart must be 16-byte aligned");
    PredArgs a;
    a.gpart = q->gpart; a.spart = q->spart; a.C = q->C;
    a.E = (int)otvm_gram_entries(q->C);
    a.nk = otvm_gram_chunks(q->P, q->C, nullptr);
    a.Mp = q->Mp; a.v = q->v; a.ws = (double*)q->ws; a.counter = q->counter;
    a.P = q->P; a.Cout = q->Cout;
    a.wscale = q->wscale; a.gamma = q->gamma; a.beta = q->beta; a.res_shift = q->res_shift;
    a.scale_eff = q->scale_eff; a.bias_eff = q->bias_eff; a.stat_out = q->stat_out;
    const int batch = q->batch > 1 ? q->batch : 1;
    a.g_bs = (int64_t)a.nk * a.E; a.s_bs = (int64_t)a.nk * q->C;
    a.ws_bs = PRED_WG * 64; a.tab_bs = batch > 1 ? q->tab_bs : 0; a.rs_bs = batch > 1 ? q->rs_bs : 0;
    a.diag = q->diag;
    int nwg = a.E / 256;                                          // 64 entry quads per workgroup and round
    if (nwg > 192) nwg = 192;                                     // (several rounds per workgroup: fewer partials for the tail)
    if (nwg < otvm_ceil_div(a.C, 64)) nwg = otvm_ceil_div(a.C, 64);
    hipLaunchKernelGGL(gn_predict_kernel, dim3(nwg, batch), dim3(256), 0, (hipStream_t)stream, a);
    OTVM_CHECK_LAUNCH("otvm_gn_predict");
    return 0;
}
