// STM memory read (reference models/trimap/STM.py:140-163) on the 16-bit matrix cores with fp32-class
// accuracy (operand splitting, see conv_f16x3.hip), over a bank stored in MFMA FRAGMENT ORDER.
//
// The memory bank is this build's own data structure (the reference concatenates fp32 tensors,
// alpha/model.py:481-493), so it is laid out for the instruction that consumes it: when a frame is memorised,
// otvm_bank_pack_f16x3 splits its key/value maps into fp16 hi/lo and writes them as 1-KiB blocks holding exactly
// the 64 x 16-byte lane fragments of one v_mfma_f32_32x32x16_f16 operand:
//   keys   Kf[kvb = hw_pad/32][db = 8][hi|lo][lane][8]:  lane l <- K[32 kvb + (l&31)][16 db + 8 (l>>5) + 0..7]   (A of K.Q^T)
//   values Vf[kvb = hw_pad/16][nb = 16][hi|lo][lane][8]: lane l <- V[16 kvb + 8 (l>>5) + 0..7][32 nb + (l&31)]   (B of P.V)
// The read kernel then streams the bank with fully coalesced 1-KiB wave loads straight into MFMA operand
// registers: no LDS staging, no transposes, no conversion of the (T times larger) memory side.  Only the query
// tile and the probabilities go through LDS.
//
// Workgroup = 64 queries x one CHUNK of the memory axis (4 waves), online softmax over the memory axis in fp32,
// 64x512 fp32 output block in accumulators, per-chunk partials merged by the combine kernel of memory_read.hip.
// The memory axis is the concatenation of the slots, cut into equal chunks of 64-row tiles (a chunk may cross slot
// boundaries) so that queries/64 x chunks fills the chip's 512 resident workgroups evenly: one workgroup per
// (64 queries, slot) gave 640 workgroups at 1080p / T = 5, i.e. a second round at 25 % occupancy (2.16 -> 1.4 ms).
#include "common.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int DK = 128, DV = 512, BQ = 64, BKV = 64;
constexpr int LDQH = DK + 8;      // halfs per Q row in LDS (272 B: conflict-free b128)
constexpr int LDPH = BKV + 8;     // halfs per P row (144 B)

// e^x for x <= 0 as one v_exp_f32 (2^y, 1 ulp) on x * log2(e): the product rounds the exponent by |x| * 6e-8, i.e. a
// relative error below 1e-6 for every term that is not negligible in the softmax sum (x > -17), 1e-7 for the dominant
// ones; results below 2^-126 flush to zero.  libm's expf spends ~20 instructions per value on the last ulp and on
// denormal results -- a third of this kernel's VALU work.  e^(-inf) = 0 as before.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

__device__ __forceinline__ void split1(float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

// ---- bank packing: fp32 [hw,128] / [hw,512] -> fragment-major split fp16 -------------------------------
__global__ __launch_bounds__(256) void bank_pack_keys_kernel(const float* __restrict__ k, int hw, _Float16* __restrict__ kf) {
    // one block per 32-row kv block: 8 d-blocks x 2 (hi/lo) x 64 lanes x 8 halfs
    const int kvb = blockIdx.x;
    for (int i = threadIdx.x; i < 8 * 64; i += 256) {
        const int db = i >> 6, l = i & 63;
        const int row = kvb * 32 + (l & 31), d0 = db * 16 + 8 * (l >> 5);
        f16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = row < hw ? k[(int64_t)row * DK + d0 + j] : 0.f;
            _Float16 h, lw;
            split1(v, h, lw);
            hi[j] = h; lo[j] = lw;
        }
        _Float16* base = kf + (((int64_t)kvb * 8 + db) * 2) * 512;
        *reinterpret_cast<f16x8*>(base + l * 8) = hi;
        *reinterpret_cast<f16x8*>(base + 512 + l * 8) = lo;
    }
}

__global__ __launch_bounds__(256) void bank_pack_vals_kernel(const float* __restrict__ v, int hw, _Float16* __restrict__ vf) {
    // one block per 16-row kv block: 16 n-blocks x 2 x 64 lanes x 8 halfs; lanes read 32 consecutive dv (coalesced)
    const int kvb = blockIdx.x;
    for (int i = threadIdx.x; i < 16 * 64; i += 256) {
        const int nb = i >> 6, l = i & 63;
        const int dv = nb * 32 + (l & 31), r0 = kvb * 16 + 8 * (l >> 5);
        f16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = (r0 + j) < hw ? v[(int64_t)(r0 + j) * DV + dv] : 0.f;
            _Float16 h, lw;
            split1(x, h, lw);
            hi[j] = h; lo[j] = lw;
        }
        _Float16* base = vf + (((int64_t)kvb * 16 + nb) * 2) * 512;
        *reinterpret_cast<f16x8*>(base + l * 8) = hi;
        *reinterpret_cast<f16x8*>(base + 512 + l * 8) = lo;
    }
}

struct Mem3Args {
    const float* q; int q_ld;
    const _Float16* kf[8]; const _Float16* vf[8];
    int hw;
    int tiles_per_slot, total_tiles, chunk_tiles;   // memory axis of this launch in 64-row tiles; tiles per workgroup
    int part0;         // index of this launch's first partial
    float* part_o;     // [partials][hw][512]
    float* part_ml;    // [partials][hw][2]
};

__global__ __launch_bounds__(256, 2) void memory_read_f16x3_kernel(const Mem3Args p) {
    __shared__ __attribute__((aligned(16))) _Float16 Qh[BQ * LDQH];
    __shared__ __attribute__((aligned(16))) _Float16 Ql[BQ * LDQH];
    __shared__ __attribute__((aligned(16))) _Float16 Ph[BQ * LDPH];
    __shared__ __attribute__((aligned(16))) _Float16 Pl[BQ * LDPH];
    __shared__ float red_m[2 * BQ], red_s[2 * BQ], alpha_l[BQ];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q0 = blockIdx.x * BQ, part = p.part0 + blockIdx.y;
    const int g0 = blockIdx.y * p.chunk_tiles;
    const int g1 = g0 + p.chunk_tiles < p.total_tiles ? g0 + p.chunk_tiles : p.total_tiles;
    const int hw = p.hw;

    // query tile -> LDS, split (rows beyond hw are zero)
    for (int i = tid; i < BQ * (DK / 4); i += 256) {
        const int r = i / (DK / 4), c = (i - r * (DK / 4)) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (q0 + r < hw) v = *reinterpret_cast<const f32x4*>(p.q + (int64_t)(q0 + r) * p.q_ld + c);
        f16x4 hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) { _Float16 h, lw; split1(v[j], h, lw); hi[j] = h; lo[j] = lw; }
        *reinterpret_cast<f16x4*>(&Qh[r * LDQH + c]) = hi;
        *reinterpret_cast<f16x4*>(&Ql[r * LDQH + c]) = lo;
    }
    __syncthreads();
    float m_run = -__builtin_huge_valf(), l_run = 0.f;   // of query sb*32 + (lane & 31)

    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    const int sa = wave >> 1, sb = wave & 1;            // S sub-tile of this wave: kv 32sa.., q 32sb..
    const int frow = lane & 31, fh = lane >> 5;
    const float scale = 1.0f / sqrtf((float)DK);        // p / math.sqrt(D_e)  (STM.py:154)

    // the query fragments are re-read from LDS every tile (16 ds_read_b128): holding them costs 64 VGPRs, which at
    // two workgroups per CU are worth more as bank fragments in flight
    const _Float16* qhp = &Qh[(sb * 32 + frow) * LDQH + 8 * fh];
    const _Float16* qlp = &Ql[(sb * 32 + frow) * LDQH + 8 * fh];

    int si = g0 / p.tiles_per_slot, t = g0 - si * p.tiles_per_slot;       // slot and tile inside it
    for (int g = g0; g < g1; ++g, ++t) {
        if (t == p.tiles_per_slot) { t = 0; ++si; }
        const _Float16* __restrict__ Kf = p.kf[si];
        const _Float16* __restrict__ Vf = p.vf[si];
        const int kv0 = t * BKV;
        // ---- S = K Q^T for this wave's 32x32 block: A fragments straight from the packed bank
        f32x16 s, s2;                                    // two accumulators: half the dependent-MFMA chain (a third
                                                         // one measured slower: its 16 registers cost prefetch depth)
#pragma unroll
        for (int e = 0; e < 16; ++e) { s[e] = 0.f; s2[e] = 0.f; }
        const _Float16* kblk = Kf + ((int64_t)(2 * t + sa) * 8 * 2) * 512 + lane * 8;
        // all 16 key fragments of the tile are requested before the first MFMA (64 VGPRs that the PV phase below does
        // not need at the same time): one exposed L2 round trip per tile instead of one per k-step
        f16x8 kh[8], kl[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            kh[ks] = *reinterpret_cast<const f16x8*>(kblk + (ks * 2) * 512);
            kl[ks] = *reinterpret_cast<const f16x8*>(kblk + (ks * 2 + 1) * 512);
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const f16x8 qh = *reinterpret_cast<const f16x8*>(qhp + 16 * ks);
            const f16x8 ql = *reinterpret_cast<const f16x8*>(qlp + 16 * ks);
            s2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl[ks], qh, s2, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[ks], qh, s, 0, 0, 0);
            s2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[ks], ql, s2, 0, 0, 0);
        }
        // (measured, round 2: K up front + V one k-step ahead 1.344 -> 1.274 ms at 1080p / 5 slots; the one-instruction
        // exponential 1.274 -> 1.229; s_setprio around the MFMA bursts 1.229 -> 1.248, rejected; loading only half of the
        // value fragments -- wrong results, timing probe -- 1.223 -> 1.149: the kernel is not bound by the CU's 64 B/clk
        // vector-memory path, so a 128-query workgroup that halves the bank traffic was not built)
        // the first value fragments of the tile travel under the softmax
        f16x8 vh[2][4], vl[2][4];
        {
            const _Float16* vblk = Vf + (((int64_t)(4 * t) * 16 + wave * 4) * 2) * 512 + lane * 8;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                vh[0][b] = *reinterpret_cast<const f16x8*>(vblk + (b * 2) * 512);
                vl[0][b] = *reinterpret_cast<const f16x8*>(vblk + (b * 2 + 1) * 512);
            }
        }
        s += s2;                                         // small terms first, then onto the hi*hi sum
        // ---- online softmax over the memory axis, in registers.  The S^T block keeps a QUERY per lane (column
        // q = lane & 31, rows kv = (e&3) + 8 (e>>2) + 4 (lane>>5)): max and sum over kv are 16 in-lane values plus one
        // exchange with lane ^ 32; the two waves that share a query block (sa = 0, 1) meet through red_m / red_s.
        // Every lane carries the running max / sum of its query in registers (both waves of a pair redundantly).
        const int qq = sb * 32 + frow;
        float v[16];
        {
            const float ninf = -__builtin_huge_valf();
            float mx = ninf;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int kvl = sa * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                v[e] = kv0 + kvl < hw ? s[e] * scale : ninf;
                mx = fmaxf(mx, v[e]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            if (fh == 0) red_m[sa * BQ + qq] = mx;
        }
        __syncthreads();                                 // (A) tile maxima visible; P / alpha of the last tile consumed
        {
            const float m_new = fmaxf(m_run, fmaxf(red_m[qq], red_m[BQ + qq]));
            const float al = fast_exp(m_run - m_new);
            float sum = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f16x4 hi, lo;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float e = fast_exp(v[4 * g + j] - m_new);
                    sum += e;
                    _Float16 h, lw;
                    split1(e, h, lw);
                    hi[j] = h; lo[j] = lw;
                }
                const int kvl = sa * 32 + 8 * g + 4 * fh;
                *reinterpret_cast<f16x4*>(&Ph[qq * LDPH + kvl]) = hi;
                *reinterpret_cast<f16x4*>(&Pl[qq * LDPH + kvl]) = lo;
            }
            sum += __shfl_xor(sum, 32);
            if (fh == 0) {
                red_s[sa * BQ + qq] = sum;
                if (sa == 0) alpha_l[qq] = al;
            }
            m_run = m_new;
            l_run *= al;
        }
        __syncthreads();                                 // (B) P, alpha and the partial sums are visible
        l_run += red_s[qq] + red_s[BQ + qq];

        // ---- rescale O, accumulate P V: wave w owns output channels [128w, 128w+128)
        {
            float alr[2][16];
            bool moved = false;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    alr[a][e] = alpha_l[a * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh];
                    moved |= alr[a][e] != 1.f;
                }
            if (__any(moved)) {                          // the running maxima settle after the first tiles
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
#pragma unroll
                        for (int b = 0; b < 4; ++b) acc[a][b][e] *= alr[a][e];
            }
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            f16x8 ph[2], pl[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                ph[a] = *reinterpret_cast<const f16x8*>(&Ph[(a * 32 + frow) * LDPH + 16 * ks + 8 * fh]);
                pl[a] = *reinterpret_cast<const f16x8*>(&Pl[(a * 32 + frow) * LDPH + 16 * ks + 8 * fh]);
            }
            if (ks + 1 < 4) {                            // the next k-step's value fragments, a k-step ahead
                const _Float16* vblk = Vf + (((int64_t)(4 * t + ks + 1) * 16 + wave * 4) * 2) * 512 + lane * 8;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    vh[(ks + 1) & 1][b] = *reinterpret_cast<const f16x8*>(vblk + (b * 2) * 512);
                    vl[(ks + 1) & 1][b] = *reinterpret_cast<const f16x8*>(vblk + (b * 2 + 1) * 512);
                }
                asm volatile("" ::: "memory");
            }
            const int vb = ks & 1;
            // three passes over the eight accumulator tiles: consecutive MFMAs never share an accumulator
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl[a], vh[vb][b], acc[a][b], 0, 0, 0);
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph[a], vl[vb][b], acc[a][b], 0, 0, 0);
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph[a], vh[vb][b], acc[a][b], 0, 0, 0);
        }
        // no barrier here: the next tile's red_m writes happen after (B), its P / alpha writes after its own (A),
        // which every wave reaches only after finishing the reads above
    }
    // partial results: un-normalised O, running max and sum
    const int dv0 = wave * 128;
    float* po = p.part_o + ((int64_t)part * hw) * DV;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int qq = q0 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
            if (qq < hw) {
#pragma unroll
                for (int b = 0; b < 4; ++b) po[(int64_t)qq * DV + dv0 + 32 * b + frow] = acc[a][b][e];
            }
        }
    if (sa == 0 && fh == 0 && q0 + sb * 32 + frow < hw) {
        float* ml = p.part_ml + ((int64_t)part * hw + q0 + sb * 32 + frow) * 2;
        ml[0] = m_run;
        ml[1] = l_run;
    }
}



}  // namespace

int otvm_memory_read_combine(const float* part_o, const float* part_ml, int T, int hw, float* out, int out_ld, void* stream);

static inline int hw_pad64(int hw) { return (hw + 63) / 64 * 64; }

// chunks of the memory axis for a launch over n slots: enough to fill 512 resident workgroups (2 per CU)
static inline int mr_chunks(int n, int hw) {
    const int qblocks = otvm_ceil_div(hw, BQ), total = n * (hw_pad64(hw) / BKV);
    int s = otvm_ceil_div(512, qblocks);
    return s < 1 ? 1 : (s > total ? total : s);
}

// partial results otvm_memory_read_f16x3 writes for a bank of T slots (sizes the workspace, memory_read.hip)
int otvm_memory_read_f16x3_partials(int T, int hw) {
    int np = 0;
    for (int s0 = 0; s0 < T; s0 += 8) np += mr_chunks(T - s0 < 8 ? T - s0 : 8, hw);
    return np;
}

extern "C" int64_t otvm_bank_slot_bytes_f16x3(int hw) {
    // keys: hw_pad x 128 x (hi,lo) fp16 ; values: hw_pad x 512 x (hi,lo) fp16
    return (int64_t)hw_pad64(hw) * (DK + DV) * 2 * sizeof(_Float16);
}

extern "C" int otvm_bank_pack_f16x3(const float* key, const float* val, int hw, void* slot, void* stream) {
    OTVM_REQUIRE(key && val && slot && hw > 0, "otvm_bank_pack_f16x3: bad arguments");
    const int hp = hw_pad64(hw);
    _Float16* kf = (_Float16*)slot;
    _Float16* vf = kf + (int64_t)hp * DK * 2;
    hipLaunchKernelGGL(bank_pack_keys_kernel, dim3(hp / 32), dim3(256), 0, (hipStream_t)stream, key, hw, kf);
    hipLaunchKernelGGL(bank_pack_vals_kernel, dim3(hp / 16), dim3(256), 0, (hipStream_t)stream, val, hw, vf);
    OTVM_CHECK_LAUNCH("otvm_bank_pack_f16x3");
    return 0;
}

// partial launches over `n` slots: writes partials [part0, return value) of a workspace laid out for `np_cap` partials
static int mr_launch_partials(const float* q_key, int q_ld, const void* const* slots, int n_slots, int hw, void* ws, int np_cap,
                              int part0, hipStream_t stream) {
    const int hp = hw_pad64(hw);
    Mem3Args a;
    a.q = q_key; a.q_ld = q_ld; a.hw = hw;
    a.part_o = (float*)ws;
    a.part_ml = a.part_o + (int64_t)np_cap * hw * DV;
    a.tiles_per_slot = hp / BKV;
    // the reference's bank holds at most 5 slots (config.py:22); larger banks (the "unbounded bank" stress knob of
    // BASELINE configs[4]) are handled 8 slots per launch, one combine over all partials
    for (int s0 = 0; s0 < n_slots; s0 += 8) {
        const int n = n_slots - s0 < 8 ? n_slots - s0 : 8;
        for (int t = 0; t < 8; ++t) {
            a.kf[t] = t < n ? (const _Float16*)slots[s0 + t] : nullptr;
            a.vf[t] = t < n ? (const _Float16*)slots[s0 + t] + (int64_t)hp * DK * 2 : nullptr;
        }
        const int chunks = mr_chunks(n, hw);
        a.total_tiles = n * a.tiles_per_slot;
        a.chunk_tiles = otvm_ceil_div(a.total_tiles, chunks);
        a.part0 = part0;
        // every chunk index < chunks owns at least one tile: chunks <= total_tiles and chunk_tiles = ceil(total/chunks)
        const int used = otvm_ceil_div(a.total_tiles, a.chunk_tiles);
        hipLaunchKernelGGL(memory_read_f16x3_kernel, dim3(otvm_ceil_div(hw, BQ), used), dim3(256), 0, stream, a);
        part0 += used;
    }
    return part0;
}

extern "C" int otvm_memory_read_f16x3(const float* q_key, int q_ld, const void* const* slots, int T, int hw, float* out,
                                      int out_ld, void* ws, void* stream) {
    OTVM_REQUIRE(T >= 1 && T <= 4096, "otvm_memory_read_f16x3: T=%d out of range [1,4096]", T);
    OTVM_REQUIRE(q_key && slots && out && ws && hw > 0, "otvm_memory_read_f16x3: bad arguments");
    OTVM_REQUIRE(q_ld % 4 == 0 && out_ld % 4 == 0, "otvm_memory_read_f16x3: views must be 16-byte aligned");
    const int np = otvm_memory_read_f16x3_partials(T, hw);
    const int used = mr_launch_partials(q_key, q_ld, slots, T, hw, ws, np, 0, (hipStream_t)stream);
    OTVM_CHECK_LAUNCH("otvm_memory_read_f16x3");
    return otvm_memory_read_combine((float*)ws, (float*)ws + (int64_t)np * hw * DV, used, hw, out, out_ld, stream);
}

// The read in two steps, for callers that know part of the bank earlier than the rest (the softmax over the memory axis
// is computed flash-style from per-chunk partials, so the bank may be visited in any grouping and order):
//   otvm_memory_read_f16x3_partial : partials over `n_slots` slots into [part0, *part_end) of a workspace laid out for
//                                    np_cap partials (np_cap >= the sum of otvm_memory_read_f16x3_partials(n, hw) over
//                                    all groups; ws >= np_cap * hw * (512 + 2) floats);
//   otvm_memory_read_f16x3_combine : merges partials [0, n_partials) into out.
extern "C" int otvm_memory_read_f16x3_partial(const float* q_key, int q_ld, const void* const* slots, int n_slots, int hw, void* ws,
                                              int np_cap, int part0, int* part_end, void* stream) {
    OTVM_REQUIRE(n_slots >= 1 && n_slots <= 4096 && q_key && slots && ws && hw > 0 && part_end && q_ld % 4 == 0,
                 "otvm_memory_read_f16x3_partial: bad arguments");
    OTVM_REQUIRE(part0 >= 0 && part0 + otvm_memory_read_f16x3_partials(n_slots, hw) <= np_cap,
                 "otvm_memory_read_f16x3_partial: workspace laid out for %d partials is too small", np_cap);
    *part_end = mr_launch_partials(q_key, q_ld, slots, n_slots, hw, ws, np_cap, part0, (hipStream_t)stream);
    OTVM_CHECK_LAUNCH("otvm_memory_read_f16x3_partial");
    return 0;
}

extern "C" int otvm_memory_read_f16x3_combine(const void* ws, int np_cap, int n_partials, int hw, float* out, int out_ld,
                                              void* stream) {
    OTVM_REQUIRE(ws && out && n_partials >= 1 && n_partials <= np_cap && out_ld % 4 == 0,
                 "otvm_memory_read_f16x3_combine: bad arguments");
    return otvm_memory_read_combine((const float*)ws, (const float*)ws + (int64_t)np_cap * hw * DV, n_partials, hw, out, out_ld, stream);
}

extern "C" int otvm_memory_read_f16x3_partial_count(int n_slots, int hw) { return otvm_memory_read_f16x3_partials(n_slots, hw); }
