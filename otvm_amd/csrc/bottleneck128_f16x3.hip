// One IDENTITY bottleneck of the STM encoders' 1/8-resolution stage (res3.1 - res3.3: planes = 128, 512 -> 512 channels,
// stride 1) as ONE kernel (round 6; the planes-64 sibling is bottleneck_f16x3.hip).
//
// torchvision Bottleneck with eval-mode BatchNorm folded (STM.py:43-51,79-87):
//     t1 = relu(W1 x + b1)  1x1, 512 -> 128      t2 = relu(W2 * t1 + b2)  3x3, 128 -> 128      y = relu(W3 t2 + b3 + x)  1x1, 128 -> 512
// As three launches the block runs 64x64 / 128x128 implicit-GEMM tiles whose K loops are 4 - 36 chunks deep: prologue, epilogue
// and launch boundaries set their time (31 + 46 + 35 us for 18.2 GFLOP at 136x240: 160 TFLOP/s, 16 - 25 % MFMA-busy,
// profiles/r05_layer_roofline_1080p.md) and the 128-channel intermediates make two round trips.  Here a workgroup of four waves
// owns a TH x TW block of output pixels and keeps everything between x and y on chip, three GEMM phases over ONE LDS arena:
//   A  conv1 on the (TH + 2) x (TW + 2) halo'd patch (the 3x3 conv needs t1 one pixel around the block; pixels outside the image
//      are ZERO, the 3x3 conv's padding): K = 512 in eight 64-channel chunks, x through buffer loads (out-of-image rows read
//      zeros without a branch) two chunks ahead -> split -> LDS, two LDS stages, one barrier per chunk; W1 fragments straight
//      from L2 into registers (every wave needs those of its own n-tiles only);
//      t1 lands in LDS as split fp16, [patch pixel][128 + 8] halfs per plane;
//   B  conv2: the nine taps read shifted windows of the t1 patch; weights per (16 channels, filter row) stage = 24 fragment
//      blocks straight from L2 into registers, one stage ahead, NO barrier in the phase; t2 replaces the t1 patch;
//   C  conv3 in two halves of 256 output channels: A fragments from the t2 tile, W3 fragments from L2 the same way; epilogue as
//      everywhere: accumulator tile -> wave-private LDS patch -> 16-byte row-major stores; the accumulators START at x / scale
//      (the per-filter scale is a power of two: exact), so the residual costs no epilogue round trip.
// x is read once (+ halo, + the residual re-read of the block's own pixels), y written once; one launch instead of three.
// fp32 contract as everywhere: operands split into fp16 hi + lo, three MFMA passes, fp32 accumulate; weights pre-split with a
// per-filter power-of-two scale (otvm_split_conv_weight_f16x3) in MFMA B-fragment order (otvm_pack_wave_weight_f16x3:
// [n/32][32-channel chunk][k-step][hi|lo][64 lanes][8 halfs]; the 3x3 conv's chunks are (channel block, tap), tap = 3 ky + kx).
//
// Why not planes = 256 (res4): the t1 patch of a 64-pixel tile alone is 135 KB of split fp16 and a 32-pixel tile streams 4.5 MB of
// weights per workgroup for one 32-row MFMA tile (48 FLOP per weight byte: bound by the L2 -> CU path at ~50 % of the MFMA rate the
// separate launches already reach) -- DESIGN.md 6.
#include "common.h"
#include <hip/hip_fp16.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifdef OTVM_BNK_TIMING
// experiment build only (tools/bottleneck_bench.py --planes128): per-stage time of wave 0 of every workgroup, 100 MHz ticks
__device__ unsigned long long g_bnk128_t[8];
#define B128_STAMP(i)                                                      \
    if (threadIdx.x == 0) {                                                \
        const unsigned long long t_now = wall_clock64();                   \
        atomicAdd(&g_bnk128_t[i], t_now - t_prev);                         \
        t_prev = t_now;                                                    \
    }
#else
#define B128_STAMP(i)
#endif

namespace {

struct Bnk128Args {
    const float* x; float* y;
    const _Float16* w1f; const _Float16* w2f; const _Float16* w3f;      // fragment-major split weights
    const float* s1; const float* s2; const float* s3;                  // per-filter scales (undo the power-of-two scaling)
    const float* b1; const float* b2; const float* b3;                  // folded BatchNorm biases
    int H, W, x_ld, y_ld, tiles_x, tiles_y; OtvmTileWalk walk;
    unsigned x_bytes;                                                   // one image's input view (buffer-resource range)
    int64_t x_bs, y_bs;                                                 // batch: image blockIdx.y
};

constexpr int PL = 128, CIN = 512, COUT = 512;
constexpr int LDT = PL + 8;          // halfs per t1 / t2 row = 272 bytes (17 x 16: consecutive rows on distinct 16-byte slots)

__device__ __forceinline__ void split4c(const f32x4 v, f16x4& hi, f16x4& lo) {
    typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
    const fp16x2 p01 = __builtin_amdgcn_cvt_pkrtz(v.x, v.y);
    const fp16x2 p23 = __builtin_amdgcn_cvt_pkrtz(v.z, v.w);
    const f16x2 h01 = __builtin_bit_cast(f16x2, p01);
    const f16x2 h23 = __builtin_bit_cast(f16x2, p23);
    hi = f16x4{h01.x, h01.y, h23.x, h23.y};
    lo = f16x4{(_Float16)(v.x - (float)h01.x), (_Float16)(v.y - (float)h01.y), (_Float16)(v.z - (float)h23.x),
               (_Float16)(v.w - (float)h23.y)};
}

__device__ __forceinline__ void split1c(float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

#define MFMA3X(ACC, AH, AL, BH, BL)                                             \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(AL, BH, ACC, 0, 0, 0);         \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, BL, ACC, 0, 0, 0);         \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, BH, ACC, 0, 0, 0)

template <int TH, int TW, int NW>
struct Bnk128Geom {
    static constexpr int NT = NW * 64;
    static constexpr int RPT = 32 / TW;                  // output rows per 32-pixel m-tile
    static constexpr int MT2 = TH / RPT;                 // m-tiles of the output block
    static constexpr int PW = TW + 2, PH = TH + 2, NPIX = PH * PW;
    static constexpr int MT1 = (NPIX + 31) / 32;         // m-tiles of the halo'd patch
    static constexpr int WGM = MT2 >= 2 ? 2 : 1, WGN = NW / WGM;         // wave grid: NW = 4: 2 x 2 or 1 x 4; NW = 8: 2 x 4
    static constexpr int TMA = MT1 / WGM, TNA = 4 / WGN;                 // stage A: m-tiles x n-tiles per wave (N = 128)
    static constexpr int TMB = MT2 / WGM, TNB = 4 / WGN;                 // stage B
    static constexpr int TNC = 8 / WGN;                                  // stage C: n-tiles per wave and half (256 channels)
    static constexpr int T1_HALFS = MT1 * 32 * LDT;                      // per hi / lo plane
    static constexpr int T2_HALFS = MT2 * 32 * LDT;
    static constexpr int CK = 64, LDA = CK + 8;                          // stage A: channels per chunk; halfs per row of an x stage (144 bytes: 9 x 16)
    static constexpr int A_STAGE = 2 * MT1 * 32 * LDA;                   // ... (hi, lo) x stage; two of them
    static constexpr int EPI_OFF = 2 * T2_HALFS;                         // halfs; NW wave-private 16 x 36 fp32 patches (half a tile)
    static constexpr int END_A = 2 * A_STAGE, END_B = 2 * T1_HALFS, END_C = EPI_OFF + NW * 16 * 36 * 2;
    static constexpr int LDS_HALFS = END_A > END_B ? (END_A > END_C ? END_A : END_C) : (END_B > END_C ? END_B : END_C);
    static constexpr int LDS_BYTES = LDS_HALFS * 2;
    static_assert(32 % TW == 0 && TH % RPT == 0 && MT1 % WGM == 0 && MT2 % WGM == 0 && WGN <= 4 && (MT1 * 32 * (CK / 4)) % NT == 0, "bad tile");
    static_assert(LDS_BYTES <= 160 * 1024, "the arena must fit the CU's LDS");
};

// four-wave forms: two workgroups per CU (the 8 x 8 and 4 x 8 arenas are 73 / 51 KiB), i.e. two waves per SIMD and 256 registers --
// the two workgroups run out of phase, one's HBM-bound stage A beside the other's MFMA-bound stages B / C
template <int TH, int TW, int NW>
__global__ __launch_bounds__(NW * 64)
__attribute__((amdgpu_waves_per_eu((NW == 8 || Bnk128Geom<TH, TW, NW>::LDS_BYTES <= 80 * 1024) ? 2 : 1, (NW == 8 || Bnk128Geom<TH, TW, NW>::LDS_BYTES <= 80 * 1024) ? 2 : 10)))
void stm_bottleneck128_f16x3_kernel(const Bnk128Args pa) {
    using G = Bnk128Geom<TH, TW, NW>;
    constexpr int NT = G::NT;
    constexpr int RPT = G::RPT, MT1 = G::MT1, PW = G::PW, NPIX = G::NPIX;
    constexpr int WGM = G::WGM, TMA = G::TMA, TNA = G::TNA, TMB = G::TMB, TNB = G::TNB, TNC = G::TNC;
    Bnk128Args p = pa;
    p.x += blockIdx.y * p.x_bs;
    p.y += blockIdx.y * p.y_bs;
    extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 31, fh = lane >> 5;
    const int wave_m = wave % WGM, wave_n = wave / WGM;
    int tile_n_, tile_x, tile_y;
    otvm_tile_decode(p.walk, blockIdx.x, gridDim.x, 1, p.tiles_x, p.tiles_y, tile_n_, tile_x, tile_y);
    const int ty0 = tile_y * TH, tx0 = tile_x * TW;
#ifdef OTVM_BNK_TIMING
    unsigned long long t_prev = wall_clock64();
#endif

    // ------------------------------------------------------------------ stage A: t1 = relu(W1 x + b1) on the halo'd patch
    {
        f32x16 acc[TMA][TNA];
#pragma unroll
        for (int a = 0; a < TMA; ++a)
#pragma unroll
            for (int b = 0; b < TNA; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
        // staging: thread -> row (tid / QR) + AR i of the patch, channels 4 (tid % QR) .. + 3 of the CK-channel chunk
        constexpr int CK = G::CK, LDA = G::LDA, QR = CK / 4, AR = NT / QR, NA = MT1 * 32 / AR, KS = CK / 16;
        const int arow = tid / QR, ak = (tid % QR) * 4;
        unsigned aoff[NA];                                               // byte offset of the row's chunk-0 quad; 2^31 = outside the image
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int pr = arow + AR * i;
            const int py = pr / PW, px = pr - py * PW;
            const int iy = ty0 - 1 + py, ix = tx0 - 1 + px;
            const bool ok = (pr < NPIX) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
            // (a value, not a load, is selected: outside rows carry EXACTLY 2^31 -- beyond the resource's range for every chunk)
            aoff[i] = ok ? ((unsigned)(iy * p.W + ix) * (unsigned)p.x_ld + (unsigned)ak) << 2 : 0x80000000u;
        }
        __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
        // x runs TWO chunks ahead of the MFMAs (two register sets: 2 x 48 KiB in flight per CU -- every workgroup of the launch
        // streams x at the same time, the phase is bound by HBM / Infinity-Cache latency x bytes in flight).  W1: every wave needs
        // the fragments of ITS n-tiles only -- straight from L2 into registers, one chunk ahead, as in stages B and C (no weight
        // stage, no LDS-DMA bookkeeping).  64-channel chunks: eight barriers in the phase instead of sixteen -- the barrier per
        // chunk is what the x stage cannot lose (its rows are shared by the waves of a row group)
        f32x4 ra0[NA], ra1[NA];
        constexpr int NCH = CIN / CK, NF1 = TNA * KS * 2;                 // fragments of a chunk: (n-tile, k-step, hi|lo)
        f16x8 uA[NF1], uB[NF1];
        auto load_x = [&](int c, f32x4 (&ra)[NA]) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < NA; ++i)
                ra[i] = __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(x_rsrc, aoff[i] + (unsigned)(c * CK * 4), 0, 0));
        };
        auto fetch1 = [&](int c, f16x8 (&w)[NF1]) __attribute__((always_inline)) {
#pragma unroll
            for (int b = 0; b < TNA; ++b)
#pragma unroll
                for (int k4 = 0; k4 < KS; ++k4)
#pragma unroll
                    for (int hl = 0; hl < 2; ++hl)
                        w[(b * KS + k4) * 2 + hl] = *reinterpret_cast<const f16x8*>(
                            p.w1f + ((int64_t)((wave_n * TNA + b) * (CIN / 32) + c * (CK / 32) + (k4 >> 1)) * 4 + (k4 & 1) * 2 + hl) * 512 + lane * 8);
        };
        auto store = [&](int buf, const f32x4 (&ra)[NA]) __attribute__((always_inline)) {
            _Float16* Ah = smem + buf * G::A_STAGE;
            _Float16* Al = Ah + MT1 * 32 * LDA;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                f16x4 hi, lo;
                split4c(ra[i], hi, lo);
                *reinterpret_cast<f16x4*>(&Ah[(arow + AR * i) * LDA + ak]) = hi;
                *reinterpret_cast<f16x4*>(&Al[(arow + AR * i) * LDA + ak]) = lo;
            }
        };
        auto compute = [&](int buf, const f16x8 (&w)[NF1]) __attribute__((always_inline)) {
            const _Float16* Ah = smem + buf * G::A_STAGE;
            const _Float16* Al = Ah + MT1 * 32 * LDA;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                f16x8 ah[TMA], al[TMA];
#pragma unroll
                for (int a = 0; a < TMA; ++a) {
                    const int o = ((wave_m * TMA + a) * 32 + frow) * LDA + ks * 16 + fh * 8;
                    ah[a] = *reinterpret_cast<const f16x8*>(&Ah[o]);
                    al[a] = *reinterpret_cast<const f16x8*>(&Al[o]);
                }
#pragma unroll
                for (int a = 0; a < TMA; ++a)
#pragma unroll
                    for (int b = 0; b < TNA; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[a], w[(b * KS + ks) * 2], acc[a][b], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < TMA; ++a)
#pragma unroll
                    for (int b = 0; b < TNA; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], w[(b * KS + ks) * 2 + 1], acc[a][b], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < TMA; ++a)
#pragma unroll
                    for (int b = 0; b < TNA; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], w[(b * KS + ks) * 2], acc[a][b], 0, 0, 0);
            }
        };
        // one chunk: `cur` (chunk c's x set, staged already) receives chunk c + 2, `nxt` holds chunk c + 1 and is staged behind the
        // MFMAs into the other LDS stage (last read a chunk ago: every wave is past that chunk's barrier)
        auto step = [&](int c, f32x4 (&cur)[NA], f32x4 (&nxt)[NA], const f16x8 (&w)[NF1], f16x8 (&wn)[NF1]) __attribute__((always_inline)) {
            // UNCONDITIONAL loads (the tail re-reads the last chunk: harmless): a branch around a load makes the number of loads in
            // flight unknown to the compiler's waitcnt pass at the join, which then drains vmcnt to the shorter path's count -- the
            // first build of this loop waited for the loads it had just issued (ISA: vmcnt(7) ... vmcnt(0) under the MFMAs)
            fetch1(c + 1 < NCH ? c + 1 : NCH - 1, wn);
            load_x(c + 2 < NCH ? c + 2 : NCH - 1, cur);
            __builtin_amdgcn_sched_barrier(0);                           // (the loads stay HERE: the scheduler would sink them to their use)
            // the two waves of a SIMD (w, w + 4) run the step in opposite orders -- one splits and stores chunk c + 1 (VALU, LDS
            // writes) while the other multiplies chunk c (matrix pipe), then they swap: in lock step both would stage, then both
            // multiply, and the phases would add.  Legal in either order: chunk c + 1's x landed a chunk ago, its LDS stage was last
            // read before the previous barrier.  (After the last chunk the store goes to a stage nobody reads any more.)
            if (NW == 8 && wave >= 4) {
                store((c + 1) & 1, nxt);
                compute(c & 1, w);
            } else {
                compute(c & 1, w);
                store((c + 1) & 1, nxt);
            }
            __syncthreads();
        };
        fetch1(0, uA);
        load_x(0, ra0);
        load_x(1, ra1);
        __builtin_amdgcn_sched_barrier(0);
        store(0, ra0);
        __syncthreads();
        for (int c = 0; c < NCH; c += 2) {
            step(c, ra0, ra1, uA, uB);
            step(c + 1, ra1, ra0, uB, uA);
        }
        B128_STAMP(0);                                                   // stage A, K loop
        // t1 -> LDS (split) over the staging area (every wave is past the loop's last barrier); patch pixels outside the image are
        // ZERO -- the 3x3 conv's padding, not relu(b1)
        _Float16* T1h = smem;
        _Float16* T1l = smem + G::T1_HALFS;
#pragma unroll
        for (int a = 0; a < TMA; ++a) {
            const int mt = wave_m * TMA + a;
            bool ok[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int pr = mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                const int py = pr / PW, px = pr - py * PW;
                const int iy = ty0 - 1 + py, ix = tx0 - 1 + px;
                ok[e] = (pr < NPIX) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
            }
#pragma unroll
            for (int b = 0; b < TNA; ++b) {
                const int n = (wave_n * TNA + b) * 32 + frow;
                const float sc = p.s1[n], bi = p.b1[n];
                const int o = (mt * 32 + 4 * fh) * LDT + n;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float v = acc[a][b][e] * sc + bi;
                    v = ok[e] ? (v > 0.f ? v : 0.f) : 0.f;
                    _Float16 h, l;
                    split1c(v, h, l);
                    T1h[o + ((e & 3) + 8 * (e >> 2)) * LDT] = h;
                    T1l[o + ((e & 3) + 8 * (e >> 2)) * LDT] = l;
                }
            }
        }
    }

    B128_STAMP(1);                                                       // t1 -> LDS
    // the residual (accumulator layout) of a half of the output channels: requested behind stage B (it arrives while t2 is written)
    // and behind the first half's GEMM (it arrives under that half's epilogue); the accumulators START at x / scale (the per-filter
    // scale is a power of two: exact), so the residual costs no epilogue round trip.  (Measured, rejected: adding x in the epilogue
    // from 16-byte loads in the stores' layout -- no accumulator-layout dword loads, 40 registers fewer -- 75.8 vs 68.0 us at 136x240:
    // the loads' latency lands on the store tail.)
    f32x16 res[TMB][TNC];
    auto load_res = [&](int half) __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < TMB; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int pp = (wave_m * TMB + a) * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                const int y = ty0 + pp / TW, x = tx0 + pp % TW;
                const bool ok = (y < p.H) & (x < p.W);
                // (32-bit element offsets from the uniform base: one address register per load instead of two; the views are < 2^31 bytes)
                const unsigned off = (ok ? (unsigned)(y * p.W + x) * (unsigned)p.x_ld : 0u) + (unsigned)((half * 8 + wave_n * TNC) * 32 + frow);
#pragma unroll
                for (int b = 0; b < TNC; ++b) res[a][b][e] = p.x[off + (unsigned)(b * 32)];
            }
    };

    // ------------------------------------------------------------------ stage B: t2 = relu(W2 * t1 + b2), 3x3 over the patch
    // The t1 patch is read-only now and every wave needs the weight fragments of ITS n-tiles only: they come straight from L2 into
    // registers, one stage (16 channels x one filter row = 3 taps) ahead of the MFMAs -- no weight stage in LDS, NO barrier in the
    // whole phase: the waves drift apart and the two waves of a SIMD cover each other's fragment-read latencies (the version with
    // LDS-DMA weight stages and one barrier per stage ran its 24 stages in lock step: 29 us against 12 us of MFMA time,
    // profiles/r06_stm_bottleneck128.txt).  The wave_m partner reads the same fragments (L1 / L2 hits).
    f32x16 acc2[TMB][TNB];
#pragma unroll
    for (int a = 0; a < TMB; ++a)
#pragma unroll
        for (int b = 0; b < TNB; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[a][b][e] = 0.f;
    {
        const _Float16* T1h = smem;
        const _Float16* T1l = smem + G::T1_HALFS;
        constexpr int NF2 = 3 * TNB * 2;                                  // fragments of a stage: (tap, n-tile, hi|lo)
        constexpr int NST = (PL / 16) * 3;
        f16x8 wA[NF2], wB[NF2];
        auto fetch = [&](int s, f16x8 (&w)[NF2]) __attribute__((always_inline)) {
            const int cb = s / 3, g = s - cb * 3;
#pragma unroll
            for (int tl = 0; tl < 3; ++tl)
#pragma unroll
                for (int b = 0; b < TNB; ++b)
#pragma unroll
                    for (int hl = 0; hl < 2; ++hl) {
                        const int64_t src = ((((int64_t)(wave_n * TNB + b) * 36 + (cb >> 1) * 9 + 3 * g + tl) * 2 + (cb & 1)) * 2 + hl) * 512 + lane * 8;
                        w[(tl * TNB + b) * 2 + hl] = *reinterpret_cast<const f16x8*>(p.w2f + src);
                    }
        };
        const int prow = frow / TW, pcol = frow % TW;
        auto stage = [&](int s, const f16x8 (&w)[NF2], f16x8 (&wn)[NF2]) __attribute__((always_inline)) {
            if (s + 1 < NST) fetch(s + 1, wn);
            __builtin_amdgcn_sched_barrier(0);                           // (the loads stay HERE: the scheduler would sink them to their use)
            const int cb = s / 3, g = s - cb * 3;
#pragma unroll
            for (int tl = 0; tl < 3; ++tl) {
                f16x8 ah[TMB], al[TMB];
#pragma unroll
                for (int a = 0; a < TMB; ++a) {
                    const int o = (((wave_m * TMB + a) * RPT + prow + g) * PW + pcol + tl) * LDT + cb * 16 + 8 * fh;
                    ah[a] = *reinterpret_cast<const f16x8*>(&T1h[o]);
                    al[a] = *reinterpret_cast<const f16x8*>(&T1l[o]);
                }
#pragma unroll
                for (int a = 0; a < TMB; ++a)
#pragma unroll
                    for (int b = 0; b < TNB; ++b) acc2[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[a], w[(tl * TNB + b) * 2], acc2[a][b], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < TMB; ++a)
#pragma unroll
                    for (int b = 0; b < TNB; ++b) acc2[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], w[(tl * TNB + b) * 2 + 1], acc2[a][b], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < TMB; ++a)
#pragma unroll
                    for (int b = 0; b < TNB; ++b) acc2[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], w[(tl * TNB + b) * 2], acc2[a][b], 0, 0, 0);
            }
        };
        fetch(0, wA);
        __syncthreads();                                                 // t1 complete
        for (int s = 0; s < NST; s += 2) {
            stage(s, wA, wB);
            stage(s + 1, wB, wA);
        }
    }
    B128_STAMP(2);                                                       // stage B
    // stage C's weights the same way: stage s = (half, 32-channel chunk c), fragments (n-tile, k-step, hi|lo) of the wave's TNC n-tiles
    constexpr int NF3 = TNC * 4;
    f16x8 vA[NF3], vB[NF3];
    auto fetch3 = [&](int s, f16x8 (&w)[NF3]) __attribute__((always_inline)) {
        const int half = s >> 2, c = s & 3;
#pragma unroll
        for (int b = 0; b < TNC; ++b)
#pragma unroll
            for (int sub = 0; sub < 4; ++sub)
                w[b * 4 + sub] = *reinterpret_cast<const f16x8*>(p.w3f + ((int64_t)((half * 8 + wave_n * TNC + b) * (PL / 32) + c) * 4 + sub) * 512 + lane * 8);
    };
    fetch3(0, vA);
    load_res(0);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                                     // every wave is done with the t1 patch: t2 tile -> LDS (split), over the t1 planes
    _Float16* T2h = smem;
    _Float16* T2l = smem + G::T2_HALFS;
#pragma unroll
    for (int a = 0; a < TMB; ++a)
#pragma unroll
        for (int b = 0; b < TNB; ++b) {
            const int n = (wave_n * TNB + b) * 32 + frow;
            const float sc = p.s2[n], bi = p.b2[n];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int pr = (wave_m * TMB + a) * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                float v = acc2[a][b][e] * sc + bi;
                v = v > 0.f ? v : 0.f;
                _Float16 h, l;
                split1c(v, h, l);
                T2h[pr * LDT + n] = h;
                T2l[pr * LDT + n] = l;
            }
        }
    __syncthreads();
    B128_STAMP(3);                                                       // t2 -> LDS

    // ------------------------------------------------------------------ stage C: y = relu(W3 t2 + b3 + x), two halves of 256 channels
    float* patch = reinterpret_cast<float*>(smem + G::EPI_OFF) + wave * (16 * 36);      // half a 32 x 32 tile at a time
    f32x16 acc[TMB][TNC];
    auto stage3 = [&](int s, const f16x8 (&w)[NF3], f16x8 (&wn)[NF3]) __attribute__((always_inline)) {
        if (s + 1 < 8) fetch3(s + 1, wn);
        __builtin_amdgcn_sched_barrier(0);
        const int c = s & 3;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 ah[TMB], al[TMB];
#pragma unroll
            for (int a = 0; a < TMB; ++a) {
                const int o = ((wave_m * TMB + a) * 32 + frow) * LDT + c * 32 + ks * 16 + 8 * fh;
                ah[a] = *reinterpret_cast<const f16x8*>(&T2h[o]);
                al[a] = *reinterpret_cast<const f16x8*>(&T2l[o]);
            }
#pragma unroll
            for (int a = 0; a < TMB; ++a)
#pragma unroll
                for (int b = 0; b < TNC; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[a], w[b * 4 + ks * 2], acc[a][b], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < TMB; ++a)
#pragma unroll
                for (int b = 0; b < TNC; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], w[b * 4 + ks * 2 + 1], acc[a][b], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < TMB; ++a)
#pragma unroll
                for (int b = 0; b < TNC; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], w[b * 4 + ks * 2], acc[a][b], 0, 0, 0);
        }
    };
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int b = 0; b < TNC; ++b) {
            const int n = (half * 8 + wave_n * TNC + b) * 32 + frow;
            const float inv = 1.0f / p.s3[n];
#pragma unroll
            for (int a = 0; a < TMB; ++a)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][b][e] = res[a][b][e] * inv;
        }
        stage3(half * 4 + 0, vA, vB);
        stage3(half * 4 + 1, vB, vA);
        stage3(half * 4 + 2, vA, vB);
        stage3(half * 4 + 3, vB, vA);
        if (half == 0) {                                                 // the second half's residual arrives under this half's epilogue
            load_res(1);
            __builtin_amdgcn_sched_barrier(0);
        }
        B128_STAMP(4 + 2 * half);                                        // GEMM of this half
        // epilogue of this half: accumulator tile -> wave-private LDS patch (16 rows at a time: registers 8 h .. 8 h + 7 of the
        // tile are its rows 16 h .. 16 h + 15) -> 16-byte row-major stores (scale, bias, ReLU)
        const int col = lane & 31, rbase = fh * 4;
        const int erow = lane >> 3, pc = (lane & 7) * 4;
#pragma unroll
        for (int b = 0; b < TNC; ++b) {
            const int n4 = (half * 8 + wave_n * TNC + b) * 32 + pc;
            const f32x4 sc4 = *reinterpret_cast<const f32x4*>(p.s3 + n4), bi4 = *reinterpret_cast<const f32x4*>(p.b3 + n4);
#pragma unroll
            for (int a = 0; a < TMB; ++a)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) patch[((e & 3) + 8 * (e >> 2) + rbase) * 36 + col] = acc[a][b][8 * h + e];
                    // (one wave writes and reads its own patch: the compiler orders the LDS accesses, no barrier needed)
#pragma unroll
                    for (int r2 = 0; r2 < 2; ++r2) {
                        const int xi = r2 * 8 + erow;
                        const int pp = (wave_m * TMB + a) * 32 + 16 * h + xi;
                        const int y = ty0 + pp / TW, x = tx0 + pp % TW;
                        f32x4 v = *reinterpret_cast<const f32x4*>(&patch[xi * 36 + pc]);
                        v = v * sc4 + bi4;
                        v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
                        v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
                        if (y < p.H && x < p.W) *reinterpret_cast<f32x4*>(p.y + ((unsigned)(y * p.W + x) * (unsigned)p.y_ld + (unsigned)n4)) = v;
                    }
                }
        }
        B128_STAMP(5 + 2 * half);                                        // epilogue of this half
    }
}

template <int TH, int TW, int NW>
int launch128(const otvm_stm_bottleneck_params* q, hipStream_t stream) {
    using G = Bnk128Geom<TH, TW, NW>;
    Bnk128Args a;
    a.x = q->x; a.y = q->y; a.w1f = (const _Float16*)q->w1f; a.w2f = (const _Float16*)q->w2f; a.w3f = (const _Float16*)q->w3f;
    a.s1 = q->s1; a.s2 = q->s2; a.s3 = q->s3; a.b1 = q->b1; a.b2 = q->b2; a.b3 = q->b3;
    a.H = q->H; a.W = q->W; a.x_ld = q->x_ld; a.y_ld = q->y_ld;
    a.tiles_x = otvm_ceil_div(q->W, TW); a.tiles_y = otvm_ceil_div(q->H, TH);
    a.walk = otvm_tile_walk_of(8);
    a.x_bytes = (unsigned)((int64_t)q->H * q->W * q->x_ld * 4);
    const int batch = q->batch > 1 ? q->batch : 1;
    a.x_bs = batch > 1 ? q->x_bs : 0; a.y_bs = batch > 1 ? q->y_bs : 0;
    static std::atomic<bool> done[OTVM_MAX_DEVICES];
    hipError_t e = otvm_reserve_lds_once(done, stm_bottleneck128_f16x3_kernel<TH, TW, NW>, G::LDS_BYTES);
    if (e != hipSuccess) {
        otvm_set_error("otvm_stm_bottleneck_f16x3 (planes 128): cannot reserve %d bytes of LDS: %s", G::LDS_BYTES, hipGetErrorString(e));
        return 2;
    }
    hipLaunchKernelGGL((stm_bottleneck128_f16x3_kernel<TH, TW, NW>), dim3(a.tiles_x * a.tiles_y, batch), dim3(NW * 64), G::LDS_BYTES, stream, a);
    OTVM_CHECK_LAUNCH("otvm_stm_bottleneck_f16x3 (planes 128)");
    return 0;
}

}  // namespace

#ifdef OTVM_BNK_TIMING
extern "C" int otvm_debug_bnk128_times(unsigned long long* out8, int reset) {
    if (out8) hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_bnk128_t), 64);
    if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_bnk128_t), z, 64); }
    return 0;
}
#endif

// tile: 0 = chosen from the map size; 1 = 8 x 16 pixels (eight waves), 2 = 8 x 8 (eight waves), 3 = 4 x 8 (four waves, two
// workgroups per CU): the tuner's candidates
int otvm_stm_bottleneck128_launch(const otvm_stm_bottleneck_params* q, int tile, void* stream) {
    OTVM_REQUIRE(q->x_ld % 4 == 0 && q->y_ld % 4 == 0 && ((uintptr_t)q->x & 15) == 0 && ((uintptr_t)q->y & 15) == 0 &&
                 q->x_ld >= CIN && q->y_ld >= COUT, "otvm_stm_bottleneck_f16x3: views must be 16-byte aligned");
    OTVM_REQUIRE((int64_t)q->H * q->W * q->x_ld * 4 < (1ll << 31) && (int64_t)q->H * q->W * q->y_ld * 4 < (1ll << 31),
                 "otvm_stm_bottleneck_f16x3: views too large for 32-bit byte offsets");
    if (tile == 0) {
        // one workgroup per CU in ONE round when the map is large enough, else the tile that still gives every CU work
        const int64_t px = (int64_t)q->H * q->W * (q->batch > 1 ? q->batch : 1);
        tile = px >= 128 * 192 ? 1 : (px >= 64 * 160 ? 2 : 3);
    }
    switch (tile) {
        case 1: return launch128<8, 16, 8>(q, (hipStream_t)stream);
        case 2: return launch128<8, 8, 8>(q, (hipStream_t)stream);
        case 3: return launch128<4, 8, 4>(q, (hipStream_t)stream);
    }
    otvm_set_error("otvm_stm_bottleneck_f16x3: unknown tile %d for the planes-128 block", tile);
    return 1;
}
