// One ResNet bottleneck of the STM encoders' 1/4-resolution stage as ONE kernel (round 3).
//
// torchvision Bottleneck with eval-mode BatchNorm folded (STM.py:43-51,79-87; planes = 64):
//     t1 = relu(W1 x + b1)  1x1, Cin -> 64        t2 = relu(W2 * t1 + b2)  3x3, 64 -> 64
//     y  = relu(W3 t2 + b3 + identity)  1x1, 64 -> 256,   identity = x (Cin = 256)  or  Wd x + bd (first block, Cin = 64)
// As three launches the block is HBM-bound, not MFMA-bound: x (134 MB at 1080p) is read twice, the 64-channel
// intermediates make two round trips, the 1x1 convs run at 3-4 TB/s and 50-90 TFLOP/s (profiles/r03_layer_roofline_1080p.md).
// Here a workgroup owns an 8x32 block of output pixels and keeps everything between x and y on chip:
//   A  conv1 on the 10x34 halo'd patch (the 3x3 conv needs t1 one pixel around the block; pixels outside the image are ZERO,
//      the 3x3 conv's padding): every wave loads the A fragments of its three 32-pixel tiles and the W1 fragments straight
//      from global memory / L2 four k-steps ahead (no staging, no barriers -- the first version staged x through LDS with
//      two barriers per 32 channels and spent its time in exposed load latency: 138 us per identity block at 1080p);
//      t1 lands in LDS as split fp16 (352 rows x 64 channels, 144-byte rows);
//   B  conv2 as in conv_patch_f16x3.hip: the nine taps read shifted windows of the t1 patch, weights staged per 16-channel
//      stage; t2 (256 pixels x 64 channels) replaces the t1 patch in LDS;
//   C  conv3 in two halves of 128 output channels: A fragments from the t2 tile, W3 fragments straight from L2; for the
//      first block of the stage the projection Wd x is the same GEMM with 64 more K (weights concatenated at load time,
//      its A fragments -- the block's own 256 pixels of x -- loaded from global memory in fragment form); identity blocks
//      add x in the epilogue.
// x is read once (plus a 33 % halo), y written once: 268 MB instead of 534 MB per identity block at 1080p, one launch
// instead of three (four).  fp32 contract as everywhere: operands split into fp16 hi + lo, three MFMA passes, fp32
// accumulate; weights pre-split with a per-filter power-of-two scale (otvm_split_conv_weight_f16x3) and re-packed in MFMA
// B-fragment order (otvm_pack_wave_weight_f16x3: [n/32][32-channel chunk][k-step][hi|lo][64 lanes][8 halfs]).
#include "common.h"
#include <hip/hip_fp16.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

#ifdef OTVM_BNK_TIMING
// experiment build only (tools/bottleneck_bench.py): per-stage time of wave 0 of every workgroup, 100 MHz ticks
__device__ unsigned long long g_bnk_t[8];
#define BNK_STAMP(i)                                                       \
    if (threadIdx.x == 0) {                                                \
        const unsigned long long t_now = wall_clock64();                   \
        atomicAdd(&g_bnk_t[i], t_now - t_prev);                            \
        t_prev = t_now;                                                    \
    }
#else
#define BNK_STAMP(i)
#endif

namespace {

struct BnkArgs {
    const float* x; float* y;
    const _Float16* w1f; const _Float16* w2f; const _Float16* w3f;      // fragment-major split weights
    const float* s1; const float* s2; const float* s3;                  // per-filter scales (undo the power-of-two scaling)
    const float* b1; const float* b2; const float* b3;                  // folded BatchNorm biases (b3 includes bd with a projection)
    int H, W, x_ld, y_ld, tiles_x, tiles_y; OtvmTileWalk walk;
    int64_t x_bs, y_bs;                                                 // batch: image blockIdx.y
};

constexpr int TH = 8, TW = 32, PWD = TW + 2, PHT = TH + 2, NPIX = PHT * PWD;   // 10 x 34 = 340 patch pixels
constexpr int LDT = 72;              // halfs per t1 / t2 row (64 channels + 8 pad) = 144 bytes
constexpr int T1_HALFS = 352 * LDT;                                      // per hi / lo plane
constexpr int STG_OFF = 2 * T1_HALFS;                                    // halfs: staging region behind the t1 planes
constexpr int STG_HALFS = 9 * 2 * 2 * 512;                               // stage-B weights of one 16-channel stage (36 KiB)
constexpr int LDS_BYTES = (STG_OFF + STG_HALFS) * 2;

__device__ __forceinline__ void split4b(const f32x4 v, f16x4& hi, f16x4& lo) {
    typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
    const fp16x2 p01 = __builtin_amdgcn_cvt_pkrtz(v.x, v.y);
    const fp16x2 p23 = __builtin_amdgcn_cvt_pkrtz(v.z, v.w);
    const f16x2 h01 = __builtin_bit_cast(f16x2, p01);
    const f16x2 h23 = __builtin_bit_cast(f16x2, p23);
    hi = f16x4{h01.x, h01.y, h23.x, h23.y};
    lo = f16x4{(_Float16)(v.x - (float)h01.x), (_Float16)(v.y - (float)h01.y), (_Float16)(v.z - (float)h23.x),
               (_Float16)(v.w - (float)h23.y)};
}

__device__ __forceinline__ void split1b(float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

#define MFMA3(ACC, AH, AL, BH, BL)                                              \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(AL, BH, ACC, 0, 0, 0);         \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, BL, ACC, 0, 0, 0);         \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, BH, ACC, 0, 0, 0)

template <bool FULL>
__device__ __forceinline__ void store_patch(const float* patch, int prow, int pc, f32x4 sc4, f32x4 bi4, float* yp, int y, int tx0,
                                            int H, int W, int y_ld, int n4) {
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const int xi = r4 * 8 + prow, x = tx0 + xi;
        f32x4 v = *reinterpret_cast<const f32x4*>(&patch[xi * 36 + pc]);
        v = v * sc4 + bi4;
        v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
        v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
        if (FULL || (y < H && x < W)) *reinterpret_cast<f32x4*>(yp + ((int64_t)y * W + x) * y_ld + n4) = v;
    }
}

// CIN = 256: identity block;  CIN = 64: first block of the stage, projection folded into the last GEMM (K = 128)
template <int CIN>
__global__ __launch_bounds__(256) void stm_bottleneck_f16x3_kernel(const BnkArgs pa) {
    constexpr bool PROJ = CIN == 64;
    BnkArgs p = pa;
    p.x += blockIdx.y * p.x_bs;
    p.y += blockIdx.y * p.y_bs;
    extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
    _Float16* T1h = smem;
    _Float16* T1l = smem + T1_HALFS;
    _Float16* STG = smem + STG_OFF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 31, fh = lane >> 5;
    int tile_n_, tile_x, tile_y;
    otvm_tile_decode(p.walk, blockIdx.x, gridDim.x, 1, p.tiles_x, p.tiles_y, tile_n_, tile_x, tile_y);
    const int ty0 = tile_y * TH, tx0 = tile_x * TW;
#ifdef OTVM_BNK_TIMING
    unsigned long long t_prev = wall_clock64();
#endif

    // ------------------------------------------------------------------ stage A: t1 = relu(W1 x + b1) on the halo'd patch
    // No LDS staging and no barriers: wave w owns the 32-row m-tiles w, w + 4, w + 8 of the patch (11 tiles; the twelfth is
    // a dummy) and loads its A fragments -- one patch pixel per lane -- straight from global memory, two 32-channel chunks
    // ahead (4 waves per CU cannot hide HBM latency by switching, so the loads are issued early: a 240-register prefetch
    // ring).  A lane takes 64 CONTIGUOUS bytes of its pixel per chunk (channels 16 fh .. 16 fh + 15; the two k-steps of the
    // chunk use the first / second 8 of them), so that every 128-byte line is consumed by back-to-back instructions; the K
    // order inside a chunk is then (k-step, lane half) <-> channel 16 * half + 8 * k-step + j, and the W1 fragment of k-step
    // s for lane half fh is the standard layout's block (k-step fh, lane half s).
    {
        constexpr int NCH = CIN / 32, DEPTH = 3;
        int64_t aoff[3];
        bool aok[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int pr = (wave + 4 * t) * 32 + frow;
            const int py = pr / PWD, px = pr - py * PWD;
            const int iy = ty0 - 1 + py, ix = tx0 - 1 + px;
            aok[t] = (pr < NPIX) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
            aoff[t] = aok[t] ? ((int64_t)iy * p.W + ix) * p.x_ld + 16 * fh : 0;
        }
        f32x16 acc[3][2];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[t][b][e] = 0.f;
        f32x4 ra[DEPTH][3][4];
        f16x8 wb[DEPTH][2][2][2];
        const int wlane = (fh * 2) * 512 + frow * 8;                     // halfs: block (k-step fh), + (32 s) * 8 per k-step s
        auto issue = [&](int c, int slot) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) ra[slot][t][q] = *reinterpret_cast<const f32x4*>(p.x + aoff[t] + c * 32 + q * 4);
#pragma unroll
            for (int sk = 0; sk < 2; ++sk)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int hl = 0; hl < 2; ++hl)
                        wb[slot][sk][b][hl] = *reinterpret_cast<const f16x8*>(p.w1f + (((int64_t)b * NCH + c) * 4 + hl) * 512 + wlane + sk * 256);
        };
#pragma unroll
        for (int d = 0; d < DEPTH - 1; ++d)
            if (d < NCH) issue(d, d);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (c + DEPTH - 1 < NCH) issue(c + DEPTH - 1, (c + DEPTH - 1) % DEPTH);
            __builtin_amdgcn_sched_barrier(0);                           // keep the loads HERE (the scheduler would sink them to their use)
            const int slot = c % DEPTH;
#pragma unroll
            for (int sk = 0; sk < 2; ++sk)
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    f16x4 h0, l0, h1, l1;
                    split4b(aok[t] ? ra[slot][t][2 * sk] : z, h0, l0);
                    split4b(aok[t] ? ra[slot][t][2 * sk + 1] : z, h1, l1);
                    const f16x8 ah = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                    const f16x8 al = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
#pragma unroll
                    for (int b = 0; b < 2; ++b) { MFMA3(acc[t][b], ah, al, wb[slot][sk][b][0], wb[slot][sk][b][1]); }
                }
        }
        // t1 -> LDS (split); pixels outside the image are ZERO (the 3x3 conv's padding) -- only blocks on the image border
        // have any, interior blocks skip the test (rows 340..351 of the patch are never read)
        const bool interior = ty0 >= 1 && tx0 >= 1 && ty0 + TH + 1 <= p.H && tx0 + TW + 1 <= p.W;   // workgroup-uniform
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            if (wave + 4 * t >= 11) continue;                            // (wave-uniform: the twelfth tile has no rows in LDS)
            bool ok[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                ok[e] = true;
                if (!interior) {
                    const int pr = (wave + 4 * t) * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                    const int py = pr / PWD, px = pr - py * PWD;
                    const int iy = ty0 - 1 + py, ix = tx0 - 1 + px;
                    ok[e] = (pr < NPIX) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
                }
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int n = b * 32 + frow;
                const float sc = p.s1[n], bi = p.b1[n];
                const int o = ((wave + 4 * t) * 32 + 4 * fh) * LDT + n;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float v = acc[t][b][e] * sc + bi;
                    v = ok[e] ? (v > 0.f ? v : 0.f) : 0.f;
                    _Float16 h, l;
                    split1b(v, h, l);
                    T1h[o + ((e & 3) + 8 * (e >> 2)) * LDT] = h;
                    T1l[o + ((e & 3) + 8 * (e >> 2)) * LDT] = l;
                }
            }
        }
    }

    BNK_STAMP(0);                                                        // stage A
    // identity blocks: the residual (this lane's 128 outputs of the first half of the channels, accumulator layout) is requested
    // NOW -- it arrives while stage B computes; the second half's is requested before the first half's GEMM
    f32x16 res[PROJ ? 1 : 2][PROJ ? 1 : 4];
    const bool full = ty0 + TH <= p.H && tx0 + TW <= p.W;                 // workgroup-uniform: no partial rows / columns
    auto load_res = [&](int half) __attribute__((always_inline)) {
        if (PROJ) return;
        if (full) {                                                      // scalar base per pixel, one 32-bit lane offset per row
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const unsigned voff = (unsigned)((ty0 + wave * 2 + a) * p.W + tx0 + 4 * fh) * (unsigned)p.x_ld + half * 128 + frow;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float* base = p.x + (size_t)((e & 3) + 8 * (e >> 2)) * p.x_ld;
#pragma unroll
                    for (int b = 0; b < 4; ++b) res[PROJ ? 0 : a][PROJ ? 0 : b][e] = base[voff + b * 32];
                }
            }
            return;
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int n = (half * 4 + b) * 32 + frow;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int y = ty0 + wave * 2 + a;
#pragma unroll
                for (int e = 0; e < 16; ++e) {                            // branch-free: 128 loads in flight, not 128 round trips
                    const int x = tx0 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                    const bool ok = (y < p.H) & (x < p.W);
                    res[PROJ ? 0 : a][PROJ ? 0 : b][e] = p.x[ok ? ((int64_t)y * p.W + x) * p.x_ld + n : 0];
                }
            }
        }
    };
    load_res(0);
    __builtin_amdgcn_sched_barrier(0);

    // ------------------------------------------------------------------ stage B: t2 = relu(W2 * t1 + b2), 3x3 over the patch
    f32x16 acc2[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[a][b][e] = 0.f;
    {
        // weights of one 16-channel stage: 9 taps x 2 n-tiles x (hi, lo) x 1 KiB = 36 x 1 KiB pieces; 9 x 16 bytes per thread.
        // fragment (n-tile b, chunk = cb32 * 9 + tap, k-step = cb & 1, hl) of the wave layout, nchunks = 18
        f16x8 rw[9];
        auto prefetch = [&](int cb) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int i = tid + k * 256;                             // piece index 0 .. 2303: [(tap * 2 + b) * 2 + hl][lane]
                const int l = i & 63, blk = i >> 6;
                const int hl = blk & 1, b = (blk >> 1) & 1, tap = blk >> 2;
                const int64_t src = ((((int64_t)b * 18 + (cb >> 1) * 9 + tap) * 2 + (cb & 1)) * 2 + hl) * 512 + l * 8;
                rw[k] = *reinterpret_cast<const f16x8*>(p.w2f + src);
            }
        };
        prefetch(0);
        for (int cb = 0; cb < 4; ++cb) {
            __syncthreads();                                             // stage A's t1 writes / the previous stage's reads are done
#pragma unroll
            for (int k = 0; k < 9; ++k) *reinterpret_cast<f16x8*>(&STG[(tid + k * 256) * 8]) = rw[k];
            __syncthreads();
            if (cb + 1 < 4) prefetch(cb + 1);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - ky * 3;
                f16x8 ah[2], al[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int o = ((wave * 2 + a + ky) * PWD + kx + frow) * LDT + cb * 16 + 8 * fh;
                    ah[a] = *reinterpret_cast<const f16x8*>(&T1h[o]);
                    al[a] = *reinterpret_cast<const f16x8*>(&T1l[o]);
                }
                f16x8 bh[2], bl[2];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    bh[b] = *reinterpret_cast<const f16x8*>(&STG[((tap * 2 + b) * 2) * 512 + lane * 8]);
                    bl[b] = *reinterpret_cast<const f16x8*>(&STG[((tap * 2 + b) * 2 + 1) * 512 + lane * 8]);
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc2[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[a], bh[b], acc2[a][b], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc2[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bl[b], acc2[a][b], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc2[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bh[b], acc2[a][b], 0, 0, 0);
            }
        }
    }
    BNK_STAMP(1);                                                        // stage B
    __syncthreads();                                                     // every wave is done with the t1 patch
    // t2 tile -> LDS (split), over the t1 planes: pixel (row 2 wave + a, column x) at row (2 wave + a) * 32 + x
    _Float16* T2h = smem;
    _Float16* T2l = smem + 256 * LDT;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int n = b * 32 + frow;
            const float sc = p.s2[n], bi = p.b2[n];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int pr = (wave * 2 + a) * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                float v = acc2[a][b][e] * sc + bi;
                v = v > 0.f ? v : 0.f;
                _Float16 h, l;
                split1b(v, h, l);
                T2h[pr * LDT + n] = h;
                T2l[pr * LDT + n] = l;
            }
        }
    __syncthreads();

    BNK_STAMP(2);                                                        // t2 -> LDS
    // ------------------------------------------------------------------ stage C: y = relu(W3 t2 [+ Wd x] + b3 [+ x])
    constexpr int KS3 = PROJ ? 8 : 4;                                    // 16-deep k-steps: t2 (4) [+ x (4)]
    constexpr int NCH3 = KS3 / 2;                                        // 32-channel chunks of the fragment layout
    float* patch = reinterpret_cast<float*>(smem + 2 * 256 * LDT) + wave * (32 * 36);   // behind the t2 planes
    // projection: A fragments of its K = 64 -- this lane's pixel of x, 8 consecutive channels per k-step -- loaded and split once
    f16x8 xh[PROJ ? 2 : 1][PROJ ? 4 : 1], xl[PROJ ? 2 : 1][PROJ ? 4 : 1];
    if (PROJ) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int y = ty0 + wave * 2 + a, x = tx0 + frow;
            const bool ok = y < p.H && x < p.W;
            const int64_t off = ok ? ((int64_t)y * p.W + x) * p.x_ld + 8 * fh : 0;
            f32x4 v[4][2];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                v[ks][0] = *reinterpret_cast<const f32x4*>(p.x + off + ks * 16);
                v[ks][1] = *reinterpret_cast<const f32x4*>(p.x + off + ks * 16 + 4);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                f16x4 h0, l0, h1, l1;
                split4b(ok ? v[ks][0] : z, h0, l0);
                split4b(ok ? v[ks][1] : z, h1, l1);
                xh[a][ks] = f16x8{h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                xl[a][ks] = f16x8{l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
            }
        }
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        // identity blocks start the accumulators at x / scale (the scale is a power of two: exact) -- the residual costs no
        // epilogue round trips.  (Pixels outside the image carry garbage here; they are never stored.)
        f32x16 acc[2][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int n = (half * 4 + b) * 32 + frow;
            const float inv = PROJ ? 0.f : 1.0f / p.s3[n];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][b][e] = PROJ ? 0.f : res[PROJ ? 0 : a][PROJ ? 0 : b][e] * inv;
        }
        if (half == 0) {
            load_res(1);
            __builtin_amdgcn_sched_barrier(0);
        }
        f16x8 wh[2][4], wl[2][4];
        auto loadw = [&](int ks, int slot) __attribute__((always_inline)) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int64_t src = ((((int64_t)(half * 4 + b) * NCH3 + (ks >> 1)) * 2 + (ks & 1)) * 2) * 512 + lane * 8;
                wh[slot][b] = *reinterpret_cast<const f16x8*>(p.w3f + src);
                wl[slot][b] = *reinterpret_cast<const f16x8*>(p.w3f + src + 512);
            }
        };
        loadw(0, 0);
#pragma unroll
        for (int ks = 0; ks < KS3; ++ks) {
            if (ks + 1 < KS3) loadw(ks + 1, (ks + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            const int slot = ks & 1;
            f16x8 ah[2], al[2];
            if (ks < 4) {
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int o = ((wave * 2 + a) * 32 + frow) * LDT + ks * 16 + 8 * fh;
                    ah[a] = *reinterpret_cast<const f16x8*>(&T2h[o]);
                    al[a] = *reinterpret_cast<const f16x8*>(&T2l[o]);
                }
            } else {
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    ah[a] = xh[PROJ ? a : 0][PROJ ? ks - 4 : 0];
                    al[a] = xl[PROJ ? a : 0][PROJ ? ks - 4 : 0];
                }
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[a], wh[slot][b], acc[a][b], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], wl[slot][b], acc[a][b], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], wh[slot][b], acc[a][b], 0, 0, 0);
        }
        BNK_STAMP(3 + 2 * half);                                         // GEMM of this half
        // epilogue of this half: accumulator tile -> wave-private LDS patch -> 16-byte row-major stores (scale, bias, ReLU)
        const int col = lane & 31, rbase = fh * 4;
        const int prow = lane >> 3, pc = (lane & 7) * 4;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int n4 = (half * 4 + b) * 32 + pc;
            const f32x4 sc4 = *reinterpret_cast<const f32x4*>(p.s3 + n4), bi4 = *reinterpret_cast<const f32x4*>(p.b3 + n4);
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int y = ty0 + wave * 2 + a;
#pragma unroll
                for (int e = 0; e < 16; ++e) patch[((e & 3) + 8 * (e >> 2) + rbase) * 36 + col] = acc[a][b][e];
                // (one wave writes and reads its own patch: the compiler orders the LDS accesses, no barrier needed)
                // stores count in vmcnt on gfx9 and the compiler waits for vmcnt(0) in front of every store that sits in a
                // divergent branch -- i.e. for the previous store's acknowledgement: interior blocks take a branch-free path
                if (full) store_patch<true>(patch, prow, pc, sc4, bi4, p.y, y, tx0, p.H, p.W, p.y_ld, n4);
                else store_patch<false>(patch, prow, pc, sc4, bi4, p.y, y, tx0, p.H, p.W, p.y_ld, n4);
            }
        }
        BNK_STAMP(4 + 2 * half);                                         // epilogue of this half
    }
}

}  // namespace

#ifdef OTVM_BNK_TIMING
extern "C" int otvm_debug_bnk_times(unsigned long long* out8, int reset) {
    if (out8) hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_bnk_t), 64);
    if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_bnk_t), z, 64); }
    return 0;
}
#endif

int otvm_stm_bottleneck128_launch(const otvm_stm_bottleneck_params* q, int tile, void* stream);

extern "C" int otvm_stm_bottleneck_f16x3(const otvm_stm_bottleneck_params* q, void* stream) {
    OTVM_REQUIRE(q && q->x && q->y && q->w1f && q->w2f && q->w3f && q->s1 && q->s2 && q->s3 && q->b1 && q->b2 && q->b3,
                 "otvm_stm_bottleneck_f16x3: null pointer");
    if (q->Cin == 512) return otvm_stm_bottleneck128_launch(q, q->tile, stream);     // planes = 128 (bottleneck128_f16x3.hip)
    OTVM_REQUIRE(q->Cin == 64 || q->Cin == 256, "otvm_stm_bottleneck_f16x3: Cin must be 64 (projection block), 256 (identity block, planes 64) "
                 "or 512 (identity block, planes 128), got %d", q->Cin);
    OTVM_REQUIRE(q->x_ld % 4 == 0 && q->y_ld % 4 == 0 && ((uintptr_t)q->x & 15) == 0 && ((uintptr_t)q->y & 15) == 0 &&
                 q->x_ld >= q->Cin && q->y_ld >= 256, "otvm_stm_bottleneck_f16x3: views must be 16-byte aligned");
    OTVM_REQUIRE((int64_t)q->H * q->W * q->x_ld < (1ll << 31), "otvm_stm_bottleneck_f16x3: input view too large for 32-bit offsets");
    BnkArgs a;
    a.x = q->x; a.y = q->y; a.w1f = (const _Float16*)q->w1f; a.w2f = (const _Float16*)q->w2f; a.w3f = (const _Float16*)q->w3f;
    a.s1 = q->s1; a.s2 = q->s2; a.s3 = q->s3; a.b1 = q->b1; a.b2 = q->b2; a.b3 = q->b3;
    a.H = q->H; a.W = q->W; a.x_ld = q->x_ld; a.y_ld = q->y_ld;
    a.tiles_x = otvm_ceil_div(q->W, TW); a.tiles_y = otvm_ceil_div(q->H, TH);
    a.walk = otvm_tile_walk_of(8);
    const int batch = q->batch > 1 ? q->batch : 1;
    a.x_bs = batch > 1 ? q->x_bs : 0; a.y_bs = batch > 1 ? q->y_bs : 0;
    {
        static std::atomic<bool> done256[OTVM_MAX_DEVICES], done64[OTVM_MAX_DEVICES];
        hipError_t e = otvm_reserve_lds_once(done256, stm_bottleneck_f16x3_kernel<256>, LDS_BYTES);
        if (e == hipSuccess) e = otvm_reserve_lds_once(done64, stm_bottleneck_f16x3_kernel<64>, LDS_BYTES);
        if (e != hipSuccess) {
            otvm_set_error("otvm_stm_bottleneck_f16x3: cannot reserve %d bytes of LDS: %s", LDS_BYTES, hipGetErrorString(e));
            return 2;
        }
    }
    const dim3 grid(a.tiles_x * a.tiles_y, batch), block(256);
    if (q->Cin == 256) hipLaunchKernelGGL(stm_bottleneck_f16x3_kernel<256>, grid, block, LDS_BYTES, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(stm_bottleneck_f16x3_kernel<64>, grid, block, LDS_BYTES, (hipStream_t)stream, a);
    OTVM_CHECK_LAUNCH("otvm_stm_bottleneck_f16x3");
    return 0;
}
