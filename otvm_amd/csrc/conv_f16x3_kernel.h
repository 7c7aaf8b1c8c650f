// The implicit-GEMM f16x3 kernel template and its launcher (internal header of conv_f16x3.hip and conv_f16x3_glds.hip: the
// register-staged tiles are instantiated in the first translation unit, the LDS-DMA ("G") tiles in the second, so the two
// compile side by side).
#pragma once
// (the LDS-DMA instructions are issued from inline asm that writes M0, a register the compiler reserves: it re-materialises M0 in
//  front of every use of its own, so the clobber is harmless, but clang warns about any reserved register in a clobber list)
#pragma clang diagnostic ignored "-Winline-asm"
#include "common.h"
#include <type_traits>
#include <hip/hip_fp16.h>
#include <stdlib.h>

#ifndef OTVM_PF_DEPTH
#define OTVM_PF_DEPTH 3
#endif
// ring depth of the single-stage tiles (see PFS below): measured neutral, default = 1 (no ring)
#ifndef OTVM_PFS_SMALL
#define OTVM_PFS_SMALL 1
#endif
#ifndef OTVM_PFS_MID
#define OTVM_PFS_MID 1
#endif
#ifndef OTVM_PFS_LARGE
#define OTVM_PFS_LARGE 1
#endif
// timing probes for tools/build_variant.sh (the results are WRONG with any of them set): which part of the K loop costs what
#ifndef OTVM_ABL_NOMFMA
#define OTVM_ABL_NOMFMA 0      // skip the MFMAs
#endif
#ifndef OTVM_ABL_NOLOAD
#define OTVM_ABL_NOLOAD 0      // no global loads after the first chunk (its registers are re-staged)
#endif
#ifndef OTVM_ABL_NOLDSRD
#define OTVM_ABL_NOLDSRD 0     // fragments are read from LDS only for the first k-step
#endif
#ifndef OTVM_ABL_NOSTAGE
#define OTVM_ABL_NOSTAGE 0     // no split + LDS writes after the first chunk
#endif
#ifndef OTVM_PF_DB
#define OTVM_PF_DB 2           // register sets of the pipelined small tiles
#endif
#ifndef OTVM_BRANCHY_LOADS
#define OTVM_BRANCHY_LOADS 1
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

struct Conv3Args {
    const float* in; const _Float16* wh; const _Float16* wl; const float* wscale; const float* bias;
    const float* residual; float* out; double* gn_stats;
    int H, W, Cin, in_ld, K_pad, res_ld, Ho, Wo, Cout, out_ld;
    int kh, kw, stride, pad, dil, in_relu, act;
    int M, taps, nchunks, tiles_m, tiles_n;
    // split-K (layers too small to fill the chip): gridDim.y workgroups share an output tile, each walks a contiguous
    // range of the K chunks and writes its un-biased partial tile to out + blockIdx.y * split_stride (the host points
    // `out` at the workspace and clears bias / residual / act / gn_stats); splitk_finish_kernel adds them up in order
    int64_t split_stride;
    // batch: image blockIdx.z of every tensor lives *_bs elements behind image 0 (split-K: each image owns gridDim.y
    // partial tiles of the workspace, out_bs = gridDim.y * split_stride)
    int64_t in_bs, out_bs, res_bs; int gn_bs;
    // wave kernel (conv_wave_f16x3_kernel): the split weights in MFMA B-fragment order, see otvm_pack_wave_weight_f16x3
    const _Float16* wf;
    // fused normalisation of the input (NORM_IN kernels: the producer's GroupNorm apply folded into the A staging):
    // x' = x * in_scale[c] + in_shift[c], then x' > 0 ? x' : x' * in_slope (1 = none, 0 = ReLU, 0.01 = LeakyReLU), zero in the
    // conv's padding; tables from otvm_gn_table, image b's table norm_bs floats behind image 0's
    const float* in_scale; const float* in_shift; float in_slope; int norm_bs;
    OtvmGnTail tail;             // ABI 16: the output's GroupNorm table, written by the last workgroup (common.h)
    // ABI 17: per-image w_scale / bias (a predicted normalisation of the output, csrc/gram.hip) and a per-channel scale on the
    // residual (the GroupNorm scale of a raw identity-path tensor; its shift is part of bias)
    int ws_bs; const float* res_scale; int rs_bs;
    unsigned in_bytes;           // GLDS tiles: size of one image's input view (the buffer resource's range; < 2^31, launch3)
    int batch;                   // images of the launch (gridDim.z)
    int npass;                   // 3 = f16x3, 1 = precision "f16" (one MFMA pass on fp16-rounded operands; conv_f16x3_p1.hip)
};

// conv_f16x3_glds.hip: the LDS-DMA form of implicit-GEMM tile `base` (the enum of conv_f16x3.hip), K split over S workgroups
int otvm_launch_glds_tile(int base, Conv3Args& a, hipStream_t s, int S);
// conv_f16x3_m16.hip: the LDS-DMA tile `base` on v_mfma_f32_16x16x32_f16 (M16)
int otvm_launch_m16_tile(int base, Conv3Args& a, hipStream_t s, int S);
// conv_f16x3_p1.hip: the single-pass ("f16") form of tile `tile` (register-staged t or LDS-DMA 32 + t)
int otvm_launch_tile_p1(int tile, Conv3Args& a, hipStream_t s, int S);

namespace {

constexpr int BK = 32;
constexpr int LDH = 40;          // halfs per LDS row (32 + 8 pad) = 80 bytes

inline bool f16x3_fast_layout(int taps, int I_pad) { return I_pad % 32 == 0 && taps <= 32; }

__device__ __forceinline__ void split4(const f32x4 v, f16x4& hi, f16x4& lo) {
    // hi: round-toward-zero pack (any rounding works, lo is computed exactly against it)
    typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
    const fp16x2 p01 = __builtin_amdgcn_cvt_pkrtz(v.x, v.y);
    const fp16x2 p23 = __builtin_amdgcn_cvt_pkrtz(v.z, v.w);
    const f16x2 h01 = __builtin_bit_cast(f16x2, p01);
    const f16x2 h23 = __builtin_bit_cast(f16x2, p23);
    hi = f16x4{h01.x, h01.y, h23.x, h23.y};
    lo = f16x4{(_Float16)(v.x - (float)h01.x), (_Float16)(v.y - (float)h01.y), (_Float16)(v.z - (float)h23.x),
               (_Float16)(v.w - (float)h23.y)};
}

// FAST: Cin % 32 == 0 and <= 32 taps -> a K chunk never straddles a tap, so the tap walk is wave-uniform
// (scalar registers) and the per-row work per chunk shrinks to one add and one mask test.
//
// GLDS (round 5, the two-stage 8-wave tiles): the B (weight) stage never passes through registers.  The split weights exist in
// MFMA B-fragment order already (otvm_pack_wave_weight_f16x3: [n/32][chunk][k-step][hi|lo][lane][8 halfs] = 1-KiB blocks, four
// consecutive blocks per 32-filter tile and chunk), which is byte for byte what an LDS-DMA instruction writes (M0 + 16 * lane):
// every wave copies its share of the chunk's BN / 8 blocks with `global_load_lds_dwordx4`, and the fragment reads become
// lane-linear ds_read_b128 (no padding, no bank conflicts).  The DMA is issued from INLINE ASM on purpose: the compiler orders
// every LDS read behind a pending LDS-DMA it knows about (SIInsertWaitcnts has no alias information for it: `s_waitcnt vmcnt(0)`
// in front of the first fragment read -- the ISA of the round-4 patch kernel shows exactly that), so a builtin DMA is waited for
// right after it is issued.  Hidden from the compiler, the copy of chunk c + 1 flies under the second k-step of chunk c and is
// waited for by a hand-counted `s_waitcnt vmcnt(N)` in front of the chunk's one barrier, N = the activation loads issued after
// it, which stay in flight across the barrier.  For N to be a constant the activation loads are buffer loads without a branch:
// a padding lane carries an out-of-range offset and the hardware returns zeros (no exec-masked load, no select afterwards).
//
// NPASS (round 5): 3 = the f16x3 operand split (fp32-class results).  1 = precision "f16", a LABELLED reduced-precision mode
// (BASELINE configs[2] as written: "bf16 MFMA conv"): operands rounded to fp16 once (round to nearest), ONE MFMA pass, fp32
// accumulate; the lo halves are neither computed, staged nor read.  Not the default and not parity-graded.
template <int BM, int BN, int WM, int WN, bool FAST, bool RELU_IN, bool DB = false, bool NORM_IN = false, bool GLDS = false, int NPASS = 3,
          bool M16T = false>
__global__ __launch_bounds__(WM* WN * 64)
__attribute__((amdgpu_waves_per_eu((BM * BN == 32768 && WM * WN == 4) ? 2 : 1, (BM * BN == 32768 && WM * WN == 4) ? 2 : 10)))
void conv_igemm_f16x3_kernel(const Conv3Args pa) {
    Conv3Args p = pa;
    {   // image of this workgroup (scalar pointer arithmetic; a batch-1 launch has gridDim.z == 1)
        const int zb = blockIdx.z;
        p.in += zb * p.in_bs;
        p.out += zb * p.out_bs;
        if (p.residual) p.residual += zb * p.res_bs;
        if (p.gn_stats) p.gn_stats += zb * p.gn_bs;
        if (NORM_IN) { p.in_scale += zb * p.norm_bs; p.in_shift += zb * p.norm_bs; }
        p.wscale += zb * p.ws_bs;
        if (p.bias) p.bias += zb * p.ws_bs;
        if (p.res_scale) p.res_scale += zb * p.rs_bs;
    }
    static_assert(!NORM_IN || (FAST && !RELU_IN), "the fused input normalisation exists on the whole-chunk path only");
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_ROWS = NT / 8, A_LD = BM / A_ROWS;        // 8 float4 per 32-wide row
    constexpr int B_ROWS = NT / 4, B_LD = (BN + B_ROWS - 1) / B_ROWS;   // 4 x 16 bytes per 32-half row
    static_assert(A_LD >= 1 && B_LD >= 1 && TM >= 1 && TN >= 1, "bad tile");
    constexpr int STAGE_A = 2 * BM * LDH;                      // halfs of A_hi, A_lo
    constexpr int STAGE = GLDS ? STAGE_A + BN * 64 : 2 * (BM + BN) * LDH;   // halfs per LDS stage (A_hi, A_lo, B_hi, B_lo | GLDS: B fragment blocks)
    // Two LDS stages + one barrier per chunk on the 256-row tiles (+5..14 % on the layers that use them; the smaller
    // tiles lose more from the halved occupancy than they gain: 32.7 vs 33.1 frames/s with DBUF everywhere).
    // Round 3: DB = the same pipelined loop on a small tile, as a separate autotuner candidate ("64x64 D", ...).  Ablation of
    // the single-stage 64x64 tile (1024->256 1x1 at 68x120, profiles/r03_small_tile_ablation.txt): 31 us in all; MFMAs +
    // barriers alone 13, the global loads add 8 (16 KB per chunk and workgroup through a 64 B/clk L1), split + LDS writes 7,
    // fragment reads 1 -- the parts ADD because a workgroup's phases are serialised by its two barriers per chunk and the
    // two workgroups of a CU run in step.  Converting chunk c+1 between the two k-steps of chunk c lets one wave's VALU /
    // LDS work issue under its own MFMAs.
    constexpr bool DBUF = DB || (BM == 256 && BN >= 128 && WM * WN == 8) || (BM == 256 && BN == 256);   // (the 4-wave 256x128 / 128x256 tiles: one stage, two workgroups per CU)
    static_assert(!GLDS || FAST, "LDS-DMA weight stages: whole-chunk layers");
    // M16 (round 5): the matrix cores of this part do ~10 % more work per joule with v_mfma_f32_16x16x32_f16 than with
    // v_mfma_f32_32x32x16_f16 (tools/probes/mfma_variants_probe.hip: 1800 vs 1615-1665 TFLOP/s f16 sustained under the power
    // limit, random operands; a quarter of the accumulator read-modify-write per FLOP), and the frame's launches are bound by that
    // limit (profiles/r05_zero_operand_power_probe.txt).  The M16 form of an LDS-DMA tile (tile 64 + t, conv_f16x3_m16.hip; a
    // candidate of its own for the plan-time tuner: it wins on the maps that keep the chip at its power limit, -2 ... -10 % per
    // launch, and loses on small maps, where its extra fragment reads are not free) multiplies a 32-deep chunk with 16x16x32
    // instructions: a 32x32 accumulator tile is four 16x16 sub-tiles (quad q = 2 si + sj of its sixteen registers: rows
    // 16 si + 4 (lane >> 4) + r, columns 16 sj + (lane & 15)); the two "halves" of a chunk that the K loop interleaves with the
    // staging are the column halves sj = 0 / 1 instead of the two k-steps.  Neither stage layout changes: an A fragment is row
    // 16 t + (lane & 15), k-octet lane >> 4 of the [row][32 + 8] stage; a B fragment is ONE ds_read_b128 across the k-step 0 and
    // k-step 1 blocks of the fragment-major weights (lanes 0-31 read the first block, 32-63 the second, slot
    // (lane & 15) + 16 sj + 32 ((lane >> 4) & 1): filter 16 sj + (lane & 15), k-octet lane >> 4 -- conflict-free).
    constexpr bool M16 = M16T;
    static_assert(!M16T || (GLDS && NPASS == 3), "the 16x16x32 form exists for the f16x3 LDS-DMA tiles");
    // LDS: two-stage tiles [A0 | B0 | A1 | B1]; single-stage GLDS tiles [A | B0 | B1] -- their weight copy of chunk c + 1 runs
    // while chunk c is multiplied, so the B stage alone is doubled (BN * 128 bytes more)
    constexpr int BST = BN * 64;                               // halfs of one GLDS weight stage
    constexpr int SMEM_HALFS = DBUF ? 2 * STAGE : (GLDS ? STAGE + BST : STAGE);
    __shared__ __attribute__((aligned(16))) _Float16 smem[SMEM_HALFS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tile_n = wgid % p.tiles_n, tile_m = wgid / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int arow = tid >> 3, ak = (tid & 7) * 4;
    const int brow = tid >> 2, bk = (tid & 3) * 8;
    int iy0[A_LD], ix0[A_LD];
    int rowoff[A_LD];            // FAST: element offset of (iy0, ix0, ak) from p.in
    unsigned tapmask[A_LD];      // FAST: bit t set <=> tap t reads inside the image for this row
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        const int m = m0 + arow + A_ROWS * i;
        if (m < p.M) {
            const int oy = m / p.Wo, ox = m - oy * p.Wo;
            iy0[i] = oy * p.stride - p.pad;
            ix0[i] = ox * p.stride - p.pad;
        } else {
            iy0[i] = -(1 << 28);
            ix0[i] = -(1 << 28);
        }
        if (FAST) {
            rowoff[i] = (iy0[i] * p.W + ix0[i]) * p.in_ld + ak;
            unsigned mk = 0;
            for (int t = 0; t < p.taps; ++t) {
                const int ky = t / p.kw, kx = t - ky * p.kw;
                const int iy = iy0[i] + ky * p.dil, ix = ix0[i] + kx * p.dil;
                if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) mk |= 1u << t;
            }
            tapmask[i] = mk;
        }
    }
    // K chunks of this workgroup: all of them, or the blockIdx.y-th share (split-K)
    const int c_begin = (int)(((int64_t)blockIdx.y * p.nchunks) / gridDim.y);
    const int c_end = (int)(((int64_t)(blockIdx.y + 1) * p.nchunks) / gridDim.y);
    float* const outp = p.out + (int64_t)blockIdx.y * p.split_stride;
    // wave-uniform tap walk (FAST), positioned on the first chunk
    int u_cb = c_begin / p.taps, u_tap = c_begin - u_cb * p.taps;
    int u_ky = u_tap / p.kw, u_kx = u_tap - u_ky * p.kw;
    const int64_t woff0 = (int64_t)(n0 + brow) * p.K_pad + bk;
    const int64_t wstep = (int64_t)B_ROWS * p.K_pad;
    // GLDS: this wave's NBL consecutive fragment blocks of every chunk (blocks g0 .. g0 + NBL - 1 of the stage's BN / 8; a
    // 32-filter tile owns four: [k-step][hi|lo]) and where they land; the activation tensor as a buffer resource
    constexpr int NBL = GLDS ? BN / 8 / (NT / 64) : 1;          // 1-KiB blocks per wave and chunk (8 waves: BN = 256: 4, 128: 2)
    static_assert(!GLDS || ((BN / 8) % (NT / 64) == 0 && (NBL == 1 || NBL == 2 || NBL == 4 || NBL == 8)), "bad LDS-DMA split");
    constexpr unsigned B_BUF_BYTES = (DBUF ? STAGE : BST) * 2;  // distance between the two weight stages
    const int g0 = wave * NBL;
    const _Float16* const dma_src = GLDS ? p.wf + ((int64_t)((n0 >> 5) + (g0 >> 2)) * p.nchunks) * 2048 + (g0 & 3) * 512 + lane * 8 : nullptr;
    const unsigned dma_dst = GLDS ? __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(smem + STAGE_A + g0 * 512)) : 0u;
    __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, GLDS ? p.in_bytes : 0, 0x00020000);
    // chunk c's blocks -> stage `buf`.  The immediate offset of an LDS-DMA instruction moves BOTH addresses, and the blocks are
    // consecutive on both sides.  (s_nop: one wait state between the SALU write of M0 and its use)
    auto dma_b = [&](int c, int buf) __attribute__((always_inline)) {
        if constexpr (GLDS && NPASS == 1) {
            // the hi blocks only (blocks 0 and 2 of a 32-filter tile's four): they keep their places in the stage
            const _Float16* src = dma_src + (int64_t)c * 2048;
            const unsigned dst = dma_dst + (unsigned)buf * B_BUF_BYTES;
            if constexpr (NBL >= 4) {
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:2048"
                             :: "s"(dst), "v"(src) : "m0", "memory");
                if constexpr (NBL == 8)
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:2048"
                                 :: "s"(dst + 4096u), "v"(src + (int64_t)p.nchunks * 2048) : "m0", "memory");
            } else if (NBL == 2 || (g0 & 1) == 0) {               // (NBL == 1: the odd waves own lo blocks -- wave-uniform)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(dst), "v"(src) : "m0", "memory");
            }
        } else if constexpr (GLDS) {
            const _Float16* src = dma_src + (int64_t)c * 2048;
            const unsigned dst = dma_dst + (unsigned)buf * B_BUF_BYTES;
            if constexpr (NBL == 8) {                                  // (immediate offsets end at 4095: two base addresses)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                             "global_load_lds_dwordx4 %1, off offset:2048\n\tglobal_load_lds_dwordx4 %1, off offset:3072"
                             :: "s"(dst), "v"(src) : "m0", "memory");
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                             "global_load_lds_dwordx4 %1, off offset:2048\n\tglobal_load_lds_dwordx4 %1, off offset:3072"
                             :: "s"(dst + 4096u), "v"(src + (int64_t)p.nchunks * 2048) : "m0", "memory");   // (the next 32-filter tile's four blocks)
            } else if constexpr (NBL == 1)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(dst), "v"(src) : "m0", "memory");
            else if constexpr (NBL == 4)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                             "global_load_lds_dwordx4 %1, off offset:2048\n\tglobal_load_lds_dwordx4 %1, off offset:3072"
                             :: "s"(dst), "v"(src) : "m0", "memory");   // ("memory": no load may be moved across -- the waits below count)
            else
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024"
                             :: "s"(dst), "v"(src) : "m0", "memory");
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    // register sets of chunks in flight (global -> registers -> LDS).  The 256x128 tile uses ~155 of its 256 VGPRs: three
    // sets = three chunks of loads under way, because with ONE workgroup per CU a chunk's MFMA time (0.7 us) is well
    // below the L2-miss latency and the K loop otherwise runs at one memory round trip per chunk
    // Round 3, measured and rejected: a register-set ring on the single-stage tiles too (the loads of chunk c + PF issued
    // while chunk c is computed; hypothesis: the small-map layers -- 25 us launches for 1.8 GFLOP at 480p -- run their K
    // loop at one exposed memory round trip per chunk).  Built branch-free so that the compiler waits with exact
    // vmcnt(4 (PF - 1)) counts (ISA checked: vmcnt 20..23 at depth 6), timed at depths 1 / 3 / 6 / 8 on 19 OS4..OS16 layer
    // shapes and on the whole frame (profiles/r03_ring_prefetch_ab.txt): no layer moved by more than the noise, the short-K
    // expanding 1x1 convs lost 20-60 % to the longer prologue, 480p 132.7 (depth 1) / 134.0 / 134.5 / 133.8 frames/s, 1080p
    // 39.4 / 38.9 / 38.6 / 38.8.  These tiles are not latency-bound: a 64x64 tile moves 48 KB through LDS per 32-deep chunk
    // for 6 MFMAs per wave (LDS time 2x the MFMA time), and below ~12 us a launch is its fixed cost (dispatch, prologue,
    // epilogue), whatever K is.  The code stays (OTVM_PFS_* > 1 switches it on).
    constexpr int SETREGS = 4 * A_LD + 8 * B_LD;
    constexpr int PFS = (BM * BN == 32768 && WM * WN == 4) ? 1 :                  // 4-wave 256x128 / 128x256: 128 VGPRs, no room
                        (SETREGS <= 16 ? OTVM_PFS_SMALL : SETREGS <= 24 ? OTVM_PFS_MID : SETREGS <= 32 ? OTVM_PFS_LARGE : 1);
    constexpr int PF = DB ? OTVM_PF_DB : (DBUF ? (BN == 128 ? OTVM_PF_DEPTH : 1) : PFS);
    constexpr bool BRANCHY = OTVM_BRANCHY_LOADS && (DBUF || PFS == 1);
    struct RegSet {
        f32x4 ra[A_LD];
        unsigned okmask;            // bit i: ra[i] holds image data (else padding -> zero)
        f16x8 rbh[GLDS ? 1 : B_LD], rbl[GLDS ? 1 : B_LD];      // (GLDS: unused, the weights go global -> LDS)
        f32x4 nsc, nsh;             // NORM_IN: scale / shift of this thread's four channels of the chunk
    };
    RegSet rs[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) rs[j].okmask = 0;
    // valid == false (ring tiles past the last chunk): the same loads are issued against harmless addresses -- element 0 of
    // the input, the last weight chunk -- so that EVERY path through the K loop issues the same number of loads per step
    auto load_chunk = [&](int c, RegSet& R, const bool valid = true) __attribute__((always_inline)) {
        f32x4 (&ra)[A_LD] = R.ra;
        unsigned& okmask = R.okmask;
        auto& rbh = R.rbh;
        auto& rbl = R.rbl;
        if (FAST) {
            const int delta = (u_ky * p.dil * p.W + u_kx * p.dil) * p.in_ld + (u_cb << 5);   // scalar
            const unsigned bit = 1u << u_tap;
            if (NORM_IN) {
                const int cn = (valid ? (u_cb << 5) : 0) + ak;
                R.nsc = *reinterpret_cast<const f32x4*>(p.in_scale + cn);
                R.nsh = *reinterpret_cast<const f32x4*>(p.in_shift + cn);
            }
#pragma unroll
            for (int i = 0; i < A_LD; ++i) {
                // UNCONDITIONAL load (padding lanes read element 0 and are zeroed afterwards): a branch around
                // the load would hide the number of outstanding loads from the compiler, which then drains
                // vmcnt(0) in the middle of the pipeline (guide 5, trap (c)).
                // The zeroing (and the optional ReLU) happen in store_chunk, NOT here: touching the loaded value
                // now would put the s_waitcnt in front of the MFMAs and serialise load latency with compute.
                const bool ok = valid && (tapmask[i] & bit) != 0;
                // The big double-buffered tiles skip the load of a padding lane (exec-masked load: the unconditional
                // form cost 40 % on the full-resolution layers).  The ring-prefetch tiles must NOT: a branch around a load
                // makes the number of loads in flight unknown to the compiler's waitcnt pass, which then drains vmcnt(0)
                // once per turn of the ring instead of waiting for the one set it stores (seen in the ISA).
                if constexpr (GLDS) {
                    // branch-free AND traffic-free for padding lanes: an offset beyond the resource's size returns zeros
                    // (pure arithmetic on purpose: a select between the two offsets comes back from the compiler as an if / else
                    //  around two loads, i.e. one or two load instructions per row depending on the wave's lanes)
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    const unsigned okb = (tapmask[i] >> u_tap) & 1u;
                    const unsigned boff = ((unsigned)(rowoff[i] + delta) << 2) | ((okb ^ 1u) << 31);
                    ra[i] = __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(in_rsrc, boff, 0, 0));
                } else if (BRANCHY) {
                    if (ok) ra[i] = *reinterpret_cast<const f32x4*>(p.in + (int64_t)(rowoff[i] + delta));
                } else {
                    ra[i] = *reinterpret_cast<const f32x4*>(p.in + (int64_t)(ok ? rowoff[i] + delta : 0));
                }
                okmask = ok ? (okmask | (1u << i)) : (okmask & ~(1u << i));
            }
            // K order of the split weights in the fast path: channel-block major, taps inner, so the taps of one
            // 32-channel block re-read the same (shifted) pixels back to back -> L1/L2 hits instead of MALL/HBM
            ++u_tap;
            if (++u_kx == p.kw) { u_kx = 0; ++u_ky; }
            if (u_tap == p.taps) { u_tap = 0; u_kx = 0; u_ky = 0; ++u_cb; }
        } else {
            const int kk = c * BK + ak;
            const int tap = kk / p.Cin;
            const int ci = kk - tap * p.Cin;
            const int ky = tap / p.kw, kx = tap - ky * p.kw;
            const int dy = ky * p.dil, dx = kx * p.dil;
            const bool tap_ok = valid && tap < p.taps;
#pragma unroll
            for (int i = 0; i < A_LD; ++i) {
                const int iy = iy0[i] + dy, ix = ix0[i] + dx;
                const bool ok = tap_ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                ra[i] = *reinterpret_cast<const f32x4*>(p.in + (ok ? ((int64_t)iy * p.W + ix) * p.in_ld + ci : 0));
                okmask = ok ? (okmask | (1u << i)) : (okmask & ~(1u << i));
            }
        }
        const int cw = valid ? c : c_end - 1;
        if constexpr (!GLDS) {
#pragma unroll
            for (int i = 0; i < B_LD; ++i) {
                if (BN % B_ROWS == 0 || brow + B_ROWS * i < BN) {
                    rbh[i] = *reinterpret_cast<const f16x8*>(p.wh + woff0 + i * wstep + cw * BK);
                    if (NPASS == 3) rbl[i] = *reinterpret_cast<const f16x8*>(p.wl + woff0 + i * wstep + cw * BK);
                }
            }
        }
    };
    // (abuf, bbuf): which A / B stage a chunk lives in.  Two-stage tiles: both = chunk parity; single-stage GLDS tiles: A always
    // stage 0, the weights alternate
    auto b_stage = [&](int abuf, int bbuf) __attribute__((always_inline)) -> _Float16* {
        if constexpr (GLDS && !DBUF) return smem + STAGE_A + bbuf * BST;
        else return smem + abuf * STAGE + STAGE_A;
    };
    auto store_chunk = [&](int buf, RegSet& R) __attribute__((always_inline)) {
        f32x4 (&ra)[A_LD] = R.ra;
        const unsigned okmask = R.okmask;
        auto& rbh = R.rbh;
        auto& rbl = R.rbl;
        _Float16* Ah = smem + buf * STAGE;
        _Float16* Al = Ah + BM * LDH;
        _Float16* Bh = Al + BM * LDH;
        _Float16* Bl = Bh + BN * LDH;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            f16x4 hi, lo;
            f32x4 v = ra[i];
            if (RELU_IN) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
            if (NORM_IN) {                                  // the same arithmetic as otvm_gn_apply (and the patch kernel)
                v = v * R.nsc + R.nsh;
                v.x = v.x > 0.f ? v.x : v.x * p.in_slope; v.y = v.y > 0.f ? v.y : v.y * p.in_slope;
                v.z = v.z > 0.f ? v.z : v.z * p.in_slope; v.w = v.w > 0.f ? v.w : v.w * p.in_slope;
            }
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            if (!GLDS || NORM_IN) v = (okmask >> i) & 1u ? v : z;      // (GLDS: a padding lane was loaded as zeros; the normalisation moves them)
            if constexpr (NPASS == 3) {
                split4(v, hi, lo);
                // M16: k-octet o of row r sits at octet o ^ f(r), f = bit 2 ^ bit 3 of r: with the plain layout the 16x16x32
                // A fragment (row lane & 15, octet lane >> 4, 80-byte rows) is a 2-way bank conflict in every lane group
                const int row = arow + A_ROWS * i;
                const int akx = M16 ? (ak ^ ((((row >> 2) ^ (row >> 3)) & 1) << 3)) : ak;
                *reinterpret_cast<f16x4*>(&Ah[row * LDH + akx]) = hi;
                *reinterpret_cast<f16x4*>(&Al[row * LDH + akx]) = lo;
            } else {
                hi = f16x4{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};        // round to nearest
                *reinterpret_cast<f16x4*>(&Ah[(arow + A_ROWS * i) * LDH + ak]) = hi;
            }
        }
        if constexpr (!GLDS) {
#pragma unroll
            for (int i = 0; i < B_LD; ++i) {
                if (BN % B_ROWS == 0 || brow + B_ROWS * i < BN) {
                    *reinterpret_cast<f16x8*>(&Bh[(brow + B_ROWS * i) * LDH + bk]) = rbh[i];
                    if (NPASS == 3) *reinterpret_cast<f16x8*>(&Bl[(brow + B_ROWS * i) * LDH + bk]) = rbl[i];
                }
            }
        }
    };
    auto compute_ks = [&](int buf, int ks, int bbuf = -1) __attribute__((always_inline)) {
        const _Float16* Ah = smem + buf * STAGE;
        const _Float16* Al = Ah + BM * LDH;
        const _Float16* Bh = b_stage(buf, bbuf < 0 ? buf : bbuf);
        const _Float16* Bl = Bh + BN * LDH;
        if constexpr (M16) {
            const int sj = ks, l15 = lane & 15, oct = lane >> 4;
            f16x8 bh[TN], bl[TN];
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int o = (((wn * TN + b) * 2 + (lane >> 5)) * 2) * 512 + (l15 + 16 * sj + 32 * (oct & 1)) * 8;
                bh[b] = *reinterpret_cast<const f16x8*>(&Bh[o]);
                bl[b] = *reinterpret_cast<const f16x8*>(&Bh[o + 512]);
            }
            auto quad = [](const f32x16& c, int q) __attribute__((always_inline)) { return f32x4{c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]}; };
            auto put = [](f32x16& c, int q, const f32x4 v) __attribute__((always_inline)) { c[4 * q] = v.x; c[4 * q + 1] = v.y; c[4 * q + 2] = v.z; c[4 * q + 3] = v.w; };
#pragma unroll
            for (int si = 0; si < 2; ++si) {
                f16x8 ah[TM], al[TM];
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    const int o = ((wm * TM + a) * 32 + 16 * si + l15) * LDH + 8 * (oct ^ (((l15 >> 2) ^ (l15 >> 3)) & 1));
                    ah[a] = *reinterpret_cast<const f16x8*>(&Ah[o]);
                    al[a] = *reinterpret_cast<const f16x8*>(&Al[o]);
                }
                const int q = 2 * si + sj;
                // three passes over the TM x TN sub-tiles: consecutive MFMAs never share an accumulator
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b) put(acc[a][b], q, __builtin_amdgcn_mfma_f32_16x16x32_f16(al[a], bh[b], quad(acc[a][b], q), 0, 0, 0));
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b) put(acc[a][b], q, __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[a], bl[b], quad(acc[a][b], q), 0, 0, 0));
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b) put(acc[a][b], q, __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[a], bh[b], quad(acc[a][b], q), 0, 0, 0));
            }
            return;
        }
        const int frow = lane & 31, fk = (lane >> 5) * 8;
        f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#if OTVM_ABL_NOLDSRD
        static_assert(true, "");
        {
            const f16x8 one = {(_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f};
#pragma unroll
            for (int a = 0; a < TM; ++a) { ah[a] = one * (_Float16)(float)(lane + ks); al[a] = one; }
#pragma unroll
            for (int b = 0; b < TN; ++b) { bh[b] = one; bl[b] = one * (_Float16)(float)lane; }
        }
        if (buf < 0)
#endif
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const int o = ((wm * TM + a) * 32 + frow) * LDH + 16 * ks + fk;
            ah[a] = *reinterpret_cast<const f16x8*>(&Ah[o]);
            if (NPASS == 3) al[a] = *reinterpret_cast<const f16x8*>(&Al[o]);
        }
#if OTVM_ABL_NOLDSRD
        if (buf < 0)
#endif
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            if constexpr (GLDS) {                       // fragment blocks [n-tile][k-step][hi|lo][lane][8]
                const int o = (((wn * TN + b) * 2 + ks) * 2) * 512 + lane * 8;
                bh[b] = *reinterpret_cast<const f16x8*>(&Bh[o]);
                if (NPASS == 3) bl[b] = *reinterpret_cast<const f16x8*>(&Bh[o + 512]);
            } else {
                const int o = ((wn * TN + b) * 32 + frow) * LDH + 16 * ks + fk;
                bh[b] = *reinterpret_cast<const f16x8*>(&Bh[o]);
                if (NPASS == 3) bl[b] = *reinterpret_cast<const f16x8*>(&Bl[o]);
            }
        }
#if OTVM_ABL_NOMFMA
        if (ah[0][0] == (_Float16)12345.f) acc[0][0][0] += (float)bh[0][0];      // keep the fragment reads alive
        return;
#endif
        // three passes over the accumulator tiles, so consecutive MFMAs never share an accumulator
        if constexpr (NPASS == 3) {
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[a], bh[b], acc[a][b], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bl[b], acc[a][b], 0, 0, 0);
        }
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bh[b], acc[a][b], 0, 0, 0);
    };

    if constexpr (GLDS && !DBUF) {
        // ---- one activation stage, two weight stages.  Per chunk: barrier (everybody is done with chunk c - 1), convert the
        // activations of chunk c into the stage -- the wait for their registers also covers this wave's copy of chunk c's
        // weights, which is OLDER -- then launch the copy of chunk c + 1 into the other weight stage and the activation loads of
        // chunk c + 1, barrier, multiply.  Both stay in flight across the second barrier and the MFMAs.
        static_assert(PF == 1, "single-stage LDS-DMA tiles: no register ring");
        dma_b(c_begin, 0);
        load_chunk(c_begin, rs[0]);
        for (int c = c_begin; c < c_end; ++c) {
            const int bb = (c - c_begin) & 1;
            __syncthreads();
            store_chunk(0, rs[0]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the DMA of chunk c is older than the loads store_chunk waited for)
            if (c + 1 < c_end) {
                dma_b(c + 1, bb ^ 1);
                load_chunk(c + 1, rs[0]);
            }
            __syncthreads();
            compute_ks(0, 0, bb);
            compute_ks(0, 1, bb);
        }
    } else if constexpr (GLDS) {
        // ---- two LDS stages; chunk c's MFMAs run out of stage c & 1 while the same wave converts the activations of chunk
        // c + 1 into the other stage and, behind that, launches the weight DMA of chunk c + 1 and the activation loads of chunk
        // c + 1 + PF.  Every vector-memory operation in this loop is counted: at store_chunk only activation loads are
        // outstanding (the compiler's own count is exact, PF - 1 sets stay in flight); in front of the barrier the A_LD loads
        // just issued are the only operations younger than the DMA, so `vmcnt(A_LD)` (tail: 0) = "my blocks have landed".
        dma_b(c_begin, 0);
        load_chunk(c_begin, rs[0]);
        store_chunk(0, rs[0]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (the DMA is older than the loads store_chunk waited for)
#pragma unroll
        for (int j = 0; j < PF; ++j)
            if (c_begin + 1 + j < c_end) load_chunk(c_begin + 1 + j, rs[j]);
        __syncthreads();
        int c = c_begin, buf = 0;
        bool more = c + 1 < c_end;
        while (more) {
#pragma unroll
            for (int j = 0; j < PF; ++j) {             // chunk c in stage buf, the activations of chunk c + 1 in register set j
                compute_ks(buf, 0);
                store_chunk(buf ^ 1, rs[j]);
                dma_b(c + 1, buf ^ 1);                 // (stage buf ^ 1 was last read before the previous barrier)
                const bool ld = c + 1 + PF < c_end;
                if (ld) load_chunk(c + 1 + PF, rs[j]);
                compute_ks(buf, 1);
                if (ld) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(A_LD + (NORM_IN ? 2 : 0)) : "memory");   // (NORM_IN: + the chunk's scale / shift vectors)
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                ++c;
                buf ^= 1;
                if (c + 1 >= c_end) { more = false; break; }
            }
        }
        compute_ks(buf, 0);
        compute_ks(buf, 1);
    } else {
    load_chunk(c_begin, rs[0]);
    if (DBUF) {
        // one barrier per chunk: while the MFMAs of chunk c run out of stage c&1, the same wave converts chunk
        // c+1 into the other stage (VALU/LDS work issues in the shadow of the 32-cycle MFMAs) and then launches
        // the global loads of chunk c+1+PF, which have PF chunks of MFMA time to land.
        store_chunk(0, rs[0]);
#pragma unroll
        for (int j = 0; j < PF; ++j)
            if (c_begin + 1 + j < c_end) load_chunk(c_begin + 1 + j, rs[j]);
        __syncthreads();
        // the conversion of chunk c+1 sits between the two k-steps of chunk c in ONE basic block (no branch around
        // it: the last chunk is peeled), so the scheduler can interleave its VALU / LDS writes with the MFMAs
        int c = c_begin, buf = 0;
        bool more = c + 1 < c_end;
        while (more) {
#pragma unroll
            for (int j = 0; j < PF; ++j) {             // chunk c in stage buf, chunk c+1 in register set j
                compute_ks(buf, 0);
                store_chunk(buf ^ 1, rs[j]);
                if (c + 1 + PF < c_end) load_chunk(c + 1 + PF, rs[j]);
                compute_ks(buf, 1);
                __syncthreads();
                ++c;
                buf ^= 1;
                if (c + 1 >= c_end) { more = false; break; }
            }
        }
        compute_ks(buf, 0);
        compute_ks(buf, 1);
    } else if (PF == 1) {
        for (int c = c_begin; c < c_end; ++c) {
            __syncthreads();
            if (!OTVM_ABL_NOSTAGE || c == c_begin) store_chunk(0, rs[0]);
            __syncthreads();
            if (c + 1 < c_end && !OTVM_ABL_NOLOAD) load_chunk(c + 1, rs[0]);
            compute_ks(0, 0);
            compute_ks(0, 1);
        }
    } else {
        // one LDS stage, PF chunks of global loads in flight: chunk c sits in register set (c - c_begin) % PF.  No branch
        // surrounds a load (chunks past the end are loaded as dummies): with conditional prologue / tail loads the
        // compiler's waitcnt pass has to merge paths in which a set's load is the newest one in flight and paths in which
        // PF - 1 sets follow it, and falls back to draining everything once per turn of the ring (seen in the ISA).
#pragma unroll
        for (int j = 1; j < PF; ++j) load_chunk(c_begin + j, rs[j], c_begin + j < c_end);
        for (int c = c_begin; c < c_end; c += PF) {
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                if (j > 0 && c + j >= c_end) break;
                __syncthreads();
                store_chunk(0, rs[j]);
                __syncthreads();
                load_chunk(c + j + PF, rs[j], c + j + PF < c_end);
                compute_ks(0, 0);
                compute_ks(0, 1);
            }
        }
    }
    }

    const int col = lane & 31, rbase = (lane >> 5) * 4;
    // where accumulator register e of a 32x32 tile sits inside the tile (M16: four 16x16 sub-tiles, see above)
    auto acc_row = [&](int e) __attribute__((always_inline)) -> int {
        return M16 ? 16 * (e >> 3) + 4 * (lane >> 4) + (e & 3) : (e & 3) + 8 * (e >> 2) + rbase;
    };
    auto acc_col = [&](int e) __attribute__((always_inline)) -> int { return M16 ? 16 * ((e >> 2) & 1) + (lane & 15) : col; };
    // ---- epilogue.  Each 32x32 accumulator tile goes through a wave-private LDS patch (144-byte rows) so that it
    // leaves as 16-byte row-major accesses: 4 store instructions per tile instead of 16, and bias / residual are
    // read as float4.  (With scalar accesses the residual read alone ran at 0.7 TB/s on the K=64 layers.)
    __syncthreads();                                   // all waves are done with the A/B stages
    const bool vec_ok = ((p.out_ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(outp) & 15) == 0) &&
                        (!p.residual || (((p.res_ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.residual) & 15) == 0)));
    // Round 3: interior tiles (all BM rows and BN columns inside the output, 16-byte accesses possible) take a copy of the
    // epilogue WITHOUT per-row predicates.  Stores count in vmcnt on gfx9, and for a store inside a divergent branch the
    // compiler cannot count what is outstanding at the join: it waits for vmcnt(0) in every predicated row block -- i.e.
    // for the previous store's acknowledgement from L2, 32 times per wave of the 256x256 tile (profiles/r03_fused_bottleneck.txt
    // is where this showed up first).
    const bool interior = vec_ok && m0 + BM <= p.M && n0 + BN <= p.Cout &&           // workgroup-uniform
                          !(p.residual && p.res_scale && p.act != OTVM_ACT_RELU);
    // vmcnt retires in order: waiting for a load that was issued AFTER a store also waits for that store.  So the interior
    // path fetches the scale / bias vectors of all TN column tiles up front and requests the residual of tile t + 1 before
    // the stores of tile t go out -- nothing in it ever waits for a store.  Activation and residual are compile-time here:
    // a uniform branch inside the tile loop would split it into basic blocks and bring the conservative vmcnt(0) back.
    auto epilogue_full = [&](auto act_c, auto res_c) __attribute__((always_inline)) {
        constexpr int ACT = decltype(act_c)::value;
        constexpr bool RES = decltype(res_c)::value != 0;         // 0: no residual, 1: residual, 2: residual * res_scale[c]
        constexpr bool RSC = decltype(res_c)::value == 2;
        float* patch = reinterpret_cast<float*>(smem) + wave * (32 * 36);
        const int prow = lane >> 3, pc = (lane & 7) * 4;
        f32x4 sc4[TN], bi4[TN];
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int n4 = n0 + (wn * TN + b) * 32 + pc;
#pragma unroll
            for (int j = 0; j < 4; ++j) sc4[b][j] = p.wscale[n4 + j];
            bi4[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (p.bias) {
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j) bi4[b][j] = p.bias[n0 + (wn * TN + b) * 32 + pc + j];
        }
        auto load_res = [&](int t, f32x4 (&r)[4]) __attribute__((always_inline)) {
            const int b = t / TM, a = t - b * TM;
            const int n4 = n0 + (wn * TN + b) * 32 + pc;
            const int mb = m0 + (wm * TM + a) * 32;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                r[r4] = *reinterpret_cast<const f32x4*>(p.residual + (int64_t)(mb + r4 * 8 + prow) * p.res_ld + n4);
        };
        f32x4 rnext[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        // ABI 17: the residual's own per-channel scale (one vector per column tile, fetched up front like scale / bias)
        f32x4 rs4[RSC ? TN : 1];
        if (RSC) {
#pragma unroll
            for (int b = 0; b < (RSC ? TN : 1); ++b) rs4[b] = *reinterpret_cast<const f32x4*>(p.res_scale + n0 + (wn * TN + b) * 32 + pc);
        }
        if (RES) load_res(0, rnext);
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int a = 0; a < TM; ++a) {                // (two short loops: a single 16-trip loop is not unrolled, and a
                const int t = b * TM + a;                 //  dynamically indexed accumulator array goes to scratch)
                const int n4 = n0 + (wn * TN + b) * 32 + pc;
                const int mb = m0 + (wm * TM + a) * 32;
                f32x4 rres[4];
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) rres[r4] = rnext[r4];
#pragma unroll
                for (int e = 0; e < 16; ++e) patch[acc_row(e) * 36 + acc_col(e)] = acc[a][b][e];
                if (RES && t + 1 < TM * TN) load_res(t + 1, rnext);
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int row = r4 * 8 + prow;
                    f32x4 v = *reinterpret_cast<const f32x4*>(&patch[row * 36 + pc]);
                    v = v * sc4[b] + bi4[b];
                    if (RSC) v += rres[r4] * rs4[RSC ? b : 0];
                    else if (RES) v += rres[r4];
                    v.x = otvm_act(v.x, ACT); v.y = otvm_act(v.y, ACT); v.z = otvm_act(v.z, ACT); v.w = otvm_act(v.w, ACT);
                    *reinterpret_cast<f32x4*>(outp + (int64_t)(mb + row) * p.out_ld + n4) = v;
                }
            }
    };
    auto epilogue = [&](auto full_c) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_c)::value;
        float* patch = reinterpret_cast<float*>(smem) + wave * (32 * 36);
        const int prow = lane >> 3, pc = (lane & 7) * 4;
        otvm_static_for<TN>([&](auto b_c) __attribute__((always_inline)) {
            constexpr int b = decltype(b_c)::value;
            const int nb = n0 + (wn * TN + b) * 32;    // first column of this tile
            if (!FULL && nb >= p.Cout) return;
            const int n4 = nb + pc;                    // this lane's 4 columns in the row-major pass
            f32x4 sc4 = {0.f, 0.f, 0.f, 0.f}, bi4 = {0.f, 0.f, 0.f, 0.f}, rs4 = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (FULL || n4 + j < p.Cout) {
                    sc4[j] = p.wscale[n4 + j];
                    bi4[j] = p.bias ? p.bias[n4 + j] : 0.f;
                    if (p.residual && p.res_scale) rs4[j] = p.res_scale[n4 + j];
                }
            }
            otvm_static_for<TM>([&](auto a_c) __attribute__((always_inline)) {
                constexpr int a = decltype(a_c)::value;
                const int mb = m0 + (wm * TM + a) * 32;
                // residual: all four 16-byte loads of this tile are issued before anything waits on them (the
                // load -> add -> store chain per row group was latency-bound: 1.5 TB/s on the K=64 residual layers)
                f32x4 rres[4];
                const bool res_vec = p.residual && vec_ok && (FULL || n4 + 3 < p.Cout);
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int m = mb + r4 * 8 + prow;
                    rres[r4] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (FULL) {
                        if (p.residual) rres[r4] = *reinterpret_cast<const f32x4*>(p.residual + (int64_t)m * p.res_ld + n4);   // (uniform branch)
                    } else if (res_vec && m < p.M) {
                        rres[r4] = *reinterpret_cast<const f32x4*>(p.residual + (int64_t)m * p.res_ld + n4);
                    }
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) patch[acc_row(e) * 36 + acc_col(e)] = acc[a][b][e];
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int row = r4 * 8 + prow;
                    const int m = mb + row;
                    f32x4 v = *reinterpret_cast<const f32x4*>(&patch[row * 36 + pc]);
                    v = v * sc4 + bi4;
                    if (FULL || m < p.M) {
                        if (FULL || (vec_ok && n4 + 3 < p.Cout)) {
                            v += rres[r4] * rs4;
                            v.x = otvm_act(v.x, p.act); v.y = otvm_act(v.y, p.act);
                            v.z = otvm_act(v.z, p.act); v.w = otvm_act(v.w, p.act);
                            *reinterpret_cast<f32x4*>(outp + (int64_t)m * p.out_ld + n4) = v;
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                if (n4 + j < p.Cout) {
                                    float x = v[j];
                                    if (p.residual) x += p.residual[(int64_t)m * p.res_ld + n4 + j] * rs4[j];
                                    outp[(int64_t)m * p.out_ld + n4 + j] = otvm_act(x, p.act);
                                }
                            }
                        }
                    }
                }
            });
        });
    };
    if (interior) {
        using std::integral_constant;
        const bool r = p.residual != nullptr;
        using I0 = integral_constant<int, 0>;
        using I1 = integral_constant<int, 1>;
        if (r && p.res_scale) epilogue_full(integral_constant<int, OTVM_ACT_RELU>{}, integral_constant<int, 2>{});   // (the bottleneck tail: always ReLU; other activations take the predicated copy)
        else if (p.act == OTVM_ACT_RELU) { if (r) epilogue_full(integral_constant<int, OTVM_ACT_RELU>{}, I1{}); else epilogue_full(integral_constant<int, OTVM_ACT_RELU>{}, I0{}); }
        else if (p.act == OTVM_ACT_LEAKY) { if (r) epilogue_full(integral_constant<int, OTVM_ACT_LEAKY>{}, I1{}); else epilogue_full(integral_constant<int, OTVM_ACT_LEAKY>{}, I0{}); }
        else { if (r) epilogue_full(integral_constant<int, OTVM_ACT_NONE>{}, I1{}); else epilogue_full(integral_constant<int, OTVM_ACT_NONE>{}, I0{}); }
    } else {
        epilogue(std::false_type{});
    }
    // ---- fused GroupNorm statistics of the tile just written (sum / sum of squares per group, fp64 atomics)
    if (p.gn_stats) {
        // (sum, sumsq) per group of the tile, at most BN/2 groups; lives behind the waves' epilogue patches in the
        // stage memory (the two-stage 256x256 tile uses the whole 160 KB of LDS)
        static_assert((NT / 64) * (32 * 36 * 4) + 2 * BN * 8 <= SMEM_HALFS * 2, "gred does not fit behind the patches");
        double* gred = reinterpret_cast<double*>(reinterpret_cast<char*>(smem) + (NT / 64) * (32 * 36 * 4));
        const int cg = p.Cout >> 5;                         // channels per group (>= 2)
        const int seg = cg < 32 ? cg : 32;                  // lanes of one 32-column tile that share a group
        for (int i = threadIdx.x; i < 2 * BN; i += blockDim.x) gred[i] = 0.0;
        __syncthreads();
        if constexpr (M16) {
            // a lane owns two columns of a 32-column tile (sj = 0 / 1) and four rows of each of its 16-row halves; the lanes
            // that share a column are lane ^ 16, lane ^ 32, lane ^ 48
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                float s2[2], ss2[2];
#pragma unroll
                for (int sj = 0; sj < 2; ++sj) {
                    const int n = n0 + (wn * TN + b) * 32 + 16 * sj + (lane & 15);
                    float s = 0.f, ss = 0.f;
                    if (n < p.Cout) {
                        const float sc_ = p.wscale[n];
                        const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
                        for (int a = 0; a < TM; ++a)
#pragma unroll
                            for (int si = 0; si < 2; ++si)
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const int m = m0 + (wm * TM + a) * 32 + 16 * si + 4 * (lane >> 4) + r;
                                    if (m < p.M) {
                                        const float v = acc[a][b][4 * (2 * si + sj) + r] * sc_ + bias;
                                        s += v;
                                        ss += v * v;
                                    }
                                }
                    }
                    s += __shfl_xor(s, 16); ss += __shfl_xor(ss, 16);
                    s += __shfl_xor(s, 32); ss += __shfl_xor(ss, 32);
                    s2[sj] = s; ss2[sj] = ss;
                }
                if (seg == 32) {                                // the tile's 32 columns lie in one group
                    float s = s2[0] + s2[1], ss = ss2[0] + ss2[1];
                    for (int off = 1; off < 16; off <<= 1) {
                        s += __shfl_xor(s, off);
                        ss += __shfl_xor(ss, off);
                    }
                    const int nl = (wn * TN + b) * 32;
                    if (lane == 0 && n0 + nl < p.Cout) {
                        atomicAdd(&gred[2 * (nl / cg)], (double)s);
                        atomicAdd(&gred[2 * (nl / cg) + 1], (double)ss);
                    }
                } else {                                        // seg = cg <= 16 consecutive columns per group
#pragma unroll
                    for (int sj = 0; sj < 2; ++sj) {
                        float s = s2[sj], ss = ss2[sj];
                        for (int off = 1; off < seg; off <<= 1) {
                            s += __shfl_xor(s, off);
                            ss += __shfl_xor(ss, off);
                        }
                        const int nl = (wn * TN + b) * 32 + 16 * sj + (lane & 15);
                        if (lane < 16 && (lane & (seg - 1)) == 0 && n0 + nl < p.Cout) {
                            atomicAdd(&gred[2 * (nl / cg)], (double)s);
                            atomicAdd(&gred[2 * (nl / cg) + 1], (double)ss);
                        }
                    }
                }
            }
        } else
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int nl = (wn * TN + b) * 32 + col;        // column inside the tile
            const int n = n0 + nl;
            float s = 0.f, ss = 0.f;
            if (n < p.Cout) {
                const float sc_ = p.wscale[n];
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
        __syncthreads();                                    // gred is free again: scratch of the table tail
        otvm_gn_table_tail(p.gn_stats, p.M, p.Cout, p.tail, blockIdx.z, gridDim.x * gridDim.y, reinterpret_cast<float*>(gred));
    }
}

// FAST_ONLY: tiles that config_ok() only offers to whole-chunk layers do not instantiate the generic-decode kernels
template <int BM, int BN, int WM, int WN, bool DB = false, bool FAST_ONLY = false, bool GLDS = false, int NPASS = 3, bool M16T = false>
int launch3(Conv3Args& a, hipStream_t s, int ksplit = 1) {
    a.tiles_m = otvm_ceil_div(a.M, BM);
    a.tiles_n = otvm_ceil_div(a.Cout, BN);
    const dim3 grid(a.tiles_m * a.tiles_n, ksplit, a.batch), block(WM * WN * 64);
    // int32 element offsets in the fast path: the whole input view must stay below 2^31 elements
    // (the split weights of such layers are stored channel-block major: the generic decode cannot read them)
    const bool fast = f16x3_fast_layout(a.taps, a.Cin);
    if (fast && (int64_t)a.H * a.W * a.in_ld >= (1ll << 31) - (1 << 20)) {
        otvm_set_error("otvm_conv2d(f16x3): input view too large for 32-bit offsets");
        return 1;
    }
    if (a.in_scale && !(fast && !a.in_relu)) {
        otvm_set_error("otvm_conv2d(f16x3): the fused input normalisation needs a whole-chunk layer (Cin %% 32 == 0) without in_relu");
        return 1;
    }
    if constexpr (GLDS) {
        const int64_t bytes = ((int64_t)a.H * a.W * a.in_ld) * (int64_t)sizeof(float);
        if (!fast || !a.wf || bytes >= (1ll << 31)) {
            otvm_set_error("otvm_conv2d(f16x3): the LDS-DMA tiles take whole-chunk layers with fragment-major weights and < 2 GiB inputs");
            return 1;
        }
        a.in_bytes = (unsigned)bytes;
    }
    if (fast) {
        if (a.in_scale) hipLaunchKernelGGL((conv_igemm_f16x3_kernel<BM, BN, WM, WN, true, false, DB, true, GLDS, NPASS, M16T>), grid, block, 0, s, a);
        else if (a.in_relu) hipLaunchKernelGGL((conv_igemm_f16x3_kernel<BM, BN, WM, WN, true, true, DB, false, GLDS, NPASS, M16T>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((conv_igemm_f16x3_kernel<BM, BN, WM, WN, true, false, DB, false, GLDS, NPASS, M16T>), grid, block, 0, s, a);
    } else if constexpr (!DB && !FAST_ONLY) {
        if (a.in_relu) hipLaunchKernelGGL((conv_igemm_f16x3_kernel<BM, BN, WM, WN, false, true, false, false, false, NPASS>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((conv_igemm_f16x3_kernel<BM, BN, WM, WN, false, false, false, false, false, NPASS>), grid, block, 0, s, a);
    } else {
        otvm_set_error("otvm_conv2d(f16x3): this tile takes whole-chunk layers only");
        return 1;
    }
    OTVM_CHECK_LAUNCH("otvm_conv2d(f16x3)");
    return 0;
}

}  // namespace
