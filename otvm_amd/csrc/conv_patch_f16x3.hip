// 3x3 (stride 1, dilation d) convolution as a "patch" kernel on the f16 MFMA with split-fp16 operands (f16x3).
//
// The implicit-GEMM kernel (conv_f16x3.hip) re-gathers and re-splits every input element once per tap: 9 global
// loads, 9 fp32->(hi,lo) conversions and 9 LDS writes per element.  For the layers with few output channels
// (full-resolution refinement / head convs, N = 16..64) that staging work, not the MFMA, sets the time.  Here a
// workgroup owns a TH x 32 block of output pixels: per 32-channel block it stages the (TH+2d) x (32+2d) input patch
// into LDS ONCE (one conversion per element), and the 9 taps read shifted windows of that patch as MFMA A operands
// (pixel rows are 80 bytes apart in LDS -> conflict-free ds_read_b128 for 32 consecutive pixels).
//
// Weights never touch LDS: like the memory bank they are packed at load time in MFMA B-fragment order
// (otvm_pack_patch_weight_f16x3: [cin/32][tap][n/32][k-step][hi|lo][lane][8 halfs] = 1-KiB blocks, taps ordered
// kx-major: tap = kx*3 + ky, so a 3-tap weight stage holds one filter COLUMN), so a wave fetches
// a B operand with one coalesced 16-byte-per-lane load from L2.
//
// One wave owns TM = TH/NW output rows (M tiles of 32 pixels) x TN = BN/32 channel tiles.  fp32 accumulate; epilogue
// as conv_f16x3.hip (filter scale, bias, residual, activation, optional fused GroupNorm statistics).
#pragma clang diagnostic ignored "-Winline-asm"      // (M0 in an asm clobber list: see conv_f16x3_kernel.h)
#include "common.h"
#include "head_math.h"
#include <type_traits>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// timing probes for tools/build_variant.sh (results are WRONG with any of them set)
#ifndef OTVM_PABL_NOMFMA
#define OTVM_PABL_NOMFMA 0
#endif
#ifndef OTVM_PABL_NOLOAD
#define OTVM_PABL_NOLOAD 0     // global loads of the first stage only
#endif
#ifndef OTVM_PABL_NOCOMMIT
#define OTVM_PABL_NOCOMMIT 0   // split + LDS writes of the first stage only
#endif
#ifndef OTVM_PABL_NOLDSRD
#define OTVM_PABL_NOLDSRD 0    // no fragment reads from LDS
#endif
#ifndef OTVM_PABL_NOEPI
#define OTVM_PABL_NOEPI 0      // no output stores / residual loads
#endif
#ifndef OTVM_PABL_MFMA16
#define OTVM_PABL_MFMA16 0     // every MFMA of the 32x32x16 forms replaced by two 16x16x32 on the same fragments (what the instruction
#endif                         // alone is worth on a tile before a real 16x16x32 form exists: profiles/r05_patch_wide_mfma16_ab.txt)
#ifndef OTVM_PATCH_NO_LEAN
#define OTVM_PATCH_NO_LEAN 0   // (A/B build, results right) M16 tiles on the generic staging code of round 5
#endif
#ifndef OTVM_PATCH_SETPRIO
#define OTVM_PATCH_SETPRIO 0   // (A/B build) M16 tiles: s_setprio 1 around a stage's fragment reads + MFMAs, 0 around its staging
#endif
#ifndef OTVM_PM16_NOKXP
#define OTVM_PM16_NOKXP 0      // (A/B build, results right) M16 tiles: taps paired (t, t + 1) on every dilation
#endif

namespace {

struct PatchArgs {
    const float* in; const _Float16* wf; const float* wscale; const float* bias; const float* residual; float* out;
    double* gn_stats;
    int H, W, Cin, in_ld, res_ld, Cout, out_ld, n_pad32, in_relu, act;
    int tiles_x, tiles_y, tiles_n;
    OtvmTileWalk walk;           // tile walk of the grid (common.h)
    // fused input normalisation (GroupNorm apply of the producer folded into the staging): x' = in_act(x * in_scale[c]
    // + in_shift[c]) for pixels inside the image, 0 for the conv's zero padding; nullptr = plain input
    const float* in_scale; const float* in_shift; int in_act;
    // batch: image blockIdx.y of every tensor lives *_bs elements behind image 0
    int64_t in_bs, out_bs, res_bs; int gn_bs, norm_bs, batch;
    OtvmGnTail tail;             // ABI 16: the output's GroupNorm table, written by the last workgroup (common.h)
    // ABI 17 (INRES kernels): the input is the raw GroupNorm input of a residual block's LAST normalisation, whose apply pass
    // was skipped: x' = in_act(x * in_scale[c] + in_shift[c] + in_res), in_res = the block's (materialised) identity
    const float* in_res; int in_res_ld; int64_t in_res_bs;
    unsigned in_bytes, in_res_bytes;     // round 5: sizes of one image's input / identity views (buffer-resource ranges)
    // ABI 17 (HEAD kernels, 16 output channels): the 1x1 head + fba_fusion of the FBA decoder / refinement run in the epilogue
    // on the pixel's 16 hidden values (head_math.h); out (the hidden state) is optional then
    OtvmHeadArgs head; int64_t head_img_bs, head_alpha_bs, head_tri_bs, head_sm_bs;
};

constexpr int CB = 16;           // channels per stage = one MFMA k-step
constexpr int LDP = 24;          // halfs per patch pixel (16 channels + 8 pad) = 48 bytes: 3r mod 16 distinct -> conflict-free b128

__device__ __forceinline__ void split4p(const f32x4 v, f16x4& hi, f16x4& lo) {
    typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
    const fp16x2 p01 = __builtin_amdgcn_cvt_pkrtz(v.x, v.y);
    const fp16x2 p23 = __builtin_amdgcn_cvt_pkrtz(v.z, v.w);
    const f16x2 h01 = __builtin_bit_cast(f16x2, p01);
    const f16x2 h23 = __builtin_bit_cast(f16x2, p23);
    hi = f16x4{h01.x, h01.y, h23.x, h23.y};
    lo = f16x4{(_Float16)(v.x - (float)h01.x), (_Float16)(v.y - (float)h01.y), (_Float16)(v.z - (float)h23.x),
               (_Float16)(v.w - (float)h23.y)};
}

#if OTVM_PABL_MFMA16
__device__ __forceinline__ f32x16 mfma_probe16(const f16x8 a, const f16x8 b, f32x16 c) {
    f32x4 q0 = {c[0], c[1], c[2], c[3]}, q1 = {c[4], c[5], c[6], c[7]};
    q0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, q0, 0, 0, 0);
    q1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, q1, 0, 0, 0);
    c[0] = q0.x; c[1] = q0.y; c[2] = q0.z; c[3] = q0.w; c[4] = q1.x; c[5] = q1.y; c[6] = q1.z; c[7] = q1.w;
    return c;
}
#define PATCH_MFMA32(a, b, c) mfma_probe16(a, b, c)
#else
#define PATCH_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#endif

// M16: MFMA row r of a 16-pixel A fragment holds pixel pi16(r) of its 16-pixel run (see the kernel's M16 comment): rows
// {0-3, 12-15} <-> pixels = 0, 1 mod 4; rows 4-11 <-> pixels = 2, 3 mod 4
__device__ __forceinline__ int pi16(int r) {
    const int k = (r < 4) ? r : (r < 12 ? r - 4 : r - 8);          // index inside the row's class (0 ... 7)
    return 4 * (k >> 1) + (k & 1) + ((r >= 4 && r < 12) ? 2 : 0);
}

// Per stage (16 input channels) a workgroup holds in LDS the split input patch and, per group of TAPG taps, the B
// fragments of its BN output channels, copied verbatim from the fragment-major weight array (1-KiB blocks).
// TAPG = 9: one weight stage per channel stage (narrow layers); TAPG = 3: wide layers (BN = 256), where nine taps of
// weights (144 KiB) would not fit beside the patch.
// NPASS = 1: precision "f16" (one MFMA pass on fp16-rounded operands; conv_f16x3_kernel.h has the definition of the mode)
// M16 (round 5): the nine-tap tiles on v_mfma_f32_16x16x32_f16 -- the instruction does ~10 % more work per joule than
// v_mfma_f32_32x32x16_f16 under this chip's power limit (profiles/r05_igemm_mfma16_ab.txt) and these launches are bound by it.
// A stage still holds 16 input channels, so the K = 32 of one instruction is filled with TWO of the stage's 27 (tap, pass)
// products: lanes 0-31 (k-octets 0, 1) carry one, lanes 32-63 (k-octets 2, 3) the other.  Taps 0 ... 7 go as four pairs
// (t, t') x {lo.hi, hi.lo, hi.hi}; tap 8 as [hi | lo] x [hi | hi] (hi.hi + lo.hi) and [hi | lo] x [lo | 0] (hi.lo): 14 instructions'
// worth of K for 13.5 (3.7 % idle).  A 32 x 32 accumulator tile is four 16 x 16 sub-tiles (registers 4 q ... 4 q + 3 of quad
// q = 2 si + sj: pixel 16 si + PI(4 (lane >> 4) + j), channel 16 sj + (lane & 15)).  The patch lies PLANAR in LDS,
// [hi | lo][k-octet][pixel] x 16 bytes (no padding: 64 instead of 96 bytes per pixel), the octet planes 64 bytes apart mod 256:
// with the row permutation PI (rows {0-3, 12-15} <-> pixels {0, 1, 4, 5, 8, 9, 12, 13}) every lane group of a ds_read_b128 A
// fragment ({0-3, 12-15, 20-27}, ...: rows {0-3, 12-15} of one octet + rows 4-11 of the other) covers sixteen distinct 16-byte
// bank slots for ANY tap offset, and the staging ds_write_b64 (four pixels x four quads per 16 lanes) is conflict-free too.
// B fragments come from the unchanged 1-KiB blocks: slot (lane & 15) + 16 sj + 32 (octet & 1) of tap t or t + 1.
template <int TH, int BN, int NW, int DIL, int TAPG, bool INRES = false, bool HEAD = false, int NWN = 1, bool GLDS = false, int NPASS = 3,
          bool M16 = false>
__global__ __launch_bounds__(NW * 64)
__attribute__((amdgpu_waves_per_eu((TAPG == 3 && BN <= 32 && !INRES) ? 3 : (M16 ? 2 : 1), (TAPG == 3 && BN <= 32 && !INRES) ? 3 : 10)))
void conv_patch_f16x3_kernel(const PatchArgs pa) {
    PatchArgs p = pa;
    {
        const int zb = blockIdx.y;
        p.in += zb * p.in_bs;
        if (INRES) p.in_res += zb * p.in_res_bs;
        if (HEAD) {
            p.head.img += zb * p.head_img_bs;
            if (p.head.alpha_out) p.head.alpha_out += zb * p.head_alpha_bs;
            if (p.head.tri_out) p.head.tri_out += zb * p.head_tri_bs;
            if (p.head.sm) p.head.sm += zb * p.head_sm_bs;
        }
        p.out += zb * p.out_bs;
        if (p.residual) p.residual += zb * p.res_bs;
        if (p.gn_stats) p.gn_stats += zb * p.gn_bs;
        if (p.in_scale) { p.in_scale += zb * p.norm_bs; p.in_shift += zb * p.norm_bs; }
    }
    constexpr int NT = NW * 64;
    // NWN = 1: a wave owns TM = TH / NW output rows x all BN / 32 channel tiles.  NWN > 1 (the 256-channel tiles, round 4): the
    // waves form an (NW / NWN) x NWN grid -- wave_m picks the rows, wave_n a share of the channel tiles: with 8 waves, TM = 1 and
    // 8 channel tiles a wave read 2 A + 16 B fragments from LDS per 24 MFMAs, all eight waves the same 16 KiB of weights; as
    // 4 x 2 waves (TM = 2, TN = 4) it reads 4 + 8 for the same 24 MFMAs on the same 128 accumulator registers
    constexpr int NWM = NW / NWN, TNW = BN / 32;
    constexpr int TM = TH / NWM, TN = TNW / NWN;
    static_assert(NW % NWN == 0 && TH % NWM == 0 && TNW % NWN == 0, "bad wave grid");
    constexpr int PW = 32 + 2 * DIL, PH = TH + 2 * DIL, NPIX = PH * PW;
    constexpr int NG = 9 / TAPG;
    static_assert(BN % 32 == 0 && 9 % TAPG == 0, "bad tile");
    static_assert(!M16 || (TAPG == 9 && NPASS == 3 && !HEAD && NWN == 1 && !GLDS), "the 16x16x32 form: nine-tap f16x3 tiles");
    constexpr int PLANE = ((NPIX + 11) / 16) * 16 + 4;                 // M16: pixels per octet plane, = 4 mod 16 (>= NPIX)
    constexpr int PATCH_HALFS = M16 ? 2 * 2 * PLANE * 8 : 2 * NPIX * LDP;   // hi + lo
    constexpr int B_PIECES = TAPG * TNW * 2 * 64;                      // 16-byte pieces of one weight stage
    constexpr int B_HALFS = B_PIECES * 8;
    constexpr int EPI_HALFS = NW * 32 * 36 * 2 * (HEAD ? 2 : 1);       // epilogue patches (fp32) expressed in halfs
    // GLDS (round 4, the 256-channel tiles): the weight stage is copied global -> LDS by the LDS-DMA path (global_load_lds_dwordx4:
    // the 1-KiB fragment blocks are lane-linear, exactly what it writes) into the OTHER of two stage buffers while the current
    // stage is multiplied -- no staging registers, no ds_write pass, and the stages that keep their input patch (two of three)
    // need ONE barrier instead of two.  Ordering: the issuing waves wait vmcnt(0), then the barrier; the fragments are read after it.
    // (LDS-DMA weight stages copy every channel tile of the block: the host offers them to layers with Cout % BN == 0 only)
    static_assert(!GLDS || TAPG == 3, "LDS-DMA weight stages: the 3-tap stages (two buffers fit beside the patch)");
    constexpr int NBUF = GLDS ? 2 : 1;
    constexpr int SM_HALFS = PATCH_HALFS + NBUF * B_HALFS > EPI_HALFS ? PATCH_HALFS + NBUF * B_HALFS : EPI_HALFS;
    __shared__ __attribute__((aligned(16))) _Float16 smem[SM_HALFS];
    _Float16* Ph = smem;
    _Float16* Pl = smem + (M16 ? 2 * PLANE * 8 : NPIX * LDP);
    _Float16* Bs = smem + PATCH_HALFS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_m = NWN == 1 ? wave : wave % NWM, nt0 = NWN == 1 ? 0 : (wave / NWM) * TN;   // first row block / channel tile of the wave
#ifdef OTVM_PATCH_STAGGER
    // timing experiment: the second workgroup of every CU starts half a tile late, so that one workgroup's epilogue / prologue
    // meets the other's MFMA phase (later workgroups inherit the offset of the slot they take over)
    if (blockIdx.x >= 256 && blockIdx.x < 512)
        for (int i = 0; i < OTVM_PATCH_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
#endif
    // tile decode: channel tile fastest, then the map's tiles in the walk of common.h (XCD-aware bands by default)
    int tile_n, tile_x, tile_y;
    otvm_tile_decode(p.walk, blockIdx.x, gridDim.x, p.tiles_n, p.tiles_x, p.tiles_y, tile_n, tile_x, tile_y);
    const int ty0 = tile_y * TH, tx0 = tile_x * 32, n0 = tile_n * BN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    const int frow = lane & 31, fh = lane >> 5;
    const int ncb = p.Cin / CB;
    const int nbs = p.n_pad32 >> 5;
    const int nb0 = n0 >> 5;
    // register prefetch of the next stage (global -> registers under the MFMAs of the current stage)
    constexpr int NP = (NPIX * 4 + NT - 1) / NT;                       // patch float4 per thread
    constexpr int NB = (B_PIECES + NT - 1) / NT;                       // B 16-byte pieces per thread
    f32x4 rp[NP];
    f32x4 rr[INRES ? NP : 1];                                          // INRES: the identity's values of the same elements
    f16x8 rb[NB];
    f32x4 rsc = {1.f, 1.f, 1.f, 1.f}, rsh = {0.f, 0.f, 0.f, 0.f};      // input-normalisation table of this thread's quad
    // round 5: the patch is loaded through buffer resources, without a branch: a pixel outside the image carries an offset
    // beyond the resource's range and the hardware returns zeros (the `if (inside) load` form was an exec-masked branch per
    // load, i.e. NP extra basic blocks in the K loop)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t res_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(INRES ? p.in_res : p.in), 0, INRES ? p.in_res_bytes : 0, 0x00020000);
    // ---- round 6: lean staging of the nine-tap 16x16x32 tiles (LEAN).  The generic staging below recomputes, per stage and per
    // element, everything that does not depend on the stage -- patch coordinates (a division by PW), the image test, the LDS
    // address, the weight piece's place -- and tests its run-time switches (in_scale, in_relu, in_act) per element: ~1200 non-MFMA
    // instructions (525 VALU, 287 SALU, 183 branches) beside the 224 MFMAs of a stage in the 64-filter tile's ISA, i.e. more issue
    // time than the MFMAs themselves (profiles/r05_patch64_sq_counters_m16.md: 51 % issue stalls).  Here the per-thread offsets
    // are computed ONCE: a stage's loads are NP buffer loads (register offset + a scalar channel offset) and NB weight loads
    // (scalar base + register offset), the normalisation is ONE uniform branch per stage around branch-free arithmetic
    // (activation as max(v, 0) + slope min(v, 0): the same values as otvm_act), the LDS addresses are one register + immediates.
    constexpr bool LEAN = M16 && !INRES && !OTVM_PATCH_NO_LEAN;
    unsigned l_voff[LEAN ? NP : 1];                                    // byte offset of element k's quad in channel block 0; 0xFFFFFFFF = outside the image / patch
                                                                       // (views reach 2^32 - 16 bytes: a 4K full-resolution layer is 2.7 GB -- no flag bit to spare)
    unsigned l_woff[LEAN ? NB : 1];                                    // byte offset of weight piece k in channel stage 0
    unsigned l_wok = 0;                                                // bit k: piece k exists (a 64-wide tile on a <= 32-filter layer has one column tile)
    int l_o0 = 0;                                                      // LDS half-offset of element 0 (element k: + 512 k)
    float l_slope = 1.f;                                               // input activation as a negative slope (1 none, 0 ReLU, 0.01 LeakyReLU)
    if constexpr (LEAN) {
        static_assert(NT % 4 == 0 && B_PIECES == NB * NT && TAPG == 9, "lean staging: whole weight pieces per thread, one stage per channel block");
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int idx = tid + k * NT;
            const int pix = idx >> 2, c4 = (idx & 3) * 4;
            const int py = pix / PW, px = pix - py * PW;
            const int iy = ty0 - DIL + py, ix = tx0 - DIL + px;
            const bool ok = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W) & (idx < NPIX * 4);
            l_voff[k] = ok ? ((unsigned)(iy * p.W + ix) * (unsigned)p.in_ld + (unsigned)c4) << 2 : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int i = tid + k * NT;
            const int l = i & 63, blk = i >> 6, hl = blk & 1, tb = blk >> 1, b = tb % TNW, tap = tb / TNW;
            const bool ok = nb0 + b < nbs;                              // (wave-uniform: a wave copies whole 1-KiB blocks)
            l_woff[k] = (unsigned)((((tap * nbs + nb0 + (ok ? b : 0)) * 2) * 2 + hl) * 512 + l * 8) * 2u;
            l_wok |= ok ? (1u << k) : 0u;
        }
        l_o0 = (((tid >> 1) & 1) * PLANE + (tid >> 2)) * 8 + (tid & 1) * 4;
        l_slope = p.in_relu ? 0.f : (p.in_scale ? (p.in_act == OTVM_ACT_RELU ? 0.f : (p.in_act == OTVM_ACT_LEAKY ? 0.01f : 1.f)) : 1.f);
    }
    auto prefetch = [&](int cb, int g) __attribute__((always_inline)) {
        if constexpr (LEAN) {
            const unsigned soff = (unsigned)(cb * CB * 4);
#pragma unroll
            for (int k = 0; k < NP; ++k)
                rp[k] = __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(in_rsrc, __builtin_elementwise_add_sat(l_voff[k], soff), 0, 0));   // (saturating: outside stays outside)
            const char* wbase = reinterpret_cast<const char*>(p.wf) + (size_t)(((cb >> 1) * 9 * nbs * 2 + (cb & 1)) * 2) * 1024;   // uniform
#pragma unroll
            for (int k = 0; k < NB; ++k) rb[k] = *reinterpret_cast<const f16x8*>(wbase + l_woff[k]);
            if (p.in_scale) {                                           // (uniform; NT % 4 == 0: the quad of element k is tid & 3 for every k)
                rsc = *reinterpret_cast<const f32x4*>(p.in_scale + cb * CB + (tid & 3) * 4);
                rsh = *reinterpret_cast<const f32x4*>(p.in_shift + cb * CB + (tid & 3) * 4);
            }
            return;
        }
        const int cb32 = cb >> 1, ks = cb & 1;
        _Float16* bdst = Bs + (GLDS ? ((cb * NG + g) & 1) * B_HALFS : 0);
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int i = tid + k * NT;
            if (i < B_PIECES) {
                const int l = i & 63, blk = i >> 6;                     // blk = (tapl*TN + b)*2 + hl
                const int hl = blk & 1, tb = blk >> 1, b = tb % TNW, tap = g * TAPG + tb / TNW;
                if (NPASS == 1 && hl) continue;                         // (wave-uniform: a wave copies whole 1-KiB blocks)
                const int64_t src = (((((int64_t)cb32 * 9 + tap) * nbs + nb0 + b) * 2 + ks) * 2 + hl) * 512 + l * 8;
                if constexpr (GLDS) {
                    // issued from inline asm (round 5): the compiler orders every LDS read behind an LDS-DMA it knows about -- the
                    // builtin was followed by `s_waitcnt vmcnt(0)` in front of the stage's first fragment read (ISA of the round-4
                    // build), i.e. the copy never overlapped the MFMAs.  Hidden, it lands under them; the `vmcnt(0)` at the top of
                    // the next stage and the barrier behind it are what the fragment reads rely on.
                    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(bdst + (i - l) * 8));
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(m0v), "v"(p.wf + src) : "m0", "memory");
                } else {
                    // a 64-channel tile on a layer with <= 32 filters (dilated layers have no 32-channel tile): the second channel
                    // tile has no weights -- zeros, not the 2 KiB behind the array (found by tools/conv_fuzz.py --seed 4: a
                    // memory fault when the array ends a mapped segment; its columns are never stored)
                    f16x8 wv = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
                    if (nb0 + b < nbs) wv = *reinterpret_cast<const f16x8*>(p.wf + src);
                    rb[k] = wv;
                }
            }
        }
        if (g == 0) {
            if (p.in_scale) {                                           // NT % 4 == 0: idx & 3 == tid & 3 for every k
                rsc = *reinterpret_cast<const f32x4*>(p.in_scale + cb * CB + (tid & 3) * 4);
                rsh = *reinterpret_cast<const f32x4*>(p.in_shift + cb * CB + (tid & 3) * 4);
            }
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int idx = tid + k * NT;
                const int pix = idx >> 2, c4 = (idx & 3) * 4;
                const int py = pix / PW, px = pix - py * PW;
                const int iy = ty0 - DIL + py, ix = tx0 - DIL + px;
                // 0 inside the image (and inside the patch), all ones outside: pure arithmetic, no select for the compiler to
                // turn back into a branch
                const unsigned oob = (unsigned)((int)((unsigned)iy < (unsigned)p.H) & (int)((unsigned)ix < (unsigned)p.W) &
                                                (int)(idx < NPIX * 4)) - 1u;
                const unsigned e = (unsigned)(iy * p.W + ix);
                rp[k] = __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(in_rsrc, ((e * (unsigned)p.in_ld + cb * CB + c4) << 2) | oob, 0, 0));
                if (INRES)
                    rr[k] = __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(res_rsrc, ((e * (unsigned)p.in_res_ld + cb * CB + c4) << 2) | oob, 0, 0));
            }
        }
    };
    auto commit = [&](int g) __attribute__((always_inline)) {
        if constexpr (LEAN) {
            const f16x8 zero8 = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
#pragma unroll
            for (int k = 0; k < NB; ++k) *reinterpret_cast<f16x8*>(&Bs[(tid + k * NT) * 8]) = ((l_wok >> k) & 1u) ? rb[k] : zero8;
            if (p.in_scale != nullptr || p.in_relu) {                   // ONE uniform branch per stage; rsc = 1, rsh = 0 without a table
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    f32x4 v = rp[k] * rsc + rsh;
                    v.x = fmaxf(v.x, 0.f) + l_slope * fminf(v.x, 0.f); v.y = fmaxf(v.y, 0.f) + l_slope * fminf(v.y, 0.f);
                    v.z = fmaxf(v.z, 0.f) + l_slope * fminf(v.z, 0.f); v.w = fmaxf(v.w, 0.f) + l_slope * fminf(v.w, 0.f);
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    rp[k] = (l_voff[k] == 0xFFFFFFFFu) ? z : v;         // the conv's zero padding is not normalised
                }
            }
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                if (k + 1 < NP || tid + k * NT < NPIX * 4) {            // (only the last element of a thread can lie behind the patch)
                    f16x4 hi, lo;
                    split4p(rp[k], hi, lo);
                    *reinterpret_cast<f16x4*>(&Ph[l_o0 + 512 * k]) = hi;
                    *reinterpret_cast<f16x4*>(&Pl[l_o0 + 512 * k]) = lo;
                }
            }
            return;
        }
        if constexpr (!GLDS) {
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int i = tid + k * NT;
                if (NPASS == 1 && ((i >> 6) & 1)) continue;
                if (i < B_PIECES) *reinterpret_cast<f16x8*>(&Bs[i * 8]) = rb[k];
            }
        }
        if (g == 0) {
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int idx = tid + k * NT;
                if (idx < NPIX * 4) {
                    f32x4 v = rp[k];
                    if (p.in_scale) {
                        const int pix = idx >> 2;
                        const int py = pix / PW, px = pix - py * PW;
                        const int iy = ty0 - DIL + py, ix = tx0 - DIL + px;
                        const bool inb = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                        v = v * rsc + rsh;
                        if (INRES && inb) v += rr[INRES ? k : 0];       // the arithmetic of otvm_gn_apply: (x a + b) + residual, then act
                        v.x = otvm_act(v.x, p.in_act); v.y = otvm_act(v.y, p.in_act);
                        v.z = otvm_act(v.z, p.in_act); v.w = otvm_act(v.w, p.in_act);
                        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                        v = inb ? v : z;
                    }
                    if (p.in_relu) {
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    }
                    f16x4 hi, lo;
                    const int pix = idx >> 2, c4 = (idx & 3) * 4;
                    if constexpr (M16) {
                        split4p(v, hi, lo);
                        const int o = (((idx >> 1) & 1) * PLANE + pix) * 8 + (idx & 1) * 4;     // [octet][pixel][8 halfs]
                        *reinterpret_cast<f16x4*>(&Ph[o]) = hi;
                        *reinterpret_cast<f16x4*>(&Pl[o]) = lo;
                    } else if constexpr (NPASS == 3) {
                        split4p(v, hi, lo);
                        *reinterpret_cast<f16x4*>(&Ph[pix * LDP + c4]) = hi;
                        *reinterpret_cast<f16x4*>(&Pl[pix * LDP + c4]) = lo;
                    } else {
                        hi = f16x4{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};    // round to nearest
                        *reinterpret_cast<f16x4*>(&Ph[pix * LDP + c4]) = hi;
                    }
                }
            }
        }
    };
    prefetch(0, 0);
    for (int cb = 0; cb < ncb; ++cb) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const _Float16* Bcur = Bs + (GLDS ? ((cb * NG + g) & 1) * B_HALFS : 0);
            if constexpr (GLDS) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's share of the stage's weights has landed
                __syncthreads();                                        // ... everybody's; and the previous stage's LDS reads are done
                if (g == 0) {                                           // a new input patch (every third stage)
                    commit(0);
                    __syncthreads();
                }
            } else {
                __syncthreads();                                        // the previous stage's LDS reads are done
                if (!OTVM_PABL_NOCOMMIT || (cb == 0 && g == 0)) commit(g);
                __syncthreads();
            }
            if (!OTVM_PABL_NOLOAD) {
                if (g + 1 < NG) prefetch(cb, g + 1);
                else if (cb + 1 < ncb) prefetch(cb + 1, 0);
            }
            // ---- TAPG taps out of LDS
            // (measured and rejected, round 2: a "row reuse" variant -- one filter column per weight stage, its 3 x TN weight
            // fragments in registers, every patch row of the column read once and used for all (output row, ky) pairs: a
            // third of the LDS fragment reads per MFMA with 4 rows per wave -- 64->64 at 1088x1920 0.543 vs 0.526 ms,
            // 64->32 0.300 vs 0.304, 320->64 at 544x960 0.591 vs 0.520: the kernel is not bound by LDS fragment reads)
            if constexpr (M16) {
                if (OTVM_PATCH_SETPRIO) __builtin_amdgcn_s_setprio(OTVM_PATCH_SETPRIO);
                const int l15 = lane & 15, oct = (lane >> 4) & 1, hs = lane >> 5;
                const int abase = (oct * PLANE + wave_m * TM * PW + pi16(l15)) * 8;
                const int bbase = (l15 + 32 * oct) * 8;
                auto quad = [](const f32x16& c, int q) __attribute__((always_inline)) { return f32x4{c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]}; };
                auto put = [](f32x16& c, int q, const f32x4 v) __attribute__((always_inline)) { c[4 * q] = v.x; c[4 * q + 1] = v.y; c[4 * q + 2] = v.z; c[4 * q + 3] = v.w; };
                // one K = 32 pass over the wave's TM x TN tiles: 4 TM TN instructions on as many accumulator quads
                auto pass = [&](const f16x8 (&A)[TM][2], const f16x8 (&B)[TN][2]) __attribute__((always_inline)) {
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int b = 0; b < TN; ++b)
#pragma unroll
                            for (int si = 0; si < 2; ++si)
#pragma unroll
                                for (int sj = 0; sj < 2; ++sj)
                                    put(acc[a][b], 2 * si + sj, __builtin_amdgcn_mfma_f32_16x16x32_f16(A[a][si], B[b][sj], quad(acc[a][b], 2 * si + sj), 0, 0, 0));
                };
                constexpr int TAPB = TNW * 2 * 512;                         // halfs of one tap's weight blocks
#pragma unroll
                for (int pr = 0; pr < 4; ++pr) {
                    // pairs (kx 0, ky) + (kx 1, ky) for ky = 0, 1, 2 (taps ky and 3 + ky: with DIL = 1 the A fragment of output row
                    // a, tap row ky is the one of row a + 1, tap row ky - 1 -- read once), then (kx 2, ky 0) + (kx 2, ky 1)
                    // (dilated tiles share nothing and pair (t, t + 1): the kx-pairs cost them 40 registers)
                    constexpr bool KXP = DIL == 1 && !OTVM_PM16_NOKXP;
                    const int t1 = !KXP ? 2 * pr : (pr < 3 ? pr : 6), t2 = !KXP ? 2 * pr + 1 : (pr < 3 ? 3 + pr : 7);
                    // (the lane-dependent part is the same for the three kx-pairs: equal addresses are equal expressions)
                    const int o1 = ((t1 % 3) * DIL * PW + (t1 / 3) * DIL) * 8, o2 = ((t2 % 3) * DIL * PW + (t2 / 3) * DIL) * 8;
                    const int ao = !KXP ? abase + (hs ? o2 : o1)
                                        : (pr < 3 ? abase + hs * (DIL * 8) + pr * DIL * PW * 8 : abase + 2 * DIL * 8 + hs * (DIL * PW * 8));
                    const int bo = bbase + (hs ? t2 : t1) * TAPB;
                    f16x8 ah[TM][2], al[TM][2], bh[TN][2], bl[TN][2];
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int si = 0; si < 2; ++si) {
                            ah[a][si] = *reinterpret_cast<const f16x8*>(&Ph[ao + (a * PW + 16 * si) * 8]);
                            al[a][si] = *reinterpret_cast<const f16x8*>(&Pl[ao + (a * PW + 16 * si) * 8]);
                        }
#pragma unroll
                    for (int b = 0; b < TN; ++b)
#pragma unroll
                        for (int sj = 0; sj < 2; ++sj) {
                            bh[b][sj] = *reinterpret_cast<const f16x8*>(&Bcur[bo + ((nt0 + b) * 2) * 512 + 16 * sj * 8]);
                            bl[b][sj] = *reinterpret_cast<const f16x8*>(&Bcur[bo + ((nt0 + b) * 2 + 1) * 512 + 16 * sj * 8]);
                        }
                    pass(al, bh);
                    pass(ah, bl);
                    pass(ah, bh);
                }
                {   // tap 8: [hi | lo] x [hi | hi], then [hi | lo] x [lo | 0]
                    constexpr int o8 = (2 * DIL * PW + 2 * DIL) * 8;
                    const _Float16* Ahl = hs ? Pl : Ph;
                    f16x8 ax[TM][2], bhh[TN][2], bl0[TN][2];
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int si = 0; si < 2; ++si) ax[a][si] = *reinterpret_cast<const f16x8*>(&Ahl[abase + o8 + (a * PW + 16 * si) * 8]);
                    const f16x8 zero8 = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
#pragma unroll
                    for (int b = 0; b < TN; ++b)
#pragma unroll
                        for (int sj = 0; sj < 2; ++sj) {
                            bhh[b][sj] = *reinterpret_cast<const f16x8*>(&Bcur[bbase + 8 * TAPB + ((nt0 + b) * 2) * 512 + 16 * sj * 8]);
                            const f16x8 l = *reinterpret_cast<const f16x8*>(&Bcur[bbase + 8 * TAPB + ((nt0 + b) * 2 + 1) * 512 + 16 * sj * 8]);
                            bl0[b][sj] = hs ? zero8 : l;
                        }
                    pass(ax, bl0);
                    pass(ax, bhh);
                }
                if (OTVM_PATCH_SETPRIO) __builtin_amdgcn_s_setprio(0);
            } else
#pragma unroll
            for (int tl = 0; tl < TAPG; ++tl) {
                const int tap = g * TAPG + tl;
                const int kx = tap / 3, ky = tap - kx * 3;        // kx-major tap order (weight packing)
                f16x8 ah[TM], al[TM];
#pragma unroll
                for (int a = 0; a < TM; ++a) {
#if OTVM_PABL_NOLDSRD
                    const f16x8 one = {(_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f};
                    ah[a] = one * (_Float16)(float)(lane + tap); al[a] = one;
#else
                    const int o = ((wave_m * TM + a + ky * DIL) * PW + kx * DIL + frow) * LDP + 8 * fh;
                    ah[a] = *reinterpret_cast<const f16x8*>(&Ph[o]);
                    if (NPASS == 3) al[a] = *reinterpret_cast<const f16x8*>(&Pl[o]);
#endif
                }
                // B fragments are read for GB channel tiles at a time, then three passes over the GB x TM accumulators:
                // consecutive MFMAs never share an accumulator (a wide tile has TM = 1: the per-tile order
                // lo*hi, hi*lo, hi*hi was a chain of three dependent MFMAs)
                constexpr int GB = TN < 4 ? TN : 4;
#pragma unroll
                for (int b0 = 0; b0 < TN; b0 += GB) {
                    f16x8 bh[GB], bl[GB];
#pragma unroll
                    for (int j = 0; j < GB; ++j) {
#if OTVM_PABL_NOLDSRD
                        const f16x8 one = {(_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f};
                        bh[j] = one; bl[j] = one * (_Float16)(float)lane;
#else
                        bh[j] = *reinterpret_cast<const f16x8*>(&Bcur[((tl * TNW + nt0 + b0 + j) * 2) * 512 + lane * 8]);
                        if (NPASS == 3) bl[j] = *reinterpret_cast<const f16x8*>(&Bcur[((tl * TNW + nt0 + b0 + j) * 2 + 1) * 512 + lane * 8]);
#endif
                    }
#if OTVM_PABL_NOMFMA
                    if (ah[0][0] == (_Float16)12345.f) acc[0][0][0] += (float)bh[0][0] + (float)bl[0][0] + (float)al[0][0];
                    continue;
#endif
                    if constexpr (NPASS == 3) {
#pragma unroll
                        for (int j = 0; j < GB; ++j)
#pragma unroll
                            for (int a = 0; a < TM; ++a)
                                acc[a][b0 + j] = PATCH_MFMA32(al[a], bh[j], acc[a][b0 + j]);
#pragma unroll
                        for (int j = 0; j < GB; ++j)
#pragma unroll
                            for (int a = 0; a < TM; ++a)
                                acc[a][b0 + j] = PATCH_MFMA32(ah[a], bl[j], acc[a][b0 + j]);
                    }
#pragma unroll
                    for (int j = 0; j < GB; ++j)
#pragma unroll
                        for (int a = 0; a < TM; ++a)
                            acc[a][b0 + j] = PATCH_MFMA32(ah[a], bh[j], acc[a][b0 + j]);
                }
            }
        }
    }

    // ---- epilogue: accumulator tile -> wave-private LDS patch -> 16-byte row-major stores (see conv_f16x3.hip)
    const int col = lane & 31, rbase = (lane >> 5) * 4;
    // where accumulator register e of a 32 x 32 tile sits inside the tile: pixel (of the 32-pixel row), channel
    const int pbase16 = pi16(4 * (lane >> 4));                     // (pi16(4 g + j) = pi16(4 g) + (j & 1) + 4 (j >> 1))
    auto acc_row = [&](int e) __attribute__((always_inline)) -> int {
        return M16 ? 16 * (e >> 3) + pbase16 + (e & 1) + 4 * ((e >> 1) & 1) : (e & 3) + 8 * (e >> 2) + rbase;
    };
    auto acc_col = [&](int e) __attribute__((always_inline)) -> int { return M16 ? 16 * ((e >> 2) & 1) + (lane & 15) : col; };
    __syncthreads();
#if OTVM_PABL_NOEPI
    if (acc[0][0][0] == 12345.678f) p.out[0] = acc[0][0][1];
    if (acc[0][0][0] != 12345.678f) return;
#endif
    if constexpr (HEAD) {
        // ---- round 4: 16 output channels (a half-empty 32-wide tile) followed by a per-pixel head (FBA/models.py:383-388,
        // 425-432: conv_up4.2 -> conv_up4.4 -> fba_fusion, pred.2 -> pred.4 -> fba_fusion / softmax).  As two launches the hidden
        // state made a round trip through HBM (134 MB written, 134 MB + the image read back by a pass of its own: 117-170 us);
        // here a wave turns its TM = 2 accumulator tiles into [pixel][channel] rows in LDS and every lane takes ONE pixel --
        // lanes 0-31 the wave's first image row, lanes 32-63 the second: filter scale, bias, activation (the arithmetic of the
        // plain epilogue), the hidden state's 64 bytes (optional), then the head on the 16 values in registers.
        static_assert(TM == 2 && TN == 1 && NWN == 1, "the head epilogue is written for two pixel rows and one channel tile per wave");
        float* patch = reinterpret_cast<float*>(smem) + wave * (2 * 32 * 36);
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) patch[a * (32 * 36) + ((e & 3) + 8 * (e >> 2) + rbase) * 36 + col] = acc[a][0][e];
        const int a = lane >> 5, px = lane & 31;
        const int y = ty0 + wave_m * TM + a, x = tx0 + px;
        float h[16];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f32x4 v = *reinterpret_cast<const f32x4*>(&patch[a * (32 * 36) + px * 36 + 4 * k]);
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
        if (y < p.H && x < p.W) {
            const int64_t m = (int64_t)y * p.W + x;
            if (p.out) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    *reinterpret_cast<f32x4*>(p.out + m * p.out_ld + 4 * k) = f32x4{h[4 * k], h[4 * k + 1], h[4 * k + 2], h[4 * k + 3]};
            }
            otvm_head_pixel(h, p.head, m);
        }
        return;
    }
    const bool vec_ok = ((p.out_ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.out) & 15) == 0) &&
                        (!p.residual || (((p.res_ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.residual) & 15) == 0)));
    // interior blocks (every row, column and channel of the block inside the output) take a copy of the epilogue without
    // per-row predicates: a store inside a divergent branch is preceded by s_waitcnt vmcnt(0), i.e. waits for the previous
    // store's acknowledgement (stores count in vmcnt on gfx9) -- see conv_f16x3.hip
    const bool interior = vec_ok && ty0 + TH <= p.H && tx0 + 32 <= p.W && n0 + BN <= p.Cout;      // workgroup-uniform
    // interior path, specialised on activation and residual (no uniform branches inside the tile loop), scale / bias of
    // all column tiles fetched up front, the residual of tile t + 1 requested before the stores of tile t: vmcnt retires
    // in order, so nothing here ever waits for a store (conv_f16x3.hip has the long version of this comment)
    auto epilogue_full = [&](auto act_c, auto res_c) __attribute__((always_inline)) {
        constexpr int ACT = decltype(act_c)::value;
        constexpr bool RES = decltype(res_c)::value;
        float* patch = reinterpret_cast<float*>(smem) + wave * (32 * 36);
        const int prow = lane >> 3, pc = (lane & 7) * 4;
        f32x4 sc4[TN], bi4[TN];
#pragma unroll
        for (int b = 0; b < TN; ++b) {
#pragma unroll
            for (int j = 0; j < 4; ++j) sc4[b][j] = p.wscale[n0 + (nt0 + b) * 32 + pc + j];
            bi4[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (p.bias) {
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j) bi4[b][j] = p.bias[n0 + (nt0 + b) * 32 + pc + j];
        }
        auto load_res = [&](int t, f32x4 (&r)[4]) __attribute__((always_inline)) {
            const int b = t / TM, a = t - b * TM;
            const int64_t m0r = (int64_t)(ty0 + wave_m * TM + a) * p.W + tx0 + prow;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                r[r4] = *reinterpret_cast<const f32x4*>(p.residual + (m0r + r4 * 8) * p.res_ld + n0 + (nt0 + b) * 32 + pc);
        };
        f32x4 rnext[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if (RES) load_res(0, rnext);
#pragma unroll
        for (int t = 0; t < TM * TN; ++t) {
            const int b = t / TM, a = t - b * TM;
            const int64_t m0r = (int64_t)(ty0 + wave_m * TM + a) * p.W + tx0 + prow;
            f32x4 rres[4];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) rres[r4] = rnext[r4];
#pragma unroll
            for (int e = 0; e < 16; ++e) patch[acc_row(e) * 36 + acc_col(e)] = acc[a][b][e];
            if (RES && t + 1 < TM * TN) load_res(t + 1, rnext);
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                f32x4 v = *reinterpret_cast<const f32x4*>(&patch[(r4 * 8 + prow) * 36 + pc]);
                v = v * sc4[b] + bi4[b];
                if (RES) v += rres[r4];
                v.x = otvm_act(v.x, ACT); v.y = otvm_act(v.y, ACT); v.z = otvm_act(v.z, ACT); v.w = otvm_act(v.w, ACT);
                *reinterpret_cast<f32x4*>(p.out + (m0r + r4 * 8) * p.out_ld + n0 + (nt0 + b) * 32 + pc) = v;
            }
        }
    };
    auto epilogue = [&](auto full_c) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_c)::value;
        float* patch = reinterpret_cast<float*>(smem) + wave * (32 * 36);
        const int prow = lane >> 3, pc = (lane & 7) * 4;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int n4 = n0 + (nt0 + b) * 32 + pc;
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
                const int y = ty0 + wave_m * TM + a;
                // residual: all four 16-byte loads of this tile are issued before anything waits on them
                f32x4 rres[4];
                const bool res_vec = p.residual && vec_ok && (FULL || n4 + 3 < p.Cout);
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int x = tx0 + r4 * 8 + prow;
                    rres[r4] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (FULL ? (p.residual != nullptr) : (res_vec && y < p.H && x < p.W))
                        rres[r4] = *reinterpret_cast<const f32x4*>(p.residual + ((int64_t)y * p.W + x) * p.res_ld + n4);
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) patch[acc_row(e) * 36 + acc_col(e)] = acc[a][b][e];
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int xi = r4 * 8 + prow;
                    const int x = tx0 + xi;
                    f32x4 v = *reinterpret_cast<const f32x4*>(&patch[xi * 36 + pc]);
                    v = v * sc4 + bi4;
                    if (FULL || (y < p.H && x < p.W)) {
                        const int64_t m = (int64_t)y * p.W + x;
                        if (FULL || (vec_ok && n4 + 3 < p.Cout)) {
                            v += rres[r4];
                            v.x = otvm_act(v.x, p.act); v.y = otvm_act(v.y, p.act);
                            v.z = otvm_act(v.z, p.act); v.w = otvm_act(v.w, p.act);
                            *reinterpret_cast<f32x4*>(p.out + m * p.out_ld + n4) = v;
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                if (n4 + j < p.Cout) {
                                    float xv = v[j];
                                    if (p.residual) xv += p.residual[m * p.res_ld + n4 + j];
                                    p.out[m * p.out_ld + n4 + j] = otvm_act(xv, p.act);
                                }
                            }
                        }
                    }
                }
            }
        }
    };
    if (interior) {
        using std::integral_constant;
        const bool r = p.residual != nullptr;
        if (p.act == OTVM_ACT_RELU) { if (r) epilogue_full(integral_constant<int, OTVM_ACT_RELU>{}, std::true_type{}); else epilogue_full(integral_constant<int, OTVM_ACT_RELU>{}, std::false_type{}); }
        else if (p.act == OTVM_ACT_LEAKY) { if (r) epilogue_full(integral_constant<int, OTVM_ACT_LEAKY>{}, std::true_type{}); else epilogue_full(integral_constant<int, OTVM_ACT_LEAKY>{}, std::false_type{}); }
        else { if (r) epilogue_full(integral_constant<int, OTVM_ACT_NONE>{}, std::true_type{}); else epilogue_full(integral_constant<int, OTVM_ACT_NONE>{}, std::false_type{}); }
    } else {
        epilogue(std::false_type{});
    }

    // ---- fused GroupNorm statistics (sum / sum of squares per group, fp64 atomics)
    if (p.gn_stats) {
        __shared__ double gred[2 * BN];
        const int cg = p.Cout >> 5;
        const int seg = cg < 32 ? cg : 32;
        for (int i = tid; i < 2 * BN; i += NT) gred[i] = 0.0;
        __syncthreads();
        if constexpr (M16) {
            // a lane owns two channels of a 32-channel tile (sj = 0 / 1) and eight pixels of each of the wave's rows; the lanes
            // that share a channel are lane ^ 16, lane ^ 32, lane ^ 48
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int sj = 0; sj < 2; ++sj) {
                    const int nl = (nt0 + b) * 32 + 16 * sj + (lane & 15);
                    const int n = n0 + nl;
                    float s = 0.f, ss = 0.f;
                    if (n < p.Cout) {
                        const float sc_ = p.wscale[n];
                        const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
                        for (int a = 0; a < TM; ++a) {
                            const int y = ty0 + wave_m * TM + a;
#pragma unroll
                            for (int si = 0; si < 2; ++si)
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    const int e = 4 * (2 * si + sj) + j;
                                    const int x = tx0 + acc_row(e);
                                    if (y < p.H && x < p.W) {
                                        const float v = acc[a][b][e] * sc_ + bias;
                                        s += v;
                                        ss += v * v;
                                    }
                                }
                        }
                    }
                    s += __shfl_xor(s, 16); ss += __shfl_xor(ss, 16);
                    s += __shfl_xor(s, 32); ss += __shfl_xor(ss, 32);
                    const int seg16 = seg < 16 ? seg : 16;
                    for (int off = 1; off < seg16; off <<= 1) {
                        s += __shfl_xor(s, off);
                        ss += __shfl_xor(ss, off);
                    }
                    // (a group of 32 channels: its two 16-channel halves arrive as two atomics)
                    if (lane < 16 && (lane & (seg16 - 1)) == 0 && n < p.Cout) {
                        const int gl = nl / cg;
                        atomicAdd(&gred[2 * gl], (double)s);
                        atomicAdd(&gred[2 * gl + 1], (double)ss);
                    }
                }
        } else
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int nl = (nt0 + b) * 32 + col;
            const int n = n0 + nl;
            float s = 0.f, ss = 0.f;
            if (n < p.Cout) {
                const float sc_ = p.wscale[n];
                const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    const int y = ty0 + wave_m * TM + a;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int x = tx0 + (e & 3) + 8 * (e >> 2) + rbase;
                        if (y < p.H && x < p.W) {
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
                const int gl = nl / cg;
                atomicAdd(&gred[2 * gl], (double)s);
                atomicAdd(&gred[2 * gl + 1], (double)ss);
            }
        }
        __syncthreads();
        const int ng = (BN + cg - 1) / cg;
        for (int i = tid; i < 2 * ng; i += NT) {
            const int g = n0 / cg + (i >> 1);
            if (g < 32 && gred[i] != 0.0) atomicAdd(&p.gn_stats[2 * g + (i & 1)], gred[i]);
        }
        __syncthreads();
        otvm_gn_table_tail(p.gn_stats, (int64_t)p.H * p.W, p.Cout, p.tail, blockIdx.y, gridDim.x, reinterpret_cast<float*>(gred));
    }
}

// packed fp32 weight [O_pad][K_pad] (k = tap*I_pad + c) -> fragment-major split fp16 for the patch kernel
__global__ __launch_bounds__(256) void pack_patch_weight_kernel(const float* __restrict__ w, int O, int K_pad, int I_pad, int n_pad32,
                                                                _Float16* __restrict__ wf, float* __restrict__ wscale) {
    // one block per output filter row n: compute its power-of-two scale, then scatter its 9*I_pad values
    const int n = blockIdx.x;
    __shared__ float red[256];
    float mx = 0.f;
    const float* row = w + (int64_t)n * K_pad;
    const bool real = n < O;
    if (real)
        for (int k = threadIdx.x; k < 9 * I_pad; k += 256) mx = fmaxf(mx, fabsf(row[k]));
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
    const int nb = n >> 5, nl = n & 31, nbs = n_pad32 >> 5;
    for (int k = threadIdx.x; k < 9 * I_pad; k += 256) {
        const int tap0 = k / I_pad, c = k - tap0 * I_pad;     // packed fp32 weight: tap0 = ky*3 + kx
        const int tap = (tap0 % 3) * 3 + tap0 / 3;            // fragment-major copy: kx-major (see the kernel)
        const float v = real ? row[k] * inv : 0.f;
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        const int cb = c >> 5, ks = (c >> 4) & 1, h = (c >> 3) & 1, j = c & 7;
        const int l = nl + 32 * h;
        const int64_t blk = ((((int64_t)cb * 9 + tap) * nbs + nb) * 2 + ks) * 2;
        wf[(blk + 0) * 512 + l * 8 + j] = hi;
        wf[(blk + 1) * 512 + l * 8 + j] = lo;
    }
}

template <int NPASS, int TH, int BN, int NW, int DIL, int TAPG = 9, bool INRES = false, bool HEAD = false, int NWN = 1, bool GLDS = false,
          bool M16 = false>
int launch_patch(PatchArgs& a, hipStream_t s) {
    a.tiles_x = otvm_ceil_div(a.W, 32);
    a.tiles_y = otvm_ceil_div(a.H, TH);
    a.tiles_n = otvm_ceil_div(a.Cout, BN);
    a.walk = otvm_tile_walk_of(1);
    hipLaunchKernelGGL((conv_patch_f16x3_kernel<TH, BN, NW, DIL, TAPG, INRES, HEAD, NWN, GLDS, NPASS, M16>), dim3(a.tiles_x * a.tiles_y * a.tiles_n, a.batch), dim3(NW * 64), 0, s, a);
    OTVM_CHECK_LAUNCH("otvm_conv2d(patch f16x3)");
    return 0;
}

}  // namespace

extern "C" int64_t otvm_patch_weight_bytes_f16x3(int O, int I_pad) {
    const int64_t n_pad32 = (O + 31) / 32 * 32;
    return (int64_t)((I_pad + 31) / 32) * 9 * n_pad32 * 32 * 2 * sizeof(_Float16);
}

extern "C" int otvm_pack_patch_weight_f16x3(const float* w_packed, int O, int K_pad, int I_pad, void* w_frag, float* w_scale,
                                            void* stream) {
    OTVM_REQUIRE(w_packed && w_frag && w_scale && I_pad % 16 == 0, "otvm_pack_patch_weight_f16x3: needs I_pad %% 16 == 0");
    const int n_pad32 = (O + 31) / 32 * 32;
    hipLaunchKernelGGL(pack_patch_weight_kernel, dim3(n_pad32), dim3(256), 0, (hipStream_t)stream, w_packed, O, K_pad, I_pad, n_pad32,
                       (_Float16*)w_frag, w_scale);
    OTVM_CHECK_LAUNCH("otvm_pack_patch_weight_f16x3");
    return 0;
}

// 0: not eligible (the implicit-GEMM kernel takes the layer), 1: narrow tiles, 2: wide (256-channel) tiles.
// forced: shape-wise eligibility only (the plan-time autotuner decides whether the wide tiles pay on this map)
static int patch_choice(const otvm_conv_params* p, bool forced = false) {
    if (!p->w_frag || p->kh != 3 || p->kw != 3 || p->stride != 1 || p->pad != p->dil || p->Cin % 16 != 0) return 0;
    if (p->dil != 1 && p->dil != 2 && p->dil != 4) return 0;
    static const int wide = otvm_probe_int("OTVM_PATCH_WIDE", 1);
    if (p->Cout <= 64) return 1;
    if (!wide || p->Cout % 256 != 0) return 0;                     // wide path: 256-channel tiles, 8 waves, 3-tap weight stages
    if (forced) return 2;
    // needs enough 8x32 x 256-channel tiles to fill 256 CUs (OS4 maps at 1080p): 327 vs 285 TFLOP/s (256->256) and
    // 390 vs 348 (512->256) against the implicit-GEMM kernel.  On smaller maps the implicit-GEMM tiles win
    // (a 4x32-tile variant of this kernel measured 160-230 TFLOP/s vs 290-315 and was dropped).
    static const int t_patch = otvm_probe_int("OTVM_T_PATCH", 400);
    const int64_t t8 = (int64_t)otvm_ceil_div(p->H, 8) * otvm_ceil_div(p->W, 32) * (p->Cout / 256);
    return t8 >= t_patch ? 2 : 0;
}

// the fused input normalisation (otvm_conv_params.in_scale): this kernel, or the whole-chunk implicit-GEMM kernels
int otvm_conv2d_igemm_accepts_input_norm(const otvm_conv_params* p);       // conv_f16x3.hip
extern "C" int otvm_conv2d_accepts_input_norm(const otvm_conv_params* p) {
    return p && otvm_prec_is_split(p->precision) && (patch_choice(p) != 0 || otvm_conv2d_igemm_accepts_input_norm(p)) ? 1 : 0;
}
// which kernel family would take it: 0 none, 1 the patch kernel (the heuristic's choice for this layer), 2 an implicit-GEMM tile.
// A host may prefer a separate apply pass over 2 for a KxK layer: the implicit GEMM stages (and normalises) every input
// element once per tap -- on the 3x3 512-channel layers that cost 40 us per launch against 20 for the pass it replaced.
extern "C" int otvm_conv2d_input_norm_kind(const otvm_conv_params* p) {
    if (!p || !otvm_prec_is_split(p->precision)) return 0;
    if (patch_choice(p) != 0) return 1;
    return otvm_conv2d_igemm_accepts_input_norm(p) ? 2 : 0;
}

// ABI 17: in_res (the identity added inside the fused input normalisation) exists on the narrow dilation-1 patch tiles
extern "C" int otvm_conv2d_accepts_input_residual(const otvm_conv_params* p) {
    return p && otvm_prec_is_split(p->precision) && patch_choice(p) == 1 && p->dil == 1 && p->in_scale && !p->in_relu ? 1 : 0;
}

int otvm_conv2d_patch_eligible(const otvm_conv_params* p) {
    if (p->in_res && !otvm_conv2d_accepts_input_residual(p)) return 0;
    return patch_choice(p, true) != 0 ? 1 : 0;
}

static int patch_run(const otvm_conv_params* p, void* stream, int choice, const otvm_head_params* hd = nullptr);
int otvm_conv2d_head16_impl(const otvm_conv_params* p, const otvm_head_params* hd, void* stream);   // conv_head16_f16x3.hip

// ABI 17: 3x3 conv to 16 channels with the FBA head in its epilogue (include/otvm_hip.h)
extern "C" int otvm_conv2d_head(const otvm_conv_params* p, const otvm_head_params* hd, void* stream) {
    OTVM_REQUIRE(p && hd && p->in && hd->w && hd->b && hd->img, "otvm_conv2d_head: null pointer");
    OTVM_REQUIRE(otvm_prec_is_split(p->precision) && p->Cout == 16 && patch_choice(p) == 1 && p->dil == 1 && !p->residual &&
                     !p->gn_stats && !p->in_res && p->w_scale,
                 "otvm_conv2d_head: a 3x3 stride-1 f16x3 layer with 16 output channels, no residual / statistics");
    OTVM_REQUIRE(hd->n_out == 7 || (hd->n_out == 10 && hd->tri_out), "otvm_conv2d_head: n_out must be 7, or 10 with tri_out");
    OTVM_REQUIRE(!p->out || ((p->out_ld & 3) == 0 && ((uintptr_t)p->out & 15) == 0), "otvm_conv2d_head: out must be 16-byte aligned");
    if (hd->w16) return otvm_conv2d_head16_impl(p, hd, stream);    // 32 -> 16 on the 16-wide matrix-core tile
    return patch_run(p, stream, 1, hd);
}

int otvm_conv2d_patch_f16x3_forced(const otvm_conv_params* p, void* stream) { return patch_run(p, stream, patch_choice(p, true)); }

// returns -1 when the layer is not eligible (caller falls back to the implicit-GEMM kernel)
int otvm_conv2d_patch_f16x3_impl(const otvm_conv_params* p, void* stream) { return patch_run(p, stream, patch_choice(p)); }

// round 5: the nine-tap 64-filter tiles on v_mfma_f32_16x16x32_f16 (the kernel's M16 comment); OTVM_PATCH_M16=0: the 32x32x16 form
// (A/B runs).  One box, alternating (profiles/r05_patch_mfma16_ab.txt): 1080p 46.17 vs 45.54 frames/s, 832x480 151.6 vs 150.1
static int patch_m16() {
    static const int m = otvm_probe_int("OTVM_PATCH_M16", 1);
    return m;
}

template <int NPASS>
static int patch_run_t(const otvm_conv_params* p, void* stream, int choice, const otvm_head_params* hd) {
    if (choice == 0) return -1;
    const bool is_wide = choice == 2;
    PatchArgs a;
    a.in = p->in; a.wf = (const _Float16*)p->w_frag; a.wscale = p->w_scale; a.bias = p->bias; a.residual = p->residual;
    a.out = p->out; a.gn_stats = p->gn_stats;
    a.H = p->H; a.W = p->W; a.Cin = p->Cin; a.in_ld = p->in_ld; a.res_ld = p->res_ld; a.Cout = p->Cout; a.out_ld = p->out_ld;
    a.n_pad32 = (p->Cout + 31) / 32 * 32; a.in_relu = p->in_relu; a.act = p->act;
    hipStream_t s = (hipStream_t)stream;
    a.in_scale = p->in_scale; a.in_shift = p->in_shift; a.in_act = p->in_act;
    a.tail = otvm_gn_tail_of(p);
    a.batch = p->batch > 1 ? p->batch : 1;
    a.in_bs = a.batch > 1 ? p->in_bs : 0; a.out_bs = a.batch > 1 ? p->out_bs : 0; a.res_bs = a.batch > 1 ? p->res_bs : 0;
    a.gn_bs = a.batch > 1 ? p->gn_bs : 0; a.norm_bs = a.batch > 1 ? p->norm_bs : 0;
    a.in_res = p->in_res; a.in_res_ld = p->in_res_ld; a.in_res_bs = a.batch > 1 ? p->in_res_bs : 0;
    {   // buffer-resource ranges (32-bit byte offsets; 0xFFFFFFFF marks a pixel outside the image)
        const int64_t ib = (int64_t)p->H * p->W * p->in_ld * 4, rb = p->in_res ? (int64_t)p->H * p->W * p->in_res_ld * 4 : 0;
        OTVM_REQUIRE(ib < 0xFFFFFFF0ll && rb < 0xFFFFFFF0ll, "otvm_conv2d(patch f16x3): input view of %lld bytes is beyond 32-bit offsets", (long long)ib);
        a.in_bytes = (unsigned)ib; a.in_res_bytes = (unsigned)rb;
    }
    if (hd) {
        a.head.w = hd->w; a.head.b = hd->b; a.head.n_out = hd->n_out; a.head.img = hd->img; a.head.img_ld = hd->img_ld;
        a.head.P = hd->P; a.head.alpha_out = hd->alpha_out; a.head.alpha_stride = hd->alpha_stride; a.head.tri_out = hd->tri_out;
        a.head.sm = hd->sm; a.head.sm_ld = hd->sm_ld; a.head.out7 = nullptr; a.head.logits_out = nullptr;
        a.head_img_bs = a.batch > 1 ? hd->img_bs : 0; a.head_alpha_bs = a.batch > 1 ? hd->alpha_bs : 0;
        a.head_tri_bs = a.batch > 1 ? hd->tri_bs : 0; a.head_sm_bs = a.batch > 1 ? hd->sm_bs : 0;
        return launch_patch<3, 8, 32, 4, 1, 3, false, true>(a, s);      // (the head epilogue: f16x3 in either mode)
    }
    if (p->in_res) {
        OTVM_REQUIRE(otvm_conv2d_accepts_input_residual(p) && !is_wide && (p->in_res_ld & 3) == 0 && ((uintptr_t)p->in_res & 15) == 0,
                     "otvm_conv2d: in_res needs in_scale on a 3x3 stride-1 dilation-1 layer with <= 64 output channels");
        if (p->Cout <= 32) return launch_patch<3, 8, 32, 4, 1, 3, true>(a, s);
        // (the matrix-core form of the plain 64-filter tile: the same products in the same order as the layer without in_res)
        return patch_m16() ? launch_patch<3, 8, 64, 4, 1, 9, true, false, 1, false, true>(a, s) : launch_patch<3, 8, 64, 4, 1, 9, true>(a, s);
    }
    if (is_wide) {
        // the eight waves as 4 x 2 (two rows x four channel tiles each) instead of 8 x 1 (one row x eight tiles): a third less
        // LDS fragment traffic per MFMA; OTVM_PATCH_WIDE_NWN=1 keeps the round-2 arrangement (same-box A/B); 2 x 4 waves (four
        // rows x two tiles) spills 108 B
        static const int nwn = otvm_probe_int("OTVM_PATCH_WIDE_NWN", 2);
        static const int glds = otvm_probe_int("OTVM_PATCH_WIDE_GLDS", 1);
        if (nwn == 2 && p->dil == 1 && glds) return launch_patch<NPASS, 8, 256, 8, 1, 3, false, false, 2, true>(a, s);
        if (nwn == 2 && p->dil == 1) return launch_patch<NPASS, 8, 256, 8, 1, 3, false, false, 2>(a, s);   // (dilated: 52 / 92 B of scratch)
        if (p->dil == 1) return launch_patch<NPASS, 8, 256, 8, 1, 3>(a, s);
        if (p->dil == 2) return launch_patch<NPASS, 8, 256, 8, 2, 3>(a, s);
        return launch_patch<NPASS, 8, 256, 8, 4, 3>(a, s);
    }
    static const int th16 = otvm_probe_int("OTVM_PATCH_TH16", 0);
    if (p->dil == 1 && th16 && (int64_t)p->H * p->W >= (1 << 18))      // 16x32 pixel blocks, 8 waves: half the weight stream per pixel
        return p->Cout <= 32 ? launch_patch<NPASS, 16, 32, 8, 1>(a, s) : launch_patch<NPASS, 16, 64, 8, 1>(a, s);
    // (measured and rejected, round 2: 16x32-pixel blocks with 8 waves -- half the weight stream and less halo per pixel --
    // 64->64 at 1088x1920 0.604 vs 0.604 ms, 64->32 0.317 vs 0.306, 32->16 0.201 vs 0.187: the weight stream is not the limit)
    // <= 32 output channels: 3-tap weight stages (46 KB of LDS instead of 58) and 158 registers -> three workgroups per CU
    // instead of two: 64->32 at 1088x1920 0.316 -> 0.302 ms, 32->16 0.187 -> 0.171.  (64 output channels the same way:
    // spills at 168 registers, 0.591 -> 0.648 ms; 16x32-pixel blocks with 4 waves x 4 rows and 3-tap stages -- half the
    // fragment reads per MFMA, but 341 registers = one wave per SIMD -- 0.611 -> 0.774 ms.  A persistent form -- resident
    // workgroups walking several tiles, the next tile's first stage requested under the last MFMAs of the current one --
    // is worth 4 % on its own code (0.710 -> 0.680 ms) but keeps 60 prefetch registers live across the epilogue: 47 spilled
    // dwords at two waves per SIMD, against 0.591 ms for this one-tile-per-workgroup kernel.)
    // round 4, <= 32 output channels on maps of >= 2^20 pixels: 16 x 32 pixel blocks, four waves x four rows -- 18 instead of 28
    // LDS fragment reads per 36 MFMAs (a wave's six patch rows serve its four output rows and three taps of a filter column;
    // the compiler already shares equal fragment loads between taps): 80->32 at 1088x1920 0.429 -> 0.415 ms, 64->32 0.310 ->
    // 0.301; no gain at 480x832 (two workgroups per CU instead of three).  Two waves x four rows on 8 x 32 blocks: 0.447 / 0.320.
    static const int rows4 = otvm_probe_int("OTVM_PATCH32_ROWS4", 1);
    static const int glds32 = otvm_probe_int("OTVM_PATCH32_GLDS", 0);   // (LDS-DMA weight stages: neutral, see below)
    if (p->dil == 1 && p->Cout <= 32 && rows4 && (int64_t)p->H * p->W >= (1 << 20))
        return glds32 ? launch_patch<NPASS, 16, 32, 4, 1, 3, false, false, 1, true>(a, s) : launch_patch<NPASS, 16, 32, 4, 1, 3>(a, s);
    // round 5: 64 output channels with 3-tap weight stages copied by LDS-DMA into alternating buffers (as the 256-channel tiles):
    // the nine-tap stage moved 36 KiB of weights per 16 input channels through registers (9 x 16 bytes per lane + 9 ds_write_b128)
    // Measured neutral (profiles/r05_patch_narrow_glds_ab.txt: 64->64 at 1088x1920 0.548 vs 0.550 ms, 320->64 0.516 vs 0.507, 64->32
    // 0.272 vs 0.275; whole frame 45.91 vs 45.92 frames/s): the tenth variant this tile does not respond to.  Off by default.
    static const int glds64 = otvm_probe_int("OTVM_PATCH64_GLDS", 0);
    if (p->dil == 1 && p->Cout > 32 && glds64) return launch_patch<NPASS, 8, 64, 4, 1, 3, false, false, 1, true>(a, s);   // (both channel tiles have weights)
    if (p->dil == 1 && p->Cout <= 32 && glds32) return launch_patch<NPASS, 8, 32, 4, 1, 3, false, false, 1, true>(a, s);
    if constexpr (NPASS == 3) {
        if (patch_m16() && p->Cout > 32) {
            if (p->dil == 1) return launch_patch<3, 8, 64, 4, 1, 9, false, false, 1, false, true>(a, s);
            if (p->dil == 2) return launch_patch<3, 8, 64, 4, 2, 9, false, false, 1, false, true>(a, s);
            return launch_patch<3, 8, 64, 4, 4, 9, false, false, 1, false, true>(a, s);
        }
    }
    if (p->dil == 1) return p->Cout <= 32 ? launch_patch<NPASS, 8, 32, 4, 1, 3>(a, s) : launch_patch<NPASS, 8, 64, 4, 1>(a, s);
    if (p->dil == 2) return launch_patch<NPASS, 8, 64, 4, 2>(a, s);
    return launch_patch<NPASS, 8, 64, 4, 4>(a, s);
}

static int patch_run(const otvm_conv_params* p, void* stream, int choice, const otvm_head_params* hd) {
    return p->precision == OTVM_PREC_F16 ? patch_run_t<1>(p, stream, choice, hd) : patch_run_t<3>(p, stream, choice, hd);
}
