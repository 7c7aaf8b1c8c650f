// Shared helpers for the OTVM gfx950 kernels (internal; the public ABI is include/otvm_hip.h).
#pragma once
#include <type_traits>
#include <utility>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../include/otvm_hip.h"

void otvm_set_error(const char* fmt, ...);

#define OTVM_CHECK_LAUNCH(name)                                                        \
    do {                                                                               \
        hipError_t e__ = hipGetLastError();                                            \
        if (e__ != hipSuccess) {                                                       \
            otvm_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
            return 2;                                                                  \
        }                                                                              \
    } while (0)

#define OTVM_REQUIRE(cond, ...)                                                        \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            otvm_set_error(__VA_ARGS__);                                               \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

static inline int otvm_ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
// Ablation switches (tile thresholds, kernel-variant selectors, tile walks: OTVM_* names).  The shipping build compiles every one
// of them to its default -- the library reads NO environment variable; `python otvm_amd/csrc/build.py --probes` builds
// libotvm_hip_probes.so with -DOTVM_PROBES, where each is read from the environment once per process (A/B runs and the two tests
// that compare kernel forms, selected with OTVM_HIP_LIB).
#ifdef OTVM_PROBES
static inline int64_t otvm_probe_int(const char* name, int64_t dflt) {
    const char* v = getenv(name);
    return v ? atoll(v) : dflt;
}
#else
static constexpr inline int64_t otvm_probe_int(const char*, int64_t dflt) { return dflt; }
#endif
// f16x3 and its single-pass sibling "f16" share kernels, weight formats and dispatch; they differ in the MFMA passes per product
static inline bool otvm_prec_is_split(int precision) { return precision == OTVM_PREC_F16X3 || precision == OTVM_PREC_F16; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE setting: done once per (kernel, device), not once per
// process (a process driving engines on several GPUs).  `done`: one zero-initialised flag array per kernel.
#include <atomic>
constexpr int OTVM_MAX_DEVICES = 64;
template <class K>
static inline hipError_t otvm_reserve_lds_once(std::atomic<bool> (&done)[OTVM_MAX_DEVICES], K kernel, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const bool track = dev >= 0 && dev < OTVM_MAX_DEVICES;
    if (track && done[dev].load(std::memory_order_acquire)) return hipSuccess;
    e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && track) done[dev].store(true, std::memory_order_release);
    return e;
}

// ---- tile walk of the spatially tiled kernels (patch / head / stem convs, fused bottleneck; round 5).  The hardware deals
// consecutive workgroup ids round-robin to the eight XCDs, each with an L2 of its own, so with a row-major tile order the
// neighbours that share halo columns / rows never meet in one L2 and every shared line is fetched by several XCDs
// (profiles/r05_tile_walk_traffic.md: the full-resolution patch tiles fetched 1.3-1.9 x their input).  walk = 1: XCD x owns a
// contiguous range of a re-ordered tile list -- bands of `band` tile rows, column-major inside a band -- so the 64-96 workgroups
// resident on an XCD cover a compact rectangle of tiles whose halos are fetched once.  walk = 0: row-major (A/B runs).
// Which tile a workgroup computes changes, the arithmetic of a tile does not: results are bit-identical in either walk.
struct OtvmTileWalk { int walk, band; };
// family: 1 patch convs, 2 stems, 4 head conv, 8 fused bottleneck (OTVM_TILE_WALK = bit mask of the families that use walk 1).
// Default 11: whole frame at 1080p 35.86 -> 33.31 GB of conv traffic (1.26 -> 1.17 x algorithmic), +0.2 ... 0.4 % frames/s; the
// head conv keeps the row-major walk -- it fetches 13 % less with walk 1 but runs 1 - 5 % slower alone on the device
static inline OtvmTileWalk otvm_tile_walk_of(int family) {
    static const int mask = otvm_probe_int("OTVM_TILE_WALK", 11);
    static const int band = otvm_probe_int("OTVM_TILE_BAND", 8);
    return OtvmTileWalk{(mask & family) ? 1 : 0, band < 1 ? 1 : band};
}
// workgroup `bid` of `nwg` -> (tile_n, tile_x, tile_y) of a tiles_x x tiles_y map with tiles_n channel tiles per position
// (channel tile fastest: the workgroups of one position run side by side on one XCD and share its patch in L2)
__device__ __forceinline__ void otvm_tile_decode(OtvmTileWalk w, int bid, int nwg, int tiles_n, int tiles_x, int tiles_y, int& tile_n,
                                                 int& tile_x, int& tile_y) {
    if (w.walk == 0) {
        tile_n = bid % tiles_n; bid /= tiles_n;
        tile_x = bid % tiles_x;
        tile_y = bid / tiles_x;
        return;
    }
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);     // XCD x owns a contiguous range of t
    tile_n = t % tiles_n; t /= tiles_n;
    const int per_band = w.band * tiles_x;
    const int bnd = t / per_band, rem = t - bnd * per_band;
    const int y0 = bnd * w.band;
    const int rows = tiles_y - y0 < w.band ? tiles_y - y0 : w.band;
    tile_x = rem / rows;
    tile_y = y0 + (rem - tile_x * rows);
}

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}).  `#pragma unroll` is a
// request the unroller may decline for a large body; an accumulator array indexed by a loop that stayed rolled lives in
// scratch memory (found on the 4x4-tile kernel: 1 KB of scratch, 10x slower)
template <class F, int... I>
__device__ __forceinline__ void otvm_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void otvm_static_for(F&& f) {
    otvm_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// ABI 16: the GroupNorm scale / shift table of a conv's OUTPUT, written by the last workgroup of the launch (see
// otvm_conv_params.gn_counter).  Same arithmetic as gn_table_kernel (groupnorm.hip).
struct OtvmGnTail {
    const float* gamma; const float* beta; float* scale; float* shift; unsigned* counter; int tab_bs;   // scale == nullptr: none
};
static inline OtvmGnTail otvm_gn_tail_of(const otvm_conv_params* p) {
    OtvmGnTail t;
    t.gamma = p->gn_gamma; t.beta = p->gn_beta; t.scale = p->gn_stats ? p->gn_scale_out : nullptr; t.shift = p->gn_shift_out;
    t.counter = p->gn_counter; t.tab_bs = p->batch > 1 ? p->gn_tab_bs : 0;
    return t;
}
// Called by ALL threads of EVERY workgroup of the launch after the workgroup's atomicAdds into `stats` (image zb's block);
// `total` = workgroups per image.  Contains __syncthreads.
// `lds`: 65 floats of LDS nobody else uses any more (the 256x256 tile has no static LDS to spare).
__device__ __forceinline__ void otvm_gn_table_tail(double* stats, int64_t P, int C, const OtvmGnTail& t, int zb, unsigned total,
                                                   float* lds) {
    if (!t.scale) return;                                  // (uniform)
    float* tail_mean = lds;
    float* tail_rstd = lds + 32;
    volatile unsigned* tail_last = reinterpret_cast<volatile unsigned*>(lds + 64);
    // this thread's statistics atomics have been acknowledged (they execute at device scope, past the XCD's L2) before the
    // workgroup takes its ticket.  NOT __threadfence(): a device-scope release on this multi-XCD part writes the XCD's whole
    // L2 back -- every workgroup of every conv did that and the frame rate fell from 42.7 to 35.2
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) *tail_last = atomicAdd(t.counter + zb, 1u) == total - 1 ? 1u : 0u;
    __syncthreads();
    if (!*tail_last) return;
    const int cg = C / 32;
    if (threadIdx.x < 32) {
        const double cnt = (double)P * cg;
        const double sum = atomicAdd(&stats[threadIdx.x * 2], 0.0), sq = atomicAdd(&stats[threadIdx.x * 2 + 1], 0.0);   // coherent reads
        const double mean = sum / cnt;
        double var = sq / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        tail_mean[threadIdx.x] = (float)mean;
        tail_rstd[threadIdx.x] = (float)(1.0 / sqrt(var + 1e-5));
    }
    __syncthreads();
    float* scale = t.scale + (int64_t)zb * t.tab_bs;
    float* shift = t.shift + (int64_t)zb * t.tab_bs;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int g = c / cg;
        const float a = tail_rstd[g] * t.gamma[c];
        scale[c] = a;
        shift[c] = t.beta[c] - tail_mean[g] * a;
    }
    // re-armed for the next launch (an atomic store: the counter is only ever touched by device-scope atomics)
    if (threadIdx.x == 0) __hip_atomic_store(t.counter + zb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ float otvm_act(float v, int act) {
    if (act == OTVM_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == OTVM_ACT_LEAKY) return v > 0.f ? v : 0.01f * v;
    return v;
}
