// 8-channel trimap encoding on device (reference models/alpha/model.py:40-53, utils/utils.py:12-39).
//
// The reference ships every trimap to the host, runs cv2.distanceTransform (exact L2) on one core and
// uploads the result -- 6-8 times per frame.  Here the exact Euclidean distance transform runs on the
// GPU in integer arithmetic (squared distances are exact integers):
//   phase 1 (columns): g[y][x] = distance to the nearest class pixel within the column;
//   phase 2 (rows)   : d2[x] = min_i (x-i)^2 + g[i]^2, exact integer minimum by an outward scan that stops at
//                      dx^2 >= best (one workgroup per row, row staged in LDS);
//   encode           : d = sqrtf(d2) (correctly rounded, like OpenCV's float output), then
//                      exp(-(d*d) / (2 (sigma*320)^2)) for sigma in {0.02, 0.08, 0.16}; an empty class
//                      produces zeros (utils/utils.py:32).
// Classes: k=0 background (argmax == 0), k=1 foreground (argmax == 2).
#include "common.h"
#include <stdlib.h>

namespace {

typedef float edt_f32x4 __attribute__((ext_vector_type(4)));
typedef float edt_f32x2 __attribute__((ext_vector_type(2)));
constexpr int EDT_INF = 1 << 14;     // > any image side handled (asserted on the host side)

// Round 4: the three kernels below were 0.30-0.34 ms of a 1080p frame on its serial chain (STM decoder -> encoding -> FBA
// encoder) for ~0.2 GB of algorithmic traffic (profiles/r03_kernel_traffic_gbps_1080p.md: 0.08-0.23 of HBM speed).  What cost:
//   classify : besides the class map it scattered the two soft channels into x11 (16 of every 48 bytes) and d80 (8 of
//              every 320 bytes) -- 4.2 M partial-sector writes;
//   columns  : 60 workgroups at 1080p, every thread walking its 68 rows three times with one dependent byte load per step;
//   rows     : one workgroup per (class, row) writing 12 of every 48 bytes of x11, twice per pixel.
// Now: classify writes the class map only (quads of pixels per thread); the column pass keeps a thread's rows in registers
// (all loads of a segment in flight at once, both classes from one read, 16-bit distances out); the row pass handles both
// classes of an image row in one workgroup and writes channels 3..11 of x11 and the two soft channels of d80 once.
__global__ __launch_bounds__(256) void classify_kernel(const float* __restrict__ probs, int64_t P,
                                                       const uint8_t* __restrict__ cls_override, uint8_t* __restrict__ cls_out,
                                                       int* __restrict__ flags) {
    int has_bg = 0, has_fg = 0;
    const bool vec = (P & 3) == 0 && ((reinterpret_cast<uintptr_t>(probs) | reinterpret_cast<uintptr_t>(cls_out) |
                                        reinterpret_cast<uintptr_t>(cls_override)) & 15) == 0;
    auto argmax3 = [](float p0, float p1, float p2) {   // tri.max(dim)[1]: first maximal index (alpha/model.py:42)
        int cls = 0;
        float m = p0;
        if (p1 > m) { m = p1; cls = 1; }
        if (p2 > m) { cls = 2; }
        return cls;
    };
    if (vec) {
        const int64_t Q = P >> 2;
        for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < Q; q += (int64_t)gridDim.x * blockDim.x) {
            uchar4 c;
            if (cls_override) {
                c = *reinterpret_cast<const uchar4*>(cls_override + 4 * q);
            } else {
                const edt_f32x4 a = *reinterpret_cast<const edt_f32x4*>(probs + 4 * q);
                const edt_f32x4 b = *reinterpret_cast<const edt_f32x4*>(probs + P + 4 * q);
                const edt_f32x4 d = *reinterpret_cast<const edt_f32x4*>(probs + 2 * P + 4 * q);
                c.x = (uint8_t)argmax3(a.x, b.x, d.x); c.y = (uint8_t)argmax3(a.y, b.y, d.y);
                c.z = (uint8_t)argmax3(a.z, b.z, d.z); c.w = (uint8_t)argmax3(a.w, b.w, d.w);
            }
            *reinterpret_cast<uchar4*>(cls_out + 4 * q) = c;
            has_bg |= (c.x == 0) | (c.y == 0) | (c.z == 0) | (c.w == 0);
            has_fg |= (c.x == 2) | (c.y == 2) | (c.z == 2) | (c.w == 2);
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
            const int cls = cls_override ? (int)cls_override[i] : argmax3(probs[i], probs[P + i], probs[2 * P + i]);
            cls_out[i] = (uint8_t)cls;
            has_bg |= (cls == 0);
            has_fg |= (cls == 2);
        }
    }
    // "class present" flags: plain stores of the same value (idempotent; visible to the next kernel of the stream).  Round 1 used
    // atomicOr from every wave that found the class, round 2 looked at the flag first -- but in a kernel this short every wave
    // looks before the first atomic lands, and 8 k atomics on two words serialised at L2: 80-100 us of a kernel that moves
    // 27 MB (measured with the kernel alone on the device, profiles/r04_glue_probes.txt)
    if (__any(has_bg) && (threadIdx.x & 63) == 0) flags[0] = 1;
    if (__any(has_fg) && (threadIdx.x & 63) == 0) flags[1] = 1;
}

// phase 1: distance to the nearest class pixel within the column, both classes from one read of the class map.  A column
// is cut into EDT_SEGS segments, one thread each (a workgroup = EDT_XB columns x EDT_SEGS segments, lanes along x).  A
// thread loads ALL rows of its segment into registers first (<= LEN_MAX independent byte loads in flight -- the first
// version walked its rows three times with a dependent load per step: 78-108 us at 1080p on 60 workgroups), notes the
// first / last seed of either class, takes the carries (nearest seed above / below the segment) from the other segments'
// entries in LDS, and runs the down- and the up-scan in registers.  g: uint16 [2][H][W] (EDT_INF = "no seed in the column").
constexpr int EDT_SEGS = 64, EDT_XB = 16;

template <int LEN_MAX>
__global__ __launch_bounds__(EDT_XB * EDT_SEGS) void edt_columns_kernel(const uint8_t* __restrict__ cls, int H, int W, int len,
                                                                         uint16_t* __restrict__ g) {
    __shared__ int first_s[2][EDT_SEGS][EDT_XB], last_s[2][EDT_SEGS][EDT_XB];
    const int lx = threadIdx.x % EDT_XB, seg = threadIdx.x / EDT_XB;
    const int x = blockIdx.x * EDT_XB + lx;
    const int y0 = seg * len;
    const int y1 = y0 + len < H ? y0 + len : H;            // (y1 <= y0: this segment lies below the image)
    const bool live = x < W;
    uint8_t c[LEN_MAX];
#pragma unroll
    for (int j = 0; j < LEN_MAX; ++j) {
        const int y = y0 + j;
        c[j] = (live && j < len && y < H) ? cls[(int64_t)y * W + x] : (uint8_t)255;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const uint8_t target = k == 0 ? 0 : 2;
        int first = -1, last = -1;
#pragma unroll
        for (int j = 0; j < LEN_MAX; ++j) {
            if (c[j] == target) {
                if (first < 0) first = y0 + j;
                last = y0 + j;
            }
        }
        first_s[k][seg][lx] = first;
        last_s[k][seg][lx] = last;
    }
    __syncthreads();
    if (!live || y1 <= y0) return;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const uint8_t target = k == 0 ? 0 : 2;
        int above = -1, below = -1;                       // nearest seed rows outside the segment
        for (int sgm = seg - 1; sgm >= 0 && above < 0; --sgm) above = last_s[k][sgm][lx];
        for (int sgm = seg + 1; sgm < EDT_SEGS && below < 0; ++sgm) below = first_s[k][sgm][lx];
        int dn[LEN_MAX];
        int d = above >= 0 ? y0 - 1 - above : EDT_INF;    // distance of row y0-1 to the seed above
#pragma unroll
        for (int j = 0; j < LEN_MAX; ++j) {
            d = c[j] == target ? 0 : (d + 1 > EDT_INF ? EDT_INF : d + 1);
            dn[j] = d;
        }
        uint16_t* gk = g + (int64_t)k * H * W + x;
        d = below >= 0 ? below - y1 : EDT_INF;            // distance of row y1 to the seed below
#pragma unroll
        for (int j = LEN_MAX - 1; j >= 0; --j) {
            if (y0 + j < y1) {
                d = c[j] == target ? 0 : (d + 1 > EDT_INF ? EDT_INF : d + 1);
                gk[(int64_t)(y0 + j) * W] = (uint16_t)(dn[j] < d ? dn[j] : d);
            }
        }
    }
}

// images taller than EDT_SEGS * 40 rows: one thread per (class, column), two dependent passes (slow, rarely used)
__global__ void edt_columns_tall_kernel(const uint8_t* __restrict__ cls, int H, int W, uint16_t* __restrict__ g) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
    if (x >= W) return;
    const uint8_t target = k == 0 ? 0 : 2;
    uint16_t* gk = g + (int64_t)k * H * W + x;
    int d = EDT_INF;
    for (int y = 0; y < H; ++y) {
        d = cls[(int64_t)y * W + x] == target ? 0 : (d + 1 > EDT_INF ? EDT_INF : d + 1);
        gk[(int64_t)y * W] = (uint16_t)d;
    }
    d = EDT_INF;
    for (int y = H - 1; y >= 0; --y) {
        const int gv = gk[(int64_t)y * W];
        d = gv == 0 ? 0 : (d + 1 > EDT_INF ? EDT_INF : d + 1);
        gk[(int64_t)y * W] = (uint16_t)(gv < d ? gv : d);
    }
}

// phase 2 + encoding: one workgroup per image row, BOTH classes.  The row's squared column distances are staged in LDS;
// every pixel scans outwards, d2(x) = min_dx dx^2 + min(g2[x-dx], g2[x+dx]), and stops as soon as dx^2 >= best -- the trip
// count equals the pixel's own distance, so the work is sum_x d(x) instead of a serial W-step scan per row (the first
// version: one thread per row, Meijster's stack in global memory, 1.6 ms at 1080p).  Integer arithmetic throughout:
// exact.  A thread then writes the pixel's channels 3..11 of x11 (six Gaussians, the two soft channels
// trimap2_soft = [tri[:,0], tri[:,2]] (alpha/model.py:51), the zero pad) and two_chan_trimap into d80[70..71]
// (FBA/models.py:378, :418) in one go.
__device__ __forceinline__ int edt_row_best(const int* __restrict__ g2, const int* __restrict__ cidx, int W, int x) {
    int best = g2[x];
    // near field: outward scan, stops at dx^2 >= best (the trip count is the pixel's own distance)
    int dx = 1;
    for (; dx < 32 && dx * dx < best; ++dx) {
        const int xl = x - dx, xr = x + dx;
        int c = EDT_INF * EDT_INF;
        if (xl >= 0) c = g2[xl];
        if (xr < W) c = min(c, g2[xr]);
        best = min(best, dx * dx + c);
    }
    if (dx == 32 && dx * dx < best) {
        // far field (round 4): the minimiser i*(x) of (x - i)^2 + g2[i] is monotone in x (the cost is Monge), so it lies between
        // the minimisers of the two coarse points around x (every 32nd column, found once per row: edt_coarse_argmin).  The work
        // of a row is bounded by 32 (W + W / 32) candidates whatever the picture; the first far field walked 32-column blocks
        // outwards from every pixel: 145 us of a 230 us kernel on a disc-shaped trimap.  |i - x| < 32 is covered above.
        const int c = x >> 5;
        const int ilo = cidx[c], ihi = cidx[c + 1];
        const int e0 = ihi < x - 32 ? ihi : x - 32;
        for (int i = ilo; i <= e0; ++i) best = min(best, (x - i) * (x - i) + g2[i]);
        const int s1 = ilo > x + 32 ? ilo : x + 32;
        for (int i = s1; i <= ihi; ++i) best = min(best, (i - x) * (i - x) + g2[i]);
    }
    return best;
}

// leftmost minimiser of (xc - i)^2 + g2[i] for the coarse points xc = min(32 c, W - 1), c = 0 .. nb: two threads per point, one
// walking the 32-column blocks to the left of xc (xc included), one to the right, each stopping when the blocks' nearest column is
// farther than its best; blocks whose best case (nearest column, smallest g2) cannot win are skipped.  Ties go to the smaller
// index on both sides and in the merge, so the bounds of the fine pass bracket the leftmost minimiser.
__device__ __forceinline__ void edt_coarse_argmin(const int* __restrict__ g2, const int* __restrict__ bmin, int W, int nb, int c,
                                                  int dir, int* __restrict__ cidx) {
    const int xc = c * 32 < W - 1 ? c * 32 : W - 1;
    int best = 0x7fffffff, arg = xc;
    if (dir == 0) {
        for (int b = xc >> 5; b >= 0; --b) {
            const int hi = b * 32 + 31 < xc ? b * 32 + 31 : xc;
            const int dm = xc - hi;
            if (dm * dm > best) break;
            if (dm * dm + bmin[b] <= best)
                for (int i = hi; i >= b * 32; --i) {
                    const int v = (xc - i) * (xc - i) + g2[i];
                    if (v <= best) { best = v; arg = i; }
                }
        }
    } else {
        for (int b = (xc + 1) >> 5; b < nb; ++b) {
            const int lo = b * 32 > xc + 1 ? b * 32 : xc + 1;
            const int dm = lo - xc;
            if (dm * dm >= best) break;
            if (dm * dm + bmin[b] < best) {
                const int e = b * 32 + 32 < W ? b * 32 + 32 : W;
                for (int i = lo; i < e; ++i) {
                    const int v = (i - xc) * (i - xc) + g2[i];
                    if (v < best) { best = v; arg = i; }
                }
            }
        }
    }
    const int ob = __shfl_xor(best, 1), oa = __shfl_xor(arg, 1);   // the partner walks the other direction (adjacent lane)
    if (dir == 0) cidx[c] = ob < best ? oa : arg;
}

__global__ __launch_bounds__(256) void edt_rows_encode_kernel(const uint16_t* __restrict__ g, const float* __restrict__ probs,
                                                              int H, int W, const int* __restrict__ flags,
                                                              float* __restrict__ x11, int x11_ld, float* __restrict__ d80, int d80_ld,
                                                              int dbg) {
    extern __shared__ int lds[];
    const int y = blockIdx.x;
    const int nb = (W + 31) >> 5;
    int* g2[2] = {lds, lds + W};
    int* bmin[2] = {lds + 2 * W, lds + 2 * W + nb};        // min of g2 over each 32-column block
    int* cidx[2] = {lds + 2 * W + 2 * nb, lds + 2 * W + 3 * nb + 1};   // minimisers of the nb + 1 coarse points
    const float den0 = (float)(2.0 * ((0.02 * 320) * (0.02 * 320)));   // 2*((sigma*L)^2), utils/utils.py:33-37
    const float den1 = (float)(2.0 * ((0.08 * 320) * (0.08 * 320)));
    const float den2 = (float)(2.0 * ((0.16 * 320) * (0.16 * 320)));
    const int64_t P = (int64_t)H * W, row = (int64_t)y * W;
    const bool on[2] = {flags[0] != 0, flags[1] != 0};     // empty class -> zeros (utils/utils.py:32)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (!on[k]) continue;
        const uint16_t* gr = g + (int64_t)k * P + row;
        for (int x = threadIdx.x; x < W; x += blockDim.x) {
            const int v = gr[x];
            g2[k][x] = v * v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (!on[k]) continue;
        for (int b = threadIdx.x; b < nb; b += blockDim.x) {
            int m = EDT_INF * EDT_INF;
            const int e = b * 32 + 32 < W ? b * 32 + 32 : W;
            for (int i = b * 32; i < e; ++i) m = min(m, g2[k][i]);
            bmin[k][b] = m;
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 4 * (nb + 1); t += blockDim.x) {     // (class, coarse point, direction); pairs on adjacent lanes
        const int k = t / (2 * (nb + 1)), r = t - k * 2 * (nb + 1);
        if (on[k]) edt_coarse_argmin(g2[k], bmin[k], W, nb, r >> 1, r & 1, cidx[k]);
    }
    __syncthreads();
    for (int x = threadIdx.x; x < W; x += blockDim.x) {
        float e[6];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (on[k]) {
                const float d = sqrtf((float)((dbg & 4) ? g2[k][x] : edt_row_best(g2[k], cidx[k], W, x)));
                const float v = -(d * d);                  // -dt(1 - tk)**2
                if (dbg & 8) { e[3 * k] = v; e[3 * k + 1] = v + 1.f; e[3 * k + 2] = v + 2.f; }
                else { e[3 * k] = expf(v / den0); e[3 * k + 1] = expf(v / den1); e[3 * k + 2] = expf(v / den2); }
            } else {
                e[3 * k] = 0.f; e[3 * k + 1] = 0.f; e[3 * k + 2] = 0.f;
            }
        }
        const float p0 = probs[row + x], p2 = probs[2 * P + row + x];
        float* dst = x11 + (row + x) * x11_ld;
        if (!(dbg & 1)) {
            dst[3] = e[0];
            *reinterpret_cast<edt_f32x4*>(dst + 4) = edt_f32x4{e[1], e[2], e[3], e[4]};
            *reinterpret_cast<edt_f32x4*>(dst + 8) = edt_f32x4{e[5], p0, p2, 0.f};
        } else if (e[0] + e[1] + e[2] + e[3] + e[4] + e[5] == 12345.f) dst[3] = 0.f;
        if (!(dbg & 2)) *reinterpret_cast<edt_f32x2*>(d80 + (row + x) * d80_ld + 70) = edt_f32x2{p0, p2};
    }
}

}  // namespace

extern "C" int64_t otvm_trimap_encode_ws_bytes(int Hp, int Wp) {
    const int64_t P = (int64_t)Hp * Wp;
    return 256 + (2 * P) * 4;         // flags | g (the kernels use 2 x P x uint16; the size is kept from ABI <= 16)
}

extern "C" int otvm_trimap_encode(const float* probs, int Hp, int Wp, const uint8_t* cls_override, uint8_t* cls_out,
                                  float* x11, int x11_ld, float* d80, int d80_ld, void* ws, void* stream) {
    OTVM_REQUIRE(probs && cls_out && x11 && d80 && ws, "otvm_trimap_encode: null pointer");
    OTVM_REQUIRE(Hp < EDT_INF / 2 && Wp < EDT_INF / 2, "otvm_trimap_encode: image too large (%dx%d)", Hp, Wp);
    OTVM_REQUIRE(x11_ld % 4 == 0 && x11_ld >= 12 && d80_ld % 2 == 0 && ((uintptr_t)x11 & 15) == 0 && ((uintptr_t)d80 & 7) == 0,
                 "otvm_trimap_encode: x11 must be a 16-byte aligned view of >= 12 channels (stride %% 4 == 0)");
    hipStream_t s = (hipStream_t)stream;
    const int64_t P = (int64_t)Hp * Wp;
    int* flags = (int*)ws;
    uint16_t* g = (uint16_t*)((char*)ws + 256);
    if (hipMemsetAsync(flags, 0, 256, s) != hipSuccess) { otvm_set_error("otvm_trimap_encode: memset failed"); return 2; }
    int64_t nb = (P / 4 + 255) / 256;
    const int grid = (int)(nb > 4096 ? 4096 : (nb < 1 ? 1 : nb));
    // timing probes (results WRONG when set) exist in -DOTVM_PROBES builds only (tools/build_variant.sh): a release library
    // does not read OTVM_EDT_DBG
#ifdef OTVM_PROBES
    static const int dbg = otvm_probe_int("OTVM_EDT_DBG", 0);
#else
    constexpr int dbg = 0;
#endif
    if (!(dbg & 32))
    hipLaunchKernelGGL(classify_kernel, dim3(grid), dim3(256), 0, s, probs, P, cls_override, cls_out, flags);
    if (dbg & 64) return 0;
    const int len = otvm_ceil_div(Hp, EDT_SEGS);
    const dim3 cgrid(otvm_ceil_div(Wp, EDT_XB)), cblock(EDT_XB * EDT_SEGS);
    if (len <= 8) hipLaunchKernelGGL(edt_columns_kernel<8>, cgrid, cblock, 0, s, cls_out, Hp, Wp, len, g);
    else if (len <= 20) hipLaunchKernelGGL(edt_columns_kernel<20>, cgrid, cblock, 0, s, cls_out, Hp, Wp, len, g);
    else if (len <= 40) hipLaunchKernelGGL(edt_columns_kernel<40>, cgrid, cblock, 0, s, cls_out, Hp, Wp, len, g);
    else hipLaunchKernelGGL(edt_columns_tall_kernel, dim3(otvm_ceil_div(Wp, 64), 2), dim3(64), 0, s, cls_out, Hp, Wp, g);
    if (!(dbg & 16))
    hipLaunchKernelGGL(edt_rows_encode_kernel, dim3(Hp), dim3(256), (2 * Wp + 4 * ((Wp + 31) / 32) + 2) * sizeof(int), s, g, probs, Hp, Wp,
                       flags, x11, x11_ld, d80, d80_ld, dbg);
    OTVM_CHECK_LAUNCH("otvm_trimap_encode");
    return 0;
}
