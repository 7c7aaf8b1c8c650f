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

namespace {

typedef float edt_f32x4 __attribute__((ext_vector_type(4)));
typedef float edt_f32x2 __attribute__((ext_vector_type(2)));
constexpr int EDT_INF = 1 << 14;     // > any image side handled (asserted on the host side)

__global__ void classify_kernel(const float* __restrict__ probs, int64_t P, const uint8_t* __restrict__ cls_override,
                                uint8_t* __restrict__ cls_out, int* __restrict__ flags, float* __restrict__ x11, int x11_ld,
                                float* __restrict__ d80, int d80_ld) {
    int has_bg = 0, has_fg = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
        const float p0 = probs[i], p1 = probs[P + i], p2 = probs[2 * P + i];
        int cls;
        if (cls_override) {
            cls = cls_override[i];
        } else {                          // tri.max(dim)[1]: first maximal index (alpha/model.py:42)
            cls = 0;
            float m = p0;
            if (p1 > m) { m = p1; cls = 1; }
            if (p2 > m) { cls = 2; }
        }
        cls_out[i] = (uint8_t)cls;
        has_bg |= (cls == 0);
        has_fg |= (cls == 2);
        // trimap2_soft = [tri[:,0], tri[:,2]] (alpha/model.py:51) -> x11 channels 9, 10; one 16-byte store over channels
        // 8..11: channel 8 (last distance encoding) is written by edt_rows_encode afterwards, 11 is the zero pad
        *reinterpret_cast<edt_f32x4*>(x11 + i * x11_ld + 8) = edt_f32x4{0.f, p0, p2, 0.f};
        *reinterpret_cast<edt_f32x2*>(d80 + i * d80_ld + 70) = edt_f32x2{p0, p2};   // two_chan_trimap (FBA/models.py:378, :418)
    }
    // one flag write per class is enough: a wave looks first (L2 read) and only the first arrivals issue the atomic
    // (16 k waves doing an atomicOr on the same two words serialised at L2: 0.16 of the kernel's 0.20 ms)
    if (__any(has_bg) && (threadIdx.x & 63) == 0 && __hip_atomic_load(&flags[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
        atomicOr(&flags[0], 1);
    if (__any(has_fg) && (threadIdx.x & 63) == 0 && __hip_atomic_load(&flags[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
        atomicOr(&flags[1], 1);
}

// phase 1: distance to the nearest class pixel within the column.  A column is cut into EDT_SEG segments, one thread
// each (a workgroup = 64 columns x EDT_SEG segments, lanes along x: coalesced rows): every thread finds the first and
// last seed of its segment, the carries (nearest seed above / below the segment) come from the other segments'
// entries in LDS, then a down-scan and an up-scan over the segment write g.  (One thread per whole column -- 3840
// threads at 1080p -- took 0.43 ms; integer arithmetic, same result.)
constexpr int EDT_SEG = 16;

__global__ __launch_bounds__(64 * EDT_SEG) void edt_columns_kernel(const uint8_t* __restrict__ cls, int H, int W,
                                                                    int* __restrict__ g) {
    __shared__ int first_s[EDT_SEG][64], last_s[EDT_SEG][64];
    const int lx = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const int cols_per_class = (W + 63) / 64;
    const int k = blockIdx.x / cols_per_class, x = (blockIdx.x - k * cols_per_class) * 64 + lx;
    const uint8_t target = k == 0 ? 0 : 2;
    const int len = (H + EDT_SEG - 1) / EDT_SEG;
    const int y0 = seg * len, y1 = y0 + len < H ? y0 + len : H;
    const bool live = x < W;
    int first = -1, last = -1;
    if (live) {
        for (int y = y0; y < y1; ++y) {
            if (cls[(int64_t)y * W + x] == target) {
                if (first < 0) first = y;
                last = y;
            }
        }
    }
    first_s[seg][lx] = first;
    last_s[seg][lx] = last;
    __syncthreads();
    if (!live) return;
    int above = -1, below = -1;                       // nearest seed rows outside the segment
    for (int sgm = seg - 1; sgm >= 0 && above < 0; --sgm) above = last_s[sgm][lx];
    for (int sgm = seg + 1; sgm < EDT_SEG && below < 0; ++sgm) below = first_s[sgm][lx];
    int* gk = g + (int64_t)k * H * W;
    int d = above >= 0 ? y0 - 1 - above : EDT_INF;    // distance of row y0-1 to the seed above
    for (int y = y0; y < y1; ++y) {
        const bool seed = cls[(int64_t)y * W + x] == target;
        d = seed ? 0 : (d + 1 > EDT_INF ? EDT_INF : d + 1);
        gk[(int64_t)y * W + x] = d;
    }
    d = below >= 0 ? below - y1 : EDT_INF;            // distance of row y1 to the seed below
    for (int y = y1 - 1; y >= y0; --y) {
        const int gv = gk[(int64_t)y * W + x];
        d = gv == 0 ? 0 : (d + 1 > EDT_INF ? EDT_INF : d + 1);
        gk[(int64_t)y * W + x] = gv < d ? gv : d;
    }
}

// phase 2 + encoding: one workgroup per (class, image row).  The row of squared column distances is staged in
// LDS; every pixel scans outwards, d2(x) = min_dx dx^2 + min(g2[x-dx], g2[x+dx]), and stops as soon as dx^2 >= best
// -- the trip count equals the pixel's own distance, so the work is sum_x d(x) instead of a serial W-step scan per
// row (the first version: one thread per row, Meijster's stack in global memory, 1.6 ms at 1080p).  Integer
// arithmetic throughout: exact.  The three Gaussians of the class are written straight into the x11 slice.
__global__ __launch_bounds__(256) void edt_rows_encode_kernel(const int* __restrict__ g, int H, int W, const int* __restrict__ flags,
                                                              float* __restrict__ x11, int x11_ld) {
    extern __shared__ int g2[];
    const int k = blockIdx.x / H, y = blockIdx.x - k * H;
    const float den0 = (float)(2.0 * ((0.02 * 320) * (0.02 * 320)));   // 2*((sigma*L)^2), utils/utils.py:33-37
    const float den1 = (float)(2.0 * ((0.08 * 320) * (0.08 * 320)));
    const float den2 = (float)(2.0 * ((0.16 * 320) * (0.16 * 320)));
    float* dst = x11 + (int64_t)y * W * x11_ld + 3 + 3 * k;
    if (!flags[k]) {                                   // empty class -> zeros (utils/utils.py:32)
        for (int x = threadIdx.x; x < W; x += blockDim.x) {
            dst[(int64_t)x * x11_ld] = 0.f; dst[(int64_t)x * x11_ld + 1] = 0.f; dst[(int64_t)x * x11_ld + 2] = 0.f;
        }
        return;
    }
    const int* gr = g + ((int64_t)k * H + y) * W;
    const int nb = (W + 31) >> 5;
    int* bmin = g2 + W;                                // min of g2 over each 32-column block
    for (int x = threadIdx.x; x < W; x += blockDim.x) {
        const int v = gr[x];
        g2[x] = v * v;
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nb; b += blockDim.x) {
        int m = EDT_INF * EDT_INF;
        const int e = b * 32 + 32 < W ? b * 32 + 32 : W;
        for (int i = b * 32; i < e; ++i) m = min(m, g2[i]);
        bmin[b] = m;
    }
    __syncthreads();
    for (int x = threadIdx.x; x < W; x += blockDim.x) {
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
            // far field: whole 32-column blocks, outwards; a block is skipped when even its best case
            // (nearest column, smallest g2 of the block) cannot improve -- pixels far from every seed cross the
            // empty columns in W/32 steps instead of W
            const int bx = x >> 5;
            for (int r = 1;; ++r) {
                bool any = false;
                const int bl = bx - r, br = bx + r;
                if (bl >= 0) {
                    const int dm = x - (bl * 32 + 31);
                    if (dm * dm < best) {
                        any = true;
                        if (dm * dm + bmin[bl] < best)
                            for (int i = bl * 32; i < bl * 32 + 32; ++i) best = min(best, (x - i) * (x - i) + g2[i]);
                    }
                }
                if (br < nb) {
                    const int dm = br * 32 - x;
                    if (dm * dm < best) {
                        any = true;
                        if (dm * dm + bmin[br] < best) {
                            const int e = br * 32 + 32 < W ? br * 32 + 32 : W;
                            for (int i = br * 32; i < e; ++i) best = min(best, (i - x) * (i - x) + g2[i]);
                        }
                    }
                }
                if (!any) break;
            }
        }
        const float d = sqrtf((float)best);
        const float v = -(d * d);                      // -dt(1 - tk)**2
        dst[(int64_t)x * x11_ld] = expf(v / den0);
        dst[(int64_t)x * x11_ld + 1] = expf(v / den1);
        dst[(int64_t)x * x11_ld + 2] = expf(v / den2);
    }
}

}  // namespace

extern "C" int64_t otvm_trimap_encode_ws_bytes(int Hp, int Wp) {
    const int64_t P = (int64_t)Hp * Wp;
    return 256 + (2 * P) * 4;         // flags | g   (int32, 2 classes)
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
    int* g = (int*)((char*)ws + 256);
    if (hipMemsetAsync(flags, 0, 256, s) != hipSuccess) { otvm_set_error("otvm_trimap_encode: memset failed"); return 2; }
    int64_t nb = (P + 255) / 256;
    const int grid = (int)(nb > 4096 ? 4096 : nb);
    hipLaunchKernelGGL(classify_kernel, dim3(grid), dim3(256), 0, s, probs, P, cls_override, cls_out, flags, x11, x11_ld, d80,
                       d80_ld);
    hipLaunchKernelGGL(edt_columns_kernel, dim3(2 * otvm_ceil_div(Wp, 64)), dim3(64 * EDT_SEG), 0, s, cls_out, Hp, Wp, g);
    hipLaunchKernelGGL(edt_rows_encode_kernel, dim3(2 * Hp), dim3(256), (Wp + (Wp + 31) / 32) * sizeof(int), s, g, Hp, Wp, flags,
                       x11, x11_ld);
    OTVM_CHECK_LAUNCH("otvm_trimap_encode");
    return 0;
}
