// Per-frame glue of EvalModel.forward (reference models/alpha/model.py:380-512): compositing and
// normalisation, trimap padding, the FBA/refinement heads with fba_fusion, output cropping, and the
// first-frame trimap from a GT alpha.  All elementwise / HBM-bound.  Compiled with
// -ffp-contract=off so the elementwise arithmetic follows the reference's operation order exactly.
#include "common.h"
#include "head_math.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

__global__ void preprocess_kernel(const otvm_preprocess_params p) {
    const int64_t P = (int64_t)p.Hp * p.Wp;
    const float s = 1.f / 255.f;                              // IMG_SCALE, alpha/model.py:27
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
        const int yp = (int)(i / p.Wp), xp = (int)(i - (int64_t)yp * p.Wp);
        const int y = yp - p.lh, x = xp - p.lw;
        float img[3] = {0.f, 0.f, 0.f};
        if ((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) {
            const int64_t o = (int64_t)y * p.W + x, plane = (int64_t)p.H * p.W;
            const float a = p.a[o];
#pragma unroll
            for (int c = 0; c < 3; ++c) {                     // BGR -> RGB flip (alpha/model.py:384-385)
                float sf, sb;
                if (p.fg_u8) {                                // decoded uint8 [H,W,3]: float(u8) == the caller's .float()
                    const int ch = p.u8_rgb ? c : 2 - c;
                    sf = (float)p.fg_u8[o * 3 + ch] * s;
                    sb = (float)p.bg_u8[o * 3 + ch] * s;
                } else {
                    sf = p.fg[(2 - c) * plane + o] * s;
                    sb = p.bg[(2 - c) * plane + o] * s;
                }
                img[c] = sf * a + sb * (1.f - a);             // alpha/model.py:386
                if (p.scaled_imgs) p.scaled_imgs[c * plane + o] = img[c];
            }
        }
        float n[3], q[3], m[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            n[c] = (img[c] - p.mean[c]) / p.std[c];           // alpha/model.py:414
            q[c] = (img[c] - p.mean_q[c]) / p.std_q[c];       // STM.py:90
            m[c] = (img[c] - p.mean_m[c]) / p.std_m[c];       // STM.py:54
        }
        // 16-byte stores into the interleaved buffers (a pixel's three scalars 48 / 320 bytes apart from the next
        // pixel's were three uncoalesced store instructions each).  The fourth lane of every vector is a channel a
        // LATER kernel of the frame owns and overwrites: x11[3] (distance encoding), sq[3] / sm[+3] (zero pad / p_un).
        // Every destination is optional (NULL = skip): the host launches the query-encoder input (sq) on its own stream
        // ahead of the rest (otvm_amd/engine.py: the query encoder of frame t+1 overlaps the alpha network of frame t).
        if (p.x11) *reinterpret_cast<f32x4*>(p.x11 + i * p.x11_ld) = f32x4{n[0], n[1], n[2], 0.f};
        if (p.d80) {
            *reinterpret_cast<f32x4*>(p.d80 + i * p.d80_ld + 64) = f32x4{n[0], n[1], n[2], img[0]};   // FBA/models.py:377
            *reinterpret_cast<f32x2*>(p.d80 + i * p.d80_ld + 68) = f32x2{img[1], img[2]};
        }
        if (p.sq) *reinterpret_cast<f32x4*>(p.sq + i * p.sq_ld) = f32x4{q[0], q[1], q[2], 0.f};
        if (p.sm) *reinterpret_cast<f32x4*>(p.sm + i * p.sm_ld) = f32x4{m[0], m[1], m[2], 0.f};
    }
}

__global__ void pad_trimap_kernel(const float* __restrict__ tri, int H, int W, float* __restrict__ out, int Hp, int Wp,
                                  int lh, int lw) {
    const int64_t P = (int64_t)Hp * Wp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
        const int yp = (int)(i / Wp), xp = (int)(i - (int64_t)yp * Wp);
        const int y = yp - lh, x = xp - lw;
        const bool in = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        const int64_t o = (int64_t)y * W + x, plane = (int64_t)H * W;
        out[i] = in ? tri[o] : 1.f;                           // bg padded with 1 (alpha/model.py:410)
        out[P + i] = in ? tri[plane + o] : 0.f;
        out[2 * P + i] = in ? tri[2 * plane + o] : 0.f;
    }
}

__global__ void upsample4_softmax3_kernel(const float* __restrict__ lg, int h4, int w4, int ld, float* __restrict__ probs,
                                          float* __restrict__ logits_out) {
    const int Hp = h4 * 4, Wp = w4 * 4;
    const int64_t P = (int64_t)Hp * Wp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
        const int oy = (int)(i / Wp), ox = (int)(i - (int64_t)oy * Wp);
        float fy = ((float)oy + 0.5f) * 0.25f - 0.5f, fx = ((float)ox + 0.5f) * 0.25f - 0.5f;
        fy = fy < 0.f ? 0.f : fy;
        fx = fx < 0.f ? 0.f : fx;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < h4 - 1 ? 1 : 0), x1 = x0 + (x0 < w4 - 1 ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
        float l[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v00 = lg[((int64_t)y0 * w4 + x0) * ld + c], v01 = lg[((int64_t)y0 * w4 + x1) * ld + c];
            const float v10 = lg[((int64_t)y1 * w4 + x0) * ld + c], v11 = lg[((int64_t)y1 * w4 + x1) * ld + c];
            l[c] = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
        }
        if (logits_out) {                                       // training forward: the cross-entropy takes the logits
            logits_out[i] = l[0];
            logits_out[P + i] = l[1];
            logits_out[2 * P + i] = l[2];
        }
        if (!probs) continue;
        const float m = fmaxf(l[0], fmaxf(l[1], l[2]));
        const float e0 = expf(l[0] - m), e1 = expf(l[1] - m), e2 = expf(l[2] - m);
        const float inv = 1.f / (e0 + e1 + e2);
        probs[i] = e0 * inv;
        probs[P + i] = e1 * inv;
        probs[2 * P + i] = e2 * inv;
    }
}

// 1x1 conv 16 -> n_out, clamp / sigmoid, fba_fusion (FBA/models.py:279-288; the B update reads the
// already-updated F), softmax of the 3 trimap-refinement logits: head_math.h (shared with the conv kernel that carries
// the head in its epilogue).
__global__ __launch_bounds__(256) void fba_head_kernel(const float* __restrict__ hid, int hid_ld, const OtvmHeadArgs q) {
    // the 1x1 weights are read through wave-uniform addresses (scalar loads, 16 SGPRs per output row at a time).  Round 1-3
    // staged them in LDS: the compiler hoisted all 170 LDS reads out of the pixel loop into registers -- 250 VGPRs, two waves
    // per SIMD for an elementwise kernel (found in the round-4 ISA audit, profiles/r04_isa_audit.txt)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < q.P; i += (int64_t)gridDim.x * blockDim.x) {
        float h[16];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(hid + i * hid_ld + 4 * k);
            h[4 * k] = v.x; h[4 * k + 1] = v.y; h[4 * k + 2] = v.z; h[4 * k + 3] = v.w;
        }
        otvm_head_pixel(h, q, i);
    }
}

__global__ void crop_outputs_kernel(const float* __restrict__ alpha_p, const float* __restrict__ tri_p, int Hp, int Wp, int H,
                                    int W, int lh, int lw, float* __restrict__ alpha, uint8_t* __restrict__ alpha_u8,
                                    float* __restrict__ tri) {
    const int64_t N = (int64_t)H * W, P = (int64_t)Hp * Wp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
        const int64_t ip = (int64_t)(y + lh) * Wp + (x + lw);
        const float a = alpha_p[ip];
        alpha[i] = a;
        if (alpha_u8) alpha_u8[i] = (uint8_t)(a * 255.f);       // (alphas*255).byte() truncates (eval.py:209)
        if (tri) {
            tri[i] = tri_p[ip];
            tri[N + i] = tri_p[P + ip];
            tri[2 * N + i] = tri_p[2 * P + ip];
        }
    }
}

__global__ void unknown_rowmax_kernel(const float* __restrict__ a, int H, int W, int r, uint8_t* __restrict__ tmp) {
    const int64_t N = (int64_t)H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
        int any = 0;
        const int lo = x - r < 0 ? 0 : x - r, hi = x + r >= W ? W - 1 : x + r;
        for (int xx = lo; xx <= hi; ++xx) {
            const float v = a[(int64_t)y * W + xx];
            any |= (v > 0.f && v < 1.f);
        }
        tmp[i] = (uint8_t)any;
    }
}

__global__ void trimap_from_alpha_kernel(const float* __restrict__ a, const uint8_t* __restrict__ tmp, int H, int W, int r,
                                         float* __restrict__ out) {
    const int64_t N = (int64_t)H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
        int any = 0;
        const int lo = y - r < 0 ? 0 : y - r, hi = y + r >= H ? H - 1 : y + r;
        for (int yy = lo; yy <= hi; ++yy) any |= tmp[(int64_t)yy * W + x];
        // trimap1 = where(dilated, 1, 2*alpha).long() -> one_hot (alpha/model.py:361-362)
        int cls = any ? 1 : (int)(2.f * a[i]);
        cls = cls < 0 ? 0 : (cls > 2 ? 2 : cls);
        out[i] = cls == 0 ? 1.f : 0.f;
        out[N + i] = cls == 1 ? 1.f : 0.f;
        out[2 * N + i] = cls == 2 ? 1.f : 0.f;
    }
}

__global__ void onehot_argmax3_kernel(const float* __restrict__ tri, int64_t P, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
        const float v0 = tri[i], v1 = tri[P + i], v2 = tri[2 * P + i];
        int cls = 0;
        float m = v0;
        if (v1 > m) { m = v1; cls = 1; }
        if (v2 > m) { cls = 2; }
        out[i] = cls == 0 ? 1.f : 0.f;
        out[P + i] = cls == 1 ? 1.f : 0.f;
        out[2 * P + i] = cls == 2 ? 1.f : 0.f;
    }
}

int grid_for(int64_t total) {
    int64_t b = (total + 255) / 256;
    return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int otvm_preprocess(const otvm_preprocess_params* p, void* stream) {
    OTVM_REQUIRE(p && ((p->fg && p->bg) || (p->fg_u8 && p->bg_u8)) && p->a, "otvm_preprocess: null input pointer");
    OTVM_REQUIRE(p->x11 || p->sq || p->sm || p->d80 || p->scaled_imgs, "otvm_preprocess: no destination given");
    OTVM_REQUIRE(p->x11_ld % 4 == 0 && p->d80_ld % 4 == 0 && p->sq_ld % 4 == 0 && p->sm_ld % 4 == 0 &&
                     (((uintptr_t)p->x11 | (uintptr_t)p->d80 | (uintptr_t)p->sq | (uintptr_t)p->sm) & 15) == 0,
                 "otvm_preprocess: x11 / d80 / sq / sm views must be 16-byte aligned with strides that are multiples of 4");
    hipLaunchKernelGGL(preprocess_kernel, dim3(grid_for((int64_t)p->Hp * p->Wp)), dim3(256), 0, (hipStream_t)stream, *p);
    OTVM_CHECK_LAUNCH("otvm_preprocess");
    return 0;
}

extern "C" int otvm_pad_trimap(const float* tri, int H, int W, float* out, int Hp, int Wp, int lh, int lw, void* stream) {
    hipLaunchKernelGGL(pad_trimap_kernel, dim3(grid_for((int64_t)Hp * Wp)), dim3(256), 0, (hipStream_t)stream, tri, H, W, out,
                       Hp, Wp, lh, lw);
    OTVM_CHECK_LAUNCH("otvm_pad_trimap");
    return 0;
}

extern "C" int otvm_upsample4_softmax3(const float* logits, int h4, int w4, int ld, float* probs, void* stream) {
    hipLaunchKernelGGL(upsample4_softmax3_kernel, dim3(grid_for((int64_t)h4 * w4 * 16)), dim3(256), 0, (hipStream_t)stream,
                       logits, h4, w4, ld, probs, (float*)nullptr);
    OTVM_CHECK_LAUNCH("otvm_upsample4_softmax3");
    return 0;
}

extern "C" int otvm_upsample4_logits3(const float* logits, int h4, int w4, int ld, float* logits_out, void* stream) {
    OTVM_REQUIRE(logits && logits_out, "otvm_upsample4_logits3: null pointer");
    hipLaunchKernelGGL(upsample4_softmax3_kernel, dim3(grid_for((int64_t)h4 * w4 * 16)), dim3(256), 0, (hipStream_t)stream,
                       logits, h4, w4, ld, (float*)nullptr, logits_out);
    OTVM_CHECK_LAUNCH("otvm_upsample4_logits3");
    return 0;
}

__global__ void trimap_to_sm_kernel(const float* __restrict__ tri, int64_t P, float* __restrict__ sm, int sm_ld) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
        sm[i * sm_ld + 3] = tri[P + i];                         // unknown -> conv1_m (STM.py:58)
        sm[i * sm_ld + 4] = tri[2 * P + i];                     // foreground -> conv1_o
    }
}

extern "C" int otvm_trimap_to_sm(const float* tri, int64_t P, float* sm, int sm_ld, void* stream) {
    OTVM_REQUIRE(tri && sm && P > 0, "otvm_trimap_to_sm: bad arguments");
    hipLaunchKernelGGL(trimap_to_sm_kernel, dim3(grid_for(P)), dim3(256), 0, (hipStream_t)stream, tri, P, sm, sm_ld);
    OTVM_CHECK_LAUNCH("otvm_trimap_to_sm");
    return 0;
}

extern "C" int otvm_fba_head(const float* hid, int hid_ld, const float* w, const float* b, int n_out, const float* img,
                             int img_ld, int64_t P, float* alpha_out, int alpha_stride, float* tri_out, float* sm,
                             int sm_ld, void* stream) {
    OTVM_REQUIRE(n_out == 7 || n_out == 10, "otvm_fba_head: n_out must be 7 or 10 (got %d)", n_out);
    OTVM_REQUIRE(n_out == 7 || tri_out, "otvm_fba_head: tri_out required when n_out == 10");
    OTVM_REQUIRE(hid_ld % 4 == 0 && ((uintptr_t)hid & 15) == 0, "otvm_fba_head: hid view must be 16-byte aligned");
    OtvmHeadArgs q;
    q.w = w; q.b = b; q.n_out = n_out; q.img = img; q.img_ld = img_ld; q.P = P; q.alpha_out = alpha_out; q.alpha_stride = alpha_stride;
    q.tri_out = tri_out; q.sm = sm; q.sm_ld = sm_ld; q.out7 = nullptr; q.logits_out = nullptr;
    hipLaunchKernelGGL(fba_head_kernel, dim3(grid_for(P)), dim3(256), 0, (hipStream_t)stream, hid, hid_ld, q);
    OTVM_CHECK_LAUNCH("otvm_fba_head");
    return 0;
}

extern "C" int otvm_fba_head_train(const float* hid, int hid_ld, const float* w, const float* b, int n_out, const float* img,
                                   int img_ld, int64_t P, float* out7, float* logits_out, void* stream) {
    OTVM_REQUIRE(n_out == 7 || n_out == 10, "otvm_fba_head_train: n_out must be 7 or 10 (got %d)", n_out);
    OTVM_REQUIRE(out7 && (n_out == 7 || logits_out), "otvm_fba_head_train: out7 (and logits_out for n_out == 10) required");
    OTVM_REQUIRE(hid_ld % 4 == 0 && ((uintptr_t)hid & 15) == 0, "otvm_fba_head_train: hid view must be 16-byte aligned");
    OtvmHeadArgs q;
    q.w = w; q.b = b; q.n_out = n_out; q.img = img; q.img_ld = img_ld; q.P = P; q.alpha_out = nullptr; q.alpha_stride = 0;
    q.tri_out = nullptr; q.sm = nullptr; q.sm_ld = 0; q.out7 = out7; q.logits_out = logits_out;
    hipLaunchKernelGGL(fba_head_kernel, dim3(grid_for(P)), dim3(256), 0, (hipStream_t)stream, hid, hid_ld, q);
    OTVM_CHECK_LAUNCH("otvm_fba_head_train");
    return 0;
}

extern "C" int otvm_crop_outputs(const float* alpha_p, const float* tri_p, int Hp, int Wp, int H, int W, int lh, int lw,
                                 float* alpha, uint8_t* alpha_u8, float* tri, void* stream) {
    hipLaunchKernelGGL(crop_outputs_kernel, dim3(grid_for((int64_t)H * W)), dim3(256), 0, (hipStream_t)stream, alpha_p, tri_p,
                       Hp, Wp, H, W, lh, lw, alpha, alpha_u8, tri);
    OTVM_CHECK_LAUNCH("otvm_crop_outputs");
    return 0;
}

extern "C" int otvm_trimap_from_alpha(const float* a, int H, int W, int r, float* out, void* ws, void* stream) {
    OTVM_REQUIRE(r >= 0 && ws, "otvm_trimap_from_alpha: bad arguments");
    const int g = grid_for((int64_t)H * W);
    hipLaunchKernelGGL(unknown_rowmax_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, a, H, W, r, (uint8_t*)ws);
    hipLaunchKernelGGL(trimap_from_alpha_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, a, (const uint8_t*)ws, H, W, r,
                       out);
    OTVM_CHECK_LAUNCH("otvm_trimap_from_alpha");
    return 0;
}

extern "C" int otvm_onehot_argmax3(const float* tri, int64_t P, float* out, void* stream) {
    hipLaunchKernelGGL(onehot_argmax3_kernel, dim3(grid_for(P)), dim3(256), 0, (hipStream_t)stream, tri, P, out);
    OTVM_CHECK_LAUNCH("otvm_onehot_argmax3");
    return 0;
}
