// The per-pixel tail of the FBA decoder / refinement heads (reference FBA/models.py:279-288, 383-388, 425-432): 1x1 conv
// 16 -> n_out (7 or 10), clamp / sigmoid, fba_fusion (the B update reads the already-updated F), softmax of the three
// trimap-refinement logits.  Shared by fba_head_kernel (glue.hip) and the 3x3 conv kernel that carries the head in its
// epilogue (conv_patch_f16x3.hip, round 4); fp contraction is off inside, so both evaluate the reference's operation order.
#pragma once

struct OtvmHeadArgs {
    const float* w; const float* b; int n_out;           // 1x1 weights [n_out][16], bias [n_out] (wave-uniform: scalar loads)
    const float* img; int img_ld;                        // composited RGB in [0,1], 3 channels at pixel stride img_ld
    int64_t P;                                           // pixels per (padded) plane of the planar outputs
    float* alpha_out; int alpha_stride;                  // fused alpha
    float* tri_out;                                      // [3][P] softmax of logits 7..9 (n_out == 10)
    float* sm; int sm_ld;                                // optional: (p_unknown, p_fg, alpha) -> sm[i * sm_ld + 3 .. 5]
    float* out7; float* logits_out;                      // training forward: fused (alpha, F, B) planar [7][P]; raw logits [3][P]
};

__device__ __forceinline__ float otvm_sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float otvm_clamp01(float x) { return fminf(fmaxf(x, 0.f), 1.f); }

// N_OUT_T = 7 / 10: the head's width known at compile time (0: q.n_out at run time); WP = where the head's weights / bias are read
// from -- `const float*` (global memory, wave-uniform addresses) or an LDS pointer (the 16-wide conv stages them once per
// workgroup, round 5); im = the pixel's composited RGB, loaded by the caller (early, so that its latency is not exposed here)
template <int N_OUT_T, typename WP>
__device__ __forceinline__ void otvm_head_pixel_w(const float (&h)[16], const OtvmHeadArgs& q, int64_t i, WP w, WP b, const float (&im)[3]) {
#pragma clang fp contract(off)
    const int n_out = N_OUT_T ? N_OUT_T : q.n_out;
    const int64_t P = q.P;
    float o[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) {
        if (j < n_out) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) acc += w[j * 16 + k] * h[k];
            o[j] = acc + b[j];
        } else {
            o[j] = 0.f;
        }
    }
    float al = otvm_clamp01(o[0]);
    float F[3], B[3];
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float f0 = otvm_sigmoidf(o[1 + c]), b0 = otvm_sigmoidf(o[4 + c]);
        float fn = al * im[c] + (1.f - al * al) * f0 - al * (1.f - al) * b0;
        float bn = (1.f - al) * im[c] + (2.f * al - al * al) * b0 - al * (1.f - al) * fn;
        F[c] = otvm_clamp01(fn);
        B[c] = otvm_clamp01(bn);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        num += (im[c] - B[c]) * (F[c] - B[c]);
        den += (F[c] - B[c]) * (F[c] - B[c]);
    }
    al = otvm_clamp01((al * 0.1f + num) / (den + 0.1f));
    if (q.alpha_out) q.alpha_out[i * q.alpha_stride] = al;
    if (q.out7) {                                           // training forward: the fused (alpha, F, B) of FBA/models.py:388
        q.out7[i] = al;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            q.out7[(1 + c) * P + i] = F[c];
            q.out7[(4 + c) * P + i] = B[c];
        }
    }
    if (q.logits_out && n_out == 10) {
        q.logits_out[i] = o[7];
        q.logits_out[P + i] = o[8];
        q.logits_out[2 * P + i] = o[9];
    }
    if (n_out == 10 && q.tri_out) {
        const float m = fmaxf(o[7], fmaxf(o[8], o[9]));
        const float e0 = expf(o[7] - m), e1 = expf(o[8] - m), e2 = expf(o[9] - m);
        const float inv = 1.f / (e0 + e1 + e2);
        q.tri_out[i] = e0 * inv;
        q.tri_out[P + i] = e1 * inv;
        q.tri_out[2 * P + i] = e2 * inv;
        if (q.sm) {                                         // Es = cat[tri, alpha, hid] (trimap/model.py:231)
            q.sm[i * q.sm_ld + 3] = e1 * inv;               // unknown prob  -> conv1_m (STM.py:58)
            q.sm[i * q.sm_ld + 4] = e2 * inv;               // fg prob       -> conv1_o
            q.sm[i * q.sm_ld + 5] = al;                     // alpha         -> conv1_a
        }
    }
}

__device__ __forceinline__ void otvm_head_pixel(const float (&h)[16], const OtvmHeadArgs& q, int64_t i) {
    const float im[3] = {q.img[i * q.img_ld], q.img[i * q.img_ld + 1], q.img[i * q.img_ld + 2]};
    otvm_head_pixel_w<0, const float*>(h, q, i, q.w, q.b, im);
}
