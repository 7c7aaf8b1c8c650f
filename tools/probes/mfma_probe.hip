// What does v_mfma_f32_32x32x16_f16 sustain on this MI355X under its power limit?  Register-resident operands, no memory
// traffic: every wave issues `iters` rounds of 8 independent MFMAs (8 accumulator tiles).  Operand data decides the power
// drawn, so three fills are timed: zeros, fp16 values drawn like activations/weights ("hi" halves), and hi/lo pairs as the
// f16x3 kernels feed them (two of three MFMAs see a small-magnitude "lo" operand).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe tools/probes/mfma_probe.hip && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ inline float rnd(unsigned& s) {            // ~N(0,1): sum of 4 uniforms
    float a = 0.f;
    for (int i = 0; i < 4; ++i) { s = s * 1664525u + 1013904223u; a += (float)(s >> 8) * (1.0f / 16777216.0f); }
    return (a - 2.0f) * 1.7320508f;
}

__global__ __launch_bounds__(512) void probe(int iters, int mode, float* out) {
    unsigned s = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 12345u;
    f16x8 ah[2], al[2], bh[4], bl[4];
    for (int t = 0; t < 2; ++t) for (int j = 0; j < 8; ++j) {
        const float v = mode ? rnd(s) : 0.f;
        const _Float16 h = (_Float16)v;
        ah[t][j] = h; al[t][j] = (_Float16)(v - (float)h);
    }
    for (int t = 0; t < 4; ++t) for (int j = 0; j < 8; ++j) {
        const float v = mode ? 0.05f * rnd(s) : 0.f;
        const _Float16 h = (_Float16)v;
        bh[t][j] = h; bl[t][j] = (_Float16)(v - (float)h);
    }
    f32x16 acc[8];
    for (int t = 0; t < 8; ++t) for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    // f16x3 (mode 2): lo*hi, hi*lo, hi*hi; otherwise the same hi*hi product three times
                    const f16x8 a = (mode == 2 && term == 0) ? al[m] : ah[m];
                    const f16x8 b = (mode == 2 && term == 1) ? bl[n] : bh[n];
                    acc[m * 4 + n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m * 4 + n], 0, 0, 0);
                }
        // keep the operands changing a little (a real kernel feeds new fragments every step)
        if (mode) { ah[0] = ah[0] + al[1]; ah[1] = ah[1] - al[0]; }
    }
    float r = 0.f;
    for (int t = 0; t < 8; ++t) for (int j = 0; j < 16; ++j) r += acc[t][j];
    if (r == 123.456f) out[0] = r;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    float* out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"zeros", "random fp16 (hi*hi only)", "f16x3 hi/lo triples"};
    for (int cus = 256; cus >= 64; cus /= 2)
        for (int mode = 0; mode < 3; ++mode) {
            hipLaunchKernelGGL(probe, dim3(cus), dim3(512), 0, 0, iters / 10, mode, out);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(probe, dim3(cus), dim3(512), 0, 0, iters, mode, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double mf = (double)cus * 8 * iters * 24;            // MFMAs
            const double tf = mf * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
            // a SIMD runs 2 of the 8 waves; one MFMA occupies it for 32 cycles (8 passes x 4)
            const double ghz = (double)iters * 24 * 2 * 32 / (ms * 1e-3) / 1e9;
            printf("%3d workgroups x 8 waves, %-28s: %8.2f ms  %7.1f TFLOP/s f16 (%6.1f fp32-equivalent f16x3)  => %.2f GHz if the pipe never idles\n",
                   cus, names[mode], ms, tf, tf / 3, ghz);
        }
    return 0;
}
