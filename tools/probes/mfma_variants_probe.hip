// Which matrix-core instruction does the most work per joule on this MI355X?  Register-resident random operands (f16x3-like
// hi / lo triples where the type allows), no memory traffic, 8 waves per CU on all 256 CUs: the sustained rate under the power
// limit of  v_mfma_f32_32x32x16_f16,  v_mfma_f32_16x16x32_f16,  v_mfma_f32_32x32x16_bf16  and  v_mfma_f32_16x16x32_bf16.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_variants tools/probes/mfma_variants_probe.hip && /tmp/mfma_variants
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline float rnd(unsigned& s) {
    float a = 0.f;
    for (int i = 0; i < 4; ++i) { s = s * 1664525u + 1013904223u; a += (float)(s >> 8) * (1.0f / 16777216.0f); }
    return (a - 2.0f) * 1.7320508f;
}

// VAR 0: 32x32x16 f16, 1: 16x16x32 f16, 2: 32x32x16 bf16, 3: 16x16x32 bf16.  Every variant issues the same FLOPs per round.
template <int VAR>
__global__ __launch_bounds__(512) void probe(int iters, int zero, float* out) {
    unsigned s = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 12345u;
    float av[2][8], bv[4][8];
    for (int t = 0; t < 2; ++t) for (int j = 0; j < 8; ++j) av[t][j] = zero ? 0.f : rnd(s);
    for (int t = 0; t < 4; ++t) for (int j = 0; j < 8; ++j) bv[t][j] = zero ? 0.f : 0.05f * rnd(s);
    float r = 0.f;
    if constexpr (VAR == 0 || VAR == 1) {
        f16x8 ah[2], al[2], bh[4], bl[4];
        for (int t = 0; t < 2; ++t) for (int j = 0; j < 8; ++j) { ah[t][j] = (_Float16)av[t][j]; al[t][j] = (_Float16)(av[t][j] - (float)ah[t][j]); }
        for (int t = 0; t < 4; ++t) for (int j = 0; j < 8; ++j) { bh[t][j] = (_Float16)bv[t][j]; bl[t][j] = (_Float16)(bv[t][j] - (float)bh[t][j]); }
        if constexpr (VAR == 0) {
            f32x16 acc[8];
            for (int t = 0; t < 8; ++t) for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int term = 0; term < 3; ++term)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < 4; ++n)
                            acc[m * 4 + n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 0 ? al[m] : ah[m], term == 1 ? bl[n] : bh[n], acc[m * 4 + n], 0, 0, 0);
                if (!zero) { ah[0] = ah[0] + al[1]; ah[1] = ah[1] - al[0]; }
            }
            for (int t = 0; t < 8; ++t) for (int j = 0; j < 16; ++j) r += acc[t][j];
        } else {
            // 16x16x32: a quarter of the outputs per instruction, twice the K: 2 instructions = 1 of the 32x32x16 in FLOPs (consecutive
            // instructions take different A operands, as a real tiling would)
            f32x4 acc[16];
            for (int t = 0; t < 16; ++t) for (int j = 0; j < 4; ++j) acc[t][j] = 0.f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int term = 0; term < 3; ++term)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < 4; ++n)
#pragma unroll
                            for (int h = 0; h < 2; ++h)
                                acc[(m * 4 + n) * 2 + h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(term == 0 ? al[(m + h) & 1] : ah[(m + h) & 1], term == 1 ? bl[n] : bh[n], acc[(m * 4 + n) * 2 + h], 0, 0, 0);
                if (!zero) { ah[0] = ah[0] + al[1]; ah[1] = ah[1] - al[0]; }
            }
            for (int t = 0; t < 16; ++t) for (int j = 0; j < 4; ++j) r += acc[t][j];
        }
    } else {
        bf16x8 ah[2], al[2], bh[4], bl[4];
        for (int t = 0; t < 2; ++t) for (int j = 0; j < 8; ++j) { ah[t][j] = (__bf16)av[t][j]; al[t][j] = (__bf16)(av[t][j] - (float)ah[t][j]); }
        for (int t = 0; t < 4; ++t) for (int j = 0; j < 8; ++j) { bh[t][j] = (__bf16)bv[t][j]; bl[t][j] = (__bf16)(bv[t][j] - (float)bh[t][j]); }
        if constexpr (VAR == 2) {
            f32x16 acc[8];
            for (int t = 0; t < 8; ++t) for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int term = 0; term < 3; ++term)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < 4; ++n)
                            acc[m * 4 + n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(term == 0 ? al[m] : ah[m], term == 1 ? bl[n] : bh[n], acc[m * 4 + n], 0, 0, 0);
                if (!zero) { ah[0] = ah[0] + al[1]; ah[1] = ah[1] - al[0]; }
            }
            for (int t = 0; t < 8; ++t) for (int j = 0; j < 16; ++j) r += acc[t][j];
        } else {
            f32x4 acc[16];
            for (int t = 0; t < 16; ++t) for (int j = 0; j < 4; ++j) acc[t][j] = 0.f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int term = 0; term < 3; ++term)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < 4; ++n)
#pragma unroll
                            for (int h = 0; h < 2; ++h)
                                acc[(m * 4 + n) * 2 + h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(term == 0 ? al[(m + h) & 1] : ah[(m + h) & 1], term == 1 ? bl[n] : bh[n], acc[(m * 4 + n) * 2 + h], 0, 0, 0);
                if (!zero) { ah[0] = ah[0] + al[1]; ah[1] = ah[1] - al[0]; }
            }
            for (int t = 0; t < 16; ++t) for (int j = 0; j < 4; ++j) r += acc[t][j];
        }
    }
    if (r == 123.456f) out[0] = r;
}

template <int VAR>
static void run(const char* name, int iters, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int zero = 1; zero >= 0; --zero) {
        hipLaunchKernelGGL(probe<VAR>, dim3(256), dim3(512), 0, 0, iters / 10, zero, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<VAR>, dim3(256), dim3(512), 0, 0, iters, zero, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fl = 256.0 * 8 * iters * 24 * 2.0 * 32 * 32 * 16;
        printf("%-28s %-8s: %8.2f ms  %7.1f TFLOP/s (%6.1f fp32-equivalent as three passes)\n", name, zero ? "zeros" : "random", ms, fl / (ms * 1e-3) / 1e12,
               fl / (ms * 1e-3) / 1e12 / 3);
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    float* out; hipMalloc(&out, 4);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("v_mfma_f32_32x32x16_f16", iters, out);
        run<1>("v_mfma_f32_16x16x32_f16", iters, out);
        run<2>("v_mfma_f32_32x32x16_bf16", iters, out);
        run<3>("v_mfma_f32_16x16x32_bf16", iters, out);
    }
    return 0;
}
