// LDS access-pattern probe (gfx950): cycles per wave-wide ds_read_b128 / ds_write_b64 / ds_write_b128 for the row-strided
// layouts of conv_f16x3.hip, as a function of the row stride.  One wave per workgroup and four waves per workgroup (one per
// SIMD, sharing the CU's LDS).
//   hipcc --offload-arch=gfx950 -O3 -o lds_probe tools/probes/lds_probe.hip && ./lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// mode 0: fragment read b128: lane -> row (l & 31), byte offset 16 * (l >> 5)
// mode 1: A staging write b64: lane -> row (l >> 3), byte offset 8 * (l & 7)
// mode 2: B staging write b128: lane -> row (l >> 2), byte offset 16 * (l & 3)
// mode 3: fragment read b128, contiguous (fragment-major: lane * 16 bytes)
// mode 4: epilogue patch write b32: row ((l >> 5) * 4) * 144 bytes + (l & 31) * 4
__global__ void probe(int mode, int stride, int iters, unsigned long long* out, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned addr;
    if (mode == 0) addr = (l & 31) * stride + 16 * (l >> 5);
    else if (mode == 1) addr = (l >> 3) * stride + 8 * (l & 7);
    else if (mode == 2) addr = (l >> 2) * stride + 16 * (l & 3);
    else if (mode == 3) addr = l * 16;
    else addr = ((l >> 5) * 4) * stride + (l & 31) * 4;
    addr += w * 16384;                                   // waves use disjoint regions
    for (int i = threadIdx.x; i < 16384 * 4 / 4; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = 1.0f;
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 v[8];
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (mode == 0 || mode == 3) {
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(v[j]) : "v"(addr + (j & 1) * 32));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += v[j];
        } else if (mode == 1) {
            f32x2 d = {acc.x, acc.y};
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(d) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (mode == 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(acc) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(acc.x) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    const unsigned long long t1 = clock64();
    if (l == 0) out[blockIdx.x * 4 + w] = t1 - t0;
    if (acc.x == 123.456f) sink[0] = acc.y;
}

int main() {
    unsigned long long* d_out; float* d_sink;
    hipMalloc(&d_out, 4096 * 8); hipMalloc(&d_sink, 64);
    const int iters = 2000;
    const char* names[5] = {"frag read b128 (row, 16B half)", "A write b64 (8 lanes/row)", "B write b128 (4 lanes/row)", "frag read b128 contiguous", "patch write b32"};
    for (int waves : {1, 4}) {
        for (int mode = 0; mode < 5; ++mode) {
            std::vector<int> strides = {64, 80, 96, 112, 128, 144, 160, 176, 208, 272};
            if (mode == 3) strides = {0};
            if (mode == 4) strides = {144, 132, 136, 140, 148};
            for (int s : strides) {
                hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
                hipLaunchKernelGGL(probe, dim3(1), dim3(64 * waves), 65536, 0, mode, s, iters, d_out, d_sink);
                hipDeviceSynchronize();
                unsigned long long h[4];
                hipMemcpy(h, d_out, 32, hipMemcpyDeviceToHost);
                printf("%d wave(s)  %-32s stride %3d B: %.1f clocks per instruction (wave 0)\n", waves, names[mode], s, (double)h[0] / (iters * 8.0));
            }
        }
    }
    return 0;
}
