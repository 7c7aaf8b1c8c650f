// STM memory read (reference models/trimap/STM.py:140-163) as a fused, flash-style kernel.
//
//   p[m, q] = softmax over m of ( K[m,:] . Q[q,:] / sqrt(128) ),  m = T slots x hw positions
//   mem[q, :] = sum_m p[m, q] V[m, :]                               (V has 512 channels)
//
// The reference materialises p ([T*hw, hw] fp32, 1.33 GB at 1080p T=5) and runs two bmm.  Here one
// workgroup owns 64 queries and ONE memory slot, streams that slot's keys/values in tiles of 64
// positions with an online softmax (running max / running sum per query) and keeps its 64x512 output
// block in MFMA accumulators; a small combine kernel merges the per-slot partials.  Splitting by slot
// gives T x hw/64 workgroups (640 at 1080p/T=5) and lets the bank live as independent per-slot
// buffers (ring buffer on the host; no torch.cat re-copy per frame as in alpha/model.py:481-493).
//
// fp32 MFMA (v_mfma_f32_32x32x2_f32, exact fp32 fma chain).  Per KV tile and workgroup (4 waves):
//   S^T tile: wave (a,b) computes S[kv 32a.., q 32b..] = K Q^T over d=128 (64 MFMA), scaled, and
//             stores it transposed into LDS as Sl[q][kv] (b128 stores);
//   softmax : 256 threads = 64 queries x 4 kv-quarters; running max/sum in LDS;
//   PV      : wave w owns output channels [128w, 128w+128) for all 64 queries: 2x4 accumulator tiles
//             (128 AGPRs); A = P from LDS (b128, K-permuted), B = V straight from global/L2 (each
//             element is used by exactly one wave, so staging it in LDS would buy nothing).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int DK = 128, DV = 512, BQ = 64, BKV = 64;
constexpr int LDQ = DK + 4;      // 132 floats: b128 reads of 16 distinct rows hit 16 distinct slots
constexpr int LDS_S = BKV + 4;   // 68

struct MemArgs {
    const float* q; int q_ld;
    const float* keys[8]; const float* vals[8];
    int T, hw, slot0;
    float* part_o;     // [T][hw][512]
    float* part_ml;    // [T][hw][2]
};

__global__ __launch_bounds__(256) void memory_read_partial_kernel(const MemArgs p) {
    __shared__ __attribute__((aligned(16))) float Ql[BQ * LDQ];
    __shared__ __attribute__((aligned(16))) float Kl[BKV * LDQ];
    __shared__ __attribute__((aligned(16))) float Sl[BQ * LDS_S];
    __shared__ float red[4 * BQ];
    __shared__ float m_run[BQ], l_run[BQ], alpha_l[BQ];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q0 = blockIdx.x * BQ, slot = p.slot0 + blockIdx.y;
    const float* __restrict__ Kg = p.keys[blockIdx.y];
    const float* __restrict__ Vg = p.vals[blockIdx.y];
    const int hw = p.hw;

    // Q tile -> LDS (rows beyond hw zero-filled)
    for (int i = tid; i < BQ * (DK / 4); i += 256) {
        const int r = i / (DK / 4), c = (i - r * (DK / 4)) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (q0 + r < hw) v = *reinterpret_cast<const f32x4*>(p.q + (int64_t)(q0 + r) * p.q_ld + c);
        *reinterpret_cast<f32x4*>(&Ql[r * LDQ + c]) = v;
    }
    if (tid < BQ) { m_run[tid] = -__builtin_huge_valf(); l_run[tid] = 0.f; }

    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    const int sa = wave >> 1, sb = wave & 1;            // S sub-tile of this wave: kv 32sa.., q 32sb..
    const int frow = lane & 31, fk = (lane >> 5) * 4;
    const float scale = 1.0f / sqrtf((float)DK);       // p / math.sqrt(D_e)  (STM.py:154)
    const int dv0 = wave * 128;
    const int ntiles = (hw + BKV - 1) / BKV;

    for (int t = 0; t < ntiles; ++t) {
        const int kv0 = t * BKV;
        __syncthreads();                                // previous tile: Kl / Sl readers are done
        for (int i = tid; i < BKV * (DK / 4); i += 256) {
            const int r = i / (DK / 4), c = (i - r * (DK / 4)) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (kv0 + r < hw) v = *reinterpret_cast<const f32x4*>(Kg + (int64_t)(kv0 + r) * DK + c);
            *reinterpret_cast<f32x4*>(&Kl[r * LDQ + c]) = v;
        }
        __syncthreads();

        // ---- S = K Q^T (this wave's 32x32 block), D[i=kv][j=q]
        f32x16 s;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll 4
        for (int j = 0; j < DK / 8; ++j) {
            const f32x4 ka = *reinterpret_cast<const f32x4*>(&Kl[(sa * 32 + frow) * LDQ + 8 * j + fk]);
            const f32x4 qb = *reinterpret_cast<const f32x4*>(&Ql[(sb * 32 + frow) * LDQ + 8 * j + fk]);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.x, qb.x, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.y, qb.y, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.z, qb.z, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.w, qb.w, s, 0, 0, 0);
        }
        // lane: col q = lane&31, rows kv = (e&3) + 8*(e>>2) + 4*(lane>>5); store transposed Sl[q][kv]
        {
            const int qq = sb * 32 + (lane & 31);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int kvl = sa * 32 + 8 * g + fk;
                f32x4 v = {s[4 * g] * scale, s[4 * g + 1] * scale, s[4 * g + 2] * scale, s[4 * g + 3] * scale};
                const float ninf = -__builtin_huge_valf();
                if (kv0 + kvl + 0 >= hw) v.x = ninf;
                if (kv0 + kvl + 1 >= hw) v.y = ninf;
                if (kv0 + kvl + 2 >= hw) v.z = ninf;
                if (kv0 + kvl + 3 >= hw) v.w = ninf;
                *reinterpret_cast<f32x4*>(&Sl[qq * LDS_S + kvl]) = v;
            }
        }
        __syncthreads();

        // ---- online softmax over the memory axis: thread = (query, kv quarter)
        {
            const int qq = tid & 63, part = tid >> 6;
            f32x4 v[4];
            float mx = -__builtin_huge_valf();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] = *reinterpret_cast<const f32x4*>(&Sl[qq * LDS_S + part * 16 + 4 * i]);
                mx = fmaxf(mx, fmaxf(fmaxf(v[i].x, v[i].y), fmaxf(v[i].z, v[i].w)));
            }
            red[part * BQ + qq] = mx;
            __syncthreads();
            const float m_old = m_run[qq];
            const float m_tile = fmaxf(fmaxf(red[qq], red[BQ + qq]), fmaxf(red[2 * BQ + qq], red[3 * BQ + qq]));
            const float m_new = fmaxf(m_old, m_tile);
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i].x = expf(v[i].x - m_new); v[i].y = expf(v[i].y - m_new);
                v[i].z = expf(v[i].z - m_new); v[i].w = expf(v[i].w - m_new);
                sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
                *reinterpret_cast<f32x4*>(&Sl[qq * LDS_S + part * 16 + 4 * i]) = v[i];
            }
            __syncthreads();                             // all m_old / red reads done
            red[part * BQ + qq] = sum;
            __syncthreads();
            if (part == 0) {
                const float al = expf(m_old - m_new);  // exp(-inf) = 0 on the first tile
                alpha_l[qq] = al;
                l_run[qq] = l_run[qq] * al + ((red[qq] + red[BQ + qq]) + (red[2 * BQ + qq] + red[3 * BQ + qq]));
                m_run[qq] = m_new;
            }
            __syncthreads();
        }

        // ---- rescale O and accumulate P V
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float al = alpha_l[a * 32 + (e & 3) + 8 * (e >> 2) + fk];
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b][e] *= al;
            }
#pragma unroll 2
        for (int j = 0; j < BKV / 8; ++j) {
            f32x4 pa[2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
                pa[a] = *reinterpret_cast<const f32x4*>(&Sl[(a * 32 + frow) * LDS_S + 8 * j + fk]);
            float vb[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int kv = kv0 + 8 * j + fk + i;
                kv = kv < hw ? kv : hw - 1;              // masked rows have p == 0
                const float* vr = Vg + (int64_t)kv * DV + dv0 + frow;
#pragma unroll
                for (int b = 0; b < 4; ++b) vb[b][i] = vr[32 * b];
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[a].x, vb[b][0], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[a].y, vb[b][1], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[a].z, vb[b][2], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[a].w, vb[b][3], acc[a][b], 0, 0, 0);
                }
        }
    }
    __syncthreads();
    // partial results: un-normalised O, running max and sum
    float* po = p.part_o + ((int64_t)slot * hw) * DV;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int qq = q0 + a * 32 + (e & 3) + 8 * (e >> 2) + fk;
            if (qq < hw) {
#pragma unroll
                for (int b = 0; b < 4; ++b) po[(int64_t)qq * DV + dv0 + 32 * b + frow] = acc[a][b][e];
            }
        }
    if (tid < BQ && q0 + tid < hw) {
        float* ml = p.part_ml + ((int64_t)slot * hw + q0 + tid) * 2;
        ml[0] = m_run[tid];
        ml[1] = l_run[tid];
    }
}

__global__ void memory_read_combine_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml, int T, int hw,
                                           float* __restrict__ out, int out_ld) {
    const int64_t total = (int64_t)hw * (DV / 4);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(i / (DV / 4)), c = (int)(i - (int64_t)q * (DV / 4)) * 4;
        float M = -__builtin_huge_valf();
        for (int t = 0; t < T; ++t) M = fmaxf(M, part_ml[((int64_t)t * hw + q) * 2]);
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        float l = 0.f;
        for (int t = 0; t < T; ++t) {
            const float w = expf(part_ml[((int64_t)t * hw + q) * 2] - M);
            l += w * part_ml[((int64_t)t * hw + q) * 2 + 1];
            o += w * *reinterpret_cast<const f32x4*>(part_o + ((int64_t)t * hw + q) * DV + c);
        }
        *reinterpret_cast<f32x4*>(out + (int64_t)q * out_ld + c) = o * (1.f / l);
    }
}

}  // namespace

int otvm_memory_read_combine(const float* part_o, const float* part_ml, int T, int hw, float* out, int out_ld, void* stream) {
    const int64_t total = (int64_t)hw * (DV / 4);
    hipLaunchKernelGGL(memory_read_combine_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, part_o,
                       part_ml, T, hw, out, out_ld);
    OTVM_CHECK_LAUNCH("otvm_memory_read(combine)");
    return 0;
}

int otvm_memory_read_f16x3_partials(int T, int hw);     // memory_read_f16x3.hip

extern "C" int64_t otvm_memory_read_ws_bytes(int hw, int T) {
    // one partial (O block + running max / sum) per slot here, per memory-axis chunk in the f16x3 kernel
    const int np3 = otvm_memory_read_f16x3_partials(T, hw);
    return (int64_t)(np3 > T ? np3 : T) * hw * (DV + 2) * sizeof(float);
}

extern "C" int otvm_memory_read(const float* q_key, int q_ld, const float* const* keys, const float* const* vals, int T,
                                int hw, float* out, int out_ld, void* ws, void* stream) {
    OTVM_REQUIRE(T >= 1 && T <= 4096, "otvm_memory_read: T=%d out of range [1,4096]", T);
    OTVM_REQUIRE(q_key && keys && vals && out && ws && hw > 0, "otvm_memory_read: bad arguments");
    OTVM_REQUIRE(q_ld % 4 == 0 && out_ld % 4 == 0, "otvm_memory_read: views must be 16-byte aligned");
    MemArgs a;
    a.q = q_key; a.q_ld = q_ld; a.T = T; a.hw = hw;
    a.part_o = (float*)ws;
    a.part_ml = a.part_o + (int64_t)T * hw * DV;
    hipStream_t s = (hipStream_t)stream;
    for (int s0 = 0; s0 < T; s0 += 8) {            // 8 slots per launch (see memory_read_f16x3.hip)
        const int n = T - s0 < 8 ? T - s0 : 8;
        a.slot0 = s0;
        for (int t = 0; t < 8; ++t) { a.keys[t] = t < n ? keys[s0 + t] : nullptr; a.vals[t] = t < n ? vals[s0 + t] : nullptr; }
        hipLaunchKernelGGL(memory_read_partial_kernel, dim3(otvm_ceil_div(hw, BQ), n), dim3(256), 0, s, a);
    }
    OTVM_CHECK_LAUNCH("otvm_memory_read");
    return otvm_memory_read_combine(a.part_o, a.part_ml, T, hw, out, out_ld, stream);
}
