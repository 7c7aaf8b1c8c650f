// The LDS-DMA implicit-GEMM f16x3 tiles on v_mfma_f32_16x16x32_f16 (round 5, "M16": the kernel and its comment live in
// conv_f16x3_kernel.h, the dispatch in conv_f16x3.hip).  A translation unit of its own: the instantiations compile beside the others.
#include "conv_f16x3_kernel.h"

// `base`: the tile enum of conv_f16x3.hip (see conv_f16x3_glds.hip)
int otvm_launch_m16_tile(int base, Conv3Args& a, hipStream_t s, int S) {
    switch (base) {
        case 0: return launch3<256, 256, 4, 2, false, true, true, 3, true>(a, s, S);
        case 1: return launch3<256, 128, 4, 2, false, true, true, 3, true>(a, s, S);
        case 2: return launch3<128, 128, 2, 2, false, true, true, 3, true>(a, s, S);
        case 3: return launch3<128, 64, 2, 2, false, true, true, 3, true>(a, s, S);
        case 4: return launch3<64, 64, 2, 2, false, true, true, 3, true>(a, s, S);
        case 5: return launch3<256, 64, 4, 1, false, true, true, 3, true>(a, s, S);
        case 6: return launch3<256, 32, 4, 1, false, true, true, 3, true>(a, s, S);
        case 7: return launch3<256, 128, 2, 2, false, true, true, 3, true>(a, s, S);
        case 8: return launch3<128, 256, 2, 2, false, true, true, 3, true>(a, s, S);
        case 10: return launch3<64, 64, 2, 2, true, true, true, 3, true>(a, s, S);
        case 11: return launch3<128, 64, 2, 2, true, true, true, 3, true>(a, s, S);
    }
    otvm_set_error("otvm_conv2d(f16x3): tile %d has no 16x16x32 form", base);
    return 1;
}
