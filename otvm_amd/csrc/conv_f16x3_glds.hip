// The implicit-GEMM f16x3 tiles with LDS-DMA weight stages (round 5; the kernel and the long comment live in conv_f16x3_kernel.h,
// the dispatch in conv_f16x3.hip).  A translation unit of its own so that the two sets of tile instantiations compile side by side.
#include "conv_f16x3_kernel.h"

// `base`: the tile enum of conv_f16x3.hip (T256x256 = 0, T256x128, T128x128, T128x64, T64x64, T256x64, T256x32, T256x128W4,
// T128x256W4, -, T64x64D = 10, T128x64D = 11)
int otvm_launch_glds_tile(int base, Conv3Args& a, hipStream_t s, int S) {
    switch (base) {
        case 0: return launch3<256, 256, 4, 2, false, true, true>(a, s, S);
        case 1: return launch3<256, 128, 4, 2, false, true, true>(a, s, S);
        case 2: return launch3<128, 128, 2, 2, false, true, true>(a, s, S);
        case 3: return launch3<128, 64, 2, 2, false, true, true>(a, s, S);
        case 4: return launch3<64, 64, 2, 2, false, true, true>(a, s, S);
        case 5: return launch3<256, 64, 4, 1, false, true, true>(a, s, S);
        case 6: return launch3<256, 32, 4, 1, false, true, true>(a, s, S);
        case 7: return launch3<256, 128, 2, 2, false, true, true>(a, s, S);
        case 8: return launch3<128, 256, 2, 2, false, true, true>(a, s, S);
        case 10: return launch3<64, 64, 2, 2, true, true, true>(a, s, S);
        case 11: return launch3<128, 64, 2, 2, true, true, true>(a, s, S);
    }
    otvm_set_error("otvm_conv2d(f16x3): tile %d has no LDS-DMA form", base);
    return 1;
}
