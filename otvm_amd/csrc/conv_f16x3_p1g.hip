// Precision "f16" (round 5): the LDS-DMA implicit-GEMM tiles with one MFMA pass (see conv_f16x3_p1.hip).
#include "conv_f16x3_kernel.h"

int otvm_launch_glds_tile_p1(int base, Conv3Args& a, hipStream_t s, int S) {
    switch (base) {
        case 0: return launch3<256, 256, 4, 2, false, true, true, 1>(a, s, S);
        case 1: return launch3<256, 128, 4, 2, false, true, true, 1>(a, s, S);
        case 2: return launch3<128, 128, 2, 2, false, true, true, 1>(a, s, S);
        case 3: return launch3<128, 64, 2, 2, false, true, true, 1>(a, s, S);
        case 4: return launch3<64, 64, 2, 2, false, true, true, 1>(a, s, S);
        case 5: return launch3<256, 64, 4, 1, false, true, true, 1>(a, s, S);
        case 6: return launch3<256, 32, 4, 1, false, true, true, 1>(a, s, S);
        case 7: return launch3<256, 128, 2, 2, false, true, true, 1>(a, s, S);
        case 8: return launch3<128, 256, 2, 2, false, true, true, 1>(a, s, S);
        case 10: return launch3<64, 64, 2, 2, true, true, true, 1>(a, s, S);
        case 11: return launch3<128, 64, 2, 2, true, true, true, 1>(a, s, S);
    }
    otvm_set_error("otvm_conv2d(f16): tile %d has no LDS-DMA form", base);
    return 1;
}
