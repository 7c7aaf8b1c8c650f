// Precision "f16" (round 5): the implicit-GEMM tiles with ONE MFMA pass on fp16-rounded operands (NPASS = 1 of the kernel template in
// conv_f16x3_kernel.h) -- a labelled reduced-precision mode, never the default.  Register-staged tiles here, the LDS-DMA
// tiles in conv_f16x3_p1g.hip; translation units of their own so that all four sets of instantiations compile side by side.
#include "conv_f16x3_kernel.h"

int otvm_launch_glds_tile_p1(int base, Conv3Args& a, hipStream_t s, int S);      // conv_f16x3_p1g.hip

// `tile`: the enum of conv_f16x3.hip (T256x256 = 0 ... T128x64D = 11, T256x256W4 = 13; 32 + t = the LDS-DMA form)
int otvm_launch_tile_p1(int tile, Conv3Args& a, hipStream_t s, int S) {
    if (tile >= 32) return otvm_launch_glds_tile_p1(tile - 32, a, s, S);
    switch (tile) {
        case 0: return launch3<256, 256, 4, 2, false, false, false, 1>(a, s, S);
        case 1: return launch3<256, 128, 4, 2, false, false, false, 1>(a, s, S);
        case 2: return launch3<128, 128, 2, 2, false, false, false, 1>(a, s, S);
        case 3: return launch3<128, 64, 2, 2, false, false, false, 1>(a, s, S);
        case 4: return launch3<64, 64, 2, 2, false, false, false, 1>(a, s, S);
        case 5: return launch3<256, 64, 4, 1, false, false, false, 1>(a, s, S);
        case 6: return launch3<256, 32, 4, 1, false, false, false, 1>(a, s, S);
        case 7: return launch3<256, 128, 2, 2, false, true, false, 1>(a, s, S);
        case 8: return launch3<128, 256, 2, 2, false, true, false, 1>(a, s, S);
        case 10: return launch3<64, 64, 2, 2, true, false, false, 1>(a, s, S);
        case 11: return launch3<128, 64, 2, 2, true, false, false, 1>(a, s, S);
    }
    otvm_set_error("otvm_conv2d(f16): tile %d has no single-pass form", tile);
    return 1;
}
