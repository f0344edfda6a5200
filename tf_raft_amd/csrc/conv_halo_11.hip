// Instantiations of the halo-tiled fp32-MFMA convolution for 1x1 kernels (see conv_halo.h).
#include "conv_halo.h"

int raft_launch_conv_halo_1x1(const ConvArgs &a, int th, int tn, int epi, hipStream_t s) {
    return raft_launch_conv_halo_epi<1, 1>(a, th, tn, epi, s);
}
