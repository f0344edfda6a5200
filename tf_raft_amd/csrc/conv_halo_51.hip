// Instantiations of the halo-tiled fp32-MFMA convolution for 5x1 kernels (see conv_halo.h).
#include "conv_halo.h"

int raft_launch_conv_halo_5x1(const ConvArgs &a, int th, int tn, int epi, hipStream_t s) {
    return raft_launch_conv_halo_epi<5, 1>(a, th, tn, epi, s);
}
