// chain_w1024_256.hip -- chain_fwd_kernel for nfft 1024 / hop 256 (see chain_shape.inc)
#define CHAIN_SHAPE_NFFT 1024
#define CHAIN_SHAPE_HOP 256
#include "chain_shape.inc"
