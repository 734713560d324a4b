// chain_w256_128.hip -- chain_fwd_kernel for nfft 256 / hop 128 (see chain_shape.inc)
#define CHAIN_SHAPE_NFFT 256
#define CHAIN_SHAPE_HOP 128
#include "chain_shape.inc"
