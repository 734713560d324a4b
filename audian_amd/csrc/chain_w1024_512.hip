// chain_w1024_512.hip -- chain_fwd_kernel for nfft 1024 / hop 512 (see chain_shape.inc)
#define CHAIN_SHAPE_NFFT 1024
#define CHAIN_SHAPE_HOP 512
#include "chain_shape.inc"
