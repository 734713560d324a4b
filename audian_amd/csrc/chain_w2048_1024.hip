// chain_w2048_1024.hip -- chain_fwd_kernel for nfft 2048 / hop 1024 (see chain_shape.inc)
#define CHAIN_SHAPE_NFFT 2048
#define CHAIN_SHAPE_HOP 1024
#include "chain_shape.inc"
