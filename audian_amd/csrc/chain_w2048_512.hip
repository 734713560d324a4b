// chain_w2048_512.hip -- chain_fwd_kernel for nfft 2048 / hop 512 (see chain_shape.inc)
#define CHAIN_SHAPE_NFFT 2048
#define CHAIN_SHAPE_HOP 512
#include "chain_shape.inc"
