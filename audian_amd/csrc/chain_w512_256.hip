// chain_w512_256.hip -- chain_fwd_kernel for nfft 512 / hop 256 (see chain_shape.inc)
#define CHAIN_SHAPE_NFFT 512
#define CHAIN_SHAPE_HOP 256
#include "chain_shape.inc"
