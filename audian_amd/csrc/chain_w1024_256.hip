// chain_w1024_256.hip -- chain_fwd_kernel for nfft 1024 / hop 256 (see chain_shape.inc)
#define CHAIN_SHAPE_NFFT 1024
#define CHAIN_SHAPE_HOP 256
// measured per window (tools/libs_sweep.py, profiles/r05h_sweep.log): with the stage's LDS reads pinned in front of its
// arithmetic this window runs 1.9 % faster (hipcc otherwise issues them one at a time here), the others 0.1 - 2.2 % slower
#define STOCKHAM_LOADS_FIRST
#include "chain_shape.inc"
