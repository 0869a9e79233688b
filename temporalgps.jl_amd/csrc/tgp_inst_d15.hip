// d = 15: out-of-line building blocks, rolled loops (matrices in the lane's private memory)
#define TGP_NO_UNROLL
#define TGP_D 15
#include "tgp_inst.inc"
