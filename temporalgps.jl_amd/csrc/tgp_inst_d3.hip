#define TGP_D 3
#include "tgp_inst.inc"
