#define TGP_D 2
#include "tgp_inst.inc"
