#define TGP_D 1
#include "tgp_inst.inc"
