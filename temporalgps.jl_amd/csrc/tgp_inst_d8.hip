#define TGP_D 8
#include "tgp_inst.inc"
