#define TGP_D 6
#include "tgp_inst.inc"
