#define TGP_D 4
#include "tgp_inst.inc"
