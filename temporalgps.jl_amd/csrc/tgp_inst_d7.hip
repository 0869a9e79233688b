#define TGP_D 7
#include "tgp_inst.inc"
