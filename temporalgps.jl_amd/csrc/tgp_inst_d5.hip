#define TGP_D 5
#include "tgp_inst.inc"
