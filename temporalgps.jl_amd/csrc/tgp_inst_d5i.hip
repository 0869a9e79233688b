// d = 5 with EVERYTHING inlined (matrices in registers / AGPRs): several times faster than the out-of-line build of
// tgp_inst_d5.hip, but it is the spill-heavy form from which silently wrong results were measured (d = 6). It is
// therefore only used after it has reproduced the out-of-line build's results in the run-time variant check.
#define TGP_NS tgp_i
#define TGP_BIG_D 99
#define TGP_NO_GROUP
#define TGP_TABLE_SUFFIX _i
#define TGP_D 5
#include "tgp_inst.inc"
