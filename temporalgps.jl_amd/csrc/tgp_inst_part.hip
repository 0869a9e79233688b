// One PART (TGP_PART, see tgp_inst.inc) of a fully inlined build for state dimension TGP_D, both given on the command line.
// The inlined d = 7, 8 builds are split this way so that their ~13 groups of kernels compile side by side; like the
// d = 5, 6 inlined builds they are only used after reproducing the out-of-line build in the run-time variant check.
#define TGP_NS tgp_i
#define TGP_BIG_D 99
#define TGP_NO_GROUP
#define TGP_TABLE_SUFFIX _i
#define TGP_AD_SCAN_FROM_SAFE
#include "tgp_inst.inc"
