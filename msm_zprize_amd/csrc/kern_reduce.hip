// Device code of one kernel family for the curve selected with -DMSMZ_CURVE (see instantiate.h).
#include "instantiate.h"
namespace msmz {
#define X(F, Fr) MSMZ_INST_REDUCE(F, Fr, MSMZ_DEFINE)
MSMZ_WEIERSTRASS_FIELDS(X)
#undef X
#define X(F, Fr) MSMZ_INST_REDUCE_TE(F, Fr, MSMZ_DEFINE)
MSMZ_TE_FIELDS(X)
#undef X
}  // namespace msmz
