// Device code of one kernel family (see instantiate.h).
#include "instantiate.h"
namespace msmz {
#define X(F, Fr) MSMZ_INST_GEN(F, Fr, MSMZ_DEFINE)
MSMZ_WEIERSTRASS_FIELDS(X)
#undef X
}  // namespace msmz
