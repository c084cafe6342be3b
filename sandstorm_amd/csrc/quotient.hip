// quotient.hip — constraint-program evaluation over the LDE domain (row Q1 of SURVEY.md §8a).
#include <hip/hip_runtime.h>
#include "../../include/sandstorm_hip.h"

extern "C" {
ss_status ss_eval_quotient(ss_ctx *, const ss_air_program *, const uint64_t *const *, uint32_t, uint32_t, uint32_t,
                           const uint64_t[4], uint64_t *) { return SS_ERR_UNSUPPORTED; }
}
