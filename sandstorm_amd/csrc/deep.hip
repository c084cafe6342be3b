// deep.hip — out-of-domain evaluation and DEEP composition (row D1 of SURVEY.md §8a).
#include <hip/hip_runtime.h>
#include "../../include/sandstorm_hip.h"

extern "C" {
// Kernels land next; until then these fail loudly (no CPU fallback).
ss_status ss_ood_eval(ss_ctx *, const uint64_t *const *, uint32_t, uint32_t, const uint32_t *, const uint32_t *,
                      uint32_t, const uint64_t[4], uint64_t *) { return SS_ERR_UNSUPPORTED; }
ss_status ss_poly_eval(ss_ctx *, const uint64_t *const *, uint32_t, uint32_t, const uint64_t[4], uint64_t *) {
    return SS_ERR_UNSUPPORTED;
}
ss_status ss_deep_compose(ss_ctx *, const uint64_t *const *, uint32_t, const uint64_t *const *, uint32_t, uint32_t,
                          uint32_t, const uint64_t[4], const uint32_t *, const uint32_t *, uint32_t,
                          const uint64_t *, const uint64_t *, const uint64_t *, const uint64_t *, const uint64_t[4],
                          uint64_t *) { return SS_ERR_UNSUPPORTED; }
}
