// Plan replay below the ABI: a caller-owned table of (entry point, packed arguments) rows is walked by ONE call, so a
// step whose launch sequence is static (fixed shapes, fixed buffers) costs the host one loop in C instead of ~750
// Python -> ctypes round trips.  No state is kept here: the table, the buffers it points to and the streams belong
// to the caller (regda_amd/plan.py records the table from an ordinary eager step).
#include <string.h>

#include "common.h"

static inline float u2f(uint64_t v) { uint32_t b = (uint32_t)v; float f; memcpy(&f, &b, 4); return f; }
static inline double u2d(uint64_t v) { double d; memcpy(&d, &v, 8); return d; }

struct PlanFn { const char* name; int (*fn)(const uint64_t*); int nargs; };
#include "plan_thunks.inc"
static const int N_PLAN_FNS = (int)(sizeof(PLAN_FNS) / sizeof(PLAN_FNS[0]));

extern "C" int rgda_plan_fn_count(void) { return N_PLAN_FNS; }

extern "C" int rgda_plan_fn_id(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < N_PLAN_FNS; ++i)
        if (!strcmp(PLAN_FNS[i].name, name)) return i;
    return -1;
}

extern "C" int rgda_plan_run(const rgda_plan_entry* entries, int n, int* failed_index) {
    if (n < 0 || (n > 0 && !entries)) return RGDA_ERR_ARG;
    for (int i = 0; i < n; ++i) {
        const rgda_plan_entry& e = entries[i];
        if (e.fn < 0 || e.fn >= N_PLAN_FNS || e.nargs != PLAN_FNS[e.fn].nargs) {
            if (failed_index) *failed_index = i;
            return RGDA_ERR_ARG;
        }
        const int rc = PLAN_FNS[e.fn].fn(e.args);
        if (rc != RGDA_OK) {
            if (failed_index) *failed_index = i;
            return rc;
        }
    }
    return RGDA_OK;
}
