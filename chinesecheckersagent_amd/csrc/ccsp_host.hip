// ccsp_host.cpp -- host-only pieces of libccsp.so: error reporting, record packing.
#include <cstdio>
#include <cstring>
#include "ccsp_common.h"

static thread_local char g_err[256] = "";

void ccsp_set_hip_error(hipError_t e, const char *what) {
    snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
}

int ccsp_alloc_status(hipError_t e, const char *what) {
    if (e == hipSuccess) return CCSP_OK;
    ccsp_set_hip_error(e, what);
    (void)hipGetLastError();                              // the failed allocation must not poison later calls
    return e == hipErrorOutOfMemory ? CCSP_ENOMEM : CCSP_EHIP;
}

extern "C" {

const char *ccsp_last_hip_error(void) { return g_err; }
const char *ccsp_version(void) { return "ccsp 0.1 (gfx950)"; }

int ccsp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int ccsp_pack_states(const uint8_t *pos12, const uint8_t *last4, int n, ccsp_state *out) {
    if (n < 0 || (n > 0 && (!pos12 || !out))) return CCSP_EINVAL;
    for (int i = 0; i < n; i++) {
        ccsp_state s;
        s.occ[0] = s.occ[1] = 0;
        for (int p = 0; p < 2; p++)
            for (int k = 0; k < 6; k++) {
                const uint8_t c = pos12[i * 12 + p * 6 + k];
                if (c >= CCSP_NCELL) return CCSP_EINVAL;
                s.pos[p][k] = c;
                s.occ[p] |= 1ULL << c;
            }
        for (int k = 0; k < 4; k++) s.last[k] = last4 ? last4[i * 4 + k] : (uint8_t)CCSP_NO_MOVE;
        out[i] = s;
    }
    return CCSP_OK;
}

}  // extern "C"
