// ccsp_common.h -- host-side plumbing shared by the translation units of libccsp.so
#pragma once
#include <hip/hip_runtime.h>
#include "ccsp_rules.h"

void ccsp_set_hip_error(hipError_t e, const char *what);

#define CCSP_HIPCHK(expr)                                                     \
    do {                                                                      \
        hipError_t e_ = (expr);                                               \
        if (e_ != hipSuccess) { ccsp_set_hip_error(e_, #expr); return CCSP_EHIP; } \
    } while (0)

// device allocation: hipErrorOutOfMemory is reported as CCSP_ENOMEM, anything else as CCSP_EHIP
int ccsp_alloc_status(hipError_t e, const char *what);
#define CCSP_ALLOCCHK(expr)                                                   \
    do {                                                                      \
        const int rc_ = ccsp_alloc_status((expr), #expr);                     \
        if (rc_ != CCSP_OK) return rc_;                                       \
    } while (0)

// a device buffer that is freed on every way out of a host function
struct ccsp_devbuf {
    void *p = nullptr;
    ccsp_devbuf() = default;
    ccsp_devbuf(const ccsp_devbuf &) = delete;
    ccsp_devbuf &operator=(const ccsp_devbuf &) = delete;
    ~ccsp_devbuf() { if (p) (void)hipFree(p); }
};

// the ray table lives in device global memory (constant data); kernels stage it into LDS
static __device__ const ccsp_ray_table CCSP_RAYS_DEV = ccsp_make_rays();

__device__ __forceinline__ void ccsp_load_rays_to_lds(uint64_t *lds /* [294] */, int tid, int nthreads) {
    const uint64_t *src = &CCSP_RAYS_DEV.ray[0][0];
    for (int i = tid; i < CCSP_NCELL * 6; i += nthreads) lds[i] = src[i];
}

static __device__ const ccsp_line_tables CCSP_LINES_DEV = ccsp_make_lines();

__device__ __forceinline__ void ccsp_load_lines_to_lds(ccsp_line_tables *lds, int tid, int nthreads) {
    static_assert(sizeof(ccsp_line_tables) % 4 == 0, "line tables are copied as dwords");
    const uint32_t *src = reinterpret_cast<const uint32_t *>(&CCSP_LINES_DEV);
    uint32_t *dst = reinterpret_cast<uint32_t *>(lds);
    for (int i = tid; i < (int)(sizeof(ccsp_line_tables) / 4); i += nthreads) dst[i] = src[i];
}

// a record moves between memory and registers as two 16-byte accesses
__device__ __forceinline__ ccsp_sr ccsp_load_sr(const ccsp_state *p) {
    const ulonglong2 *q = reinterpret_cast<const ulonglong2 *>(p);
    const ulonglong2 lo = q[0], hi = q[1];
    ccsp_sr r; r.occ0 = lo.x; r.occ1 = lo.y; r.a = hi.x; r.b = hi.y;
    return r;
}
__device__ __forceinline__ void ccsp_store_sr(ccsp_state *p, const ccsp_sr &r) {
    ulonglong2 *q = reinterpret_cast<ulonglong2 *>(p);
    q[0] = make_ulonglong2(r.occ0, r.occ1);
    q[1] = make_ulonglong2(r.a, r.b);
}
