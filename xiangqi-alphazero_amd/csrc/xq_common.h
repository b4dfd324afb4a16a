// xq_common.h -- host-side helpers shared by the ABI translation units.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/xq_hip.h"

namespace xq {

extern thread_local hipError_t g_last_error;

inline int check(hipError_t e) {
    if (e != hipSuccess) {
        g_last_error = e;
        return XQ_ERR_HIP;
    }
    return XQ_OK;
}

inline int launch_status() { return check(hipGetLastError()); }

}  // namespace xq

#define XQ_TRY(expr)                              \
    do {                                          \
        int _rc = xq::check((expr));              \
        if (_rc != XQ_OK) return _rc;             \
    } while (0)
