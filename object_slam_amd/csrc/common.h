// Shared host-side helpers for the gfx950 hot-path library (product code; never includes oracle/).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/oslam_hip.h"

namespace oslam {

void set_error(const char* fmt, ...);

#define OSLAM_HIP_CHECK(expr)                                                              \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            oslam::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return OSLAM_E_HIP;                                                            \
        }                                                                                  \
    } while (0)

static inline int div_up(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

}  // namespace oslam
