// Shared helpers for the gfx950 kernels of libmatten_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "matten_hip.h"

#define MATTEN_LAUNCH_CHECK()                              \
    do {                                                   \
        if (hipGetLastError() != hipSuccess) return MATTEN_ELAUNCH; \
    } while (0)

static inline int64_t matten_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
