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

// Per-edge tensors of the training step (radial weights w[E,W], their gradient) may be stored as bf16 (opt-in): fp32
// arithmetic everywhere, round-to-nearest-even on the store.
__device__ __forceinline__ float matten_ld_edge(const void* base, int64_t idx, int is_bf16) {
    return is_bf16 ? __uint_as_float((uint32_t)reinterpret_cast<const uint16_t*>(base)[idx] << 16)
                   : reinterpret_cast<const float*>(base)[idx];
}
__device__ __forceinline__ uint16_t matten_f32_to_bf16(float v) {
    uint32_t u = __float_as_uint(v);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ void matten_st_edge(void* base, int64_t idx, float v, int is_bf16) {
    if (is_bf16) reinterpret_cast<uint16_t*>(base)[idx] = matten_f32_to_bf16(v);
    else reinterpret_cast<float*>(base)[idx] = v;
}
