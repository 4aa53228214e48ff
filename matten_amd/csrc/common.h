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

// Which (species, row range) does workgroup `b` own, when the rows of every species (seg[s] .. seg[s+1]) are cut into
// blocks of `rpb` rows and the blocks are numbered species-major?  Cooperative: every thread of the workgroup calls it
// (THREADS = blockDim.x, a power of two), `sh` is >= THREADS + 4 ints of LDS that nobody else uses until it returns.
// A chain of dependent global loads -- one per species, up to the block's species -- cost a batch with 73 species more
// than the whole rest of a small launch (12 us per call at 130 rows); here the species' block counts are fetched in
// parallel and scanned in LDS: one memory round trip + log2(THREADS) barriers per THREADS species.
template <int THREADS>
__device__ __forceinline__ bool matten_block_species(const int32_t* __restrict__ seg, int n_species, int rpb, int b, int* sh,
                                                     int& s_out, int& lo, int& hi) {
    const int t = threadIdx.x;
    int carry = 0;
    if (t == 0) sh[THREADS] = -1;
    for (int base = 0; base < n_species; base += THREADS) {
        const int s = base + t;
        int beg = 0, end = 0;
        if (s < n_species) beg = seg[s], end = seg[s + 1];
        const int nb = (end - beg + rpb - 1) / rpb;
        sh[t] = nb;
        __syncthreads();
        for (int off = 1; off < THREADS; off <<= 1) {   // inclusive scan (Hillis-Steele)
            const int v = t >= off ? sh[t - off] : 0;
            __syncthreads();
            sh[t] += v;
            __syncthreads();
        }
        const int incl = carry + sh[t], total = sh[THREADS - 1];
        if (nb > 0 && b >= incl - nb && b < incl) {
            sh[THREADS] = s;
            sh[THREADS + 1] = beg + (b - (incl - nb)) * rpb;
            sh[THREADS + 2] = end;
        }
        __syncthreads();
        if (sh[THREADS] >= 0) break;
        carry += total;
        __syncthreads();
    }
    __syncthreads();
    s_out = sh[THREADS];
    if (s_out < 0) return false;
    lo = sh[THREADS + 1];
    hi = min(sh[THREADS + 2], lo + rpb);
    __syncthreads();   // the scratch may be reused by the caller
    return true;
}
