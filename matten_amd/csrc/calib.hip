// libmatten_lab.so (make lab; include/matten_lab.h) -- NOT part of libmatten_hip.so.
// Calibration kernels for bench.py (SURVEY.md 8d "measurement"): two fixed pieces of work whose cost depends on the box only,
// timed next to the benchmark so that numbers taken on different pool machines (or at different DVFS states of one
// machine: MI355X clocks to its power budget, /opt/skills/guides/MI355X_MICROARCH.md "DVFS give-back") can be compared.
//   matten_calib_valu  a register-only chain of fp32 FMAs at 8 waves per SIMD on every CU: its duration gives the
//                      sustained VALU issue rate (ns per wave64 instruction and SIMD); block 0 also samples the shader
//                      clock (s_memtime) against the constant 100 MHz reference (s_memrealtime), i.e. the effective sclk
//                      under a full VALU load.
//   matten_calib_copy  a 16-byte-per-lane streaming copy: sustained HBM read + write rate.
// Neither touches the model; they replace nothing in the reference (there the host's wall clock is the only timer).
#include "common.h"
#include "matten_lab.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CALIB_CHAINS = 8;        // independent accumulators per lane: the FMA latency is hidden inside one wave
constexpr int CALIB_UNROLL = 16;

__global__ __launch_bounds__(256) void calib_valu_kernel(int64_t iters, float seed, float* __restrict__ out,
                                                         unsigned long long* __restrict__ clocks) {
    float acc[CALIB_CHAINS];
#pragma unroll
    for (int k = 0; k < CALIB_CHAINS; ++k) acc[k] = seed + (float)(threadIdx.x + k);
    const float m = 1.0f - 1e-7f * seed, a = 1e-9f * seed;   // run-time operands in VGPRs: VOP3 v_fma_f32 v, v, v, v
    // the clock reads are pinned to both ends of the loop: asm volatile statements keep their order, and the empty ones make
    // every accumulator an input / output of that point (a plain readcyclecounter floats freely around register-only code)
    unsigned long long t0 = 0, r0 = 0, t1 = 0, r1 = 0;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
#pragma unroll
    for (int k = 0; k < CALIB_CHAINS; ++k) asm volatile("" : "+v"(acc[k]));
    for (int64_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < CALIB_UNROLL; ++u)
#pragma unroll
            for (int k = 0; k < CALIB_CHAINS; ++k) acc[k] = __builtin_fmaf(acc[k], m, a);
    }
#pragma unroll
    for (int k = 0; k < CALIB_CHAINS; ++k) asm volatile("" : "+v"(acc[k]));
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < CALIB_CHAINS; ++k) s += acc[k];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clocks[0] = t1 - t0;      // s_memtime: shader clock
        clocks[1] = r1 - r0;      // s_memrealtime: constant 100 MHz
    }
    if (s == 12345.678f) out[0] = s;                  // keeps the chain alive
}

__global__ __launch_bounds__(256) void calib_copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, int64_t n16) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}

// One wave that does nothing but watch the two clocks for `ticks_100mhz` reference ticks (sleeping in between): launched on
// a side stream while the benchmark's forwards run, it reports the AVERAGE shader clock the chip sustained under that load
// (clocks[0] / clocks[1] x 100 MHz) -- what the fixed kernels above cannot tell: a box may run them at full clock and the
// power-hungry tensor-product kernels 7 % lower.
__global__ __launch_bounds__(64) void calib_clock_probe_kernel(unsigned long long ticks_100mhz, unsigned long long* __restrict__ clocks) {
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
    do {
        __builtin_amdgcn_s_sleep(127);
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    } while (r1 - r0 < ticks_100mhz);
    if (threadIdx.x == 0) {
        clocks[0] = t1 - t0;
        clocks[1] = r1 - r0;
    }
}

}  // namespace

extern "C" int matten_calib_clock_probe(int64_t ticks_100mhz, uint64_t* clocks, matten_stream_t stream_) {
    if (ticks_100mhz <= 0 || ticks_100mhz > 100000000 || !clocks) return MATTEN_EINVAL;   // at most one second
    calib_clock_probe_kernel<<<1, 64, 0, (hipStream_t)stream_>>>((unsigned long long)ticks_100mhz, (unsigned long long*)clocks);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}

extern "C" int64_t matten_calib_valu_insts_per_simd(int64_t iters) {
    // 2048 workgroups x 4 waves over 1024 SIMDs = 8 waves per SIMD, each iters x UNROLL x CHAINS FMAs
    return 8 * iters * CALIB_UNROLL * CALIB_CHAINS;
}

extern "C" int matten_calib_valu(int64_t iters, float* out, uint64_t* clocks, matten_stream_t stream_) {
    if (iters <= 0 || !out || !clocks) return MATTEN_EINVAL;
    calib_valu_kernel<<<2048, 256, 0, (hipStream_t)stream_>>>(iters, 1.0f, out, (unsigned long long*)clocks);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}

extern "C" int matten_calib_copy(const float* src, float* dst, int64_t n_floats, matten_stream_t stream_) {
    if (n_floats < 0 || (n_floats & 3)) return MATTEN_EINVAL;
    if (n_floats == 0) return MATTEN_OK;
    if (!src || !dst || ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15)) return MATTEN_EINVAL;
    calib_copy_kernel<<<256 * 16, 256, 0, (hipStream_t)stream_>>>((const f32x4*)src, (f32x4*)dst, n_floats / 4);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}
