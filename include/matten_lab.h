/*
 * matten_lab.h -- C ABI of libmatten_lab.so (`make -C matten_amd/csrc lab`): measurement helpers of the benchmark harness.
 * NOT part of the product: nothing here replaces a function of the reference, no reference-side binding calls it, and
 * libmatten_hip.so does not contain it.  bench.py loads it (when present) for its `calibration` block; lab builds of
 * tp_fused.hip (-DMATTEN_LAB ..., tools/) add their own experiment hooks to their own copy of libmatten_hip.so.
 */
#ifndef MATTEN_LAB_H
#define MATTEN_LAB_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef void* matten_lab_stream_t;   /* hipStream_t */

/* Calibration work for the benchmark harness (bench.py "calibration"; nothing of the reference corresponds -- there the
 * host's wall clock is the only timer): fixed, model-independent kernels timed next to the benchmark so that lines taken
 * on different machines / DVFS states can be compared.
 *   matten_calib_valu: `iters` x 128 dependent-chain fp32 FMAs per lane at 8 waves per SIMD on every CU
 *     (matten_calib_valu_insts_per_simd(iters) wave64 instructions per SIMD); clocks[0] = shader-clock ticks (s_memtime),
 *     clocks[1] = 100 MHz reference ticks (s_memrealtime) the first wave spent in the loop.  out: one float, never written.
 *   matten_calib_copy: dst[0..n) = src[0..n), 16 bytes per lane (n a multiple of 4, pointers 16-byte aligned).
 *   matten_calib_clock_probe: ONE wave that watches s_memtime against s_memrealtime for ticks_100mhz reference ticks
 *     (<= 1 s), sleeping in between: on a side stream beside the benchmark's forwards it reports the average shader clock
 *     under THAT load (clocks[0] / clocks[1] x 100 MHz). */
int64_t matten_calib_valu_insts_per_simd(int64_t iters);
int matten_calib_clock_probe(int64_t ticks_100mhz, uint64_t* clocks, matten_lab_stream_t stream);
int matten_calib_valu(int64_t iters, float* out, uint64_t* clocks, matten_lab_stream_t stream);
int matten_calib_copy(const float* src, float* dst, int64_t n_floats, matten_lab_stream_t stream);


#ifdef __cplusplus
}
#endif
#endif
