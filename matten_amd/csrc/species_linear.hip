// Species-indexed per-irrep linear on the fp32 matrix cores, register-streaming form.
//   e3nn FullyConnectedTensorProduct(x, one_hot)  reference nn/conv.py:59-61,77-79,84-86
//   e3nn o3.Linear (order == NULL)                reference nn/nodewise.py:111-117, tfn_scalar_tensor.py:49-51
//
// For one irrep block (mul_in -> mo channels, d = 2l+1 components) the op is, per node n of species s,
//      out[n, o_off + v*d + m] = sum_u W_s[u, v] x[n, x_off + u*d + m].
// The operator is HBM-bound (each input row, up to 16.7 KB, is read once; ~5 MAC per input float) and, on
// MI355X, latency-bound unless ~30 MB of loads are in flight.  The kernel is therefore built around the input
// stream:
//   * a workgroup = 4 waves = 64 rows of ONE species (rows are visited in species-sorted order); the species'
//     packed weight table (<= 128 KB, typically 32 KB) is copied to LDS once;
//   * a wave owns 16 rows.  Rows are the N dimension of v_mfma_f32_16x16x4_f32, output channels v the M
//     dimension, input channels u the contraction: B[k = g][n = c] = x[row_c, x_off + u*d + m] with the
//     contraction index enumerated as u = 16 st + 4g + j for the j-th MFMA of step st (g = lane>>4, c = lane&15),
//     A[m = c][k = g] = W_s[u, 16 vt + c] from LDS with the same enumeration;
//   * the row is cut into CHUNKS: windows of 16*d*KS contiguous floats per row (KS = 10/3/2/1/1 steps for
//     d = 1/3/5/7/9, <= 640 B), fetched by exactly NB = 10 16-byte buffer loads per lane -- load t takes the t-th
//     64-byte piece of each of the 16 rows, so every instruction touches 16 full lines (a lane-contiguous layout,
//     4d floats per lane, touches 64 lines per instruction and streams at 2.5 TB/s for d = 9 instead of 5.8,
//     tools/ubench/rowstream.hip); unused slots are dummy loads of a resident line.  The landed registers are
//     transposed through a wave-private LDS tile (ds_write_b128 in load order, ds_read_b128 of the 4d contiguous
//     floats lane (g, c) needs), no barrier involved;
//   * two scalar cursors walk the chunks: the load cursor runs two chunks ahead of the MFMA cursor ACROSS
//     irrep-block boundaries (three register buffers).  Because every chunk issues the same number of loads the
//     compiler's s_waitcnt vmcnt() counts are exact and a wave keeps 20 KB in flight while it multiplies;
//   * buffer (SRD) loads give hardware bounds checking: reads past the tensor return 0, tails past a block's last
//     channel are masked, so no NaN can leak between rows;
//   * the optional addend (self-connection + message: out = add + W x) is streamed the same way, as one more chunk
//     per output tile multiplied by an identity A operand, so the epilogue issues no loads;
//   * D[row = 4g + r][col = c] is channel v = 16 vt + 4g + r of row c: each lane stores 4d contiguous floats.
#include "common.h"

namespace {

struct LinSeg {  // 8 x int32: one irrep block of one pass (matten_amd/plan.py:_plan_linear_like)
    int x_off, d, mul_in, w_off, mo, o_off, pad0, pad1;
};
constexpr int SL_ROWS = 16;
constexpr int SL_WAVES = 4;
#ifndef SL_NB
#define SL_NB 10
#endif
constexpr int NB = SL_NB;     // 16-byte loads per lane and chunk
constexpr int NACC = 9;       // accumulator tiles: max(NVT * d)
constexpr int XS_RS = 16 * NB + 4;  // LDS row stride of the transpose tile (== 4 mod 32: conflict-free b128)
constexpr uint32_t SRD_WORD3 = 0x00020000u;
constexpr int64_t SRD_SPAN = ((int64_t)1 << 32) - (1 << 20);  // addressable bytes per descriptor, with slack
typedef float sl_f32x4 __attribute__((ext_vector_type(4)));
typedef int sl_i32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) sl_f4u {
    float v[4];
};
}  // namespace
// 16-byte buffer load (bounds-checked against the descriptor: out-of-range lanes read 0)
__device__ sl_f32x4 sl_buffer_load_x4(sl_i32x4 srsrc, int voffset, int soffset, int aux) __asm(
    "llvm.amdgcn.raw.buffer.load.v4f32");
namespace {

// v_mfma_f32_16x16x4_f32 accumulating in place (vDst tied to SrcC), issued through inline asm.
// Why not the builtin: with __builtin_amdgcn_mfma_f32_16x16x4f32 clang moves the accumulators between register
// tuples inside this (17 k instruction, 256 VGPR) kernel, and two builds of that form produced accumulators whose
// upper half was stale for a few shapes (d = 1 blocks with an addend; tools/sl_check.py finds them).  The hardware
// was ruled out -- overlapping vDst/SrcA/SrcB/SrcC, dependent chains and MFMA -> VALU reads are all interlocked
// on gfx950, only MFMA -> VMEM-store needs 8 wait states and clang leaves 10 (tools/ubench/mfma_overlap.hip,
// mfma_latency.hip).  Root cause (found in round 2 when a variant of this kernel reproduced it deterministically: the
// last matrix instructions of a step lost for some accumulators): SimplifyCFG sinks the accumulator updates of
// different template paths into one store through a PHI of POINTERS, which keeps those accumulators in scratch memory
// (load - MFMA - scratch_store).  With the builtin the compiler pads the MFMA -> store hazard itself but shuffles
// tuples; with inline asm its hazard recogniser is blind and the scratch store reads a result that is not there yet.
// The file is therefore built with -mllvm -simplifycfg-sink-common=false (Makefile): every accumulator stays in
// registers.  The tied form below leaves the register allocator nothing to move; it passes the full sweep
// (tests/test_gpu_parity.py::test_species_linear_shape_sweep, tools/sl_fuzz.py).
// The hazard recogniser does not look inside inline asm, hence the explicit wait states: s_nop 1 in front (VALU
// write of an operand -> MFMA read) and mfma_drain() after the last MFMA of every straight-line group, before
// anything else may read, copy or store an accumulator.
__device__ __forceinline__ void mfma_16x16x4(sl_f32x4& acc, float a, float b) {
#ifdef SL_ABL_NO_MFMA
    asm volatile("v_add_f32 %0, %1, %0" : "+v"(acc[0]) : "v"(a + b));   // timing experiment (tools/sl_ab2.sh): operands stay alive
#else
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#endif
}
__device__ __forceinline__ void mfma_drain() {
#ifndef SL_ABL_NO_MFMA
    asm volatile("s_nop 15" ::: "memory");
#endif
}

// steps (16 input channels each) per chunk: max(1, NB / d).  d is 2l+1 <= 9 (the host plan rejects anything else), so
// the run-time value is a select chain: the integer division cost ~30 instructions and a VALU -> SALU round trip three
// times per chunk on the wave's serial path.
__device__ __forceinline__ constexpr int steps_per_chunk(int d) {
    return d <= 1 ? NB : d <= 3 ? (NB / 3 > 0 ? NB / 3 : 1) : d <= 5 ? (NB / 5 > 0 ? NB / 5 : 1)
                       : d <= 7 ? (NB / 7 > 0 ? NB / 7 : 1) : (NB / 9 > 0 ? NB / 9 : 1);
}

struct Cursor {  // position of a chunk in the walk over (irrep block, channel-tile group, step); wave-uniform
    int sg, vt0, st;
};

// st == -1 is the addend chunk that opens an output tile group when add != NULL
__device__ __forceinline__ void advance(Cursor& cu, const LinSeg& L, int st_first) {
    // an empty block (mul_in == 0: output irrep without an input path) still has one, empty, step: it stores zeros / the addend
    const int n_st = max(1, (L.mul_in + 15) >> 4), n_vt = (L.mo + 15) >> 4;
    cu.st = cu.st < 0 ? 0 : cu.st + steps_per_chunk(L.d);
    if (cu.st >= n_st) {
        cu.st = st_first;
        cu.vt0 += L.d <= 3 ? 2 : 1;
        if (cu.vt0 >= n_vt) {
            cu.vt0 = 0;
            ++cu.sg;
        }
    }
}

struct Stream {  // descriptor + per-lane row offset of the two streamed operands
    sl_i32x4 x_rsrc, a_rsrc;
    int x_voff, a_voff;
};

// NB loads, always: load t fetches floats [16 t + 4g, +4) of the chunk's window in row c.
__device__ __forceinline__ void load_chunk(float (&buf)[4 * NB], const Stream& sm, int g, const LinSeg& L,
                                           const Cursor& cu, bool live) {
    const bool is_add = cu.st < 0;
    const int n_st = (L.mul_in + 15) >> 4, n_vt = (L.mo + 15) >> 4;
    // window: steps of the input row, or (addend chunk) the output tiles of the group
    const int n_run = is_add ? ((L.d <= 3 && cu.vt0 + 1 < n_vt) ? 2 : 1) : min(steps_per_chunk(L.d), n_st - cu.st);
    const int n_real = live ? n_run * L.d : 0;
    const int base = is_add ? L.o_off + 16 * cu.vt0 * L.d : L.x_off + 16 * cu.st * L.d;
    const sl_i32x4 rsrc = is_add ? sm.a_rsrc : sm.x_rsrc;
    const int voff = (is_add ? sm.a_voff : sm.x_voff) + 16 * g;
#pragma unroll
    for (int t = 0; t < NB; ++t) {
        const int soff = t < n_real ? 4 * (base + 16 * t) : 0;
#ifdef SL_DUMMY_RESIDENT
        const sl_f32x4 v = sl_buffer_load_x4(rsrc, voff, soff, 0);
#else
        // unused slots: an offset past every descriptor's range -- the bounds check answers 0 and no request leaves
        // the CU (a "resident" dummy line is not resident in a 1 GB stream: it was 40 % extra HBM fetch)
        const sl_f32x4 v = sl_buffer_load_x4(rsrc, t < n_real ? voff : (int)0xFFFFFFF0u, soff, 0);
#endif
        buf[4 * t + 0] = v[0];
        buf[4 * t + 1] = v[1];
        buf[4 * t + 2] = v[2];
        buf[4 * t + 3] = v[3];
    }
}

// registers (load order) -> wave-private LDS tile [16 rows][XS_RS]
__device__ __forceinline__ void stage_chunk(const float (&buf)[4 * NB], float* xs, int g, int c) {
    float* p = xs + c * XS_RS + 4 * g;
#pragma unroll
    for (int t = 0; t < NB; ++t)
        *reinterpret_cast<sl_f32x4*>(p + 16 * t) = sl_f32x4{buf[4 * t], buf[4 * t + 1], buf[4 * t + 2], buf[4 * t + 3]};
}

template <int D, int NVT>
__device__ __forceinline__ void mfma_chunk(const float* xs, sl_f32x4 (&acc)[NACC],
                                           const float* __restrict__ ws, const LinSeg& L, int vt0, int st, int g,
                                           int c) {
    constexpr int KS = steps_per_chunk(D);
    const int n_st = (L.mul_in + 15) >> 4;
    const float* xl = xs + c * XS_RS + 4 * g * D;  // lane (g, c): floats [16 D r + 4 g D, + 4 D) of row c's window
#pragma unroll
    for (int r = 0; r < KS; ++r) {
        if (st + r < n_st) {
            float xr[4 * D];
#pragma unroll
            for (int q = 0; q < D; ++q) {
                const sl_f32x4 t = *reinterpret_cast<const sl_f32x4*>(xl + 16 * D * r + 4 * q);
                xr[4 * q] = t[0];
                xr[4 * q + 1] = t[1];
                xr[4 * q + 2] = t[2];
                xr[4 * q + 3] = t[3];
            }
            const int ub = 16 * (st + r) + 4 * g;
            float a[NVT][4];
#pragma unroll
            for (int vt = 0; vt < NVT; ++vt) {
                const int v = 16 * (vt0 + vt) + c;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool ok = ub + j < L.mul_in && v < L.mo;
                    const float wv = ws[ok ? L.w_off + (ub + j) * L.mo + v : 0];
                    a[vt][j] = ok ? wv : 0.0f;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool u_ok = ub + j < L.mul_in;
#pragma unroll
                for (int m = 0; m < D; ++m) {
                    const float xv = u_ok ? xr[j * D + m] : 0.0f;
#pragma unroll
                    for (int vt = 0; vt < NVT; ++vt) mfma_16x16x4(acc[vt * D + m], a[vt][j], xv);
                }
            }
            mfma_drain();
        }
    }
}

// addend chunk: run vt holds add[row c, o_off + (16 (vt0+vt) + 4g + j)*D + m]; multiply by A = identity
template <int D, int NVT>
__device__ __forceinline__ void mfma_addend(const float* xs, sl_f32x4 (&acc)[NACC], const LinSeg& L,
                                            int vt0, int g, int c) {
    const float* xl = xs + c * XS_RS + 4 * g * D;
#pragma unroll
    for (int vt = 0; vt < NVT; ++vt) {
        float xr[4 * D];
#pragma unroll
        for (int q = 0; q < D; ++q) {
            const sl_f32x4 t = *reinterpret_cast<const sl_f32x4*>(xl + 16 * D * vt + 4 * q);
            xr[4 * q] = t[0];
            xr[4 * q + 1] = t[1];
            xr[4 * q + 2] = t[2];
            xr[4 * q + 3] = t[3];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = (c == 4 * g + j) ? 1.0f : 0.0f;
            const bool v_ok = 16 * (vt0 + vt) + 4 * g + j < L.mo;
#pragma unroll
            for (int m = 0; m < D; ++m) {
                const float xv = v_ok ? xr[j * D + m] : 0.0f;
                mfma_16x16x4(acc[vt * D + m], a, xv);
            }
        }
        mfma_drain();
    }
}

template <int D, int NVT>
__device__ __forceinline__ void store_tiles(sl_f32x4 (&acc)[NACC], const LinSeg& L, int vt0, int g, bool row_ok,
                                            float* __restrict__ orow) {
#pragma unroll
    for (int vt = 0; vt < NVT; ++vt) {
        const int vb = 16 * (vt0 + vt) + 4 * g;
        float* op = orow + L.o_off + vb * D;
        if (row_ok && vb + 4 <= L.mo) {
            float o[4 * D];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int m = 0; m < D; ++m) o[r * D + m] = acc[vt * D + m][r];
#pragma unroll
            for (int q = 0; q < D; ++q) {
                sl_f4u t;
#pragma unroll
                for (int e = 0; e < 4; ++e) t.v[e] = o[4 * q + e];
                *reinterpret_cast<sl_f4u*>(op + 4 * q) = t;
            }
        } else if (row_ok) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (vb + r < L.mo) {
#pragma unroll
                    for (int m = 0; m < D; ++m) op[r * D + m] = acc[vt * D + m][r];
                }
        }
    }
#pragma unroll
    for (int i = 0; i < NVT * D; ++i) acc[i] = sl_f32x4{0.f, 0.f, 0.f, 0.f};
}

template <int D, int NVT>
__device__ __forceinline__ void consume_dn(const float* buf, sl_f32x4 (&acc)[NACC],
                                           const float* __restrict__ ws, const LinSeg& L, const Cursor& cu, int g,
                                           int c, bool row_ok, float* __restrict__ orow) {
    const int n_st = (L.mul_in + 15) >> 4;
    if (cu.st < 0) {
        mfma_addend<D, NVT>(buf, acc, L, cu.vt0, g, c);
    } else {
        mfma_chunk<D, NVT>(buf, acc, ws, L, cu.vt0, cu.st, g, c);
        if (cu.st + steps_per_chunk(D) >= n_st) store_tiles<D, NVT>(acc, L, cu.vt0, g, row_ok, orow);
    }
}

template <int D>
__device__ __forceinline__ void consume_d(const float* buf, sl_f32x4 (&acc)[NACC],
                                          const float* __restrict__ ws, const LinSeg& L, const Cursor& cu, int g, int c,
                                          bool row_ok, float* __restrict__ orow) {
    const int n_vt = (L.mo + 15) >> 4;
    if (D <= 3 && cu.vt0 + 1 < n_vt) consume_dn<D, (D <= 3 ? 2 : 1)>(buf, acc, ws, L, cu, g, c, row_ok, orow);
    else consume_dn<D, 1>(buf, acc, ws, L, cu, g, c, row_ok, orow);
}

__device__ __forceinline__ void consume(const float (&regs)[4 * NB], float* buf, sl_f32x4 (&acc)[NACC],
                                        const float* __restrict__ ws, const LinSeg& L, const Cursor& cu, int g, int c,
                                        bool row_ok, float* __restrict__ orow) {
#ifdef SL_ABL_LOADS_ONLY
    {   // timing experiment (tools/sl_ab2.sh): the load stream + the cursor walk only (results are wrong)
        float t = 0.0f;
#pragma unroll
        for (int i = 0; i < 4 * NB; ++i) t += regs[i];
        acc[0][0] += t;
        if (cu.st >= 0 && cu.st + steps_per_chunk(L.d) >= ((L.mul_in + 15) >> 4) && acc[0][0] == 12345.678f) orow[0] = t;
        return;
    }
#endif
    stage_chunk(regs, buf, g, c);
#ifdef SL_ABL_STAGE_ONLY
    {   // timing experiment: + the LDS staging writes
        float t = buf[c * XS_RS + 4 * g];
        acc[0][0] += t;
        if (cu.st >= 0 && cu.st + steps_per_chunk(L.d) >= ((L.mul_in + 15) >> 4) && acc[0][0] == 12345.678f) orow[0] = t;
        return;
    }
#endif
    switch (L.d) {
        case 1: consume_d<1>(buf, acc, ws, L, cu, g, c, row_ok, orow); break;
        case 3: consume_d<3>(buf, acc, ws, L, cu, g, c, row_ok, orow); break;
        case 5: consume_d<5>(buf, acc, ws, L, cu, g, c, row_ok, orow); break;
        case 7: consume_d<7>(buf, acc, ws, L, cu, g, c, row_ok, orow); break;
        case 9: consume_d<9>(buf, acc, ws, L, cu, g, c, row_ok, orow); break;
        default: break;  // 2l+1 > 9: rejected by the host plan
    }
}

#ifndef SL_MIN_BLOCKS
#define SL_MIN_BLOCKS 2
#endif

// W_LDS = false: the species' packed table does not fit next to the transpose tiles (> ~118 KB); the A operand is
// then read from global memory (L2-resident) -- same arithmetic, slower, no size limit.
template <bool W_LDS>
__global__ __launch_bounds__(SL_WAVES * 64, SL_MIN_BLOCKS) void species_linear_kernel(
    const float* __restrict__ x, int d_in, const int32_t* __restrict__ order, const int32_t* __restrict__ seg,
    int n_species, const float* __restrict__ wp, int w_stride, const LinSeg* __restrict__ segs, int n_segs,
    int segs_per_block, int d_out, const float* __restrict__ add, int add_ld, int n_rows, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ws_lds = lds + SL_WAVES * SL_ROWS * XS_RS;  // [w_stride] packed table of this block's species
    // ---- block -> (species, 64 rows) ----
    int b = blockIdx.x, s = 0, blo = 0, bhi = 0;
    constexpr int BR = SL_ROWS * SL_WAVES;
    if (seg) {
        if (!matten_block_species<SL_WAVES * 64>(seg, n_species, BR, b, reinterpret_cast<int*>(lds), s, blo, bhi)) return;
    } else {
        blo = b * BR;
        bhi = min(n_rows, blo + BR);
        if (blo >= bhi) return;
    }
    const int sg0 = blockIdx.y * segs_per_block, sg1 = min(n_segs, sg0 + segs_per_block);
    const float* wsp = wp + (int64_t)s * w_stride;
    if (W_LDS) {   // the slice of the weight table this block's irrep blocks touch
        int w_lo = w_stride, w_hi = 0;
        for (int sg = sg0; sg < sg1; ++sg) {
            w_lo = min(w_lo, segs[sg].w_off);
            w_hi = max(w_hi, segs[sg].w_off + segs[sg].mul_in * segs[sg].mo);
        }
        for (int i = w_lo + threadIdx.x; i < w_hi; i += blockDim.x) ws_lds[i] = wsp[i];
        __syncthreads();
    }
    const float* ws = W_LDS ? ws_lds : wsp;

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, c = lane & 15;
    float* xs = lds + wave * SL_ROWS * XS_RS;  // this wave's transpose tile
    const int lo = blo + wave * SL_ROWS;
    if (lo >= bhi) return;
    const int hi = min(bhi, lo + SL_ROWS);
    const bool row_in = lo + c < hi;
    const int row = row_in ? lo + c : lo;
    const int node = order ? order[row] : row;
    float* orow = out + (int64_t)node * d_out;
    const int64_t x_bytes = (int64_t)n_rows * d_in * 4, a_bytes = ((int64_t)(n_rows - 1) * add_ld + d_out) * 4;
    const int st_first = add ? -1 : 0;

    // Rows of a tile ascend in memory; a descriptor addresses 4 GB from its base.  Tiles whose rows are further
    // apart than that (only possible for > 4 GB inputs and very rare species) take several passes.
    bool pending = row_in;
    while (true) {
        int nmin = pending ? node : 0x7fffffff;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nmin = min(nmin, __shfl_xor(nmin, off));
        if (nmin == 0x7fffffff) break;
        nmin = __builtin_amdgcn_readfirstlane(nmin);
        const int64_t rel = (int64_t)node - nmin;
        const bool row_ok = pending && rel >= 0 && rel * max(d_in, max(d_out, add_ld)) * 4 < SRD_SPAN;
        Stream sm;
        {
            const int64_t xb = (int64_t)nmin * d_in * 4, ab = (int64_t)nmin * add_ld * 4;
            const uint64_t xp = (uint64_t)(reinterpret_cast<const char*>(x) + xb);
            const uint64_t ap = (uint64_t)(reinterpret_cast<const char*>(add ? add : x) + (add ? ab : 0));
            const int64_t xl = x_bytes - xb, al = add ? a_bytes - ab : 0;
            sm.x_rsrc = sl_i32x4{(int)(uint32_t)xp, (int)((uint32_t)(xp >> 32) & 0xffffu),
                                 (int)(uint32_t)(xl > 0xffffffffll ? 0xffffffffll : xl), (int)SRD_WORD3};
            sm.a_rsrc = sl_i32x4{(int)(uint32_t)ap, (int)((uint32_t)(ap >> 32) & 0xffffu),
                                 (int)(uint32_t)(al > 0xffffffffll ? 0xffffffffll : al), (int)SRD_WORD3};
            sm.x_voff = row_ok ? (int)(rel * d_in * 4) : 0;
            sm.a_voff = row_ok ? (int)(rel * add_ld * 4) : 0;
        }

        sl_f32x4 acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = sl_f32x4{0.f, 0.f, 0.f, 0.f};
        float bufA[4 * NB], bufB[4 * NB], bufC[4 * NB];
        Cursor lc{sg0, 0, st_first}, cc{sg0, 0, st_first};
        LinSeg Ll = segs[sg0], Lc = Ll;
#define SL_STEP_LOAD(buf)                                             \
        load_chunk(buf, sm, g, Ll, lc, lc.sg < sg1);                  \
        if (lc.sg < sg1) {                                            \
            const int sgp = lc.sg;                                    \
            advance(lc, Ll, st_first);                                \
            if (lc.sg != sgp && lc.sg < sg1) Ll = segs[lc.sg];        \
        }
#define SL_STEP_MFMA(buf)                                             \
        consume(buf, xs, acc, ws, Lc, cc, g, c, row_ok, orow);            \
        {                                                             \
            const int sgp = cc.sg;                                    \
            advance(cc, Lc, st_first);                                \
            if (cc.sg >= sg1) break;                                  \
            if (cc.sg != sgp) Lc = segs[cc.sg];                       \
        }
        SL_STEP_LOAD(bufA)
        SL_STEP_LOAD(bufB)
        while (true) {  // two chunks (20 KB per wave) in flight while the third is multiplied
            SL_STEP_LOAD(bufC)
            SL_STEP_MFMA(bufA)
            SL_STEP_LOAD(bufA)
            SL_STEP_MFMA(bufB)
            SL_STEP_LOAD(bufB)
            SL_STEP_MFMA(bufC)
        }
#undef SL_STEP_LOAD
#undef SL_STEP_MFMA
        pending = pending && !row_ok;
    }
}

}  // namespace

extern "C" int matten_species_linear(const float* x, int64_t d_in, const int32_t* order, const int32_t* seg,
                                     int64_t n_species, const float* wp, int64_t w_stride, const int32_t* segs,
                                     int64_t n_segs, int64_t d_out, const float* add, int64_t add_ld, int64_t n_rows,
                                     float* out, matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_rows < 0 || d_in <= 0 || d_out <= 0 || n_species <= 0 || w_stride < 0 || n_segs < 0 ||
        n_rows >= ((int64_t)1 << 31))
        return MATTEN_EINVAL;
    if (n_rows == 0 || n_segs == 0) return MATTEN_OK;
    if (!x || !wp || !out || !segs) return MATTEN_EINVAL;
    if (add && add_ld < d_out) return MATTEN_EINVAL;
    if (!add) add_ld = d_out;
    if ((order == nullptr) != (seg == nullptr)) return MATTEN_EINVAL;
    if (!order && n_species != 1) return MATTEN_EINVAL;
    const size_t lds_tiles = sizeof(float) * SL_WAVES * SL_ROWS * XS_RS;
    const bool w_lds = lds_tiles + sizeof(float) * (size_t)((w_stride + 3) & ~3) <= 160 * 1024;
    const size_t lds = lds_tiles + (w_lds ? sizeof(float) * (size_t)((w_stride + 3) & ~3) : 0);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)species_linear_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess)
            return MATTEN_ELAUNCH;
        attr_set = true;
    }
    constexpr int BR = SL_ROWS * SL_WAVES;
    const int64_t blocks = matten_cdiv(n_rows, BR) + (order ? n_species : 0);
    // few rows: spread the irrep blocks over blockIdx.y so the chip still sees enough waves
    const int segs_per_block = blocks >= 512 ? (int)n_segs : 1;
    dim3 grid((unsigned)blocks, (unsigned)matten_cdiv(n_segs, segs_per_block));
    if (w_lds)
        species_linear_kernel<true><<<grid, SL_WAVES * 64, lds, stream>>>(
            x, (int)d_in, order, seg, (int)n_species, wp, (int)w_stride, (const LinSeg*)segs, (int)n_segs,
            segs_per_block, (int)d_out, add, (int)add_ld, (int)n_rows, out);
    else
        species_linear_kernel<false><<<grid, SL_WAVES * 64, lds, stream>>>(
            x, (int)d_in, order, seg, (int)n_species, wp, (int)w_stride, (const LinSeg*)segs, (int)n_segs,
            segs_per_block, (int)d_out, add, (int)add_ld, (int)n_rows, out);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}
