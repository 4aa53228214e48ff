// Species-indexed per-irrep linear on the fp32 matrix cores, register-streaming form.
//   e3nn FullyConnectedTensorProduct(x, one_hot)  reference nn/conv.py:59-61,77-79,84-86
//   e3nn o3.Linear (order == NULL)                reference nn/nodewise.py:111-117, tfn_scalar_tensor.py:49-51
//
// For one irrep block (mul_in -> mo channels, d = 2l+1 components) the op is, per node n of species s,
//      out[n, o_off + v*d + m] = sum_u W_s[u, v] x[n, x_off + u*d + m].
// The operator is HBM-bound (each input row, up to 16.7 KB, is read once; ~5 MAC per input float), so the
// kernel is built around the input stream:
//   * a wave owns 16 rows of ONE species (rows are visited in species-sorted order) and walks the irrep blocks;
//   * rows are the N dimension of v_mfma_f32_16x16x4_f32, output channels v the M dimension, input channels u
//     the contraction.  Lane (g = lane>>4, c = lane&15) loads, per step of 16 input channels, the 4d CONTIGUOUS
//     floats x[row_c, x_off + (16 st + 4g)*d .. +4d) straight from global memory with 16-byte loads -- the four
//     lanes of a row cover 16d contiguous floats, so every fetched line is used in full -- and feeds them to the
//     matrix core as B[k = g][n = c] with the contraction index enumerated as u = 16 st + 4g + j for the j-th
//     MFMA of the step (A uses the same enumeration, so no shuffle and no LDS is needed);
//   * the weights (<= 89 KB per species, L2-resident) are read as A[m = c][k = g] = W_s[u, 16 vt + c];
//   * D[row = 4g + r][col = c] is channel v = 16 vt + 4g + r of row c: each lane stores 4d contiguous floats.
// The next step's operands are loaded before the current step's MFMAs (two register buffers), there are no
// barriers, and the only LDS use is none at all.
#include "common.h"

namespace {

struct LinSeg {  // 8 x int32: one irrep block of one pass (matten_amd/plan.py:_plan_linear_like)
    int x_off, d, mul_in, w_off, mo, o_off, pad0, pad1;
};
constexpr int SL_ROWS = 16;
constexpr int SL_WAVES = 4;
#ifndef SL_MIN_BLOCKS
#define SL_MIN_BLOCKS 3
#endif
typedef float sl_f32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) sl_f4u {
    float v[4];
};

template <int D, int NVT>
struct Operands {
    float x[4 * D];
    float a[NVT][4];
};

template <int D, int NVT>
__device__ __forceinline__ void load_operands(Operands<D, NVT>& o, const float* __restrict__ xp,
                                              const float* __restrict__ wl, int u0, int mul_in, int mo, int v0) {
    // xp -> x[row, x_off + u0*D], wl -> W_s[u0, v0]  (u0 = 16 st + 4g, v0 = 16 vt0 + c)
    if (u0 + 4 <= mul_in) {
#pragma unroll
        for (int q = 0; q < D; ++q) {
#ifdef SL_ABLATE_COALESCED
            const sl_f4u t = *reinterpret_cast<const sl_f4u*>(xp - (u0 & 15) * (D - 1) + 16 * q);
#else
            const sl_f4u t = *reinterpret_cast<const sl_f4u*>(xp + 4 * q);
#endif
#pragma unroll
            for (int e = 0; e < 4; ++e) o.x[4 * q + e] = t.v[e];
        }
#pragma unroll
        for (int vt = 0; vt < NVT; ++vt) {
            const bool v_ok = v0 + 16 * vt < mo;
#pragma unroll
#ifdef SL_ABLATE_NO_W
            for (int j = 0; j < 4; ++j) o.a[vt][j] = v_ok ? 1.0f : 0.0f;
#else
            for (int j = 0; j < 4; ++j) o.a[vt][j] = v_ok ? wl[j * mo + 16 * vt] : 0.0f;
#endif
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool u_ok = u0 + j < mul_in;
#pragma unroll
            for (int m = 0; m < D; ++m) o.x[j * D + m] = u_ok ? xp[j * D + m] : 0.0f;
#pragma unroll
            for (int vt = 0; vt < NVT; ++vt)
                o.a[vt][j] = (u_ok && v0 + 16 * vt < mo) ? wl[j * mo + 16 * vt] : 0.0f;
        }
    }
}

template <int D, int NVT>
__device__ __forceinline__ void mfma_step(const Operands<D, NVT>& o, sl_f32x4 (&acc)[NVT][D]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int vt = 0; vt < NVT; ++vt)
#pragma unroll
            for (int m = 0; m < D; ++m)
#ifdef SL_ABLATE_NO_MFMA
                acc[vt][m][0] += o.a[vt][j] * o.x[j * D + m];
#else
                acc[vt][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[vt][j], o.x[j * D + m], acc[vt][m], 0, 0, 0);
#endif
}

template <int D, int NVT>
__device__ __forceinline__ void run_block(const float* __restrict__ xrow, const float* __restrict__ wsp,
                                          const LinSeg& L, int g, int c, int vt0, bool row_ok,
                                          float* __restrict__ orow, const float* __restrict__ arow) {
    sl_f32x4 acc[NVT][D];
#pragma unroll
    for (int vt = 0; vt < NVT; ++vt)
#pragma unroll
        for (int m = 0; m < D; ++m) acc[vt][m] = sl_f32x4{0.f, 0.f, 0.f, 0.f};
    const int v0 = 16 * vt0 + c;
    const float* xp = xrow + L.x_off + 4 * g * D;
    const float* wl = wsp + L.w_off + 4 * g * L.mo + v0;
    const int n_st = (L.mul_in + 15) >> 4;
    Operands<D, NVT> A, B;
    load_operands<D, NVT>(A, xp, wl, 4 * g, L.mul_in, L.mo, v0);
    for (int st = 0; st < n_st; st += 2) {
        if (st + 1 < n_st)
            load_operands<D, NVT>(B, xp + 16 * (st + 1) * D, wl + 16 * (st + 1) * L.mo, 16 * (st + 1) + 4 * g,
                                  L.mul_in, L.mo, v0);
        mfma_step<D, NVT>(A, acc);
        if (st + 1 < n_st) {
            if (st + 2 < n_st)
                load_operands<D, NVT>(A, xp + 16 * (st + 2) * D, wl + 16 * (st + 2) * L.mo, 16 * (st + 2) + 4 * g,
                                      L.mul_in, L.mo, v0);
            mfma_step<D, NVT>(B, acc);
        }
    }
    if (!row_ok) return;
#pragma unroll
    for (int vt = 0; vt < NVT; ++vt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int v = 16 * (vt0 + vt) + 4 * g + r;
            if (v < L.mo) {
#pragma unroll
                for (int m = 0; m < D; ++m) {
                    const int oi = L.o_off + v * D + m;
                    orow[oi] = (arow ? arow[oi] : 0.0f) + acc[vt][m][r];
                }
            }
        }
}

template <int D>
__device__ __forceinline__ void run_segment(const float* __restrict__ xrow, const float* __restrict__ wsp,
                                            const LinSeg& L, int g, int c, bool row_ok, float* __restrict__ orow,
                                            const float* __restrict__ arow) {
    const int n_vt = (L.mo + 15) >> 4;
    if constexpr (D <= 3) {
        for (int vt0 = 0; vt0 < n_vt; vt0 += 2) {
            if (vt0 + 1 < n_vt) run_block<D, 2>(xrow, wsp, L, g, c, vt0, row_ok, orow, arow);
            else run_block<D, 1>(xrow, wsp, L, g, c, vt0, row_ok, orow, arow);
        }
    } else {
        for (int vt0 = 0; vt0 < n_vt; ++vt0) run_block<D, 1>(xrow, wsp, L, g, c, vt0, row_ok, orow, arow);
    }
}

__global__ __launch_bounds__(SL_WAVES * 64, SL_MIN_BLOCKS) void species_linear_kernel(
    const float* __restrict__ x, int d_in, const int32_t* __restrict__ order, const int32_t* __restrict__ seg,
    int n_species, const float* __restrict__ wp, int w_stride, const LinSeg* __restrict__ segs, int n_segs,
    int segs_per_block, int d_out, const float* __restrict__ add, int n_rows, float* __restrict__ out) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    int t = blockIdx.x * SL_WAVES + wave, s = 0, lo = 0, hi = 0;
    if (seg) {
        bool found = false;
        for (s = 0; s < n_species; ++s) {
            const int beg = seg[s], end = seg[s + 1];
            const int nt = (end - beg + SL_ROWS - 1) / SL_ROWS;
            if (t < nt) {
                lo = beg + t * SL_ROWS;
                hi = min(end, lo + SL_ROWS);
                found = true;
                break;
            }
            t -= nt;
        }
        if (!found) return;
    } else {
        lo = t * SL_ROWS;
        hi = min(n_rows, lo + SL_ROWS);
        if (lo >= hi) return;
    }
    const float* wsp = wp + (int64_t)s * w_stride;
    const int g = lane >> 4, c = lane & 15;
    const bool row_ok = lo + c < hi;
    const int row = row_ok ? lo + c : lo;
    const int node = order ? order[row] : row;
    const float* xrow = x + (int64_t)node * d_in;
    float* orow = out + (int64_t)node * d_out;
    const float* arow = add ? add + (int64_t)node * d_out : nullptr;

    const int sg0 = blockIdx.y * segs_per_block, sg1 = min(n_segs, sg0 + segs_per_block);
    for (int sg = sg0; sg < sg1; ++sg) {
        const LinSeg L = segs[sg];
        switch (L.d) {
            case 1: run_segment<1>(xrow, wsp, L, g, c, row_ok, orow, arow); break;
            case 3: run_segment<3>(xrow, wsp, L, g, c, row_ok, orow, arow); break;
            case 5: run_segment<5>(xrow, wsp, L, g, c, row_ok, orow, arow); break;
            case 7: run_segment<7>(xrow, wsp, L, g, c, row_ok, orow, arow); break;
            case 9: run_segment<9>(xrow, wsp, L, g, c, row_ok, orow, arow); break;
            default: break;  // rejected on the host
        }
    }
}

}  // namespace

extern "C" int matten_species_linear(const float* x, int64_t d_in, const int32_t* order, const int32_t* seg,
                                     int64_t n_species, const float* wp, int64_t w_stride, const int32_t* segs,
                                     int64_t n_segs, int64_t d_out, const float* add, int64_t n_rows, float* out,
                                     matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_rows < 0 || d_in <= 0 || d_out <= 0 || n_species <= 0 || w_stride < 0 || n_segs < 0 ||
        n_rows >= ((int64_t)1 << 31))
        return MATTEN_EINVAL;
    if (n_rows == 0 || n_segs == 0) return MATTEN_OK;
    if (!x || !wp || !out || !segs) return MATTEN_EINVAL;
    if ((order == nullptr) != (seg == nullptr)) return MATTEN_EINVAL;
    if (!order && n_species != 1) return MATTEN_EINVAL;
    const int64_t tiles = matten_cdiv(n_rows, SL_ROWS) + (order ? n_species : 0);
    // few rows: spread the irrep blocks over blockIdx.y so the chip still sees enough waves
    const int segs_per_block = tiles >= 2048 ? (int)n_segs : 1;
    dim3 grid((unsigned)matten_cdiv(tiles, SL_WAVES), (unsigned)matten_cdiv(n_segs, segs_per_block));
    species_linear_kernel<<<grid, SL_WAVES * 64, 0, stream>>>(x, (int)d_in, order, seg, (int)n_species, wp,
                                                            (int)w_stride, (const LinSeg*)segs, (int)n_segs,
                                                            segs_per_block, (int)d_out, add, (int)n_rows, out);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}
