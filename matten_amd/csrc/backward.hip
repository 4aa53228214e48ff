// Backward kernels of the conv stack (training step).
// The reference gets these from autograd through e3nn's einsums and torch_scatter (model/model.py:276-372);
// here each forward operator has a hand-written adjoint.  Training graphs are small (batch 32 crystals,
// ~150 nodes, ~4.5 k edges), so these kernels favour simplicity: one thread per output element or per
// (edge, channel), atomics only where contributions genuinely collide (gather adjoints).
#include <algorithm>

#include "cg_gen.h"
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// 'uvu' TP + gather + scatter, adjoint.  forward (per sorted edge e, weight column q = (path p, u)):
//   agg[dst, out_base + k] += norm(dst) * w[e,q] * sum_ij C_ijk x[src, x_base + i] Y[e, y_off + j]
// backward given G = d agg:
//   dw[e,q]              = norm * sum_{ijk} C_ijk x_i Y_j G[dst, out_base + k]
//   dx[src, x_base + i] += norm * w[e,q] * sum_{jk} C_ijk Y_j G[dst, out_base + k]      (atomic: many edges per src)
// col_meta[W,4]  = {x_base, out_base, nnz_begin, nnz_count | y_off << 16}
// nnz_ijk[nnz,4] = {i, j, k, 0} (uint8), nnz_c[nnz] = sqrt(2 l3+1) C_ijk
// ------------------------------------------------------------------------------------------------
__global__ void tp_backward_kernel(const float* __restrict__ x, int d_in, const void* __restrict__ w_edge, int w_ld,
                                   const float* __restrict__ sh, int sh_stride, const int32_t* __restrict__ src_sorted,
                                   const int32_t* __restrict__ dst_sorted, const int4* __restrict__ col_meta, int W,
                                   const uchar4* __restrict__ nnz_ijk, const float* __restrict__ nnz_c,
                                   const float* __restrict__ g_agg, int d_mid, float avg_nn,
                                   const float* __restrict__ num_neigh, int64_t E, float* __restrict__ dx,
                                   void* __restrict__ dw, int dw_ld, int edge_bf16) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= E * W) return;
    const int64_t e = idx / W;
    const int q = (int)(idx - e * W);
    const int4 m = col_meta[q];
    const int cnt = m.w & 0xffff, y_off = m.w >> 16;
    const int src = src_sorted[e], dst = dst_sorted[e];
    const float norm = 1.0f / sqrtf(avg_nn > 0.0f ? avg_nn : num_neigh[dst]);
    const float* xp = x + (int64_t)src * d_in + m.x;
    const float* yp = sh + e * sh_stride + y_off;
    const float* gp = g_agg + (int64_t)dst * d_mid + m.y;
    const float wv = matten_ld_edge(w_edge, e * w_ld + q, edge_bf16);
    float* dxp = dx + (int64_t)src * d_in + m.x;
    const float s = wv * norm;
    // The non-zeros of a coupling are stored i-major (plan.py: np.nonzero order), so the terms of one input component
    // are consecutive: sum them in a scalar and flush per component -- one x load and one atomic per component, no
    // per-term scatter into a register array (that was nine compare / select pairs per term).
    float dwv = 0.0f, acc = 0.0f;
    int cur = cnt > 0 ? (int)nnz_ijk[m.z].x : 0;
    for (int t = 0; t < cnt; ++t) {
        const uchar4 ijk = nnz_ijk[m.z + t];
        const float c = nnz_c[m.z + t];
        if ((int)ijk.x != cur) {
            dwv = fmaf(acc, xp[cur], dwv);
            if (acc != 0.0f) atomicAdd(dxp + cur, s * acc);
            cur = ijk.x;
            acc = 0.0f;
        }
        acc = fmaf(c * yp[ijk.y], gp[ijk.z], acc);
    }
    if (cnt > 0) {
        dwv = fmaf(acc, xp[cur], dwv);
        if (acc != 0.0f) atomicAdd(dxp + cur, s * acc);
    }
    matten_st_edge(dw, e * dw_ld + q, dwv * norm, edge_bf16);
}

// The same adjoint with the weight columns grouped by the INPUT channel they read (in_ptr / in_cols: columns of channel
// c are in_cols[in_ptr[c] .. in_ptr[c+1])): a thread owns (edge, input channel), walks that channel's 5-9 paths and adds
// their contributions to dx in registers, so the gather adjoint costs one atomic per (edge, channel, component)
// instead of one per (edge, path, channel, component).  The atomics were most of the per-column kernel's time
// (~120 M of them per layer at 290 k edges).
__global__ void tp_backward_grouped_kernel(const float* __restrict__ x, int d_in, const void* __restrict__ w_edge, int w_ld,
                                           const float* __restrict__ sh, int sh_stride,
                                           const int32_t* __restrict__ src_sorted, const int32_t* __restrict__ dst_sorted,
                                           const int4* __restrict__ col_meta, const int32_t* __restrict__ in_ptr,
                                           const int32_t* __restrict__ in_cols, int n_in,
                                           const uchar4* __restrict__ nnz_ijk, const float* __restrict__ nnz_c,
                                           const float* __restrict__ g_agg, int d_mid, float avg_nn,
                                           const float* __restrict__ num_neigh, int64_t E, float* __restrict__ dx,
                                           void* __restrict__ dw, int dw_ld, int edge_bf16) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= E * n_in) return;
    const int64_t e = idx / n_in;
    const int ch = (int)(idx - e * n_in);
    const int src = src_sorted[e], dst = dst_sorted[e];
    const float norm = 1.0f / sqrtf(avg_nn > 0.0f ? avg_nn : num_neigh[dst]);
    const float* yrow = sh + e * sh_stride;
    const float* grow = g_agg + (int64_t)dst * d_mid;
    float dxi[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) dxi[i] = 0.0f;
    const int c0 = in_ptr[ch], c1 = in_ptr[ch + 1];
    int x_base = 0;
    for (int cc = c0; cc < c1; ++cc) {
        const int q = in_cols[cc];
        const int4 m = col_meta[q];
        const int cnt = m.w & 0xffff;
        x_base = m.x;
        const float* xp = x + (int64_t)src * d_in + m.x;
        const float* yp = yrow + (m.w >> 16);
        const float* gp = grow + m.y;
        const float wv = matten_ld_edge(w_edge, e * w_ld + q, edge_bf16);
        float dwv = 0.0f, acc = 0.0f;
        int cur = cnt > 0 ? (int)nnz_ijk[m.z].x : 0;
        for (int t = 0; t <= cnt; ++t) {   // one extra round flushes the last component
            const bool more = t < cnt;
            const uchar4 ijk = more ? nnz_ijk[m.z + t] : uchar4{255, 0, 0, 0};
            if ((int)ijk.x != cur) {
                dwv = fmaf(acc, xp[cur], dwv);
                const float contrib = wv * acc;
#pragma unroll
                for (int i = 0; i < 9; ++i)
                    if (i == cur) dxi[i] += contrib;
                cur = ijk.x;
                acc = 0.0f;
            }
            if (more) acc = fmaf(nnz_c[m.z + t] * yp[ijk.y], gp[ijk.z], acc);
        }
        matten_st_edge(dw, e * dw_ld + q, dwv * norm, edge_bf16);
    }
    if (c1 > c0) {
        float* dxp = dx + (int64_t)src * d_in + x_base;
#pragma unroll
        for (int i = 0; i < 9; ++i)
            if (dxi[i] != 0.0f) atomicAdd(dxp + i, norm * dxi[i]);
    }
}

// ------------------------------------------------------------------------------------------------
// species linear, weight gradient:  dWp[s, w_off + u*mo + w] = sum_{rows n of species s} sum_k x[n, x_off+u*d+k] dY[n, o_off+w*d+k]
// A GEMM whose K dimension is (row, component): v_mfma_f32_16x16x4_f32 with the input channels u as M, the output
// channels w as N and four rows of the species per instruction (lane group g = row), one instruction per component k.
// A workgroup owns (species, row slice, segment); its four waves take every fourth row quad and keep up to 4 x 4 output
// tiles in registers, then add their fragments through LDS in wave order: the summation order is fixed (no atomics).
// At large batches a species' rows are cut into slices whose partial sums wgrad_reduce_kernel adds in slice order.
// (The first version was one thread per weight walking the species' rows with 4-byte strided reads: 2.0 ms of an
// 11 ms batch-2048 step.)
// ------------------------------------------------------------------------------------------------
struct LinSeg {
    int x_off, d, mul_in, w_off, mo, o_off, pad0, pad1;
};
typedef float wg_f32x4 __attribute__((ext_vector_type(4)));
#ifndef MATTEN_WG_TB
#define MATTEN_WG_TB 4
#endif
constexpr int WG_TB = MATTEN_WG_TB;   // tile block: WG_TB x WG_TB tiles of 16 x 16 weights per pass over the rows
// lab switches (tools/wgrad_ablate.sh): only with -DMATTEN_LAB
#if !defined(MATTEN_LAB) && (defined(MATTEN_WG_NO_MFMA) || defined(MATTEN_WG_NO_LOAD) || defined(MATTEN_WG_FORCE_SLICES))
#error "MATTEN_WG_* ablation switches need -DMATTEN_LAB"
#endif
#ifdef MATTEN_WG_NO_MFMA
#define WG_MFMA(a, b, c) ((c) + wg_f32x4{(a) * (b), 0.f, 0.f, 0.f})
#else
#define WG_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#endif
#ifdef MATTEN_WG_NO_LOAD
#define WG_LD(p) ((float)(uintptr_t)(&(p)) * 1e-9f)
#else
#define WG_LD(p) (p)
#endif
// Row slices.  Real data sets are far from uniform over species (the commonest of the 73 elements of the reference's
// sample owns 7 % of the rows, the rarest 0.2 %), and a workgroup that leaves at once still costs ~12 ns of chip time here
// (it reads its segment table and row range first; 16 slices for every species = 7008 workgroups per call: +84 us), so a
// species' rows are cut
// into slices of WG_SLICE_ROWS rows and the grid is the COMPACT list of (species, slice) items: item i belongs to the
// species whose running slice count covers i (every species owns at least one item and writes zeros when it has no
// rows).  n_items <= n_species + n_rows / WG_SLICE_ROWS, the grid size; the excess items leave.
#ifndef MATTEN_WG_SLICE_ROWS
#define MATTEN_WG_SLICE_ROWS 128
#endif
constexpr int WG_SLICE_ROWS = MATTEN_WG_SLICE_ROWS;
constexpr int WG_SLICED_MIN_ROWS = 4 * WG_SLICE_ROWS;   // below: one workgroup per species writing dwp directly
__host__ __device__ __forceinline__ int wgrad_items_of(int count) {
    return count > WG_SLICE_ROWS ? (count + WG_SLICE_ROWS - 1) / WG_SLICE_ROWS : 1;
}
// (species, slice) of item `item`, found by wave 0 with a 64-wide prefix sum over the species' item counts; species -1:
// no such item.  seg == nullptr: one species of n_rows rows.
__device__ __forceinline__ int2 wgrad_find_item(const int32_t* __restrict__ seg, int n_species, int n_rows, int item) {
    __shared__ int2 found;
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        int run = 0;
        int2 f = make_int2(-1, 0);
        for (int base = 0; base < n_species; base += 64) {
            const int t = base + lane;
            const int cnt = t < n_species ? (seg ? seg[t + 1] - seg[t] : n_rows) : 0;
            const int mine = t < n_species ? wgrad_items_of(cnt) : 0;
            int incl = mine;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int up = __shfl_up(incl, d, 64);
                if (lane >= d) incl += up;
            }
            const int rel = item - run;
            const unsigned long long hit = __ballot(t < n_species && rel >= incl - mine && rel < incl);
            if (hit) {
                const int l = __ffsll((long long)hit) - 1;
                f = make_int2(base + l, rel - (__shfl(incl, l, 64) - __shfl(mine, l, 64)));
                break;
            }
            run += __shfl(incl, 63, 64);
        }
        if (lane == 0) found = f;
    }
    __syncthreads();
    return found;
}
// rows lo + 4 wave + g, + 16, ... of the slice: D = 2 l + 1 components per channel (0: run-time count)
template <int D>
__device__ __forceinline__ void wgrad_rows(const float* __restrict__ x, int d_in, const float* __restrict__ dy, int d_out,
                                           const int32_t* __restrict__ order, const LinSeg& L, int lo, int hi, int wave,
                                           int g, int c, int mt0, int nt0, int MT, int NT, wg_f32x4 (&acc)[WG_TB][WG_TB]) {
    const int d = D ? D : L.d;
    const int r_first = lo + 4 * wave + g;
    int n_next = r_first < hi ? (order ? order[r_first] : r_first) : 0;
    for (int r = r_first; r - g < hi; r += 16) {
        const bool ok = r < hi;
        const int n = n_next;
        n_next = r + 16 < hi ? (order ? order[r + 16] : r + 16) : 0;   // the next quad's row index, one iteration ahead
        const float* xr = x + (int64_t)n * d_in + L.x_off;
        const float* gr = dy + (int64_t)n * d_out + L.o_off;
        if constexpr (D != 0) {
            float a[D][WG_TB], b[D][WG_TB];
#pragma unroll
            for (int k = 0; k < D; ++k) {
#pragma unroll
                for (int i = 0; i < WG_TB; ++i) {
                    const int u = 16 * (mt0 + i) + c;
                    a[k][i] = (ok && u < L.mul_in) ? WG_LD(xr[u * D + k]) : 0.0f;
                }
#pragma unroll
                for (int j = 0; j < WG_TB; ++j) {
                    const int w = 16 * (nt0 + j) + c;
                    b[k][j] = (ok && w < L.mo) ? WG_LD(gr[w * D + k]) : 0.0f;
                }
            }
#pragma unroll
            for (int i = 0; i < WG_TB; ++i)
#pragma unroll
                for (int j = 0; j < WG_TB; ++j)
                    if (mt0 + i < MT && nt0 + j < NT) {
#pragma unroll
                        for (int k = 0; k < D; ++k)
                            acc[i][j] = WG_MFMA(a[k][i], b[k][j], acc[i][j]);
                    }
        } else {
            for (int k = 0; k < d; ++k) {
                float a[WG_TB], b[WG_TB];
#pragma unroll
                for (int i = 0; i < WG_TB; ++i) {
                    const int u = 16 * (mt0 + i) + c;
                    a[i] = (ok && u < L.mul_in) ? WG_LD(xr[u * d + k]) : 0.0f;
                }
#pragma unroll
                for (int j = 0; j < WG_TB; ++j) {
                    const int w = 16 * (nt0 + j) + c;
                    b[j] = (ok && w < L.mo) ? WG_LD(gr[w * d + k]) : 0.0f;
                }
#pragma unroll
                for (int i = 0; i < WG_TB; ++i)
#pragma unroll
                    for (int j = 0; j < WG_TB; ++j)
                        if (mt0 + i < MT && nt0 + j < NT)
                            acc[i][j] = WG_MFMA(a[i], b[j], acc[i][j]);
            }
        }
    }
}

__global__ __launch_bounds__(256) void species_linear_wgrad_kernel(const float* __restrict__ x, int d_in,
                                                                   const float* __restrict__ dy, int d_out,
                                                                   const int32_t* __restrict__ order,
                                                                   const int32_t* __restrict__ seg, int n_rows,
                                                                   const LinSeg* __restrict__ segs, int w_stride,
                                                                   float* __restrict__ dwp, float* __restrict__ partial,
                                                                   int32_t* __restrict__ first, int n_species) {
    __shared__ __attribute__((aligned(16))) float red[4][256];
    // partial != nullptr (sliced): blockIdx.x is an item of the compact (species, slice) list; a species of several items
    // writes partial row blockIdx.x and notes its first item for wgrad_reduce_kernel, a species of one item (most of
    // them) and the unsliced form (blockIdx.x = species) write dwp
    const bool sliced = partial != nullptr;
    int s = blockIdx.x, z = 0;
    if (sliced) {
        const int2 it = wgrad_find_item(seg, n_species, n_rows, (int)blockIdx.x);
        if (it.x < 0) return;
        s = it.x, z = it.y;
    }
    const LinSeg L = segs[blockIdx.y];
    int lo = seg ? seg[s] : 0, hi = seg ? seg[s + 1] : n_rows;
    float* outp = dwp + (int64_t)s * w_stride + L.w_off;
    if (sliced) {
        if (hi - lo > WG_SLICE_ROWS) {
            outp = partial + (int64_t)blockIdx.x * w_stride + L.w_off;
            if (z == 0 && blockIdx.y == 0 && threadIdx.x == 0) first[s] = (int)blockIdx.x;
        }
        lo += z * WG_SLICE_ROWS;
        hi = min(hi, lo + WG_SLICE_ROWS);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int MT = (L.mul_in + 15) >> 4, NT = (L.mo + 15) >> 4;
    for (int mt0 = 0; mt0 < MT; mt0 += WG_TB) {
        for (int nt0 = 0; nt0 < NT; nt0 += WG_TB) {
            wg_f32x4 acc[WG_TB][WG_TB];
#pragma unroll
            for (int i = 0; i < WG_TB; ++i)
#pragma unroll
                for (int j = 0; j < WG_TB; ++j) acc[i][j] = wg_f32x4{0.f, 0.f, 0.f, 0.f};
            switch (L.d) {   // the component loop unrolled: all of a row quad's loads are in flight before its first MFMA
                case 1: wgrad_rows<1>(x, d_in, dy, d_out, order, L, lo, hi, wave, g, c, mt0, nt0, MT, NT, acc); break;
                case 3: wgrad_rows<3>(x, d_in, dy, d_out, order, L, lo, hi, wave, g, c, mt0, nt0, MT, NT, acc); break;
                case 5: wgrad_rows<5>(x, d_in, dy, d_out, order, L, lo, hi, wave, g, c, mt0, nt0, MT, NT, acc); break;
                case 7: wgrad_rows<7>(x, d_in, dy, d_out, order, L, lo, hi, wave, g, c, mt0, nt0, MT, NT, acc); break;
                case 9: wgrad_rows<9>(x, d_in, dy, d_out, order, L, lo, hi, wave, g, c, mt0, nt0, MT, NT, acc); break;
                default: wgrad_rows<0>(x, d_in, dy, d_out, order, L, lo, hi, wave, g, c, mt0, nt0, MT, NT, acc); break;
            }
            // D fragment of lane (g, c): rows u = 4 g + r, column w = c.  Waves add up in wave order.
#pragma unroll
            for (int i = 0; i < WG_TB; ++i)
#pragma unroll
                for (int j = 0; j < WG_TB; ++j) {
                    if (mt0 + i < MT && nt0 + j < NT) {   // uniform over the workgroup
                        *reinterpret_cast<wg_f32x4*>(&red[wave][4 * lane]) = acc[i][j];
                        __syncthreads();
                        const int t = threadIdx.x, ln = t >> 2, rr = t & 3;
                        const int u = 16 * (mt0 + i) + 4 * (ln >> 4) + rr, w = 16 * (nt0 + j) + (ln & 15);
                        const float v = ((red[0][t] + red[1][t]) + red[2][t]) + red[3][t];
                        if (u < L.mul_in && w < L.mo) outp[u * L.mo + w] = v;
                        __syncthreads();
                    }
                }
        }
    }
}

// dWp[s, q] = sum over the species' items (in slice order) of partial[item, q], for the weights of this call's segments
// and the species of several items (the others wrote dwp themselves)
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ partial, const int32_t* __restrict__ first,
                                    const LinSeg* __restrict__ segs, int w_stride, const int32_t* __restrict__ seg,
                                    int n_rows, float* __restrict__ dwp) {
    const int s = blockIdx.x;
    const int nz = wgrad_items_of(seg ? seg[s + 1] - seg[s] : n_rows);
    if (nz == 1) return;
    const LinSeg L = segs[blockIdx.y];
    const float* p0 = partial + (int64_t)first[s] * w_stride + L.w_off;
    for (int p = threadIdx.x; p < L.mul_in * L.mo; p += blockDim.x) {
        float v = 0.0f;
        for (int z0 = 0; z0 < nz; z0 += 8) {   // eight loads in flight, added in slice order
            float t[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] = z0 + i < nz ? p0[(int64_t)(z0 + i) * w_stride + p] : 0.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) v += t[i];
        }
        dwp[(int64_t)s * w_stride + L.w_off + p] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// Gate, adjoint (forward: node.hip gate_bn_kernel without BatchNorm)
// meta[d_out] int4 {src, gate(-1: scalar), act | gate_act<<8, unused}
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }
__device__ __forceinline__ float act_f(int code, float v) {
    switch (code) {
        case 1: return v * sigmoidf_(v);
        case 2: return tanhf(v);
        case 3: return sigmoidf_(v);
        case 4: return (v > 20.0f ? v : log1pf(expf(v))) - 0.6931471805599453f;
        case 5: return fabsf(v);
        default: return v;
    }
}
__device__ __forceinline__ float act_df(int code, float v) {
    switch (code) {
        case 1: { float s = sigmoidf_(v); return s * (1.0f + v * (1.0f - s)); }
        case 2: { float t = tanhf(v); return 1.0f - t * t; }
        case 3: { float s = sigmoidf_(v); return s * (1.0f - s); }
        case 4: return sigmoidf_(v);
        case 5: return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f);
        default: return 1.0f;
    }
}

__global__ void gate_bwd_kernel(const float* __restrict__ x, int d_in, const int4* __restrict__ meta, int d_out,
                                const float* __restrict__ act_cst, const float* __restrict__ dy, int64_t n_rows,
                                float* __restrict__ dx) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_rows * d_out) return;
    int64_t n = idx / d_out;
    int o = (int)(idx - n * d_out);
    int4 m = meta[o];
    const float* xr = x + n * d_in;
    float* dxr = dx + n * d_in;
    const float g = dy[idx];
    const int act = m.z & 0xff, gact = (m.z >> 8) & 0xff;
    if (m.y < 0) {
        const float v = xr[m.x];
        dxr[m.x] = act ? g * act_df(act, v) * act_cst[act] : g;  // one output per scalar input: plain store
    } else {
        const float v = xr[m.x], gt = xr[m.y];
        const float a = gact ? act_f(gact, gt) * act_cst[gact] : gt;
        const float da = gact ? act_df(gact, gt) * act_cst[gact] : 1.0f;
        dxr[m.x] = g * a;
        // the 2l+1 components of a gated channel share one gate (and no other channel uses it): the thread of the first
        // component sums the channel's terms in component order -- a plain store, no atomics, every input column written
        if (o == 0 || meta[o - 1].y != m.y) {
            float sum = 0.0f;
            for (int kk = 0; o + kk < d_out && meta[o + kk].y == m.y; ++kk) sum = fmaf(dy[idx + kk], xr[meta[o + kk].x], sum);
            dxr[m.y] = sum * da;
        }
        (void)v;
    }
}

// ------------------------------------------------------------------------------------------------
// e3nn NormActivation (reference nn/utils.py:142-150: nonlinearity_type "norm"; normalize = True, epsilon = 1e-8, no
// bias, the even-scalar activation used as given): channel c (2l+1 components at chan[c].x) is scaled by f(n) / n,
// n = sqrt(max(sum_k x_k^2, eps^2)); optionally followed by eval-mode BatchNorm (per-channel scale, shift on 0e).
// One thread per (row, channel).  chan[C] int4 {offset, d, is_0e, mean index} (plan_batchnorm).
// ------------------------------------------------------------------------------------------------
__global__ void norm_act_kernel(const float* __restrict__ x, int dim, int64_t n_rows, const int4* __restrict__ chan,
                                int n_chan, int act, float eps2, const float* __restrict__ running_mean,
                                const float* __restrict__ running_var, const float* __restrict__ bn_weight,
                                const float* __restrict__ bn_bias, float bn_eps, float* __restrict__ y) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_rows * n_chan) return;
    const int64_t n = idx / n_chan;
    const int c = (int)(idx - n * n_chan);
    const int4 ch = chan[c];
    const float* xp = x + n * dim + ch.x;
    float n2 = 0.0f;
    for (int k = 0; k < ch.y; ++k) n2 = fmaf(xp[k], xp[k], n2);
    const float nn = sqrtf(fmaxf(n2, eps2));
    float s = act_f(act, nn) / nn;
    float shift = 0.0f;
    if (bn_weight) {
        const float bs = bn_weight[c] / sqrtf(running_var[c] + bn_eps);
        if (ch.z) shift = bn_bias[ch.w] - running_mean[ch.w] * bs;
        s *= bs;
    }
    float* yp = y + n * dim + ch.x;
    for (int k = 0; k < ch.y; ++k) yp[k] = fmaf(xp[k], s, shift);
}

// adjoint (no BatchNorm folded in): dx_k = s g_k + [n^2 >= eps^2] (f'(n) n - f(n)) / n^3 (sum_j g_j x_j) x_k
__global__ void norm_act_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, int dim, int64_t n_rows,
                                    const int4* __restrict__ chan, int n_chan, int act, float eps2,
                                    float* __restrict__ dx) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_rows * n_chan) return;
    const int64_t n = idx / n_chan;
    const int c = (int)(idx - n * n_chan);
    const int4 ch = chan[c];
    const float* xp = x + n * dim + ch.x;
    const float* gp = dy + n * dim + ch.x;
    float n2 = 0.0f, gx = 0.0f;
    for (int k = 0; k < ch.y; ++k) {
        n2 = fmaf(xp[k], xp[k], n2);
        gx = fmaf(gp[k], xp[k], gx);
    }
    const float nn = sqrtf(fmaxf(n2, eps2));
    const float f = act_f(act, nn), s = f / nn;
    const float t = n2 < eps2 ? 0.0f : (act_df(act, nn) * nn - f) / (nn * nn * nn) * gx;   // clamped norm: constant
    float* dp = dx + n * dim + ch.x;
    for (int k = 0; k < ch.y; ++k) dp[k] = fmaf(t, xp[k], s * gp[k]);
}

// ------------------------------------------------------------------------------------------------
// e3nn BatchNorm, training mode (reference nn/utils.py:418 -> e3nn BatchNorm.forward with self.training).
// Channel c = one multiplicity index of one irrep block (d components, offset off[c]); 0e channels are
// centred.  stats: mean[c] (0 for non-scalars), nu[c] = mean_n mean_k (x - mean)^2.
// chan[C] int4 {offset, d, is_scalar, mean_idx(-1)}
// ------------------------------------------------------------------------------------------------
// Segmented form (instance / graph normalisation, reference nn/utils.py:448-588: one set of statistics per crystal):
// seg_ptr[B+1] != NULL, grid.y = B, rows [seg_ptr[b], seg_ptr[b+1]) -> mean / nu [B, C]; seg_ptr == NULL: the whole batch.
__global__ void bn_stats_kernel(const float* __restrict__ x, int dim, int64_t n_rows_all, const int4* __restrict__ chan,
                                float* __restrict__ mean, float* __restrict__ nu, const int64_t* __restrict__ seg_ptr,
                                float* __restrict__ running_mean, float* __restrict__ running_var, float momentum) {
    __shared__ float red[256];
    const int c = blockIdx.x;
    const int4 ch = chan[c];
    const int d = ch.y;
    const int64_t r0 = seg_ptr ? seg_ptr[blockIdx.y] : 0;
    const int64_t n_rows = seg_ptr ? seg_ptr[blockIdx.y + 1] - r0 : n_rows_all;
    x += r0 * dim;
    mean += (int64_t)blockIdx.y * gridDim.x;
    nu += (int64_t)blockIdx.y * gridDim.x;
    if (n_rows <= 0) {   // an empty crystal: no rows will read these
        if (threadIdx.x == 0) mean[c] = 0.0f, nu[c] = 0.0f;
        return;
    }
    float s = 0.0f;
    if (ch.z) {
        {   // four rows in flight per thread (a thread's rows are dim floats apart: every load is its own cache line, the
            // loop is a chain of load latencies), partial sums combined in a fixed order
            float s4[4] = {0.f, 0.f, 0.f, 0.f};
            const int64_t step = blockDim.x;
            int64_t n = threadIdx.x;
            for (; n + 3 * step < n_rows; n += 4 * step) {
#pragma unroll
                for (int r = 0; r < 4; ++r) s4[r] += x[(n + r * step) * dim + ch.x];
            }
            for (int r = 0; n < n_rows; n += step, ++r) s4[r] += x[n * dim + ch.x];
            s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        }
        red[threadIdx.x] = s;
        __syncthreads();
        for (int o = blockDim.x / 2; o > 0; o >>= 1) {
            if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        s = red[0] / (float)n_rows;
        __syncthreads();
    }
    const float mu = ch.z ? s : 0.0f;
    float q = 0.0f;
    {
        float q4[4] = {0.f, 0.f, 0.f, 0.f};
        const int64_t step = blockDim.x;
        int64_t n = threadIdx.x;
        for (; n + 3 * step < n_rows; n += 4 * step)
            for (int k = 0; k < d; ++k) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = x[(n + r * step) * dim + ch.x + k] - mu;
#pragma unroll
                for (int r = 0; r < 4; ++r) q4[r] = fmaf(v[r], v[r], q4[r]);
            }
        for (int r = 0; n < n_rows; n += step, ++r)
            for (int k = 0; k < d; ++k) {
                const float v = x[n * dim + ch.x + k] - mu;
                q4[r] = fmaf(v, v, q4[r]);
            }
        q = (q4[0] + q4[1]) + (q4[2] + q4[3]);
    }
    red[threadIdx.x] = q;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float v = red[0] / ((float)n_rows * (float)d);
        mean[c] = mu;
        nu[c] = v;
        if (running_var) {   // e3nn: running = (1 - momentum) running + momentum batch (whole-batch statistics only)
            running_var[c] = (1.0f - momentum) * running_var[c] + momentum * v;
            if (ch.z) running_mean[ch.w] = (1.0f - momentum) * running_mean[ch.w] + momentum * mu;
        }
    }
}

// ---- whole-batch statistics at large row counts: two stages, thread = column -------------------------------------
// bn_stats_kernel gives a channel to a workgroup whose threads walk rows: neighbouring lanes read addresses one ROW
// apart, every 4-byte load pulls its own cache line (32x the bytes through L1/L2: 47 us at 9652 rows x 200 columns).
// Here a workgroup takes BN_RB consecutive rows and thread j the column j: coalesced row reads, one partial record per
// (row block, column); a second launch merges a channel's records in a fixed order.  Centred (0e) columns carry
// (block mean, block sum of squared deviations) merged pairwise with the parallel-variance formula (no E[x^2] - mean^2
// cancellation), the others the block sum of squares.
constexpr int BN_RB = 16;             // rows per block
constexpr int BN_COLS_MIN_ROWS = 2048;   // below: the one-launch kernels
__global__ __launch_bounds__(256) void bn_stats_cols_kernel(const float* __restrict__ x, int dim, int64_t n_rows,
                                                            const int32_t* __restrict__ col2chan,
                                                            const int4* __restrict__ chan, float2* __restrict__ part) {
    const int64_t r0 = (int64_t)blockIdx.x * BN_RB;
    const int nr = (int)min((int64_t)BN_RB, n_rows - r0);
    for (int col = threadIdx.x; col < dim; col += blockDim.x) {
        float v[BN_RB];
#pragma unroll
        for (int r = 0; r < BN_RB; ++r) v[r] = r < nr ? x[(r0 + r) * dim + col] : 0.0f;
        float m = 0.0f, q = 0.0f;
        if (chan[col2chan[col]].z) {
#pragma unroll
            for (int r = 0; r < BN_RB; ++r) m += v[r];
            m /= (float)nr;
#pragma unroll
            for (int r = 0; r < BN_RB; ++r) {
                const float dlt = r < nr ? v[r] - m : 0.0f;
                q = fmaf(dlt, dlt, q);
            }
        } else {
#pragma unroll
            for (int r = 0; r < BN_RB; ++r) q = fmaf(v[r], v[r], q);
        }
        part[(int64_t)blockIdx.x * dim + col] = make_float2(m, q);
    }
}

// (count, mean, M2) of two disjoint sets -> of their union
__device__ __forceinline__ void bn_merge(float& na, float& ma, float& qa, float nb, float mb, float qb) {
    if (nb == 0.0f) return;
    const float n = na + nb, dlt = mb - ma;
    ma = fmaf(dlt, nb / n, ma);
    qa = qa + qb + dlt * dlt * (na * nb / n);
    na = n;
}

__global__ __launch_bounds__(256) void bn_stats_finish_kernel(const float2* __restrict__ part, int dim, int64_t n_rows,
                                                              int n_blocks, const int4* __restrict__ chan,
                                                              float* __restrict__ mean, float* __restrict__ nu,
                                                              float* __restrict__ running_mean,
                                                              float* __restrict__ running_var, float momentum) {
    __shared__ float rn[256], rm[256], rq[256];
    const int c = blockIdx.x, t = threadIdx.x;
    const int4 ch = chan[c];
    float n = 0.0f, m = 0.0f, q = 0.0f;
    if (ch.z) {   // one column: thread t merges blocks t, t + 256, ... in order
        for (int b = t; b < n_blocks; b += 256) {
            const float2 p = part[(int64_t)b * dim + ch.x];
            bn_merge(n, m, q, (float)min((int64_t)BN_RB, n_rows - (int64_t)b * BN_RB), p.x, p.y);
        }
    } else {
        const int items = n_blocks * ch.y;
        for (int i = t; i < items; i += 256) q += part[(int64_t)(i / ch.y) * dim + ch.x + i % ch.y].y;
    }
    rn[t] = n, rm[t] = m, rq[t] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) {
            if (ch.z) {
                bn_merge(rn[t], rm[t], rq[t], rn[t + o], rm[t + o], rq[t + o]);
            } else {
                rq[t] += rq[t + o];
            }
        }
        __syncthreads();
    }
    if (t == 0) {
        const float mu = ch.z ? rm[0] : 0.0f;
        const float v = rq[0] / ((float)n_rows * (float)ch.y);
        mean[c] = mu;
        nu[c] = v;
        if (running_var) {
            running_var[c] = (1.0f - momentum) * running_var[c] + momentum * v;
            if (ch.z) running_mean[ch.w] = (1.0f - momentum) * running_mean[ch.w] + momentum * mu;
        }
    }
}

// adjoint reductions, same two stages: part[block, column] = (sum dy (x - mean), sum dy) over the block's rows
__global__ __launch_bounds__(256) void bn_bwd_cols_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          int dim, int64_t n_rows, const int32_t* __restrict__ col2chan,
                                                          const float* __restrict__ mean, float2* __restrict__ part) {
    const int64_t r0 = (int64_t)blockIdx.x * BN_RB;
    const int nr = (int)min((int64_t)BN_RB, n_rows - r0);
    for (int col = threadIdx.x; col < dim; col += blockDim.x) {
        const float mu = mean[col2chan[col]];
        float xv[BN_RB], g[BN_RB];
#pragma unroll
        for (int r = 0; r < BN_RB; ++r) {
            xv[r] = r < nr ? x[(r0 + r) * dim + col] : mu;
            g[r] = r < nr ? dy[(r0 + r) * dim + col] : 0.0f;
        }
        float a = 0.0f, b = 0.0f;
#pragma unroll
        for (int r = 0; r < BN_RB; ++r) {
            a = fmaf(g[r], xv[r] - mu, a);
            b += g[r];
        }
        part[(int64_t)blockIdx.x * dim + col] = make_float2(a, b);
    }
}

__global__ __launch_bounds__(256) void bn_bwd_finish_kernel(const float2* __restrict__ part, int dim, int n_blocks,
                                                            const int4* __restrict__ chan, float* __restrict__ A,
                                                            float* __restrict__ B, const float* __restrict__ nu,
                                                            float eps, float* __restrict__ dweight,
                                                            float* __restrict__ dbias) {
    __shared__ float ra[256], rb[256];
    const int c = blockIdx.x, t = threadIdx.x;
    const int4 ch = chan[c];
    float a = 0.0f, b = 0.0f;
    const int items = n_blocks * ch.y;
    for (int i = t; i < items; i += 256) {
        const float2 p = part[(int64_t)(i / ch.y) * dim + ch.x + i % ch.y];
        a += p.x;
        b += p.y;
    }
    ra[t] = a, rb[t] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) ra[t] += ra[t + o], rb[t] += rb[t + o];
        __syncthreads();
    }
    if (t == 0) {
        A[c] = ra[0];
        B[c] = rb[0];
        if (dweight) {
            dweight[c] = ra[0] * rsqrtf(nu[c] + eps);
            if (ch.z) dbias[ch.w] = rb[0];
        }
    }
}

// y = (x - mean) * rsqrt(nu + eps) * weight + bias(0e only)
__global__ void bn_apply_kernel(const float* __restrict__ x, int dim, int64_t n_rows, const int32_t* __restrict__ col2chan,
                                const int4* __restrict__ chan, const float* __restrict__ mean,
                                const float* __restrict__ nu, const float* __restrict__ weight,
                                const float* __restrict__ bias, float eps, float* __restrict__ y,
                                const int64_t* __restrict__ seg_of_row, int n_chan) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_rows * dim) return;
    const int col = (int)(idx % dim);
    const int c = col2chan[col];
    const int4 ch = chan[c];
    const int64_t sc = seg_of_row ? seg_of_row[idx / dim] * n_chan + c : c;   // statistics of the row's crystal
    float v = (x[idx] - mean[sc]) * rsqrtf(nu[sc] + eps) * weight[c];
    if (ch.z) v += bias[ch.w];
    y[idx] = v;
}

// reductions of the adjoint: A[c] = sum dy (x - mean), B[c] = sum dy   (over rows and components)
__global__ void bn_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dy, int dim, int64_t n_rows,
                                     const int4* __restrict__ chan, const float* __restrict__ mean,
                                     float* __restrict__ A, float* __restrict__ B, const int64_t* __restrict__ seg_ptr,
                                     const float* __restrict__ nu, float eps, float* __restrict__ dweight,
                                     float* __restrict__ dbias) {
    __shared__ float ra[256], rb[256];
    const int c = blockIdx.x;
    const int4 ch = chan[c];
    if (seg_ptr) {   // segment blockIdx.y (see bn_stats_kernel)
        const int64_t r0 = seg_ptr[blockIdx.y];
        n_rows = seg_ptr[blockIdx.y + 1] - r0;
        x += r0 * dim;
        dy += r0 * dim;
        mean += (int64_t)blockIdx.y * gridDim.x;
        A += (int64_t)blockIdx.y * gridDim.x;
        B += (int64_t)blockIdx.y * gridDim.x;
    }
    const float mu = mean[c];
    float a = 0.0f, b = 0.0f;
    {   // four rows in flight per thread (see bn_stats_kernel), partial sums combined in a fixed order
        float a4[4] = {0.f, 0.f, 0.f, 0.f}, b4[4] = {0.f, 0.f, 0.f, 0.f};
        const int64_t step = blockDim.x;
        int64_t n = threadIdx.x;
        for (; n + 3 * step < n_rows; n += 4 * step)
            for (int k = 0; k < ch.y; ++k) {
                float g[4], xv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    g[r] = dy[(n + r * step) * dim + ch.x + k];
                    xv[r] = x[(n + r * step) * dim + ch.x + k];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    a4[r] = fmaf(g[r], xv[r] - mu, a4[r]);
                    b4[r] += g[r];
                }
            }
        for (int r = 0; n < n_rows; n += step, ++r)
            for (int k = 0; k < ch.y; ++k) {
                const float g = dy[n * dim + ch.x + k];
                a4[r] = fmaf(g, x[n * dim + ch.x + k] - mu, a4[r]);
                b4[r] += g;
            }
        a = (a4[0] + a4[1]) + (a4[2] + a4[3]);
        b = (b4[0] + b4[1]) + (b4[2] + b4[3]);
    }
    ra[threadIdx.x] = a;
    rb[threadIdx.x] = b;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            ra[threadIdx.x] += ra[threadIdx.x + o];
            rb[threadIdx.x] += rb[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        A[c] = ra[0];
        B[c] = rb[0];
        if (dweight) {   // whole-batch statistics only: the parameter gradients right here
            dweight[c] = ra[0] * rsqrtf(nu[c] + eps);
            if (ch.z) dbias[ch.w] = rb[0];
        }
    }
}

// dx = weight * s * [ dy - (x-mean) * s^2 * A/(N d) - (0e ? B/N : 0) ],  s = rsqrt(nu+eps)
__global__ void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy, int dim, int64_t n_rows,
                                    const int32_t* __restrict__ col2chan, const int4* __restrict__ chan,
                                    const float* __restrict__ mean, const float* __restrict__ nu,
                                    const float* __restrict__ weight, const float* __restrict__ A,
                                    const float* __restrict__ B, float eps, float* __restrict__ dx,
                                    const int64_t* __restrict__ seg_of_row, const int64_t* __restrict__ seg_ptr,
                                    int n_chan) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_rows * dim) return;
    const int col = (int)(idx % dim);
    const int c = col2chan[col];
    const int4 ch = chan[c];
    int64_t sc = c;
    float rows = (float)n_rows;
    if (seg_of_row) {
        const int64_t b = seg_of_row[idx / dim];
        sc = b * n_chan + c;
        rows = (float)(seg_ptr[b + 1] - seg_ptr[b]);
    }
    const float s = rsqrtf(nu[sc] + eps);
    const float nd = rows * (float)ch.y;
    float v = dy[idx] - (x[idx] - mean[sc]) * s * s * A[sc] / nd;
    if (ch.z) v -= B[sc] / rows;
    dx[idx] = weight[c] * s * v;
}

// NodewiseReduce adjoint: dx[n,:] = dy[batch(n),:] / (mean ? max(count,1) : 1)
__global__ void segment_reduce_bwd_kernel(const float* __restrict__ dy, int dim, const int64_t* __restrict__ ptr,
                                          int64_t n_seg, int mean, float* __restrict__ dx) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_seg * dim) return;
    const int64_t b = idx / dim;
    const int c = (int)(idx - b * dim);
    const int64_t beg = ptr[b], end = ptr[b + 1];
    float g = dy[idx];
    if (mean) {
        float cnt = (float)(end - beg);
        g /= (cnt < 1.0f ? 1.0f : cnt);
    }
    for (int64_t n = beg; n < end; ++n) dx[n * dim + c] = g;
}

}  // namespace

extern "C" int matten_tp_backward(const float* x, int64_t d_in, const void* w_edge, int64_t w_ld, const float* sh_sorted,
                                  int64_t sh_stride, const int32_t* src_sorted, const int32_t* dst_sorted,
                                  const int32_t* col_meta, int64_t n_cols, const uint8_t* nnz_ijk, const float* nnz_c,
                                  const float* g_agg, int64_t d_mid, float avg_num_neighbors, const float* num_neigh,
                                  int64_t n_edges, float* dx /*zero-initialised [N,d_in]*/, void* dw, int64_t dw_ld,
                                  const int32_t* in_ptr, const int32_t* in_cols, int64_t n_in, int edge_is_bf16,
                                  matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_edges < 0 || d_in <= 0 || n_cols <= 0 || d_mid <= 0 || w_ld < n_cols || dw_ld < n_cols) return MATTEN_EINVAL;
    if (n_edges == 0) return MATTEN_OK;
    if (!x || !w_edge || !sh_sorted || !src_sorted || !dst_sorted || !col_meta || !nnz_ijk || !nnz_c || !g_agg || !dx || !dw)
        return MATTEN_EINVAL;
    if (!(avg_num_neighbors > 0.0f) && !num_neigh) return MATTEN_EINVAL;
    const int T = 256;
    if ((in_ptr == nullptr) != (in_cols == nullptr) || (in_ptr && n_in <= 0)) return MATTEN_EINVAL;
    if (in_ptr) {
        if (matten_cdiv(n_edges * n_in, T) >= ((int64_t)1 << 31)) return MATTEN_EINVAL;
        tp_backward_grouped_kernel<<<(unsigned)matten_cdiv(n_edges * n_in, T), T, 0, stream>>>(
            x, (int)d_in, w_edge, (int)w_ld, sh_sorted, (int)sh_stride, src_sorted, dst_sorted, (const int4*)col_meta,
            in_ptr, in_cols, (int)n_in, (const uchar4*)nnz_ijk, nnz_c, g_agg, (int)d_mid, avg_num_neighbors, num_neigh,
            n_edges, dx, dw, (int)dw_ld, edge_is_bf16);
        MATTEN_LAUNCH_CHECK();
        return MATTEN_OK;
    }
    if (matten_cdiv(n_edges * n_cols, T) >= ((int64_t)1 << 31)) return MATTEN_EINVAL;
    tp_backward_kernel<<<(unsigned)matten_cdiv(n_edges * n_cols, T), T, 0, stream>>>(
        x, (int)d_in, w_edge, (int)w_ld, sh_sorted, (int)sh_stride, src_sorted, dst_sorted, (const int4*)col_meta,
        (int)n_cols, (const uchar4*)nnz_ijk, nnz_c, g_agg, (int)d_mid, avg_num_neighbors, num_neigh, n_edges, dx, dw,
        (int)dw_ld, edge_is_bf16);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}

// Floats of scratch a call needs.  0: none -- one workgroup per species writes dwp directly; else the compact
// (species, slice) item list (see WG_SLICE_ROWS) is in use: [items, w_stride] partial sums, written by the species of
// several items and added in slice order by a second launch, + one int per species (its first item).  Either way
// every packed weight of the table's segments is written exactly once (dwp need not be initialised) and the summation
// order is fixed.
static int64_t wgrad_items_bound(int64_t n_rows, int64_t n_species) {
#ifdef MATTEN_WG_FORCE_SLICES
    if (MATTEN_WG_FORCE_SLICES == 1) return 0;
#endif
    if (n_rows < WG_SLICED_MIN_ROWS) return 0;
    return std::max<int64_t>(1, n_species) + n_rows / WG_SLICE_ROWS;
}
extern "C" int64_t matten_species_linear_wgrad_scratch_floats(int64_t n_rows, int64_t n_species, int64_t w_stride) {
    const int64_t items = wgrad_items_bound(n_rows, n_species);
    return items ? items * w_stride + std::max<int64_t>(1, n_species) : 0;
}

extern "C" int matten_species_linear_wgrad(const float* x, int64_t d_in, const float* dy, int64_t d_out,
                                           const int32_t* order, const int32_t* seg, int64_t n_species, int64_t n_rows,
                                           const int32_t* segs, int64_t n_segs, int64_t w_stride, float* dwp,
                                           float* scratch, matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_rows < 0 || d_in <= 0 || d_out <= 0 || n_species <= 0 || n_segs < 0 || w_stride < 0) return MATTEN_EINVAL;
    if (n_segs == 0 || w_stride == 0) return MATTEN_OK;
    if (!x || !dy || !segs || !dwp) return MATTEN_EINVAL;
    if ((order == nullptr) != (seg == nullptr)) return MATTEN_EINVAL;
    if (!order && n_species != 1) return MATTEN_EINVAL;
    const int64_t items = wgrad_items_bound(n_rows, n_species);
    if (items && !scratch) return MATTEN_EINVAL;
    if (items >= ((int64_t)1 << 31) || n_species >= ((int64_t)1 << 31) || n_segs > 65535) return MATTEN_EINVAL;
    int32_t* first = items ? reinterpret_cast<int32_t*>(scratch + items * w_stride) : nullptr;
    dim3 grid((unsigned)(items ? items : n_species), (unsigned)n_segs);
    species_linear_wgrad_kernel<<<grid, 256, 0, stream>>>(x, (int)d_in, dy, (int)d_out, order, seg, (int)n_rows,
                                                          (const LinSeg*)segs, (int)w_stride, dwp,
                                                          items ? scratch : nullptr, first, (int)n_species);
    MATTEN_LAUNCH_CHECK();
    if (items) {
        wgrad_reduce_kernel<<<dim3((unsigned)n_species, (unsigned)n_segs), 1024, 0, stream>>>(
            scratch, first, (const LinSeg*)segs, (int)w_stride, seg, (int)n_rows, dwp);
        MATTEN_LAUNCH_CHECK();
    }
    return MATTEN_OK;
}

extern "C" int matten_gate_bwd(const float* x, int64_t d_in, const int32_t* meta, int64_t d_out, const float* act_cst,
                               const float* dy, int64_t n_rows, float* dx /*every column written*/, matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_rows < 0 || d_in <= 0 || d_out <= 0) return MATTEN_EINVAL;
    if (n_rows == 0) return MATTEN_OK;
    if (!x || !meta || !act_cst || !dy || !dx) return MATTEN_EINVAL;
    const int T = 256;
    gate_bwd_kernel<<<(unsigned)matten_cdiv(n_rows * d_out, T), T, 0, stream>>>(x, (int)d_in, (const int4*)meta,
                                                                                (int)d_out, act_cst, dy, n_rows, dx);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}

extern "C" int matten_norm_act(const float* x, int64_t dim, int64_t n_rows, const int32_t* chan, int64_t n_chan, int act,
                               float epsilon, const float* running_mean, const float* running_var,
                               const float* bn_weight, const float* bn_bias, float bn_eps, float* y,
                               matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_rows < 0 || dim <= 0 || n_chan <= 0 || act < 1 || act > 5 || !(epsilon > 0.0f)) return MATTEN_EINVAL;
    if (n_rows == 0) return MATTEN_OK;
    if (!x || !chan || !y) return MATTEN_EINVAL;
    if (bn_weight && (!running_var || !running_mean || !bn_bias)) return MATTEN_EINVAL;
    const int T = 256;
    norm_act_kernel<<<(unsigned)matten_cdiv(n_rows * n_chan, T), T, 0, stream>>>(
        x, (int)dim, n_rows, (const int4*)chan, (int)n_chan, act, epsilon * epsilon, running_mean, running_var, bn_weight,
        bn_bias, bn_eps, y);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}

extern "C" int matten_norm_act_bwd(const float* x, const float* dy, int64_t dim, int64_t n_rows, const int32_t* chan,
                                   int64_t n_chan, int act, float epsilon, float* dx, matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_rows < 0 || dim <= 0 || n_chan <= 0 || act < 1 || act > 5 || !(epsilon > 0.0f)) return MATTEN_EINVAL;
    if (n_rows == 0) return MATTEN_OK;
    if (!x || !dy || !chan || !dx) return MATTEN_EINVAL;
    const int T = 256;
    norm_act_bwd_kernel<<<(unsigned)matten_cdiv(n_rows * n_chan, T), T, 0, stream>>>(
        x, dy, (int)dim, n_rows, (const int4*)chan, (int)n_chan, act, epsilon * epsilon, dx);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}

// floats of scratch the whole-batch BatchNorm calls need (0: none -- small batches run the one-launch reductions)
extern "C" int64_t matten_bn_scratch_floats(int64_t n_rows, int64_t dim) {
    return n_rows >= BN_COLS_MIN_ROWS ? 2 * matten_cdiv(n_rows, BN_RB) * dim : 0;
}

extern "C" int matten_bn_train_fwd(const float* x, int64_t dim, int64_t n_rows, const int32_t* col2chan,
                                   const int32_t* chan, int64_t n_chan, const float* weight, const float* bias,
                                   float eps, float* mean, float* nu, float* y, float* running_mean,
                                   float* running_var, float momentum, float* scratch, matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_rows <= 0 || dim <= 0 || n_chan <= 0) return MATTEN_EINVAL;
    if (!x || !col2chan || !chan || !weight || !bias || !mean || !nu || !y) return MATTEN_EINVAL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return MATTEN_EINVAL;
    if (matten_bn_scratch_floats(n_rows, dim) > 0) {
        if (!scratch) return MATTEN_EINVAL;
        const int64_t n_blocks = matten_cdiv(n_rows, BN_RB);
        if (n_blocks >= ((int64_t)1 << 31)) return MATTEN_EINVAL;
        bn_stats_cols_kernel<<<(unsigned)n_blocks, 256, 0, stream>>>(x, (int)dim, n_rows, col2chan, (const int4*)chan,
                                                                     (float2*)scratch);
        MATTEN_LAUNCH_CHECK();
        bn_stats_finish_kernel<<<(unsigned)n_chan, 256, 0, stream>>>((const float2*)scratch, (int)dim, n_rows,
                                                                     (int)n_blocks, (const int4*)chan, mean, nu,
                                                                     running_mean, running_var, momentum);
    } else {
        bn_stats_kernel<<<(unsigned)n_chan, 256, 0, stream>>>(x, (int)dim, n_rows, (const int4*)chan, mean, nu, nullptr,
                                                              running_mean, running_var, momentum);
    }
    MATTEN_LAUNCH_CHECK();
    const int T = 256;
    bn_apply_kernel<<<(unsigned)matten_cdiv(n_rows * dim, T), T, 0, stream>>>(x, (int)dim, n_rows, col2chan,
                                                                              (const int4*)chan, mean, nu, weight, bias,
                                                                              eps, y, nullptr, (int)n_chan);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}

// Instance (graph) normalisation, reference nn/utils.py:448-588: the statistics of matten_bn_train_fwd per crystal.
// seg_ptr[n_seg + 1]: the crystals' row ranges (rows grouped per crystal), seg_of_row[n_rows]: crystal of every row;
// mean / nu [n_seg, n_chan] are outputs (kept for the adjoint).
extern "C" int matten_instance_norm_fwd(const float* x, int64_t dim, int64_t n_rows, const int64_t* seg_ptr,
                                        const int64_t* seg_of_row, int64_t n_seg, const int32_t* col2chan,
                                        const int32_t* chan, int64_t n_chan, const float* weight, const float* bias,
                                        float eps, float* mean, float* nu, float* y, matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_rows < 0 || dim <= 0 || n_chan <= 0 || n_seg <= 0 || n_seg > 65535) return MATTEN_EINVAL;
    if (!seg_ptr || !col2chan || !chan || !weight || !bias || !mean || !nu) return MATTEN_EINVAL;
    if (n_rows > 0 && (!x || !y || !seg_of_row)) return MATTEN_EINVAL;
    bn_stats_kernel<<<dim3((unsigned)n_chan, (unsigned)n_seg), 256, 0, stream>>>(x, (int)dim, n_rows, (const int4*)chan, mean,
                                                                                  nu, seg_ptr, nullptr, nullptr, 0.0f);
    MATTEN_LAUNCH_CHECK();
    if (n_rows == 0) return MATTEN_OK;
    const int T = 256;
    bn_apply_kernel<<<(unsigned)matten_cdiv(n_rows * dim, T), T, 0, stream>>>(x, (int)dim, n_rows, col2chan,
                                                                              (const int4*)chan, mean, nu, weight, bias,
                                                                              eps, y, seg_of_row, (int)n_chan);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}

// adjoint: A, B [n_seg, n_chan] (sum dy (x - mean), sum dy per crystal and channel), dx
extern "C" int matten_instance_norm_bwd(const float* x, const float* dy, int64_t dim, int64_t n_rows, const int64_t* seg_ptr,
                                        const int64_t* seg_of_row, int64_t n_seg, const int32_t* col2chan,
                                        const int32_t* chan, int64_t n_chan, const float* mean, const float* nu,
                                        const float* weight, float eps, float* A, float* B, float* dx,
                                        matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_rows <= 0 || dim <= 0 || n_chan <= 0 || n_seg <= 0 || n_seg > 65535) return MATTEN_EINVAL;
    if (!x || !dy || !seg_ptr || !seg_of_row || !col2chan || !chan || !mean || !nu || !weight || !A || !B || !dx)
        return MATTEN_EINVAL;
    bn_bwd_reduce_kernel<<<dim3((unsigned)n_chan, (unsigned)n_seg), 256, 0, stream>>>(x, dy, (int)dim, n_rows,
                                                                                       (const int4*)chan, mean, A, B, seg_ptr,
                                                                                       nullptr, 0.0f, nullptr, nullptr);
    MATTEN_LAUNCH_CHECK();
    const int T = 256;
    bn_bwd_apply_kernel<<<(unsigned)matten_cdiv(n_rows * dim, T), T, 0, stream>>>(
        x, dy, (int)dim, n_rows, col2chan, (const int4*)chan, mean, nu, weight, A, B, eps, dx, seg_of_row, seg_ptr,
        (int)n_chan);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}

extern "C" int matten_bn_train_bwd(const float* x, const float* dy, int64_t dim, int64_t n_rows,
                                   const int32_t* col2chan, const int32_t* chan, int64_t n_chan, const float* mean,
                                   const float* nu, const float* weight, float eps, float* A, float* B, float* dx,
                                   float* dweight, float* dbias, float* scratch, matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_rows <= 0 || dim <= 0 || n_chan <= 0) return MATTEN_EINVAL;
    if (!x || !dy || !col2chan || !chan || !mean || !nu || !weight || !A || !B || !dx) return MATTEN_EINVAL;
    if ((dweight == nullptr) != (dbias == nullptr)) return MATTEN_EINVAL;
    if (matten_bn_scratch_floats(n_rows, dim) > 0) {
        if (!scratch) return MATTEN_EINVAL;
        const int64_t n_blocks = matten_cdiv(n_rows, BN_RB);
        if (n_blocks >= ((int64_t)1 << 31)) return MATTEN_EINVAL;
        bn_bwd_cols_kernel<<<(unsigned)n_blocks, 256, 0, stream>>>(x, dy, (int)dim, n_rows, col2chan, mean,
                                                                   (float2*)scratch);
        MATTEN_LAUNCH_CHECK();
        bn_bwd_finish_kernel<<<(unsigned)n_chan, 256, 0, stream>>>((const float2*)scratch, (int)dim, (int)n_blocks,
                                                                   (const int4*)chan, A, B, nu, eps, dweight, dbias);
    } else {
        bn_bwd_reduce_kernel<<<(unsigned)n_chan, 256, 0, stream>>>(x, dy, (int)dim, n_rows, (const int4*)chan, mean, A, B,
                                                                   nullptr, nu, eps, dweight, dbias);
    }
    MATTEN_LAUNCH_CHECK();
    const int T = 256;
    bn_bwd_apply_kernel<<<(unsigned)matten_cdiv(n_rows * dim, T), T, 0, stream>>>(
        x, dy, (int)dim, n_rows, col2chan, (const int4*)chan, mean, nu, weight, A, B, eps, dx, nullptr, nullptr, (int)n_chan);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}

extern "C" int matten_segment_reduce_bwd(const float* dy, int64_t dim, const int64_t* ptr, int64_t n_segments, int mean,
                                         float* dx, matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_segments < 0 || dim <= 0) return MATTEN_EINVAL;
    if (n_segments == 0) return MATTEN_OK;
    if (!dy || !ptr || !dx) return MATTEN_EINVAL;
    const int T = 256;
    segment_reduce_bwd_kernel<<<(unsigned)matten_cdiv(n_segments * dim, T), T, 0, stream>>>(dy, (int)dim, ptr,
                                                                                            n_segments, mean, dx);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}

// ------------------------------------------------------------------------------------------------
// 'uvu' TP adjoint with the literal-coefficient coupling code (cg_gen.h CG<l1,l2,l3>::adjoint), the form the forward
// kernels use.  A thread owns (edge, channel u of ONE input block) and walks that block's paths (the list is uniform
// over the block): per path  t_i = sum_jk C_ijk Y_j G_k  (straight-line code, coefficients as literals),
//     dw[e, w_off + u] = norm * sum_i x_i t_i,      dx_i += w * t_i   (registers; ONE atomic per component at the end).
// The table-driven kernels above spend their time fetching (i, j, k, c) records and scattering into dx through compare /
// select chains: 2.14 ms per layer at 294 k edges against 0.28 ms for the forward of the same layer.
// blocks[n_blocks,4] = {x_off, mul, l1, first path | n_paths << 16}; paths[n_paths,4] = {l1*25 + l2*5 + l3, w_off, out_off, 0}
// ------------------------------------------------------------------------------------------------
namespace {

struct LitArgs {
    const float* x;
    const void* w_edge;
    const float* sh;
    const int32_t* src;
    const int32_t* dst;
    const float* g_agg;
    const float* num_neigh;
    float* dx;          // [N, d_in] accumulated with atomics, or (dx_per_edge) [E, d_in]: row e = edge e's contribution
    void* dw;
    int64_t E;
    int d_in, w_ld, sh_stride, d_mid, dw_ld, edge_bf16;
    float avg_nn;
    int dx_per_edge;
};

template <int L1, int L2, int L3>
__device__ __forceinline__ void lit_path(const LitArgs& a, const int4 pth, float wv, int64_t e, int u,
                                         const float* __restrict__ x, const float* __restrict__ grow,
                                         const float* __restrict__ yrow, float norm, float* __restrict__ dxi) {
    constexpr int D1 = 2 * L1 + 1, D2 = 2 * L2 + 1, D3 = 2 * L3 + 1;
    float g[D3], y[D2], t[D1];
    const float* gp = grow + pth.z + u * D3;
#pragma unroll
    for (int k = 0; k < D3; ++k) g[k] = gp[k];
#pragma unroll
    for (int j = 0; j < D2; ++j) y[j] = yrow[L2 * L2 + j];
#pragma unroll
    for (int i = 0; i < D1; ++i) t[i] = 0.0f;
    matten::CG<L1, L2, L3>::adjoint(y, g, t);
    float dwv = 0.0f;
#pragma unroll
    for (int i = 0; i < D1; ++i) {
        dwv = fmaf(x[i], t[i], dwv);
        dxi[i] = fmaf(wv, t[i], dxi[i]);
    }
    matten_st_edge(a.dw, e * a.dw_ld + pth.y + u, dwv * norm, a.edge_bf16);
}

#define MATTEN_LIT_CASE(L1, L2, L3) \
    case (L1 * 25 + L2 * 5 + L3): lit_path<L1, L2, L3>(a, pth, wv, e, u, x, grow, yrow, norm, dxi); break;

// the case lists per input-block degree (paths[].x = l1 * 25 + l2 * 5 + l3)
#define MATTEN_LIT_LIST_0 MATTEN_LIT_CASE(0, 0, 0) MATTEN_LIT_CASE(0, 1, 1) MATTEN_LIT_CASE(0, 2, 2) MATTEN_LIT_CASE(0, 3, 3) MATTEN_LIT_CASE(0, 4, 4)
#define MATTEN_LIT_LIST_1 MATTEN_LIT_CASE(1, 0, 1) MATTEN_LIT_CASE(1, 1, 0) MATTEN_LIT_CASE(1, 1, 1) MATTEN_LIT_CASE(1, 1, 2) MATTEN_LIT_CASE(1, 2, 1) MATTEN_LIT_CASE(1, 2, 2) MATTEN_LIT_CASE(1, 2, 3) MATTEN_LIT_CASE(1, 3, 2) MATTEN_LIT_CASE(1, 3, 3) MATTEN_LIT_CASE(1, 3, 4) MATTEN_LIT_CASE(1, 4, 3) MATTEN_LIT_CASE(1, 4, 4)
#define MATTEN_LIT_LIST_2 MATTEN_LIT_CASE(2, 0, 2) MATTEN_LIT_CASE(2, 1, 1) MATTEN_LIT_CASE(2, 1, 2) MATTEN_LIT_CASE(2, 1, 3) MATTEN_LIT_CASE(2, 2, 0) MATTEN_LIT_CASE(2, 2, 1) MATTEN_LIT_CASE(2, 2, 2) MATTEN_LIT_CASE(2, 2, 3) MATTEN_LIT_CASE(2, 2, 4) MATTEN_LIT_CASE(2, 3, 1) MATTEN_LIT_CASE(2, 3, 2) MATTEN_LIT_CASE(2, 3, 3) MATTEN_LIT_CASE(2, 3, 4) MATTEN_LIT_CASE(2, 4, 2) MATTEN_LIT_CASE(2, 4, 3) MATTEN_LIT_CASE(2, 4, 4)
#define MATTEN_LIT_LIST_3 MATTEN_LIT_CASE(3, 0, 3) MATTEN_LIT_CASE(3, 1, 2) MATTEN_LIT_CASE(3, 1, 3) MATTEN_LIT_CASE(3, 1, 4) MATTEN_LIT_CASE(3, 2, 1) MATTEN_LIT_CASE(3, 2, 2) MATTEN_LIT_CASE(3, 2, 3) MATTEN_LIT_CASE(3, 2, 4) MATTEN_LIT_CASE(3, 3, 0) MATTEN_LIT_CASE(3, 3, 1) MATTEN_LIT_CASE(3, 3, 2) MATTEN_LIT_CASE(3, 3, 3) MATTEN_LIT_CASE(3, 3, 4) MATTEN_LIT_CASE(3, 4, 1) MATTEN_LIT_CASE(3, 4, 2) MATTEN_LIT_CASE(3, 4, 3) MATTEN_LIT_CASE(3, 4, 4)
#define MATTEN_LIT_LIST_4 MATTEN_LIT_CASE(4, 0, 4) MATTEN_LIT_CASE(4, 1, 3) MATTEN_LIT_CASE(4, 1, 4) MATTEN_LIT_CASE(4, 2, 2) MATTEN_LIT_CASE(4, 2, 3) MATTEN_LIT_CASE(4, 2, 4) MATTEN_LIT_CASE(4, 3, 1) MATTEN_LIT_CASE(4, 3, 2) MATTEN_LIT_CASE(4, 3, 3) MATTEN_LIT_CASE(4, 3, 4) MATTEN_LIT_CASE(4, 4, 0) MATTEN_LIT_CASE(4, 4, 1) MATTEN_LIT_CASE(4, 4, 2) MATTEN_LIT_CASE(4, 4, 3) MATTEN_LIT_CASE(4, 4, 4)

#define MATTEN_LIT_LIST2_0 MATTEN_LIT_CASE(0, 0, 0) MATTEN_LIT_CASE(0, 1, 1) MATTEN_LIT_CASE(0, 2, 2)
#define MATTEN_LIT_LIST2_1 MATTEN_LIT_CASE(1, 0, 1) MATTEN_LIT_CASE(1, 1, 0) MATTEN_LIT_CASE(1, 1, 1) MATTEN_LIT_CASE(1, 1, 2) MATTEN_LIT_CASE(1, 2, 1) MATTEN_LIT_CASE(1, 2, 2)
#define MATTEN_LIT_LIST2_2 MATTEN_LIT_CASE(2, 0, 2) MATTEN_LIT_CASE(2, 1, 1) MATTEN_LIT_CASE(2, 1, 2) MATTEN_LIT_CASE(2, 2, 0) MATTEN_LIT_CASE(2, 2, 1) MATTEN_LIT_CASE(2, 2, 2)

// one path of an input block for one (edge, channel): dw written, the edge's dx contribution added into dxi
// LMAX = 2: the instantiation for tensor products whose degrees all stay <= 2 (the kernel's registers are those of its widest
// case: 60 instead of 108, twice the resident waves)
template <int L1, int LMAX>
__device__ __forceinline__ void lit_one_path(const LitArgs& a, const int4 pth, float wv, int64_t e, int u,
                                             const float* __restrict__ x, const float* __restrict__ grow,
                                             const float* __restrict__ yrow, float norm, float* __restrict__ dxi) {
    // the code is uniform over the block: every thread of the launch row walks the same list
    if constexpr (LMAX <= 2) {
        if constexpr (L1 == 0) { switch (pth.x) { MATTEN_LIT_LIST2_0 default: break; } }
        if constexpr (L1 == 1) { switch (pth.x) { MATTEN_LIT_LIST2_1 default: break; } }
        if constexpr (L1 == 2) { switch (pth.x) { MATTEN_LIT_LIST2_2 default: break; } }
    } else {
        if constexpr (L1 == 0) { switch (pth.x) { MATTEN_LIT_LIST_0 default: break; } }
        if constexpr (L1 == 1) { switch (pth.x) { MATTEN_LIT_LIST_1 default: break; } }
        if constexpr (L1 == 2) { switch (pth.x) { MATTEN_LIT_LIST_2 default: break; } }
        if constexpr (L1 == 3) { switch (pth.x) { MATTEN_LIT_LIST_3 default: break; } }
        if constexpr (L1 == 4) { switch (pth.x) { MATTEN_LIT_LIST_4 default: break; } }
    }
}

template <int L1, int LMAX>
__device__ __forceinline__ void lit_block(const LitArgs& a, const int4 blk, const int4* __restrict__ paths, int64_t e, int u) {
    constexpr int D1 = 2 * L1 + 1;
    const int src = a.src[e], dst = a.dst[e];
    const float norm = 1.0f / sqrtf(a.avg_nn > 0.0f ? a.avg_nn : a.num_neigh[dst]);
    const float* xp = a.x + (int64_t)src * a.d_in + blk.x + u * D1;
    float x[D1], dxi[D1];
#pragma unroll
    for (int i = 0; i < D1; ++i) {
        x[i] = xp[i];
        dxi[i] = 0.0f;
    }
    const float* grow = a.g_agg + (int64_t)dst * a.d_mid;
    const float* yrow = a.sh + e * a.sh_stride;
    const int p0 = blk.w & 0xffff, np = blk.w >> 16;
    for (int p = p0; p < p0 + np; ++p) {
        const int4 pth = paths[p];
        const float wv = matten_ld_edge(a.w_edge, e * a.w_ld + pth.y + u, a.edge_bf16);
        lit_one_path<L1, LMAX>(a, pth, wv, e, u, x, grow, yrow, norm, dxi);
    }
    float* dxp = a.dx + (a.dx_per_edge ? e : (int64_t)src) * a.d_in + blk.x + u * D1;
#pragma unroll
    for (int i = 0; i < D1; ++i) {
        if (a.dx_per_edge) dxp[i] = norm * dxi[i];   // summed per source node in a fixed order by rows_segment_sum_kernel
        else if (dxi[i] != 0.0f) atomicAdd(dxp + i, norm * dxi[i]);
    }
}

// ---- the same adjoint WITHOUT w[E, W] in memory (training on the fused forward) --------------------------------------------
// The workgroup (256 / cu edges x cu channel lanes of one input block) first evaluates the weights it is about to use,
//      w[e, w_off_p + u] = sum_k h2[e, k] W2p[k, w_off_p + u],
// on the matrix cores exactly as the forward does (split fp16: hi hi + 2^-11 (lo hi + hi lo), tp_fused.hip header): per path
// ceil(mul / 16) column tiles of ready-made A fragments (matten_split_a_tiles over pseudo-entries, one per path, REFERENCE
// column order), the edges of the workgroup as the N dimension in tiles of 16, B = the hidden-feature rows h2s[E, 2, 32] the
// forward used.  The D fragments (4 columns of one edge per lane) go to an LDS tile [edge][path][column]; after a barrier
// the (edge, channel) threads run the literal coupling code with w read from there.  Paths are taken in rounds so that
// the tile stays within the launch's LDS (lds_floats); dx accumulates in registers across rounds.  What it saves: matten_radial_mlp in the
// backward (0.36 ms at 293 k edges) and the 4 W bytes per edge of w written and read back; w never exists.
typedef _Float16 wf_f16x8 __attribute__((ext_vector_type(8)));
typedef float wf_f32x4 __attribute__((ext_vector_type(4)));
#ifndef WF_BATCH_N
#define WF_BATCH_N 2
#endif
constexpr int WF_BATCH = WF_BATCH_N;   // matrix jobs of a wave whose operand loads are in flight together
struct WFreeArgs {
    const _Float16* h2s;      // [E, 2, 32]
    const _Float16* frag;     // [n_tiles, 64, 16]: hi(kk 0..7) | lo(kk 0..7) per lane
    const float* w_inv;       // [n_paths] 1 / (the path's fragment scale x the hidden-feature scale)
    int lds_floats;           // capacity of the workgroup's tile (plan.bw_wfree_lds_floats: what the widest block needs, capped)
};

template <int L1, int LMAX>
__device__ __forceinline__ void lit_block_wfree(const LitArgs& a, const WFreeArgs& wf, const int4 blk, const int4* __restrict__ paths,
                                                int cu, int64_t e0, float* __restrict__ wt) {
    constexpr int D1 = 2 * L1 + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int epw = 256 / cu;                       // edges of this workgroup
    const int n_et = (epw + 15) >> 4;               // 16-edge MFMA tiles
    const int n_mt = (blk.y + 15) >> 4;             // 16-column tiles per path
    const int tw = (blk.y + 3) & ~3;                // floats of the LDS tile per (edge, path)
    const int p0 = blk.w & 0xffff, np = blk.w >> 16;
    const int per_round = max(1, (wf.lds_floats / epw - 4) / tw);
    const int row = min(np, per_round) * tw + 4;    // floats per edge row of the tile (+4: rows start on different banks)
    // phase-2 role; its edge-invariant loads go out in front of phase 1
    const int el = tid / cu, u = tid & (cu - 1);
    const int64_t e = e0 + el;
    const bool active = e < a.E && u < blk.y;
    int src = 0, dst = 0;
    float norm = 0.0f;
    float x[D1], dxi[D1];
#pragma unroll
    for (int i = 0; i < D1; ++i) x[i] = 0.0f, dxi[i] = 0.0f;
    if (active) {
        src = a.src[e], dst = a.dst[e];
        norm = 1.0f / sqrtf(a.avg_nn > 0.0f ? a.avg_nn : a.num_neigh[dst]);
        const float* xp = a.x + (int64_t)src * a.d_in + blk.x + u * D1;
#pragma unroll
        for (int i = 0; i < D1; ++i) x[i] = xp[i];
    }
    const float* grow = a.g_agg + (int64_t)dst * a.d_mid;
    const float* yrow = a.sh + (active ? e : 0) * a.sh_stride;
    const wf_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int r0 = 0; r0 < np; r0 += per_round) {
        const int nr = min(per_round, np - r0);
        // ---- phase 1: jobs (path, column tile, edge tile) dealt to the four waves in batches of WF_BATCH: all of a batch's
        // operand loads (two 16-byte A pieces per job, the edge tile's two B pieces) go out before the first matrix instruction
        const int n_jobs = nr * n_mt * n_et;
        for (int j0 = wave * WF_BATCH; j0 < n_jobs; j0 += 4 * WF_BATCH) {
            wf_f16x8 ah[WF_BATCH], al[WF_BATCH], bh[WF_BATCH], bl[WF_BATCH];
            float inv[WF_BATCH];
            int dst_off[WF_BATCH];
#pragma unroll
            for (int q = 0; q < WF_BATCH; ++q) {
                const int j = min(j0 + q, n_jobs - 1);
                const int jj = j / n_et, et = j - jj * n_et;
                const int pl = jj / n_mt, mt = jj - pl * n_mt;
                const int4 pth = paths[p0 + r0 + pl];
                const int64_t er = min(e0 + 16 * et + c, a.E - 1);
                const _Float16* fr = wf.frag + ((int64_t)(pth.w + mt) * 64 + lane) * 16;
                ah[q] = *reinterpret_cast<const wf_f16x8*>(fr);
                al[q] = *reinterpret_cast<const wf_f16x8*>(fr + 8);
                bh[q] = *reinterpret_cast<const wf_f16x8*>(wf.h2s + er * 64 + g * 8);
                bl[q] = *reinterpret_cast<const wf_f16x8*>(wf.h2s + er * 64 + 32 + g * 8);
                inv[q] = wf.w_inv[p0 + r0 + pl];
                const int col = 16 * mt + 4 * g;
                dst_off[q] = (j0 + q < n_jobs && 16 * et + c < epw && col < tw) ? (16 * et + c) * row + pl * tw + col : -1;
            }
#pragma unroll
            for (int q = 0; q < WF_BATCH; ++q) {
                wf_f32x4 dl = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[q], bh[q], zero, 0, 0, 0);
                wf_f32x4 dh = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[q], bh[q], zero, 0, 0, 0);
                dl = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[q], bl[q], dl, 0, 0, 0);
                const wf_f32x4 w4 = (dh + (1.0f / 2048.0f) * dl) * inv[q];
                if (dst_off[q] >= 0) *reinterpret_cast<wf_f32x4*>(wt + dst_off[q]) = w4;
            }
        }
        __syncthreads();
        // ---- phase 2: the literal adjoint with w from the tile
        if (active) {
            for (int pl = 0; pl < nr; ++pl) {
                const int4 pth = paths[p0 + r0 + pl];
                lit_one_path<L1, LMAX>(a, pth, wt[el * row + pl * tw + u], e, u, x, grow, yrow, norm, dxi);
            }
        }
        if (r0 + per_round < np) __syncthreads();   // the next round overwrites the tile
    }
    if (active) {
        float* dxp = a.dx + (a.dx_per_edge ? e : (int64_t)src) * a.d_in + blk.x + u * D1;
#pragma unroll
        for (int i = 0; i < D1; ++i) {
            if (a.dx_per_edge) dxp[i] = norm * dxi[i];
            else if (dxi[i] != 0.0f) atomicAdd(dxp + i, norm * dxi[i]);
        }
    }
}

}  // namespace

namespace {
// out[n, c] = sum over k in [ptr[n], ptr[n+1]) of rows[perm[k], c], in k order (fixed: no atomics).  A thread owns one
// column of one segment; two rows are in flight per step (the row index of the step after next is fetched early).
__global__ void rows_segment_sum_kernel(const float* __restrict__ rows, int d, const int32_t* __restrict__ ptr,
                                        const int32_t* __restrict__ perm, int64_t n_seg, float* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_seg * d) return;
    const int64_t n = idx / d;
    const int c = (int)(idx - n * d);
    const int beg = ptr[n], end = ptr[n + 1];
    float acc = 0.0f;
    int k = beg;
    for (; k + 4 <= end; k += 4) {   // four independent loads per step, added in k order
        const float v0 = rows[(int64_t)perm[k] * d + c], v1 = rows[(int64_t)perm[k + 1] * d + c];
        const float v2 = rows[(int64_t)perm[k + 2] * d + c], v3 = rows[(int64_t)perm[k + 3] * d + c];
        acc = (((acc + v0) + v1) + v2) + v3;
    }
    for (; k < end; ++k) acc += rows[(int64_t)perm[k] * d + c];
    out[idx] = acc;
}

// The grid is the input blocks' own workgroup ranges (cdiv(E * lanes per edge, 256) each) laid end to end: a 2-D grid sized
// for the widest block left half of its workgroups without an edge (blocks of 4 channels beside blocks of 32: 100 k
// empty workgroups per launch at 293 k edges).
template <int LMAX>
__global__ __launch_bounds__(256) void tp_backward_lit_kernel(LitArgs a, const int4* __restrict__ blocks, int n_blocks,
                                                              const int4* __restrict__ paths) {
    int b = 0, cu = 1;
    int64_t first = 0;   // first workgroup of block b
    for (;; ++b) {
        cu = 1;
        while (cu < blocks[b].y) cu <<= 1;             // lanes per edge: the block's channels rounded up to a power of two
        const int64_t n = (a.E * cu + 255) / 256;
        if (b + 1 >= n_blocks || (int64_t)blockIdx.x < first + n) break;
        first += n;
    }
    const int4 blk = blocks[b];
    const int64_t idx = ((int64_t)blockIdx.x - first) * blockDim.x + threadIdx.x;
    const int64_t e = idx / cu;
    const int u = (int)(idx - e * cu);
    if (e >= a.E || u >= blk.y) return;
    switch (blk.z) {
        case 0: lit_block<0, LMAX>(a, blk, paths, e, u); break;
        case 1: lit_block<1, LMAX>(a, blk, paths, e, u); break;
        case 2: lit_block<2, LMAX>(a, blk, paths, e, u); break;
        case 3: if constexpr (LMAX > 2) lit_block<3, LMAX>(a, blk, paths, e, u); break;
        case 4: if constexpr (LMAX > 2) lit_block<4, LMAX>(a, blk, paths, e, u); break;
        default: break;
    }
}

// the same grid as tp_backward_lit_kernel; every thread of a workgroup stays to the end (barriers)
#ifndef WF_MIN_WGS
#define WF_MIN_WGS 6   // l <= 2 instantiation: six workgroups per CU (<= 80 registers; 7 and 8 spill: 1.82 / 2.68 vs 1.53 ms)
#endif
template <int LMAX>
__global__ __launch_bounds__(256, (LMAX <= 2 ? WF_MIN_WGS : 1)) void tp_backward_lit_wfree_kernel(LitArgs a, WFreeArgs wf, const int4* __restrict__ blocks,
                                                                    int n_blocks, const int4* __restrict__ paths) {
    extern __shared__ __attribute__((aligned(16))) float wt[];
    int b = 0, cu = 1;
    int64_t first = 0;
    for (;; ++b) {
        cu = 1;
        while (cu < blocks[b].y) cu <<= 1;
        const int64_t n = (a.E * cu + 255) / 256;
        if (b + 1 >= n_blocks || (int64_t)blockIdx.x < first + n) break;
        first += n;
    }
    const int4 blk = blocks[b];
    if (cu > 256) return;    // (the entry point refuses max_mul > 256; a table that lies about it computes nothing rather than dividing by zero)
    const int64_t e0 = ((int64_t)blockIdx.x - first) * (256 / cu);
    if (e0 >= a.E) return;   // (workgroup-uniform: the excess workgroups behind the last block)
    switch (blk.z) {
        case 0: lit_block_wfree<0, LMAX>(a, wf, blk, paths, cu, e0, wt); break;
        case 1: lit_block_wfree<1, LMAX>(a, wf, blk, paths, cu, e0, wt); break;
        case 2: lit_block_wfree<2, LMAX>(a, wf, blk, paths, cu, e0, wt); break;
        case 3: if constexpr (LMAX > 2) lit_block_wfree<3, LMAX>(a, wf, blk, paths, cu, e0, wt); break;
        case 4: if constexpr (LMAX > 2) lit_block_wfree<4, LMAX>(a, wf, blk, paths, cu, e0, wt); break;
        default: break;
    }
}
}  // namespace

extern "C" int matten_tp_backward_lit(const float* x, int64_t d_in, const void* w_edge, int64_t w_ld, const float* sh_sorted,
                                      int64_t sh_stride, const int32_t* src_sorted, const int32_t* dst_sorted,
                                      const int32_t* blocks, int64_t n_blocks, int64_t sum_lanes, const int32_t* paths,
                                      int64_t n_paths, const float* g_agg, int64_t d_mid, float avg_num_neighbors,
                                      const float* num_neigh, int64_t n_edges, float* dx, void* dw, int64_t dw_ld,
                                      int edge_is_bf16, int64_t n_nodes, const int32_t* out_ptr, const int32_t* out_perm,
                                      float* dx_edges, int max_l, matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_edges < 0 || d_in <= 0 || n_blocks <= 0 || n_blocks > 65535 || n_paths <= 0 || d_mid <= 0 || sum_lanes <= 0 ||
        sum_lanes > 4096 * n_blocks || w_ld <= 0 || dw_ld <= 0)
        return MATTEN_EINVAL;
    if (dx_edges && (!out_ptr || !out_perm || n_nodes < 0)) return MATTEN_EINVAL;
    if (n_edges == 0) {
        if (dx_edges && n_nodes > 0) {
            if (!dx) return MATTEN_EINVAL;
            if (hipMemsetAsync(dx, 0, sizeof(float) * (size_t)n_nodes * (size_t)d_in, stream) != hipSuccess) return MATTEN_ELAUNCH;
        }
        return MATTEN_OK;
    }
    if (!x || !w_edge || !sh_sorted || !src_sorted || !dst_sorted || !blocks || !paths || !g_agg || !dx || !dw)
        return MATTEN_EINVAL;
    if (!(avg_num_neighbors > 0.0f) && !num_neigh) return MATTEN_EINVAL;
    // sum_lanes = sum over the blocks of their lanes per edge (channels rounded up to a power of two): the blocks'
    // workgroup ranges laid end to end need at most this many workgroups (the excess ones find no edge and leave)
    const int64_t gx = matten_cdiv(n_edges * sum_lanes, 256) + n_blocks;
    if (gx >= ((int64_t)1 << 31)) return MATTEN_EINVAL;
    LitArgs a{x, w_edge, sh_sorted, src_sorted, dst_sorted, g_agg, num_neigh, dx_edges ? dx_edges : dx, dw, n_edges, (int)d_in,
              (int)w_ld, (int)sh_stride, (int)d_mid, (int)dw_ld, edge_is_bf16, avg_num_neighbors, dx_edges ? 1 : 0};
    if (max_l <= 2) tp_backward_lit_kernel<2><<<(unsigned)gx, 256, 0, stream>>>(a, (const int4*)blocks, (int)n_blocks, (const int4*)paths);
    else tp_backward_lit_kernel<4><<<(unsigned)gx, 256, 0, stream>>>(a, (const int4*)blocks, (int)n_blocks, (const int4*)paths);
    MATTEN_LAUNCH_CHECK();
    if (dx_edges && n_nodes > 0) {
        if (matten_cdiv(n_nodes * d_in, 256) >= ((int64_t)1 << 31)) return MATTEN_EINVAL;
        rows_segment_sum_kernel<<<(unsigned)matten_cdiv(n_nodes * d_in, 256), 256, 0, stream>>>(dx_edges, (int)d_in, out_ptr,
                                                                                              out_perm, n_nodes, dx);
        MATTEN_LAUNCH_CHECK();
    }
    return MATTEN_OK;
}

// matten_tp_backward_lit without w[E, W]: the weights are re-evaluated per workgroup on the matrix cores from the hidden
// features (see lit_block_wfree).  paths[p].w = first A-fragment tile of path p; frag / w_inv from matten_split_a_tiles over one
// pseudo-entry per path (words 5, 6, 7 = w_off, first tile, ceil(mul / 16)) on the last radial layer in REFERENCE column order.
extern "C" int matten_tp_backward_lit_wfree(const float* x, int64_t d_in, const uint16_t* h2s, const uint16_t* frag,
                                            const float* w_inv, const float* sh_sorted, int64_t sh_stride,
                                            const int32_t* src_sorted, const int32_t* dst_sorted, const int32_t* blocks,
                                            int64_t n_blocks, int64_t sum_lanes, const int32_t* paths, int64_t n_paths,
                                            const float* g_agg, int64_t d_mid, float avg_num_neighbors, const float* num_neigh,
                                            int64_t n_edges, float* dx, void* dw, int64_t dw_ld, int edge_is_bf16,
                                            int64_t n_nodes, const int32_t* out_ptr, const int32_t* out_perm, float* dx_edges,
                                            int64_t lds_floats, int max_l, int max_mul, matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    // max_mul: a workgroup is 256 threads = (256 / lanes per edge) edges, so a block wider than 256 channels has no edge per
    // workgroup (the materialised-w adjoint takes such a layer).  lds_floats >= 2048: the narrowest blocks put 256 edges in
    // a workgroup with rows of round4(mul) + 4 floats each (1 channel: 256 x 8), and every block needs at least one path per round.
    if (n_edges < 0 || d_in <= 0 || n_blocks <= 0 || n_blocks > 65535 || n_paths <= 0 || d_mid <= 0 || sum_lanes <= 0 ||
        sum_lanes > 256 * n_blocks || dw_ld <= 0 || lds_floats < 2048 || lds_floats > 15 * 1024 || max_mul < 1 || max_mul > 256)
        return MATTEN_EINVAL;
    if (dx_edges && (!out_ptr || !out_perm || n_nodes < 0)) return MATTEN_EINVAL;
    if (n_edges == 0) {
        if (dx_edges && n_nodes > 0) {
            if (!dx) return MATTEN_EINVAL;
            if (hipMemsetAsync(dx, 0, sizeof(float) * (size_t)n_nodes * (size_t)d_in, stream) != hipSuccess) return MATTEN_ELAUNCH;
        }
        return MATTEN_OK;
    }
    if (!x || !h2s || !frag || !w_inv || !sh_sorted || !src_sorted || !dst_sorted || !blocks || !paths || !g_agg || !dx || !dw)
        return MATTEN_EINVAL;
    if (!(avg_num_neighbors > 0.0f) && !num_neigh) return MATTEN_EINVAL;
    const int64_t gx = matten_cdiv(n_edges * sum_lanes, 256) + n_blocks;
    if (gx >= ((int64_t)1 << 31)) return MATTEN_EINVAL;
    LitArgs a{x, nullptr, sh_sorted, src_sorted, dst_sorted, g_agg, num_neigh, dx_edges ? dx_edges : dx, dw, n_edges, (int)d_in,
              0, (int)sh_stride, (int)d_mid, (int)dw_ld, edge_is_bf16, avg_num_neighbors, dx_edges ? 1 : 0};
    WFreeArgs wf{(const _Float16*)h2s, (const _Float16*)frag, w_inv, (int)lds_floats};
    if (max_l <= 2)
        tp_backward_lit_wfree_kernel<2><<<(unsigned)gx, 256, sizeof(float) * (size_t)lds_floats, stream>>>(
            a, wf, (const int4*)blocks, (int)n_blocks, (const int4*)paths);
    else
        tp_backward_lit_wfree_kernel<4><<<(unsigned)gx, 256, sizeof(float) * (size_t)lds_floats, stream>>>(
            a, wf, (const int4*)blocks, (int)n_blocks, (const int4*)paths);
    MATTEN_LAUNCH_CHECK();
    if (dx_edges && n_nodes > 0) {
        if (matten_cdiv(n_nodes * d_in, 256) >= ((int64_t)1 << 31)) return MATTEN_EINVAL;
        rows_segment_sum_kernel<<<(unsigned)matten_cdiv(n_nodes * d_in, 256), 256, 0, stream>>>(dx_edges, (int)d_in, out_ptr,
                                                                                              out_perm, n_nodes, dx);
        MATTEN_LAUNCH_CHECK();
    }
    return MATTEN_OK;
}

// ---- packed[s, j] = weight[gather[s, j]] * scale[j]  (and its adjoint through the inverse permutation) in one launch.
// The species-indexed linears keep the reference's flat e3nn parameter; every step re-packs 14 of them and maps 14
// gradients back: as torch index + multiply that was 56 tiny launches per optimisation step.
namespace {
__global__ void gather_scale_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx,
                                    const float* __restrict__ scale, int64_t n, int64_t scale_period, int scale_by_source,
                                    float* __restrict__ out, const int64_t* __restrict__ perm2, float* __restrict__ out2) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t j = idx[i];
    out[i] = src[j] * scale[(scale_by_source ? j : i) % scale_period];
    if (out2) {   // a second copy with the columns of every period permuted: out2[r, q] = out[r, perm2[q]]
        const int64_t r = i / scale_period, q = i - r * scale_period;
        const int64_t p = r * scale_period + perm2[q];
        out2[i] = src[idx[p]] * scale[(scale_by_source ? idx[p] : p) % scale_period];
    }
}
}  // namespace

extern "C" int matten_gather_scale(const float* src, const int64_t* idx, const float* scale, int64_t n, int64_t scale_period,
                                   int scale_by_source, float* out, const int64_t* perm2, float* out2,
                                   matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0 || scale_period <= 0) return MATTEN_EINVAL;
    if (n == 0) return MATTEN_OK;
    if (!src || !idx || !scale || !out) return MATTEN_EINVAL;
    if ((perm2 == nullptr) != (out2 == nullptr) || (out2 && n % scale_period)) return MATTEN_EINVAL;
    gather_scale_kernel<<<(unsigned)matten_cdiv(n, 256), 256, 0, stream>>>(src, idx, scale, n, scale_period, scale_by_source,
                                                                           out, perm2, out2);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}
