// Node-side operators of the conv stack.
//   species_linear : e3nn FullyConnectedTensorProduct(x, one_hot)  reference nn/conv.py:59-61,77-79,84-86
//                    and e3nn o3.Linear (species == NULL)           reference nn/nodewise.py:111-117
//   gate_bn        : e3nn Gate + BatchNorm(eval)                    reference nn/conv.py:209-211
//   segment_reduce : NodewiseReduce                                 reference nn/nodewise.py:142-148
//   dense_rows     : CartesianTensor.to_cartesian                   reference utils.py:123-124
#include "common.h"

namespace {

// Species-indexed per-irrep linear on the fp32 matrix cores.
// For one irrep block (mul_in -> mul_out channels, 2l+1 components) and one component k the op is a
// plain GEMM over rows:  out[row, w] = sum_u x[row, u] W_s[u, w]  with strides (x: d, out: d).  The host
// enumerates "items" = (irrep block, component, 16-column tile of w); a wave takes an item for a tile
// of 16 rows and runs ceil(mul_in/4) v_mfma_f32_16x16x4_f32:
//      A[m = row][k = u]  = x[row, x_off + u*x_step]          (global, L2-resident gathers)
//      B[k = u][n = w]    = Ws[w_off + u*w_step + n]          (LDS: the species' packed table, staged once)
//      D[row][w]         -> out[row, o_off + n*o_step] (+ add)
// Rows are visited in species-sorted order (order/seg) so a workgroup sees ONE weight table.
struct LinItem {  // 8 x int32
    int x_off, x_step, mul_in, w_off, w_step, n_cols, o_off, o_step;
};
constexpr int SL_ROWS_PER_BLOCK = 64;  // 4 row tiles of 16
constexpr int SL_THREADS = 512;
typedef float sl_f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(SL_THREADS) void species_linear_kernel(
    const float* __restrict__ x, int d_in, const int32_t* __restrict__ order, const int32_t* __restrict__ seg,
    int n_species, const float* __restrict__ wp, int w_stride, int w_in_lds, const LinItem* __restrict__ items,
    int n_items, int d_out, const float* __restrict__ add, int n_rows, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float ws[];

    int b = blockIdx.x, s = 0, lo = 0, hi = 0;
    if (seg) {
        bool found = false;
        for (s = 0; s < n_species; ++s) {
            const int beg = seg[s], end = seg[s + 1];
            const int nblk = (end - beg + SL_ROWS_PER_BLOCK - 1) / SL_ROWS_PER_BLOCK;
            if (b < nblk) {
                lo = beg + b * SL_ROWS_PER_BLOCK;
                hi = min(end, lo + SL_ROWS_PER_BLOCK);
                found = true;
                break;
            }
            b -= nblk;
        }
        if (!found) return;
    } else {
        lo = b * SL_ROWS_PER_BLOCK;
        hi = min(n_rows, lo + SL_ROWS_PER_BLOCK);
        if (lo >= hi) return;
    }
    const float* wsp = wp + (int64_t)s * w_stride;
    if (w_in_lds) {
        for (int i = threadIdx.x; i < w_stride; i += blockDim.x) ws[i] = wsp[i];
        __syncthreads();
        wsp = ws;
    }

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int n_row_tiles = (hi - lo + 15) >> 4;
    // work = (row tile, item); waves stride through it
    for (int wk = wave; wk < n_row_tiles * n_items; wk += SL_THREADS / 64) {
        const int rt = wk / n_items;
        const LinItem it = items[wk - rt * n_items];
        // A role: row m = c of the tile
        const int ra = lo + rt * 16 + c;
        const bool ra_ok = ra < hi;
        const int na = ra_ok ? (order ? order[ra] : ra) : 0;
        const float* xa = x + (int64_t)na * d_in + it.x_off;
        const float* wb = wsp + it.w_off + c;
        const bool col_ok = c < it.n_cols;
        sl_f32x4 d = {0.f, 0.f, 0.f, 0.f};
        const int ksteps = (it.mul_in + 3) >> 2;
        // batches of 16 k-steps: 16 independent gathers in flight per lane, then 16 MFMAs
        for (int k0 = 0; k0 < ksteps; k0 += 16) {
            float av[16], bv[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int u = 4 * (k0 + q) + g;
                const bool u_ok = u < it.mul_in;
                av[q] = (ra_ok && u_ok) ? xa[u * it.x_step] : 0.0f;
                bv[q] = (col_ok && u_ok) ? wb[u * it.w_step] : 0.0f;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (k0 + q < ksteps) d = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bv[q], d, 0, 0, 0);
        }
        // D[row = 4g + r][col = c]
        if (col_ok) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = lo + rt * 16 + 4 * g + r;
                if (rr < hi) {
                    const int nn = order ? order[rr] : rr;
                    const int64_t oi = (int64_t)nn * d_out + it.o_off + c * it.o_step;
                    out[oi] = (add ? add[oi] : 0.0f) + d[r];
                }
            }
        }
    }
}

__device__ __forceinline__ float apply_act(int code, float v) {
    switch (code) {
        case 1: return v / (1.0f + expf(-v));                          // silu
        case 2: return tanhf(v);                                       // tanh
        case 3: return 1.0f / (1.0f + expf(-v));                       // sigmoid
        case 4: return (v > 20.0f ? v : log1pf(expf(v))) - 0.6931471805599453f;  // shifted softplus
        case 5: return fabsf(v);                                       // abs
        default: return v;
    }
}

__global__ void gate_bn_kernel(const float* __restrict__ x, int d_in, const int4* __restrict__ meta, int d_out,
                               const float* __restrict__ act_cst, const float* __restrict__ running_mean,
                               const float* __restrict__ running_var, const float* __restrict__ bn_weight,
                               const float* __restrict__ bn_bias, float eps, int64_t n_rows,
                               float* __restrict__ out) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_rows * d_out) return;
    int64_t n = idx / d_out;
    int o = (int)(idx - n * d_out);
    int4 m = meta[o];
    const float* xr = x + n * d_in;
    float v = xr[m.x];
    int act = m.z & 0xff, gact = (m.z >> 8) & 0xff;
    if (m.y < 0) {
        if (act) v = apply_act(act, v) * act_cst[act];
    } else {
        float gte = xr[m.y];
        if (gact) gte = apply_act(gact, gte) * act_cst[gact];
        v = v * gte;
    }
    if (bn_weight) {
        int bn_idx = m.w & 0xffff, mean_idx = (m.w >> 16) & 0xffff;
        float scale = bn_weight[bn_idx] / sqrtf(running_var[bn_idx] + eps);
        if (mean_idx != 0xffff) v = (v - running_mean[mean_idx]) * scale + bn_bias[mean_idx];
        else v = v * scale;
    }
    out[idx] = v;
}

__global__ void segment_reduce_kernel(const float* __restrict__ x, int dim, const int64_t* __restrict__ ptr,
                                      int64_t n_seg, int mean, float* __restrict__ out) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_seg * dim) return;
    int64_t b = idx / dim;
    int c = (int)(idx - b * dim);
    int64_t beg = ptr[b], end = ptr[b + 1];
    float s = 0.0f;
    for (int64_t n = beg; n < end; ++n) s += x[n * dim + c];
    if (mean) {
        float cnt = (float)(end - beg);
        s = s / (cnt < 1.0f ? 1.0f : cnt);
    }
    out[idx] = s;
}

__global__ void dense_rows_kernel(const float* __restrict__ x, int n_in, const float* __restrict__ q, int n_out,
                                  int64_t n_rows, float* __restrict__ out) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_rows * n_out) return;
    int64_t b = idx / n_out;
    int c = (int)(idx - b * n_out);
    float s = 0.0f;
    for (int k = 0; k < n_in; ++k) s += x[b * n_in + k] * q[k * n_out + c];
    out[idx] = s;
}

}  // namespace

extern "C" int matten_species_linear(const float* x, int64_t d_in, const int32_t* order, const int32_t* seg,
                                     int64_t n_species, const float* wp, int64_t w_stride, const int32_t* items,
                                     int64_t n_items, int64_t d_out, const float* add, int64_t n_rows, float* out,
                                     matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_rows < 0 || d_in <= 0 || d_out <= 0 || n_species <= 0 || w_stride < 0 || n_items < 0 ||
        n_rows >= ((int64_t)1 << 31))
        return MATTEN_EINVAL;
    if (n_rows == 0) return MATTEN_OK;
    if (!x || !wp || !out || (n_items > 0 && !items)) return MATTEN_EINVAL;
    if ((order == nullptr) != (seg == nullptr)) return MATTEN_EINVAL;
    if (!order && n_species != 1) return MATTEN_EINVAL;
    const size_t w_bytes = sizeof(float) * (size_t)((w_stride + 3) & ~3);
    const int w_in_lds = w_bytes <= 64 * 1024 ? 1 : 0;
    const size_t lds = w_in_lds ? w_bytes : 16;
    const int64_t grid = matten_cdiv(n_rows, SL_ROWS_PER_BLOCK) + (order ? n_species : 0);
    species_linear_kernel<<<(unsigned)grid, SL_THREADS, lds, stream>>>(x, (int)d_in, order, seg, (int)n_species, wp,
                                                                       (int)w_stride, w_in_lds,
                                                                       (const LinItem*)items, (int)n_items,
                                                                       (int)d_out, add, (int)n_rows, out);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}

extern "C" int matten_gate_bn(const float* x, int64_t d_in, const int32_t* meta, int64_t d_out, const float* act_cst,
                              const float* running_mean, const float* running_var, const float* bn_weight,
                              const float* bn_bias, float eps, int64_t n_rows, float* out, matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_rows < 0 || d_in <= 0 || d_out <= 0) return MATTEN_EINVAL;
    if (n_rows == 0) return MATTEN_OK;
    if (!x || !meta || !act_cst || !out) return MATTEN_EINVAL;
    if (bn_weight && (!running_var || !running_mean || !bn_bias)) return MATTEN_EINVAL;
    const int T = 256;
    gate_bn_kernel<<<(unsigned)matten_cdiv(n_rows * d_out, T), T, 0, stream>>>(
        x, (int)d_in, (const int4*)meta, (int)d_out, act_cst, running_mean, running_var, bn_weight, bn_bias, eps,
        n_rows, out);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}

extern "C" int matten_segment_reduce(const float* x, int64_t dim, const int64_t* ptr, int64_t n_segments, int mean,
                                     float* out, matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_segments < 0 || dim <= 0) return MATTEN_EINVAL;
    if (n_segments == 0) return MATTEN_OK;
    if (!x || !ptr || !out) return MATTEN_EINVAL;
    const int T = 256;
    segment_reduce_kernel<<<(unsigned)matten_cdiv(n_segments * dim, T), T, 0, stream>>>(x, (int)dim, ptr, n_segments,
                                                                                        mean, out);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}

extern "C" int matten_dense_rows(const float* x, int64_t n_in, const float* q, int64_t n_out, int64_t n_rows,
                                 float* out, matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_rows < 0 || n_in <= 0 || n_out <= 0) return MATTEN_EINVAL;
    if (n_rows == 0) return MATTEN_OK;
    if (!x || !q || !out) return MATTEN_EINVAL;
    const int T = 256;
    dense_rows_kernel<<<(unsigned)matten_cdiv(n_rows * n_out, T), T, 0, stream>>>(x, (int)n_in, q, (int)n_out, n_rows,
                                                                                  out);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}

extern "C" int matten_abi_version(void) { return 5; }
