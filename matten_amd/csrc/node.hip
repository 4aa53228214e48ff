// Node-side operators of the conv stack.
//   (species_linear lives in species_linear.hip)
//   gate_bn        : e3nn Gate + BatchNorm(eval)                    reference nn/conv.py:209-211
//   segment_reduce : NodewiseReduce                                 reference nn/nodewise.py:142-148
//   dense_rows     : CartesianTensor.to_cartesian                   reference utils.py:123-124
#include "common.h"

namespace {

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

// Gate + (eval) BatchNorm.  A thread owns output column o -- its source / gate column, activation codes and the
// BatchNorm affine folded to (scale, shift) live in registers -- and walks the workgroup's GB_ROWS rows GB_UNROLL at a
// time, all loads of a batch issued before the first activation.  Consecutive threads, consecutive columns: coalesced
// like the earlier row-per-wave form (a wave walked a row with the column tables in LDS: 46 us per launch), without
// its five LDS reads per element and with 2 x GB_UNROLL loads in flight per thread: 32 us (4.3 TB/s).  16 or 32 rows
// per workgroup measure the same, 64 rows or an unroll of 16 are 25 % slower (tools/gb_ab.sh).
#ifndef GB_R
#define GB_R 16
#endif
#ifndef GB_U
#define GB_U 8
#endif
constexpr int GB_ROWS = GB_R, GB_UNROLL = GB_U;
__global__ __launch_bounds__(256) void gate_bn_kernel(const float* __restrict__ x, int d_in,
                                                           const int4* __restrict__ meta, int d_out,
                                                           const float* __restrict__ act_cst,
                                                           const float* __restrict__ running_mean,
                                                           const float* __restrict__ running_var,
                                                           const float* __restrict__ bn_weight,
                                                           const float* __restrict__ bn_bias, float eps, int64_t n_rows,
                                                           float* __restrict__ out) {
    const int o = blockIdx.y * blockDim.x + threadIdx.x;
    if (o >= d_out) return;
    const int4 m = meta[o];
    const int act = m.z & 0xff, gact = (m.z >> 8) & 0xff;
    const float ca = act ? act_cst[act] : 1.0f, cg = gact ? act_cst[gact] : 1.0f;
    float scale = 1.0f, shift = 0.0f;
    if (bn_weight) {
        const int bn_idx = m.w & 0xffff, mean_idx = (m.w >> 16) & 0xffff;
        scale = bn_weight[bn_idx] / sqrtf(running_var[bn_idx] + eps);
        if (mean_idx != 0xffff) shift = bn_bias[mean_idx] - running_mean[mean_idx] * scale;
    }
    const int gcol = m.y < 0 ? m.x : m.y;   // scalars read their own column twice (no branch around a load)
    const int64_t row0 = (int64_t)blockIdx.x * GB_ROWS;
    const int rows = (int)min((int64_t)GB_ROWS, n_rows - row0);
    for (int r0 = 0; r0 < rows; r0 += GB_UNROLL) {
        float a[GB_UNROLL], b[GB_UNROLL];
#pragma unroll
        for (int i = 0; i < GB_UNROLL; ++i) {
            const int64_t r = row0 + min(r0 + i, rows - 1);
            a[i] = x[r * d_in + m.x];
            b[i] = x[r * d_in + gcol];
        }
#pragma unroll
        for (int i = 0; i < GB_UNROLL; ++i) {
            if (r0 + i < rows) {
                float v = a[i];
                if (m.y < 0) {
                    if (act) v = apply_act(act, v) * ca;
                } else {
                    float gte = b[i];
                    if (gact) gte = apply_act(gact, gte) * cg;
                    v = v * gte;
                }
                out[(row0 + r0 + i) * d_out + o] = bn_weight ? fmaf(v, scale, shift) : v;
            }
        }
    }
}


__global__ void segment_reduce_kernel(const float* __restrict__ x, int dim, const int64_t* __restrict__ ptr,
                                      int64_t n_seg, int mean, float* __restrict__ out) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_seg * dim) return;
    int64_t b = idx / dim;
    int c = (int)(idx - b * dim);
    int64_t beg = ptr[b], end = ptr[b + 1];
    // same left-to-right sum as before, the loads of eight rows issued together (the dependent adds then run on
    // values that are already there: 22 -> 6 us for 1000 crystals of 64 atoms)
    float s = 0.0f;
    int64_t n = beg;
    for (; n + 8 <= end; n += 8) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = x[(n + i) * dim + c];
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    for (; n < end; ++n) s += x[n * dim + c];
    if (mean) {
        float cnt = (float)(end - beg);
        s = s / (cnt < 1.0f ? 1.0f : cnt);
    }
    out[idx] = s;
}

// reduce = min | max (torch_scatter.scatter_min / _max): value and the row it came from (first of equals; an empty
// crystal gives 0 and row -1)
__global__ void segment_minmax_kernel(const float* __restrict__ x, int dim, const int64_t* __restrict__ ptr, int64_t n_seg,
                                      int take_max, float* __restrict__ out, int64_t* __restrict__ arg) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_seg * dim) return;
    const int64_t b = idx / dim;
    const int c = (int)(idx - b * dim);
    const int64_t beg = ptr[b], end = ptr[b + 1];
    float best = 0.0f;
    int64_t at = -1;
    for (int64_t n = beg; n < end; ++n) {
        const float v = x[n * dim + c];
        if (at < 0 || (take_max ? v > best : v < best)) best = v, at = n;
    }
    out[idx] = best;
    if (arg) arg[idx] = at;
}

// its adjoint: the gradient of (crystal b, column c) goes to the row the value came from (dx zero-initialised)
__global__ void segment_minmax_bwd_kernel(const float* __restrict__ dy, int dim, const int64_t* __restrict__ arg,
                                          int64_t n_seg, float* __restrict__ dx) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_seg * dim) return;
    const int64_t at = arg[idx];
    if (at >= 0) dx[at * dim + (idx % dim)] = dy[idx];
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

extern "C" int matten_gate_bn(const float* x, int64_t d_in, const int32_t* meta, int64_t d_out, const float* act_cst,
                              const float* running_mean, const float* running_var, const float* bn_weight,
                              const float* bn_bias, float eps, int64_t n_rows, float* out, matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_rows < 0 || d_in <= 0 || d_out <= 0) return MATTEN_EINVAL;
    if (n_rows == 0) return MATTEN_OK;
    if (!x || !meta || !act_cst || !out) return MATTEN_EINVAL;
    if (bn_weight && (!running_var || !running_mean || !bn_bias)) return MATTEN_EINVAL;
    const int TC = (int)(d_out >= 256 ? 256 : 64 * matten_cdiv(d_out, 64));   // whole waves, at most one of them part idle
    dim3 grid((unsigned)matten_cdiv(n_rows, GB_ROWS), (unsigned)matten_cdiv(d_out, TC));
    gate_bn_kernel<<<grid, TC, 0, stream>>>(x, (int)d_in, (const int4*)meta, (int)d_out, act_cst, running_mean, running_var,
                                            bn_weight, bn_bias, eps, n_rows, out);
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

extern "C" int matten_segment_minmax(const float* x, int64_t dim, const int64_t* ptr, int64_t n_segments, int take_max,
                                     float* out, int64_t* arg, matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_segments < 0 || dim <= 0) return MATTEN_EINVAL;
    if (n_segments == 0) return MATTEN_OK;
    if (!x || !ptr || !out) return MATTEN_EINVAL;
    const int T = 256;
    segment_minmax_kernel<<<(unsigned)matten_cdiv(n_segments * dim, T), T, 0, stream>>>(x, (int)dim, ptr, n_segments,
                                                                                        take_max, out, arg);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}

extern "C" int matten_segment_minmax_bwd(const float* dy, int64_t dim, const int64_t* arg, int64_t n_segments, float* dx,
                                         matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_segments < 0 || dim <= 0) return MATTEN_EINVAL;
    if (n_segments == 0) return MATTEN_OK;
    if (!dy || !arg || !dx) return MATTEN_EINVAL;
    const int T = 256;
    segment_minmax_bwd_kernel<<<(unsigned)matten_cdiv(n_segments * dim, T), T, 0, stream>>>(dy, (int)dim, arg, n_segments, dx);
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

extern "C" int matten_abi_version(void) { return 44; }
