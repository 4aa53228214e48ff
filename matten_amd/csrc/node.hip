// Node-side operators of the conv stack.
//   species_linear : e3nn FullyConnectedTensorProduct(x, one_hot)  reference nn/conv.py:59-61,77-79,84-86
//                    and e3nn o3.Linear (species == NULL)           reference nn/nodewise.py:111-117
//   gate_bn        : e3nn Gate + BatchNorm(eval)                    reference nn/conv.py:209-211
//   segment_reduce : NodewiseReduce                                 reference nn/nodewise.py:142-148
//   dense_rows     : CartesianTensor.to_cartesian                   reference utils.py:123-124
#include "common.h"

namespace {

// v3: nodes are visited in species-sorted order so that a workgroup works on rows of ONE species:
// it stages that species' packed weight table in LDS once (<= ~36 KB at the paper config), then walks
// its share of the species' rows in chunks of NB rows staged in LDS with coalesced copies.  The inner
// product loop touches LDS only: weights W[w_base(o) + u*w_step] (consecutive lanes -> consecutive
// banks) and features xs[r][x_base(o) + u*x_step] (broadcast).  Every input row is read from HBM once.
//   order[N]   node ids sorted by species (NULL: identity, single weight table)
//   seg[S+1]   offsets of each species' run in `order`
constexpr int SL_ROWS_PER_BLOCK = 16;
constexpr int SL_THREADS = 512;

__global__ __launch_bounds__(SL_THREADS) void species_linear_kernel(
    const float* __restrict__ x, int d_in, int nb, const int32_t* __restrict__ order,
    const int32_t* __restrict__ seg, int n_species, const float* __restrict__ wp, int w_stride, int w_in_lds,
    const int4* __restrict__ out_meta, int d_out, const float* __restrict__ add, int n_rows,
    float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ws = lds;                                   // [w_stride] if w_in_lds
    float* xs = lds + (w_in_lds ? ((w_stride + 3) & ~3) : 0);  // [nb][d_in]

    // which species / which slice of its rows does this block own?
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
        wsp = ws;
    }

    for (int c0 = lo; c0 < hi; c0 += nb) {
        const int rows = min(nb, hi - c0);
        __syncthreads();  // previous chunk fully consumed (and ws visible on the first pass)
        for (int r = 0; r < rows; ++r) {
            const int n = order ? order[c0 + r] : (c0 + r);
            const float* src = x + (int64_t)n * d_in;
            for (int i = threadIdx.x; i < d_in; i += blockDim.x) xs[r * d_in + i] = src[i];
        }
        __syncthreads();
        const int total_out = rows * d_out;
        for (int idx = threadIdx.x; idx < total_out; idx += blockDim.x) {
            const int r = idx / d_out;
            const int o = idx - r * d_out;
            const int4 m = out_meta[o];
            const int x_step = m.y & 0xffff, mul_in = m.y >> 16;
            const float* xp = xs + r * d_in + m.x;
            const float* w = wsp + m.z;
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
            int u = 0;
            for (; u + 4 <= mul_in; u += 4) {
                a0 = fmaf(w[(u + 0) * m.w], xp[(u + 0) * x_step], a0);
                a1 = fmaf(w[(u + 1) * m.w], xp[(u + 1) * x_step], a1);
                a2 = fmaf(w[(u + 2) * m.w], xp[(u + 2) * x_step], a2);
                a3 = fmaf(w[(u + 3) * m.w], xp[(u + 3) * x_step], a3);
            }
            for (; u < mul_in; ++u) a0 = fmaf(w[u * m.w], xp[u * x_step], a0);
            const int n = order ? order[c0 + r] : (c0 + r);
            const int64_t oi = (int64_t)n * d_out + o;
            out[oi] = (add ? add[oi] : 0.0f) + ((a0 + a1) + (a2 + a3));
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
                                     int64_t n_species, const float* wp, int64_t w_stride, const int32_t* out_meta,
                                     int64_t d_out, const float* add, int64_t n_rows, float* out,
                                     matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_rows < 0 || d_in <= 0 || d_out <= 0 || n_species <= 0 || w_stride < 0 || n_rows >= ((int64_t)1 << 31))
        return MATTEN_EINVAL;
    if (n_rows == 0) return MATTEN_OK;
    if (!x || !wp || !out_meta || !out) return MATTEN_EINVAL;
    if ((order == nullptr) != (seg == nullptr)) return MATTEN_EINVAL;
    if (!order && n_species != 1) return MATTEN_EINVAL;
    // LDS plan: weight table (if it fits in 64 KiB) + as many rows as fit in the rest of ~96 KiB, <= 8
    const size_t w_bytes = sizeof(float) * (size_t)((w_stride + 3) & ~3);
    const int w_in_lds = w_bytes <= 64 * 1024 ? 1 : 0;
    const size_t budget = 78 * 1024 - (w_in_lds ? w_bytes : 0);  // two workgroups per CU
    int nb = (int)(budget / (sizeof(float) * (size_t)d_in));
    nb = nb > 8 ? 8 : nb;
    if (nb < 1) nb = 1;
    const size_t lds = (w_in_lds ? w_bytes : 0) + sizeof(float) * (size_t)nb * (size_t)d_in;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)species_linear_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess)
            return MATTEN_ELAUNCH;
        attr_set = true;
    }
    const int64_t grid = matten_cdiv(n_rows, SL_ROWS_PER_BLOCK) + (order ? n_species : 0);
    species_linear_kernel<<<(unsigned)grid, SL_THREADS, lds, stream>>>(x, (int)d_in, nb, order, seg, (int)n_species, wp,
                                                                 (int)w_stride, w_in_lds, (const int4*)out_meta,
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

extern "C" int matten_abi_version(void) { return 4; }
