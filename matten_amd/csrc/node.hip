// Node-side operators of the conv stack.
//   species_linear : e3nn FullyConnectedTensorProduct(x, one_hot)  reference nn/conv.py:59-61,77-79,84-86
//                    and e3nn o3.Linear (species == NULL)           reference nn/nodewise.py:111-117
//   gate_bn        : e3nn Gate + BatchNorm(eval)                    reference nn/conv.py:209-211
//   segment_reduce : NodewiseReduce                                 reference nn/nodewise.py:142-148
//   dense_rows     : CartesianTensor.to_cartesian                   reference utils.py:123-124
#include "common.h"

namespace {

// Species-indexed per-irrep linear on the fp32 matrix cores.
// For one irrep block (mul_in -> mul_out channels, d = 2l+1 components) and one component k the op is a
// plain GEMM over rows:  out[row, w, k] = sum_u x[row, u, k] W_s[u, w].  A workgroup owns 16 rows of ONE
// species (rows are visited in species-sorted order) and keeps that species' packed weight table in LDS.
// Per irrep block it streams the rows' slice x[row, x_off + u*d + k] through LDS in chunks of <= 256
// floats per row with coalesced row-segment loads (every input element is fetched from HBM exactly once,
// as full cache lines), and its 4 waves run v_mfma_f32_16x16x4_f32 on (component k, 16-column tile) items:
//      A[m = row][k = u]  = xs[m][(u - u0)*d + k]        LDS, row stride == 2 (mod 32): conflict-free for odd d
//      B[k = u][n = w]    = ws[w_off + u*mo + 16 nt + n]  LDS
//      D[row][w]         -> out[row, o_off + (16 nt + n)*d + k] (+ add)
struct LinSeg {  // 8 x int32: one irrep block of one pass
    int x_off, d, mul_in, w_off, mo, o_off, pad0, pad1;
};
constexpr int SL_ROWS = 16;
constexpr int SL_THREADS = 256;
constexpr int SL_CHUNK = 256;          // floats per row per chunk
constexpr int SL_RS = SL_CHUNK + 2;    // LDS row stride
constexpr int SL_ITEMS_PER_WAVE = 3;
typedef float sl_f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(SL_THREADS) void species_linear_kernel(
    const float* __restrict__ x, int d_in, const int32_t* __restrict__ order, const int32_t* __restrict__ seg,
    int n_species, const float* __restrict__ wp, int w_stride, int w_in_lds, const LinSeg* __restrict__ segs,
    int n_segs, int d_out, const float* __restrict__ add, int n_rows, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;                      // [SL_ROWS][SL_RS]
    float* ws = lds + SL_ROWS * SL_RS;    // [w_stride] if w_in_lds

    int b = blockIdx.x, s = 0, lo = 0, hi = 0;
    if (seg) {
        bool found = false;
        for (s = 0; s < n_species; ++s) {
            const int beg = seg[s], end = seg[s + 1];
            const int nblk = (end - beg + SL_ROWS - 1) / SL_ROWS;
            if (b < nblk) {
                lo = beg + b * SL_ROWS;
                hi = min(end, lo + SL_ROWS);
                found = true;
                break;
            }
            b -= nblk;
        }
        if (!found) return;
    } else {
        lo = b * SL_ROWS;
        hi = min(n_rows, lo + SL_ROWS);
        if (lo >= hi) return;
    }
    const float* wsp = wp + (int64_t)s * w_stride;
    if (w_in_lds) {
        for (int i = threadIdx.x; i < w_stride; i += blockDim.x) ws[i] = wsp[i];
        wsp = ws;
    }

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    // staging role: 16 threads per row
    const int sr = threadIdx.x >> 4, sc = threadIdx.x & 15;
    const int srow = lo + sr;
    const int snode = (srow < hi) ? (order ? order[srow] : srow) : -1;

    for (int sg = 0; sg < n_segs; ++sg) {
        const LinSeg L = segs[sg];
        const int d = L.d;
        const int nt_count = (L.mo + 15) >> 4;
        const int n_items = d * nt_count;
        const int uc = max(4, (SL_CHUNK / d) & ~3);  // channels per chunk
        for (int ig = 0; ig < n_items; ig += 4 * SL_ITEMS_PER_WAVE) {
            sl_f32x4 acc[SL_ITEMS_PER_WAVE];
#pragma unroll
            for (int q = 0; q < SL_ITEMS_PER_WAVE; ++q) acc[q] = sl_f32x4{0.f, 0.f, 0.f, 0.f};
            for (int u0 = 0; u0 < L.mul_in; u0 += uc) {
                const int ucnt = min(uc, L.mul_in - u0);
                const int nfl = ucnt * d;
                __syncthreads();  // previous chunk consumed (and ws staged on the first pass)
                if (snode >= 0) {
                    const float* src = x + (int64_t)snode * d_in + L.x_off + u0 * d;
                    for (int i = sc; i < nfl; i += 16) xs[sr * SL_RS + i] = src[i];
                } else {
                    for (int i = sc; i < nfl; i += 16) xs[sr * SL_RS + i] = 0.0f;
                }
                __syncthreads();
                const int ksteps = (ucnt + 3) >> 2;
#pragma unroll
                for (int q = 0; q < SL_ITEMS_PER_WAVE; ++q) {
                    const int item = ig + wave + 4 * q;
                    if (item < n_items) {
                        const int k = item % d, nt = item / d;
                        const float* ap = xs + c * SL_RS + k;
                        const float* bp = wsp + L.w_off + (int64_t)u0 * L.mo + nt * 16 + c;
                        const bool col_ok = nt * 16 + c < L.mo;
                        sl_f32x4 dacc = acc[q];
                        for (int k0 = 0; k0 < ksteps; k0 += 8) {
                            float av[8], bv[8];
#pragma unroll
                            for (int t = 0; t < 8; ++t) {
                                const int ul = 4 * (k0 + t) + g;
                                const bool u_ok = ul < ucnt;
                                av[t] = u_ok ? ap[ul * d] : 0.0f;
                                bv[t] = (u_ok && col_ok) ? bp[ul * L.mo] : 0.0f;
                            }
#pragma unroll
                            for (int t = 0; t < 8; ++t)
                                if (k0 + t < ksteps) dacc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv[t], dacc, 0, 0, 0);
                        }
                        acc[q] = dacc;
                    }
                }
            }
            // D[row = 4g + r][col = c]
#pragma unroll
            for (int q = 0; q < SL_ITEMS_PER_WAVE; ++q) {
                const int item = ig + wave + 4 * q;
                if (item < n_items) {
                    const int k = item % d, nt = item / d;
                    if (nt * 16 + c < L.mo) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int rr = lo + 4 * g + r;
                            if (rr < hi) {
                                const int nn = order ? order[rr] : rr;
                                const int64_t oi = (int64_t)nn * d_out + L.o_off + (nt * 16 + c) * d + k;
                                out[oi] = (add ? add[oi] : 0.0f) + acc[q][r];
                            }
                        }
                    }
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
                                     int64_t n_species, const float* wp, int64_t w_stride, const int32_t* segs,
                                     int64_t n_segs, int64_t d_out, const float* add, int64_t n_rows, float* out,
                                     matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_rows < 0 || d_in <= 0 || d_out <= 0 || n_species <= 0 || w_stride < 0 || n_segs < 0 ||
        n_rows >= ((int64_t)1 << 31))
        return MATTEN_EINVAL;
    if (n_rows == 0) return MATTEN_OK;
    if (!x || !wp || !out || (n_segs > 0 && !segs)) return MATTEN_EINVAL;
    if ((order == nullptr) != (seg == nullptr)) return MATTEN_EINVAL;
    if (!order && n_species != 1) return MATTEN_EINVAL;
    const size_t w_bytes = sizeof(float) * (size_t)((w_stride + 3) & ~3);
    const int w_in_lds = w_bytes <= 96 * 1024 ? 1 : 0;
    const size_t lds = sizeof(float) * SL_ROWS * SL_RS + (w_in_lds ? w_bytes : 0);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)species_linear_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess)
            return MATTEN_ELAUNCH;
        attr_set = true;
    }
    const int64_t grid = matten_cdiv(n_rows, SL_ROWS) + (order ? n_species : 0);
    species_linear_kernel<<<(unsigned)grid, SL_THREADS, lds, stream>>>(x, (int)d_in, order, seg, (int)n_species, wp,
                                                                       (int)w_stride, w_in_lds, (const LinSeg*)segs,
                                                                       (int)n_segs, (int)d_out, add, (int)n_rows,
                                                                       out);
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

extern "C" int matten_abi_version(void) { return 7; }
