// Per-edge weighted 'uvu' Clebsch-Gordan tensor product fused with the gather of x[src] and the
// neighbour sum (reference nn/utils.py:230-237,263 + nn/conv.py:113-120).
//
// v1 layout: one workgroup per destination node, walking its CSR segment in fixed order (the
// neighbour sum is a sequential segmented reduction: deterministic, no atomics).  Per edge:
//   1. stage x[src,:], w[e,:], Y(e) in LDS                (coalesced row reads)
//   2. M[t][i,k] = sqrt(2l3+1) sum_j C_ijk Y_j            (CG non-zeros from the plan tables)
//   3. each thread owns outputs o = tid + 256 r and accumulates
//        acc[r] += w[wi(o)] * sum_i x[xb(o)+i] * M[mb(o) + i*d3(o)]
// Outputs beyond 256*R per pass are handled by further passes over the segment.
#include "common.h"

namespace {

constexpr int TPB = 256;

template <int R>
__global__ __launch_bounds__(TPB) void tp_scatter_kernel(
    const float* __restrict__ x, int d_in, const float* __restrict__ w_edge, int w_pad, int w_used,
    const float* __restrict__ sh_sorted, int sh_dim, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ src_sorted, const uint8_t* __restrict__ m_idx, const float* __restrict__ m_coef,
    int m_total, int m_nterms, const int4* __restrict__ out_meta, int d_mid, int o_begin, float avg_nn,
    const float* __restrict__ num_neigh, float* __restrict__ agg) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;                       // [d_in]
    float* ws = xs + ((d_in + 3) & ~3);    // [w_used]
    float* ys = ws + ((w_used + 3) & ~3);  // [sh_dim]
    float* ms = ys + ((sh_dim + 3) & ~3);  // [m_total]

    const int n = blockIdx.x;
    const int tid = threadIdx.x;
    const int beg = rowptr[n], end = rowptr[n + 1];

    int4 meta[R];
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int o = o_begin + tid + TPB * r;
        meta[r] = (o < d_mid) ? out_meta[o] : make_int4(0, 0, 0, 0);
        acc[r] = 0.0f;
    }

    for (int e = beg; e < end; ++e) {
        const int src = src_sorted[e];
        const float* xrow = x + (int64_t)src * d_in;
        const float* wrow = w_edge + (int64_t)e * w_pad;
        for (int i = tid; i < d_in; i += TPB) xs[i] = xrow[i];
        for (int i = tid; i < w_used; i += TPB) ws[i] = wrow[i];
        if (tid < sh_dim) ys[tid] = sh_sorted[(int64_t)e * sh_dim + tid];
        __syncthreads();
        for (int m = tid; m < m_total; m += TPB) {
            float s = 0.0f;
            for (int t = 0; t < m_nterms; ++t) s += m_coef[m * m_nterms + t] * ys[m_idx[m * m_nterms + t]];
            ms[m] = s;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int d1 = meta[r].w & 0xff, d3 = meta[r].w >> 8;
            const float* xp = xs + meta[r].x;
            const float* mp = ms + meta[r].z;
            float s = 0.0f;
            for (int i = 0; i < d1; ++i) s += xp[i] * mp[i * d3];
            acc[r] += ws[meta[r].y] * s;
        }
        __syncthreads();
    }

    const float norm = 1.0f / sqrtf(avg_nn > 0.0f ? avg_nn : num_neigh[n]);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int o = o_begin + tid + TPB * r;
        if (o < d_mid) agg[(int64_t)n * d_mid + o] = acc[r] * norm;
    }
}

}  // namespace

extern "C" int matten_tp_scatter(const float* x, int64_t d_in, const float* w_edge, int64_t w_pad,
                                 const float* sh_sorted, int64_t sh_dim, const int32_t* rowptr,
                                 const int32_t* src_sorted, int64_t n_nodes, const uint8_t* m_terms_idx,
                                 const float* m_terms_coef, int64_t m_total, int64_t m_nterms,
                                 const int32_t* out_meta, int64_t d_mid, float avg_num_neighbors,
                                 const float* num_neigh, float* agg, matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_nodes < 0 || d_in <= 0 || w_pad <= 0 || sh_dim <= 0 || sh_dim > TPB || m_total <= 0 || m_nterms <= 0 ||
        d_mid <= 0)
        return MATTEN_EINVAL;
    if (n_nodes == 0) return MATTEN_OK;
    if (!x || !w_edge || !sh_sorted || !rowptr || !src_sorted || !m_terms_idx || !m_terms_coef || !out_meta || !agg)
        return MATTEN_EINVAL;
    if (!(avg_num_neighbors > 0.0f) && !num_neigh) return MATTEN_EINVAL;
    const int w_used = (int)w_pad;
    size_t lds = sizeof(float) * (((d_in + 3) & ~3) + ((w_used + 3) & ~3) + ((sh_dim + 3) & ~3) + m_total);
    if (lds > 64 * 1024) return MATTEN_EINVAL;

    constexpr int RMAX = 20;
    for (int64_t o_begin = 0; o_begin < d_mid; o_begin += (int64_t)RMAX * TPB) {
        int64_t rem = d_mid - o_begin;
        int r_need = (int)matten_cdiv(rem < (int64_t)RMAX * TPB ? rem : (int64_t)RMAX * TPB, TPB);
#define LAUNCH(RR)                                                                                              \
    tp_scatter_kernel<RR><<<(unsigned)n_nodes, TPB, lds, stream>>>(                                              \
        x, (int)d_in, w_edge, (int)w_pad, w_used, sh_sorted, (int)sh_dim, rowptr, src_sorted, m_terms_idx,      \
        m_terms_coef, (int)m_total, (int)m_nterms, (const int4*)out_meta, (int)d_mid, (int)o_begin,             \
        avg_num_neighbors, num_neigh, agg)
        if (r_need <= 2) LAUNCH(2);
        else if (r_need <= 6) LAUNCH(6);
        else if (r_need <= 10) LAUNCH(10);
        else if (r_need <= 15) LAUNCH(15);
        else LAUNCH(20);
#undef LAUNCH
        MATTEN_LAUNCH_CHECK();
    }
    return MATTEN_OK;
}
