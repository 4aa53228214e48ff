// Destination-sorted CSR of the edge list (stable radix sort by dst => deterministic neighbour sums).
// Replaces the unordered atomics of torch_scatter.scatter on the reference's path (nn/conv.py:114).
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"

namespace {

__global__ void csr_keys_kernel(const int64_t* __restrict__ edge_index, int64_t E, int64_t N,
                                int32_t* __restrict__ keys, int32_t* __restrict__ vals, int32_t* err_flag) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    int64_t s = edge_index[e];
    int64_t d = edge_index[E + e];
    if (s < 0 || s >= N || d < 0 || d >= N) {
        atomicOr(err_flag, 1);
        d = d < 0 ? 0 : (d >= N ? N - 1 : d);
    }
    keys[e] = (int32_t)d;
    vals[e] = (int32_t)e;
}

__global__ void csr_rowptr_kernel(const int32_t* __restrict__ keys_sorted, int64_t E, int64_t N,
                                  int32_t* __restrict__ rowptr) {
    int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n > N) return;
    // rowptr[n] = first position whose key >= n
    int64_t lo = 0, hi = E;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (keys_sorted[mid] < (int32_t)n) lo = mid + 1; else hi = mid;
    }
    rowptr[n] = (int32_t)lo;
}

__global__ void csr_src_kernel(const int64_t* __restrict__ edge_index, const int32_t* __restrict__ perm, int64_t E,
                               int64_t N, int32_t* __restrict__ src_sorted) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    int64_t s = edge_index[perm[e]];
    s = s < 0 ? 0 : (s >= N ? N - 1 : s);
    src_sorted[e] = (int32_t)s;
}

// keys of a plain grouping (matten_group_by_key): value = position, key clamped into [0, n_keys)
__global__ void group_keys_kernel(const int64_t* __restrict__ key, int64_t n, int64_t n_keys,
                                  int32_t* __restrict__ keys, int32_t* __restrict__ vals, int32_t* err_flag) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t k = key[i];
    if (k < 0 || k >= n_keys) {
        atomicOr(err_flag, 1);
        k = k < 0 ? 0 : n_keys - 1;
    }
    keys[i] = (int32_t)k;
    vals[i] = (int32_t)i;
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

inline int key_bits(int64_t N) {
    int b = 1;
    while (((int64_t)1 << b) < N && b < 31) ++b;
    return b;
}

}  // namespace

extern "C" size_t matten_csr_workspace_bytes(int64_t E, int64_t N) {
    if (E < 0 || N < 0) return 0;
    size_t tmp = 0;
    int32_t* p = nullptr;
    // size query only (no launch)
    (void)rocprim::radix_sort_pairs(nullptr, tmp, p, p, p, p, (size_t)(E > 0 ? E : 1), 0, key_bits(N), 0, false);
    return 3 * align256((size_t)(E > 0 ? E : 1) * sizeof(int32_t)) + align256(tmp) + 256;
}

extern "C" int matten_csr_build(const int64_t* edge_index, int64_t E, int64_t N, int32_t* perm, int32_t* rowptr,
                                int32_t* src_sorted, void* workspace, size_t workspace_bytes, int32_t* err_flag,
                                matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (E < 0 || N < 0 || N >= ((int64_t)1 << 31) || E >= ((int64_t)1 << 31)) return MATTEN_EINVAL;
    if (!rowptr || !err_flag) return MATTEN_EINVAL;
    if (E == 0) {
        if (hipMemsetAsync(rowptr, 0, (size_t)(N + 1) * sizeof(int32_t), stream) != hipSuccess) return MATTEN_ELAUNCH;
        return MATTEN_OK;
    }
    if (!edge_index || !perm || !src_sorted || !workspace) return MATTEN_EINVAL;
    size_t need = matten_csr_workspace_bytes(E, N);
    if (workspace_bytes < need) return MATTEN_ENOMEM;

    char* ws = (char*)workspace;
    size_t seg = align256((size_t)E * sizeof(int32_t));
    int32_t* keys_in = (int32_t*)ws;
    int32_t* keys_out = (int32_t*)(ws + seg);
    int32_t* vals_in = (int32_t*)(ws + 2 * seg);
    void* tmp = ws + 3 * seg;
    size_t tmp_bytes = workspace_bytes - 3 * seg;

    const int T = 256;
    csr_keys_kernel<<<(unsigned)matten_cdiv(E, T), T, 0, stream>>>(edge_index, E, N, keys_in, vals_in, err_flag);
    MATTEN_LAUNCH_CHECK();
    if (rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, vals_in, perm, (size_t)E, 0, key_bits(N), stream,
                                  false) != hipSuccess)
        return MATTEN_ELAUNCH;
    csr_rowptr_kernel<<<(unsigned)matten_cdiv(N + 1, T), T, 0, stream>>>(keys_out, E, N, rowptr);
    MATTEN_LAUNCH_CHECK();
    csr_src_kernel<<<(unsigned)matten_cdiv(E, T), T, 0, stream>>>(edge_index, perm, E, N, src_sorted);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}

// order[n] = the positions 0..n-1 stably sorted by key[i] in [0, n_keys); seg[n_keys+1] = first position of each key.
// Used for the species grouping the species-indexed linears walk (reference nn/conv.py:59-86 evaluates them densely
// over the one-hot).  err_flag bit 0 is set when a key is out of range (the item is then grouped with the nearest key).
extern "C" int matten_group_by_key(const int64_t* key, int64_t n, int64_t n_keys, int32_t* order, int32_t* seg,
                                   void* workspace, size_t workspace_bytes, int32_t* err_flag, matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0 || n_keys <= 0 || n_keys >= ((int64_t)1 << 31) || n >= ((int64_t)1 << 31)) return MATTEN_EINVAL;
    if (!seg || !err_flag) return MATTEN_EINVAL;
    if (n == 0) {
        if (hipMemsetAsync(seg, 0, (size_t)(n_keys + 1) * sizeof(int32_t), stream) != hipSuccess) return MATTEN_ELAUNCH;
        return MATTEN_OK;
    }
    if (!key || !order || !workspace) return MATTEN_EINVAL;
    if (workspace_bytes < matten_csr_workspace_bytes(n, n_keys)) return MATTEN_ENOMEM;
    char* ws = (char*)workspace;
    size_t sg = align256((size_t)n * sizeof(int32_t));
    int32_t* keys_in = (int32_t*)ws;
    int32_t* keys_out = (int32_t*)(ws + sg);
    int32_t* vals_in = (int32_t*)(ws + 2 * sg);
    void* tmp = ws + 3 * sg;
    size_t tmp_bytes = workspace_bytes - 3 * sg;
    const int T = 256;
    group_keys_kernel<<<(unsigned)matten_cdiv(n, T), T, 0, stream>>>(key, n, n_keys, keys_in, vals_in, err_flag);
    MATTEN_LAUNCH_CHECK();
    if (rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, vals_in, order, (size_t)n, 0, key_bits(n_keys),
                                  stream, false) != hipSuccess)
        return MATTEN_ELAUNCH;
    csr_rowptr_kernel<<<(unsigned)matten_cdiv(n_keys + 1, T), T, 0, stream>>>(keys_out, n, n_keys, seg);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}
