// Destination-sorted CSR of the edge list (stable radix sort by dst => deterministic neighbour sums).
// Replaces the unordered atomics of torch_scatter.scatter on the reference's path (nn/conv.py:114).
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

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

// ---- counting build (sparse graphs: a few dozen edges per node) -----------------------------------------------------
// The radix sort spends ~135 us in 14 launches on the 1.15 M edges of 1000 fcc-64 crystals.  With short segments the
// same stable order comes cheaper: count the in-degrees, scan them into rowptr, drop every edge into its node's segment
// in arrival order (the counting atomic's return value: any order), then rank the ids INSIDE each segment (16 lanes per node, degree^2 / 16
// comparisons) -- sorted by (dst, edge id), i.e. exactly the stable sort's permutation.
// (a kernel, not hipMemsetAsync: a memset node captured into a hipGraph was not re-executed on replay on ROCm 7.2 --
// the counters then kept the previous replay's totals and the placement wrote past its segment)
__global__ void zero_i32_kernel(int32_t* __restrict__ p, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
}

__global__ void csr_count_kernel(const int64_t* __restrict__ edge_index, int64_t E, int64_t N,
                                 int32_t* __restrict__ deg, int32_t* __restrict__ slot, int32_t* err_flag) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    int64_t s = edge_index[e];
    int64_t d = edge_index[E + e];
    if (s < 0 || s >= N || d < 0 || d >= N) {
        atomicOr(err_flag, 1);
        d = d < 0 ? 0 : (d >= N ? N - 1 : d);
    }
    slot[e] = atomicAdd(&deg[d], 1);   // arrival order inside the segment (any order: ranked below)
}

__global__ void csr_place_kernel(const int64_t* __restrict__ edge_index, int64_t E, int64_t N,
                                 const int32_t* __restrict__ rowptr, const int32_t* __restrict__ slot,
                                 int32_t* __restrict__ ids) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    int64_t d = edge_index[E + e];
    d = d < 0 ? 0 : (d >= N ? N - 1 : d);
    ids[rowptr[d] + slot[e]] = (int32_t)e;
}

// Segments longer than this are not ranked quadratically by 16 lanes (one hub node with 10^5-10^6 in-edges would be
// 10^10-10^12 serial comparisons in a half-wave): csr_rank_kernel queues them and csr_hub_kernel sorts each with a
// whole workgroup, O(d log^2 d / 1024) per thread.
constexpr int CSR_HUB_DEGREE = 2048;
constexpr int CSR_HUB_THREADS = 1024;
constexpr int CSR_HUB_BLOCKS = 64;

// hubs[0..N): queue of hub nodes, hubs[N]: its length (the in-degree counters, dead after the scan; counter N is zero)
__global__ void csr_rank_kernel(const int64_t* __restrict__ edge_index, const int32_t* __restrict__ rowptr,
                                const int32_t* __restrict__ ids, int64_t E, int64_t N, int32_t* __restrict__ perm,
                                int32_t* __restrict__ src_sorted, int32_t* __restrict__ hubs) {
    const int64_t node = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    if (node >= N) return;
    const int sub = threadIdx.x & 15;
    const int beg = rowptr[node], d = rowptr[node + 1] - beg;
    if (d > CSR_HUB_DEGREE) {
        if (sub == 0) hubs[atomicAdd(&hubs[N], 1)] = (int32_t)node;
        return;
    }
    for (int i = sub; i < d; i += 16) {
        const int id = ids[beg + i];
        int rank = 0;
        int j = 0;
        for (; j + 8 <= d; j += 8) {   // eight loads in flight (one per step was a chain of round trips: 23 us for the
            int v[8];                  // 80-edge segments of a 473-node batch)
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = ids[beg + j + q];
#pragma unroll
            for (int q = 0; q < 8; ++q) rank += v[q] < id;
        }
        for (; j < d; ++j) rank += ids[beg + j] < id;
        perm[beg + rank] = id;
        int64_t s = edge_index[id];
        s = s < 0 ? 0 : (s >= N ? N - 1 : s);
        src_sorted[beg + rank] = (int32_t)s;
    }
}

// In-place bitonic sort of a hub segment's edge ids by one workgroup (the all-ascending network: first step of a merge
// pairs i with its mirror i ^ (k - 1), the rest with i ^ j; partners past the end count as +inf and never move, so any
// length works), then the same perm / src_sorted outputs as the ranking above.
__global__ __launch_bounds__(CSR_HUB_THREADS) void csr_hub_kernel(const int64_t* __restrict__ edge_index,
                                                                  const int32_t* __restrict__ rowptr, int32_t* ids,
                                                                  int64_t N, int32_t* __restrict__ perm,
                                                                  int32_t* __restrict__ src_sorted,
                                                                  const int32_t* __restrict__ hubs) {
    const int n_hubs = hubs[N];
    for (int h = blockIdx.x; h < n_hubs; h += gridDim.x) {
        const int node = hubs[h];
        const int beg = rowptr[node], d = rowptr[node + 1] - beg;
        int32_t* a = ids + beg;
        for (int64_t k = 2; (k >> 1) < d; k <<= 1) {
            for (int64_t j = k >> 1; j > 0; j >>= 1) {
                const int64_t m = (j == (k >> 1)) ? k - 1 : j;   // mirror step, then butterfly steps
                for (int64_t i = threadIdx.x; i < d; i += CSR_HUB_THREADS) {
                    const int64_t l = i ^ m;
                    if (l > i && l < d) {
                        const int32_t x = a[i], y = a[l];
                        if (y < x) a[i] = y, a[l] = x;
                    }
                }
                __syncthreads();
            }
        }
        for (int64_t i = threadIdx.x; i < d; i += CSR_HUB_THREADS) {
            const int32_t id = a[i];
            perm[beg + i] = id;
            int64_t s = edge_index[id];
            s = s < 0 ? 0 : (s >= N ? N - 1 : s);
            src_sorted[beg + i] = (int32_t)s;
        }
        __syncthreads();
    }
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

// ---- grouping by a small key (species): counting form ----------------------------------------------------------------
// cnt[k][w] = items with key k in the w-th run of 64 items; an exclusive scan of the key-major table is the stable
// position of (k, w); inside a run the rank among equal keys comes from wave ballots.  3 launches + a scan instead
// of the radix sort's 9.
constexpr int GROUP_COUNTING_MAX_KEYS = 256;
constexpr int GROUP_WAVES = 4;

__device__ __forceinline__ int group_key_of(const int64_t* __restrict__ key, int64_t i, int64_t n, int64_t n_keys,
                                            int32_t* err_flag) {
    if (i >= n) return -1;
    int64_t k = key[i];
    if (k < 0 || k >= n_keys) {
        atomicOr(err_flag, 1);
        k = k < 0 ? 0 : n_keys - 1;
    }
    return (int)k;
}

__global__ __launch_bounds__(GROUP_WAVES * 64) void group_hist_kernel(const int64_t* __restrict__ key, int64_t n,
                                                                      int64_t n_keys, int64_t n_runs,
                                                                      int32_t* __restrict__ cnt, int32_t* err_flag) {
    __shared__ int h[GROUP_WAVES][GROUP_COUNTING_MAX_KEYS];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t run = (int64_t)blockIdx.x * GROUP_WAVES + wv;
    for (int kk = lane; kk < n_keys; kk += 64) h[wv][kk] = 0;
    __builtin_amdgcn_wave_barrier();
    const int k = group_key_of(key, run * 64 + lane, n, n_keys, err_flag);
    if (k >= 0) atomicAdd(&h[wv][k], 1);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    if (run < n_runs)
        for (int kk = lane; kk < n_keys; kk += 64) cnt[(int64_t)kk * n_runs + run] = h[wv][kk];
}

__global__ __launch_bounds__(GROUP_WAVES * 64) void group_place_kernel(const int64_t* __restrict__ key, int64_t n,
                                                                       int64_t n_keys, int64_t n_runs,
                                                                       const int32_t* __restrict__ offs,
                                                                       int32_t* __restrict__ order,
                                                                       int32_t* __restrict__ seg, int32_t* err_flag) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t run = i >> 6;
    if (i <= n_keys) seg[i] = i < n_keys ? offs[i * n_runs] : (int32_t)n;
    const int k = group_key_of(key, i, n, n_keys, err_flag);
    unsigned long long todo = __ballot(k >= 0);
    int rank = 0;
    while (todo) {   // one round per distinct key of the run
        const int kf = __shfl(k, __ffsll((long long)todo) - 1);
        const unsigned long long m = __ballot(k == kf);
        if (k == kf) rank = __popcll(m & ((1ull << lane) - 1ull));
        todo &= ~m;
    }
    if (k >= 0) order[offs[(int64_t)k * n_runs + run] + rank] = (int32_t)i;
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

inline int key_bits(int64_t N) {
    int b = 1;
    while (((int64_t)1 << b) < N && b < 31) ++b;
    return b;
}

}  // namespace

// ---- long CSR segments cut into VIRTUAL nodes (small batches) ------------------------------------------------------------
// The tensor-product kernels give a lane one (destination node, channel) and walk the node's segment edge by edge: the
// launch lasts as long as the longest segment.  In a large regular batch that is the degree (18-30 everywhere); in a small
// batch of real crystals it is the one hub (a one- or two-atom cell has 100-300 neighbours inside the cutoff while the
// average is 30: the reference's n100 sample spends 65 % of its forward walking a handful of such segments on a few CUs).
// Here every segment is cut into the fewest pieces of at most `max_len` edges, of equal length up to one edge (the launch
// lasts as long as the LONGEST piece); the pieces tile the sorted edge list in order, so
// vrowptr[] is itself a CSR row pointer over virtual nodes, the kernels run on it unchanged, and the real node's sum is
// the ordered sum of its pieces (matten_segment_reduce over vseg): fixed order, independent of the rest of the batch.
// One workgroup (the batches this is for have a few thousand nodes): block scans of 1024 nodes with a running carry.
__global__ __launch_bounds__(1024) void csr_split_kernel(const int* __restrict__ rowptr, int N, int max_len, int nv_bound,
                                                         int* __restrict__ vrowptr, int64_t* __restrict__ vseg,
                                                         const float* __restrict__ num_neigh, float* __restrict__ vnn) {
    __shared__ int sh[1024];
    __shared__ int carry_s;
    const int t = threadIdx.x;
    if (t == 0) carry_s = 0;
    __syncthreads();
    const int E = rowptr[N];
    for (int base = 0; base < N; base += 1024) {
        const int n = base + t;
        int beg = 0, k = 0, d = 0;
        if (n < N) {
            beg = rowptr[n];
            d = rowptr[n + 1] - beg;
            k = max(1, (d + max_len - 1) / max_len);
        }
        sh[t] = k;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {   // inclusive scan (Hillis-Steele)
            const int v = t >= off ? sh[t - off] : 0;
            __syncthreads();
            sh[t] += v;
            __syncthreads();
        }
        const int carry = carry_s;
        const int first = carry + sh[t] - k;
        if (n < N) {
            vseg[n] = first;
            for (int i = 0; i < k; ++i) {
                const int v = first + i;
                if (v < nv_bound) {
                    vrowptr[v] = beg + (int)((int64_t)i * d / k);   // balanced pieces (18 edges at max_len 16: 9 + 9, not 16 + 2)
                    if (vnn) vnn[v] = num_neigh[n];
                }
            }
        }
        __syncthreads();
        if (t == 1023) carry_s = carry + sh[1023];
        __syncthreads();
    }
    const int nv = min(carry_s, nv_bound);
    if (t == 0) vseg[N] = nv;
    for (int v = nv + t; v <= nv_bound; v += 1024) {   // unused virtual nodes: empty segments behind the last edge
        vrowptr[v] = E;
        if (vnn && v < nv_bound) vnn[v] = 1.0f;
    }
}

extern "C" size_t matten_csr_workspace_bytes(int64_t E, int64_t N) {
    if (E < 0 || N < 0) return 0;
    size_t tmp = 0;
    int32_t* p = nullptr;
    // size query only (no launch)
    (void)rocprim::radix_sort_pairs(nullptr, tmp, p, p, p, p, (size_t)(E > 0 ? E : 1), 0, key_bits(N), 0, false);
    size_t scan_tmp = 0;
    (void)rocprim::exclusive_scan(nullptr, scan_tmp, p, p, 0, (size_t)(N + 1), rocprim::plus<int32_t>());
    // radix path: 3 edge-sized arrays + its temporaries; counting path: 2 edge-sized arrays + N+1 counters + scan temporaries
    return 3 * align256((size_t)(E > 0 ? E : 1) * sizeof(int32_t)) + align256(tmp) + align256((size_t)(N + 1) * sizeof(int32_t)) +
           align256(scan_tmp) + 256;
}

// workspace of matten_group_by_key(n items, n_keys): the sort's, or the counting form's two [n_keys][ceil(n / 64)] tables
extern "C" size_t matten_group_workspace_bytes(int64_t n, int64_t n_keys) {
    if (n < 0 || n_keys <= 0) return 0;
    size_t need = matten_csr_workspace_bytes(n, n_keys);
    if (n_keys <= GROUP_COUNTING_MAX_KEYS) {
        const size_t table = (size_t)n_keys * (size_t)((n + 63) / 64 + 1);
        size_t scan_tmp = 0;
        int32_t* p = nullptr;
        (void)rocprim::exclusive_scan(nullptr, scan_tmp, p, p, 0, table, rocprim::plus<int32_t>());
        const size_t counting = 2 * align256(table * sizeof(int32_t)) + align256(scan_tmp) + 256;
        if (counting > need) need = counting;
    }
    return need;
}

// average in-degree up to which the counting build is used (the ranking step is quadratic in a node's degree)
constexpr int64_t CSR_COUNTING_MAX_AVG_DEGREE = 64;

extern "C" int matten_csr_counting_max_avg_degree(void) { return (int)CSR_COUNTING_MAX_AVG_DEGREE; }

extern "C" int matten_csr_build(const int64_t* edge_index, int64_t E, int64_t N, int32_t* perm, int32_t* rowptr,
                                int32_t* src_sorted, void* workspace, size_t workspace_bytes, int32_t* err_flag,
                                matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (E < 0 || N < 0 || N >= ((int64_t)1 << 31) || E >= ((int64_t)1 << 31)) return MATTEN_EINVAL;
    if (!rowptr || !err_flag) return MATTEN_EINVAL;
    if (E == 0) {
        zero_i32_kernel<<<(unsigned)matten_cdiv(N + 1, 256), 256, 0, stream>>>(rowptr, N + 1);
        MATTEN_LAUNCH_CHECK();
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
    if (N > 0 && E <= CSR_COUNTING_MAX_AVG_DEGREE * N) {
        int32_t* ids = (int32_t*)ws;
        int32_t* slot = (int32_t*)(ws + seg);
        int32_t* deg = (int32_t*)(ws + 2 * seg);
        const size_t deg_bytes = align256((size_t)(N + 1) * sizeof(int32_t));
        void* scan_tmp = ws + 2 * seg + deg_bytes;
        size_t scan_bytes = workspace_bytes - 2 * seg - deg_bytes;
        zero_i32_kernel<<<(unsigned)matten_cdiv(N + 1, T), T, 0, stream>>>(deg, N + 1);
        MATTEN_LAUNCH_CHECK();
        csr_count_kernel<<<(unsigned)matten_cdiv(E, T), T, 0, stream>>>(edge_index, E, N, deg, slot, err_flag);
        MATTEN_LAUNCH_CHECK();
        if (rocprim::exclusive_scan(scan_tmp, scan_bytes, deg, rowptr, 0, (size_t)(N + 1), rocprim::plus<int32_t>(),
                                    stream) != hipSuccess)
            return MATTEN_ELAUNCH;
        csr_place_kernel<<<(unsigned)matten_cdiv(E, T), T, 0, stream>>>(edge_index, E, N, rowptr, slot, ids);
        MATTEN_LAUNCH_CHECK();
        csr_rank_kernel<<<(unsigned)matten_cdiv(N * 16, T), T, 0, stream>>>(edge_index, rowptr, ids, E, N, perm,
                                                                           src_sorted, deg);
        MATTEN_LAUNCH_CHECK();
        if (E > CSR_HUB_DEGREE) {  // a segment can only be that long if the graph has that many edges
            const int64_t max_hubs = E / CSR_HUB_DEGREE;
            csr_hub_kernel<<<(unsigned)(max_hubs < CSR_HUB_BLOCKS ? max_hubs : CSR_HUB_BLOCKS), CSR_HUB_THREADS, 0,
                             stream>>>(edge_index, rowptr, ids, N, perm, src_sorted, deg);
            MATTEN_LAUNCH_CHECK();
        }
        return MATTEN_OK;
    }
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
        zero_i32_kernel<<<(unsigned)matten_cdiv(n_keys + 1, 256), 256, 0, stream>>>(seg, n_keys + 1);
        MATTEN_LAUNCH_CHECK();
        return MATTEN_OK;
    }
    if (!key || !order || !workspace) return MATTEN_EINVAL;
    if (workspace_bytes < matten_group_workspace_bytes(n, n_keys)) return MATTEN_ENOMEM;
    char* ws = (char*)workspace;
    if (n_keys <= GROUP_COUNTING_MAX_KEYS) {
        const int64_t n_runs = (n + 63) / 64;
        const size_t table = (size_t)n_keys * (size_t)n_runs;
        const size_t tb = align256((size_t)n_keys * (size_t)(n_runs + 1) * sizeof(int32_t));
        int32_t* cnt = (int32_t*)ws;
        int32_t* offs = (int32_t*)(ws + tb);
        void* scan_tmp = ws + 2 * tb;
        size_t scan_bytes = workspace_bytes - 2 * tb;
        const int TB = GROUP_WAVES * 64;
        group_hist_kernel<<<(unsigned)matten_cdiv(n_runs, GROUP_WAVES), TB, 0, stream>>>(key, n, n_keys, n_runs, cnt,
                                                                                        err_flag);
        MATTEN_LAUNCH_CHECK();
        if (rocprim::exclusive_scan(scan_tmp, scan_bytes, cnt, offs, 0, table, rocprim::plus<int32_t>(), stream) !=
            hipSuccess)
            return MATTEN_ELAUNCH;
        const int64_t threads = n > n_keys + 1 ? n : n_keys + 1;
        group_place_kernel<<<(unsigned)matten_cdiv(threads, TB), TB, 0, stream>>>(key, n, n_keys, n_runs, offs, order,
                                                                                 seg, err_flag);
        MATTEN_LAUNCH_CHECK();
        return MATTEN_OK;
    }
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

extern "C" int64_t matten_csr_split_bound(int64_t n_nodes, int64_t n_edges, int64_t max_len) {
    return max_len > 0 ? n_nodes + n_edges / max_len : n_nodes;   // every node one piece + one more per max_len edges
}

extern "C" int matten_csr_split(const int32_t* rowptr, int64_t n_nodes, int64_t n_edges, int64_t max_len, int32_t* vrowptr,
                                int64_t* vseg, const float* num_neigh, float* vnn, matten_stream_t stream_) {
    if (n_nodes < 0 || n_edges < 0 || max_len <= 0 || n_nodes >= ((int64_t)1 << 30) || n_edges >= ((int64_t)1 << 31))
        return MATTEN_EINVAL;
    if (!rowptr || !vrowptr || !vseg || ((num_neigh == nullptr) != (vnn == nullptr))) return MATTEN_EINVAL;
    const int64_t bound = matten_csr_split_bound(n_nodes, n_edges, max_len);
    if (bound >= ((int64_t)1 << 31)) return MATTEN_EINVAL;
    csr_split_kernel<<<1, 1024, 0, (hipStream_t)stream_>>>(rowptr, (int)n_nodes, (int)max_len, (int)bound, vrowptr, vseg,
                                                          num_neigh, vnn);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}
