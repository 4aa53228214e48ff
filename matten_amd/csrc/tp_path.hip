// 'uvu' Clebsch-Gordan tensor product + gather + neighbour sum, v2: one wave per (path, node group).
// (reference nn/utils.py:230-237,263 + nn/conv.py:113-120)
//
// Mapping.  A lane owns ONE output channel (path p, multiplicity index u) of ONE destination node and
// walks that node's CSR segment: the neighbour sum is a per-lane sequential segmented reduction --
// fixed order, no atomics, no cross-lane traffic -- and the 2*l3+1 accumulators never leave registers.
// Everything a lane needs per edge is contiguous:
//      w[e, w_off+u]                     1 dword      (lanes of a node: consecutive u  -> coalesced)
//      x[src, x_off + u*d1 .. +d1)       d1 dwords    (lanes of a node: mul*d1 contiguous floats)
//      Y_l2(e) = sh[e, l2^2 .. +d2)      d2 dwords    (lanes of a node: broadcast)
// The coupling coefficients are compile-time literals (cg_gen.h): each (l1,l2,l3) is its own unrolled
// body, selected once per wave by a wave-uniform switch.  ~40-64 VGPRs => 8 waves/SIMD.
//
// Work decomposition.  Nodes are cut into tiles of TILE_NODES; all (path, node-group) units of a tile
// are adjacent in the grid and pinned to one XCD (blockIdx % 8), so the tile's per-edge weight rows
// (the only operand streamed from HBM) are fetched into that XCD's L2 once and reused by every path.
#include "cg_gen.h"
#include "common.h"

namespace {

constexpr int TILE_NODES = 64;
constexpr int WAVES_PER_BLOCK = 4;
constexpr int N_XCD = 8;

struct PathEntry {  // 8 x int32, built by matten_amd/plan.py
    int triple;     // l1*25 + l2*5 + l3
    int x_off;      // offset of channel 0 of this entry in the node feature row
    int w_off;      // offset of channel 0 in the per-edge weight row
    int out_off;    // offset of channel 0 in the message row
    int mul;        // channels in this entry (<= 64)
    int cu_log2;    // lanes per node = 1 << cu_log2 >= mul
    int pad0, pad1;
};

struct Args {
    const float* x;
    const void* w_edge;   // fp32, or bf16 when w_bf16
    const float* sh;
    const int* rowptr;
    const int* src_sorted;
    const float* num_neigh;
    float* agg;
    int d_in, w_pad, sh_dim, d_mid, n_nodes;
    float avg_nn;
    int w_bf16;
};

template <int L1, int L2, int L3>
__device__ __forceinline__ void run_path(const Args& a, const PathEntry& pe, int node, int u, bool valid, int beg,
                                         int deg, int maxdeg) {
    constexpr int D1 = 2 * L1 + 1, D2 = 2 * L2 + 1, D3 = 2 * L3 + 1;
    float acc[D3];
#pragma unroll
    for (int k = 0; k < D3; ++k) acc[k] = 0.0f;

    const int wcol = pe.w_off + u;
    const int xcol = pe.x_off + u * D1;
    for (int s = 0; s < maxdeg; ++s) {
        if (s < deg) {
            const int e = beg + s;
            const int src = a.src_sorted[e];
            const float w = matten_ld_edge(a.w_edge, (int64_t)e * a.w_pad + wcol, a.w_bf16);
            const float* xp = a.x + (int64_t)src * a.d_in + xcol;
            const float* yp = a.sh + (int64_t)e * a.sh_dim + L2 * L2;
            float xw[D1], y[D2];
#pragma unroll
            for (int i = 0; i < D1; ++i) xw[i] = xp[i];
#pragma unroll
            for (int j = 0; j < D2; ++j) y[j] = yp[j];
#pragma unroll
            for (int i = 0; i < D1; ++i) xw[i] *= w;
            matten::CG<L1, L2, L3>::apply(xw, y, acc);
        }
    }
    if (valid) {
        const float nn = a.avg_nn > 0.0f ? a.avg_nn : a.num_neigh[node];
        const float norm = 1.0f / sqrtf(nn);
        float* op = a.agg + (int64_t)node * a.d_mid + pe.out_off + u * D3;
#pragma unroll
        for (int k = 0; k < D3; ++k) op[k] = acc[k] * norm;
    }
}

#define MATTEN_CASE(L1, L2, L3) \
    case (L1 * 25 + L2 * 5 + L3): run_path<L1, L2, L3>(a, pe, node, u, valid, beg, deg, maxdeg); break;

__global__ __launch_bounds__(WAVES_PER_BLOCK * 64) void tp_path_kernel(Args a, const PathEntry* __restrict__ entries,
                                                                       const int* __restrict__ ustart, int n_entries,
                                                                       int units_per_tile, int blocks_per_tile,
                                                                       int n_tiles) {
    // XCD-aware decode: all blocks of a tile share blockIdx % 8
    const int xcd = blockIdx.x % N_XCD;
    const int q = blockIdx.x / N_XCD;
    const int tile = (q / blocks_per_tile) * N_XCD + xcd;
    if (tile >= n_tiles) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int unit = (q % blocks_per_tile) * WAVES_PER_BLOCK + wave;
    if (unit >= units_per_tile) return;
    const int lane = threadIdx.x & 63;

    // entry = last e with ustart[e] <= unit (wave-uniform binary search, scalar loads)
    int lo = 0, hi = n_entries;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (ustart[mid] <= unit) lo = mid; else hi = mid;
    }
    const PathEntry pe = entries[lo];
    const int r = unit - ustart[lo];

    const int cu = 1 << pe.cu_log2;
    const int nodes_per_wave = cu >= 64 ? 1 : (64 >> pe.cu_log2);
    const int g = lane >> pe.cu_log2;
    const int u = lane & (cu - 1);
    const int g_in_tile = r * nodes_per_wave + g;
    const int node = tile * TILE_NODES + g_in_tile;
    const bool valid = (g_in_tile < TILE_NODES) && (node < a.n_nodes) && (u < pe.mul);
    int beg = 0, deg = 0;
    if (valid) {
        beg = a.rowptr[node];
        deg = a.rowptr[node + 1] - beg;
    }
    int maxdeg = deg;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) maxdeg = max(maxdeg, __shfl_xor(maxdeg, off));

    switch (pe.triple) {
        MATTEN_CASE(0, 0, 0) MATTEN_CASE(0, 1, 1) MATTEN_CASE(0, 2, 2) MATTEN_CASE(0, 3, 3) MATTEN_CASE(0, 4, 4)
        MATTEN_CASE(1, 0, 1) MATTEN_CASE(1, 1, 0) MATTEN_CASE(1, 1, 1) MATTEN_CASE(1, 1, 2) MATTEN_CASE(1, 2, 1)
        MATTEN_CASE(1, 2, 2) MATTEN_CASE(1, 2, 3) MATTEN_CASE(1, 3, 2) MATTEN_CASE(1, 3, 3) MATTEN_CASE(1, 3, 4)
        MATTEN_CASE(1, 4, 3) MATTEN_CASE(1, 4, 4)
        MATTEN_CASE(2, 0, 2) MATTEN_CASE(2, 1, 1) MATTEN_CASE(2, 1, 2) MATTEN_CASE(2, 1, 3) MATTEN_CASE(2, 2, 0)
        MATTEN_CASE(2, 2, 1) MATTEN_CASE(2, 2, 2) MATTEN_CASE(2, 2, 3) MATTEN_CASE(2, 2, 4) MATTEN_CASE(2, 3, 1)
        MATTEN_CASE(2, 3, 2) MATTEN_CASE(2, 3, 3) MATTEN_CASE(2, 3, 4) MATTEN_CASE(2, 4, 2) MATTEN_CASE(2, 4, 3)
        MATTEN_CASE(2, 4, 4)
        MATTEN_CASE(3, 0, 3) MATTEN_CASE(3, 1, 2) MATTEN_CASE(3, 1, 3) MATTEN_CASE(3, 1, 4) MATTEN_CASE(3, 2, 1)
        MATTEN_CASE(3, 2, 2) MATTEN_CASE(3, 2, 3) MATTEN_CASE(3, 2, 4) MATTEN_CASE(3, 3, 0) MATTEN_CASE(3, 3, 1)
        MATTEN_CASE(3, 3, 2) MATTEN_CASE(3, 3, 3) MATTEN_CASE(3, 3, 4) MATTEN_CASE(3, 4, 1) MATTEN_CASE(3, 4, 2)
        MATTEN_CASE(3, 4, 3) MATTEN_CASE(3, 4, 4)
        MATTEN_CASE(4, 0, 4) MATTEN_CASE(4, 1, 3) MATTEN_CASE(4, 1, 4) MATTEN_CASE(4, 2, 2) MATTEN_CASE(4, 2, 3)
        MATTEN_CASE(4, 2, 4) MATTEN_CASE(4, 3, 1) MATTEN_CASE(4, 3, 2) MATTEN_CASE(4, 3, 3) MATTEN_CASE(4, 3, 4)
        MATTEN_CASE(4, 4, 0) MATTEN_CASE(4, 4, 1) MATTEN_CASE(4, 4, 2) MATTEN_CASE(4, 4, 3) MATTEN_CASE(4, 4, 4)
        default: break;
    }
}

}  // namespace

extern "C" int matten_tp_paths(const float* x, int64_t d_in, const void* w_edge, int64_t w_pad,
                               const float* sh_sorted, int64_t sh_dim, const int32_t* rowptr,
                               const int32_t* src_sorted, int64_t n_nodes, const int32_t* path_entries,
                               const int32_t* unit_start, int64_t n_entries, int64_t units_per_tile, int64_t d_mid,
                               float avg_num_neighbors, const float* num_neigh, float* agg, int w_is_bf16,
                               matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_nodes < 0 || d_in <= 0 || w_pad <= 0 || sh_dim <= 0 || sh_dim > 32 || n_entries <= 0 || units_per_tile <= 0 ||
        d_mid <= 0)
        return MATTEN_EINVAL;
    if (n_nodes == 0) return MATTEN_OK;
    if (!x || !w_edge || !sh_sorted || !rowptr || !src_sorted || !path_entries || !unit_start || !agg)
        return MATTEN_EINVAL;
    if (!(avg_num_neighbors > 0.0f) && !num_neigh) return MATTEN_EINVAL;
    Args a{x, w_edge, sh_sorted, rowptr, src_sorted, num_neigh, agg, (int)d_in, (int)w_pad, (int)sh_dim, (int)d_mid,
           (int)n_nodes, avg_num_neighbors, w_is_bf16};
    const int n_tiles = (int)matten_cdiv(n_nodes, TILE_NODES);
    const int blocks_per_tile = (int)matten_cdiv(units_per_tile, WAVES_PER_BLOCK);
    const int64_t grid = matten_cdiv(n_tiles, N_XCD) * N_XCD * (int64_t)blocks_per_tile;
    if (grid >= ((int64_t)1 << 31)) return MATTEN_EINVAL;
    tp_path_kernel<<<(unsigned)grid, WAVES_PER_BLOCK * 64, 0, stream>>>(a, (const PathEntry*)path_entries, unit_start,
                                                                        (int)n_entries, (int)units_per_tile,
                                                                        blocks_per_tile, n_tiles);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}

extern "C" int matten_tp_tile_nodes(void) { return TILE_NODES; }
