// 'uvu' Clebsch-Gordan tensor product + gather + neighbour sum, v3: one wave per
// (input block, l2 group, node group); all couplings that read the same input block are fused.
// (reference nn/utils.py:230-237,263 + nn/conv.py:113-120)
//
// A lane owns ONE channel u of ONE input irrep block (l1,p1) for ONE destination node and walks the
// node's CSR segment (sequential segmented reduction: fixed order, no atomics, no cross-lane traffic).
// Per edge it loads, once,
//      x[src, x_off + u*d1 .. +d1)       its channel of the source node's features
//      Y_l2(e), l2 = lo..hi              the edge harmonics of its l2 group
//      w[e, w_base + u*NC .. +NC)        the radial weights of all NC couplings of the group: the MLP
//                                        writes its output columns in this [entry][u][c] order, so a
//                                        lane's weights are contiguous (vector loads) and the lanes of
//                                        a node read one contiguous mul*NC run
// and feeds every coupling (l1,l2,l3) of the group from those registers, so the gathers are
// amortised over up to 12 couplings instead of being repeated per path: ~60 cache-line touches per
// edge and layer instead of ~400, which is what bounded the per-path kernel (texture-address path,
// not VALU).  CG coefficients are compile-time literals (cg_gen.h).  Accumulators: <= 52 per lane.
//
// Grid: node tiles of TILE_NODES pinned to one XCD (blockIdx % 8) so the tile's weight rows are
// pulled from HBM once and reused from that XCD's L2 by all groups.
#include "cg_gen.h"
#include "common.h"

namespace {

constexpr int TILE_NODES = 64;
constexpr int WAVES_PER_BLOCK = 4;
constexpr int N_XCD = 8;
constexpr int MAXC = matten::GROUP_MAX_COMBOS;

struct GroupEntry {  // 32 x int32, built by matten_amd/plan.py
    int kind;        // l1*GROUP_KIND_STRIDE + group index
    int x_off;       // offset of channel 0 of this entry in the node feature row
    int mul;         // channels in this entry (<= 64)
    int cu_log2;     // lanes per node = 1 << cu_log2 >= mul
    unsigned mask;   // bit c set <=> coupling c of the group exists in this layer
    int w_base;      // offset of this entry's [u][c] weight block in the per-edge weight row
    int pad[2];
    int reserved[MAXC];
    int out_off[MAXC];  // offset of channel 0's output of coupling c in the message row
};
static_assert(sizeof(GroupEntry) == 32 * 4, "GroupEntry layout");

struct Args {
    const float* x;
    const float* w_edge;
    const float* sh;
    const int* rowptr;
    const int* src_sorted;
    const float* num_neigh;
    float* agg;
    int d_in, w_pad, sh_stride, d_mid, n_nodes;
    float avg_nn;
};

template <int L1, int GI>
__device__ __forceinline__ void run_group(const Args& a, const GroupEntry& ge, int node, int u, bool valid, int beg,
                                          int deg, int maxdeg) {
    using G = matten::Group<L1, GI>;
    float acc[G::NACC];
#pragma unroll
    for (int k = 0; k < G::NACC; ++k) acc[k] = 0.0f;

    const unsigned mask = ge.mask;
    const int xcol = ge.x_off + u * G::D1;
    for (int s = 0; s < maxdeg; ++s) {
        if (s < deg) {
            const int e = beg + s;
            const int src = a.src_sorted[e];
            const float* xp = a.x + (int64_t)src * a.d_in + xcol;
            const float* yp = a.sh + (int64_t)e * a.sh_stride + G::Y0;
            const float* wp = a.w_edge + (int64_t)e * a.w_pad + ge.w_base + u * G::NC;
            float x[G::D1], y[G::NY], w[G::NC];
#pragma unroll
            for (int i = 0; i < G::D1; ++i) x[i] = xp[i];
#pragma unroll
            for (int j = 0; j < G::NY; ++j) y[j] = yp[j];
#pragma unroll
            for (int c = 0; c < G::NC; ++c) w[c] = wp[c];
            G::apply(mask, x, y, w, acc);
        }
    }
    if (valid) {
        const float nn = a.avg_nn > 0.0f ? a.avg_nn : a.num_neigh[node];
        const float norm = 1.0f / sqrtf(nn);
        float* orow = a.agg + (int64_t)node * a.d_mid;
#pragma unroll
        for (int c = 0; c < G::NC; ++c) {
            if ((mask >> c) & 1u) {
                const int d3 = 2 * G::L3[c] + 1;
                float* op = orow + ge.out_off[c] + u * d3;
#pragma unroll
                for (int k = 0; k < 2 * matten::CG_LMAX + 1; ++k)
                    if (k < d3) op[k] = acc[G::OFF[c] + k] * norm;
            }
        }
    }
}

#define MATTEN_GROUP_CASE(L1, GI) \
    case (L1 * matten::GROUP_KIND_STRIDE + GI): run_group<L1, GI>(a, ge, node, u, valid, beg, deg, maxdeg); break;

__global__ __launch_bounds__(WAVES_PER_BLOCK * 64) void tp_block_kernel(Args a, const GroupEntry* __restrict__ entries,
                                                                        const int* __restrict__ ustart,
                                                                        int n_entries, int units_per_tile,
                                                                        int blocks_per_tile, int n_tiles) {
    const int xcd = blockIdx.x % N_XCD;
    const int q = blockIdx.x / N_XCD;
    const int tile = (q / blocks_per_tile) * N_XCD + xcd;
    if (tile >= n_tiles) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int unit = (q % blocks_per_tile) * WAVES_PER_BLOCK + wave;
    if (unit >= units_per_tile) return;
    const int lane = threadIdx.x & 63;

    int lo = 0, hi = n_entries;  // last entry with ustart[entry] <= unit (wave-uniform)
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (ustart[mid] <= unit) lo = mid; else hi = mid;
    }
    const GroupEntry& ge = entries[lo];
    const int r = unit - ustart[lo];

    const int cu_log2 = ge.cu_log2;
    const int cu = 1 << cu_log2;
    const int nodes_per_wave = cu >= 64 ? 1 : (64 >> cu_log2);
    const int g = lane >> cu_log2;
    const int u = lane & (cu - 1);
    const int g_in_tile = r * nodes_per_wave + g;
    const int node = tile * TILE_NODES + g_in_tile;
    const bool valid = (g_in_tile < TILE_NODES) && (node < a.n_nodes) && (u < ge.mul);
    int beg = 0, deg = 0;
    if (valid) {
        beg = a.rowptr[node];
        deg = a.rowptr[node + 1] - beg;
    }
    int maxdeg = deg;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) maxdeg = max(maxdeg, __shfl_xor(maxdeg, off));

    switch (ge.kind) {
        MATTEN_FOR_EACH_GROUP(MATTEN_GROUP_CASE)
        default: break;
    }
}

}  // namespace

extern "C" int matten_tp_blocks(const float* x, int64_t d_in, const float* w_edge, int64_t w_pad,
                                const float* sh_sorted, int64_t sh_stride, const int32_t* rowptr,
                                const int32_t* src_sorted, int64_t n_nodes, const int32_t* group_entries,
                                const int32_t* unit_start, int64_t n_entries, int64_t units_per_tile, int64_t d_mid,
                                float avg_num_neighbors, const float* num_neigh, float* agg,
                                matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_nodes < 0 || d_in <= 0 || w_pad <= 0 || sh_stride < 25 || n_entries <= 0 || units_per_tile <= 0 ||
        d_mid <= 0)
        return MATTEN_EINVAL;
    if (n_nodes == 0) return MATTEN_OK;
    if (!x || !w_edge || !sh_sorted || !rowptr || !src_sorted || !group_entries || !unit_start || !agg)
        return MATTEN_EINVAL;
    if (!(avg_num_neighbors > 0.0f) && !num_neigh) return MATTEN_EINVAL;
    Args a{x, w_edge, sh_sorted, rowptr, src_sorted, num_neigh, agg, (int)d_in, (int)w_pad, (int)sh_stride,
           (int)d_mid, (int)n_nodes, avg_num_neighbors};
    const int n_tiles = (int)matten_cdiv(n_nodes, TILE_NODES);
    const int blocks_per_tile = (int)matten_cdiv(units_per_tile, WAVES_PER_BLOCK);
    const int64_t grid = matten_cdiv(n_tiles, N_XCD) * N_XCD * (int64_t)blocks_per_tile;
    if (grid >= ((int64_t)1 << 31)) return MATTEN_EINVAL;
    tp_block_kernel<<<(unsigned)grid, WAVES_PER_BLOCK * 64, 0, stream>>>(
        a, (const GroupEntry*)group_entries, unit_start, (int)n_entries, (int)units_per_tile, blocks_per_tile, n_tiles);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}
