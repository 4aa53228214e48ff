// Periodic neighbour list on the GPU, canonical order.
// Contract of the reference's graph builder (data/data.py:285-413, which calls ASE): every ordered triple
// (i, j, S) with | r_j + S.cell - r_i | < r_cut (strict, fp64), minus the true self edge (i == j, S == 0);
// edge_index[0] = i (centre), edge_index[1] = j, edge_cell_shift = S.  Edges are emitted in the
// lexicographic order (i, j, Sx, Sy, Sz) so the result is deterministic and identical to the host builder.
//
// One thread per centre atom walks (j, S) in that order twice: a counting pass, then (after an exclusive scan
// of the counts by the caller) a fill pass that writes its edges contiguously.  Distances use the same fp64
// expression as the host code, with contraction disabled, so the edge set is bit-identical.
#include "common.h"

#pragma clang fp contract(off)

namespace {

struct Cry {
    const double* pos;      // [N,3] all crystals concatenated
    const double* cell;     // [B,9] rows = lattice vectors
    const int64_t* ptr;     // [B+1]
    const int32_t* reach;   // [B,3] images needed along each lattice direction
    const int64_t* batch;   // [N]
};

template <bool FILL>
__global__ void neighbor_kernel(Cry c, double r_cut, int64_t n_nodes, int32_t* __restrict__ counts,
                                const int64_t* __restrict__ offsets, int64_t* __restrict__ edge_index, int64_t n_edges,
                                float* __restrict__ shifts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const int64_t b = c.batch[i];
    const int64_t lo = c.ptr[b], hi = c.ptr[b + 1];
    const double* cl = c.cell + 9 * b;
    const int rx = c.reach[3 * b], ry = c.reach[3 * b + 1], rz = c.reach[3 * b + 2];
    const double pix = c.pos[3 * i], piy = c.pos[3 * i + 1], piz = c.pos[3 * i + 2];
    int64_t out = FILL ? offsets[i] : 0;
    int cnt = 0;
    for (int64_t j = lo; j < hi; ++j) {
        const double pjx = c.pos[3 * j], pjy = c.pos[3 * j + 1], pjz = c.pos[3 * j + 2];
        for (int sx = -rx; sx <= rx; ++sx)
            for (int sy = -ry; sy <= ry; ++sy)
                for (int sz = -rz; sz <= rz; ++sz) {
                    if (i == j && sx == 0 && sy == 0 && sz == 0) continue;
                    // T = S @ cell : ((sx*c0 + sy*c1) + sz*c2) per component
                    const double tx = ((double)sx * cl[0] + (double)sy * cl[3]) + (double)sz * cl[6];
                    const double ty = ((double)sx * cl[1] + (double)sy * cl[4]) + (double)sz * cl[7];
                    const double tz = ((double)sx * cl[2] + (double)sy * cl[5]) + (double)sz * cl[8];
                    const double dx = (pjx + tx) - pix, dy = (pjy + ty) - piy, dz = (pjz + tz) - piz;
                    const double d2 = (dx * dx + dy * dy) + dz * dz;
                    if (sqrt(d2) < r_cut) {
                        if (FILL) {
                            edge_index[out] = i;
                            edge_index[n_edges + out] = j;
                            shifts[3 * out] = (float)sx;
                            shifts[3 * out + 1] = (float)sy;
                            shifts[3 * out + 2] = (float)sz;
                            ++out;
                        }
                        ++cnt;
                    }
                }
    }
    if (!FILL) counts[i] = cnt;
}

}  // namespace

extern "C" int matten_neighbor_count(const double* pos, const double* cell, const int64_t* ptr, const int32_t* reach,
                                     const int64_t* batch, double r_cut, int64_t n_nodes, int32_t* counts,
                                     matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_nodes < 0 || !(r_cut > 0.0)) return MATTEN_EINVAL;
    if (n_nodes == 0) return MATTEN_OK;
    if (!pos || !cell || !ptr || !reach || !batch || !counts) return MATTEN_EINVAL;
    Cry c{pos, cell, ptr, reach, batch};
    neighbor_kernel<false><<<(unsigned)matten_cdiv(n_nodes, 64), 64, 0, stream>>>(c, r_cut, n_nodes, counts, nullptr,
                                                                                 nullptr, 0, nullptr);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}

extern "C" int matten_neighbor_fill(const double* pos, const double* cell, const int64_t* ptr, const int32_t* reach,
                                    const int64_t* batch, double r_cut, int64_t n_nodes, const int64_t* offsets,
                                    int64_t n_edges, int64_t* edge_index, float* edge_cell_shift,
                                    matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_nodes < 0 || n_edges < 0 || !(r_cut > 0.0)) return MATTEN_EINVAL;
    if (n_nodes == 0 || n_edges == 0) return MATTEN_OK;
    if (!pos || !cell || !ptr || !reach || !batch || !offsets || !edge_index || !edge_cell_shift) return MATTEN_EINVAL;
    Cry c{pos, cell, ptr, reach, batch};
    neighbor_kernel<true><<<(unsigned)matten_cdiv(n_nodes, 64), 64, 0, stream>>>(c, r_cut, n_nodes, nullptr, offsets,
                                                                                edge_index, n_edges, edge_cell_shift);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}
