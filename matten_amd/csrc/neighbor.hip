// Periodic neighbour list on the GPU, canonical order.
// Contract of the reference's graph builder (data/data.py:285-413, which calls ASE): every ordered triple
// (i, j, S) with | r_j + S.cell - r_i | < r_cut (strict, fp64), minus the true self edge (i == j, S == 0);
// edge_index[0] = i (centre), edge_index[1] = j, edge_cell_shift = S.  Edges are emitted in the
// lexicographic order (i, j, Sx, Sy, Sz) so the result is deterministic and identical to the host builder.
//
// One thread per ordered atom PAIR (i, j) of a crystal walks the image shifts S in lexicographic order twice: a
// counting pass, then (after an exclusive scan of the per-pair counts by the caller) a fill pass that writes the
// pair's edges contiguously.  Pairs are numbered crystal by crystal, i-major, so the scan order IS the canonical
// edge order.  Distances use the host builder's fp64 expression with contraction disabled; the square root is
// only evaluated for the pairs within 1e-15 (relative) of the cutoff, which leaves the decision bit-identical.
#include "common.h"

#pragma clang fp contract(off)

namespace {

struct Cry {
    const double* pos;        // [N,3] all crystals concatenated
    const double* cell;       // [B,9] rows = lattice vectors
    const int64_t* ptr;       // [B+1] first atom of each crystal
    const int32_t* reach;     // [B,3] images needed along each lattice direction
    const int64_t* pair_ptr;  // [B+1] first pair of each crystal (sum of n^2)
};

template <bool FILL>
__global__ __launch_bounds__(256) void neighbor_kernel(Cry c, double r_cut, int32_t* __restrict__ counts,
                                                       const int64_t* __restrict__ offsets,
                                                       int64_t* __restrict__ edge_index, int64_t n_edges,
                                                       float* __restrict__ shifts) {
    const int64_t b = blockIdx.y;
    const int64_t lo = c.ptr[b];
    const int64_t n = c.ptr[b + 1] - lo;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * n) return;
    const int64_t i = lo + t / n, j = lo + t % n;
    const int64_t pair = c.pair_ptr[b] + t;
    const double* cl = c.cell + 9 * b;
    const int rx = c.reach[3 * b], ry = c.reach[3 * b + 1], rz = c.reach[3 * b + 2];
    const double pix = c.pos[3 * i], piy = c.pos[3 * i + 1], piz = c.pos[3 * i + 2];
    const double pjx = c.pos[3 * j], pjy = c.pos[3 * j + 1], pjz = c.pos[3 * j + 2];
    const double r2 = r_cut * r_cut;
    const double r2_in = r2 * (1.0 - 1e-15), r2_out = r2 * (1.0 + 1e-15);
    int64_t out = FILL ? offsets[pair] : 0;
    int cnt = 0;
    for (int sx = -rx; sx <= rx; ++sx)
        for (int sy = -ry; sy <= ry; ++sy) {
            // T = S @ cell : ((sx*c0 + sy*c1) + sz*c2) per component, as the host builder's matmul
            const double ax = (double)sx * cl[0] + (double)sy * cl[3];
            const double ay = (double)sx * cl[1] + (double)sy * cl[4];
            const double az = (double)sx * cl[2] + (double)sy * cl[5];
            for (int sz = -rz; sz <= rz; ++sz) {
                const double tx = ax + (double)sz * cl[6];
                const double ty = ay + (double)sz * cl[7];
                const double tz = az + (double)sz * cl[8];
                const double dx = (pjx + tx) - pix, dy = (pjy + ty) - piy, dz = (pjz + tz) - piz;
                const double d2 = (dx * dx + dy * dy) + dz * dz;
                bool hit = d2 < r2_in;
                if (!hit && d2 <= r2_out) hit = sqrt(d2) < r_cut;  // the reference's test, needed only at the boundary
                if (hit && !(i == j && sx == 0 && sy == 0 && sz == 0)) {
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
    if (!FILL) counts[pair] = cnt;
}

}  // namespace

extern "C" int matten_neighbor_count(const double* pos, const double* cell, const int64_t* ptr, const int32_t* reach,
                                     const int64_t* pair_ptr, double r_cut, int64_t n_crystals, int64_t max_atoms,
                                     int32_t* counts, matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_crystals < 0 || max_atoms < 0 || !(r_cut > 0.0) || n_crystals > 65535) return MATTEN_EINVAL;
    if (n_crystals == 0 || max_atoms == 0) return MATTEN_OK;
    if (!pos || !cell || !ptr || !reach || !pair_ptr || !counts) return MATTEN_EINVAL;
    Cry c{pos, cell, ptr, reach, pair_ptr};
    dim3 grid((unsigned)matten_cdiv(max_atoms * max_atoms, 256), (unsigned)n_crystals);
    neighbor_kernel<false><<<grid, 256, 0, stream>>>(c, r_cut, counts, nullptr, nullptr, 0, nullptr);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}

extern "C" int matten_neighbor_fill(const double* pos, const double* cell, const int64_t* ptr, const int32_t* reach,
                                    const int64_t* pair_ptr, double r_cut, int64_t n_crystals, int64_t max_atoms,
                                    const int64_t* offsets, int64_t n_edges, int64_t* edge_index,
                                    float* edge_cell_shift, matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_crystals < 0 || max_atoms < 0 || n_edges < 0 || !(r_cut > 0.0) || n_crystals > 65535) return MATTEN_EINVAL;
    if (n_crystals == 0 || max_atoms == 0 || n_edges == 0) return MATTEN_OK;
    if (!pos || !cell || !ptr || !reach || !pair_ptr || !offsets || !edge_index || !edge_cell_shift)
        return MATTEN_EINVAL;
    Cry c{pos, cell, ptr, reach, pair_ptr};
    dim3 grid((unsigned)matten_cdiv(max_atoms * max_atoms, 256), (unsigned)n_crystals);
    neighbor_kernel<true><<<grid, 256, 0, stream>>>(c, r_cut, nullptr, offsets, edge_index, n_edges, edge_cell_shift);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}
