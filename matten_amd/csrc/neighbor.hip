// Periodic neighbour list on the GPU, canonical order.
// Contract of the reference's graph builder (data/data.py:285-413, which calls ASE): every ordered triple
// (i, j, S) with | r_j + S.cell - r_i | < r_cut (strict, fp64), minus the true self edge (i == j, S == 0);
// edge_index[0] = i (centre), edge_index[1] = j, edge_cell_shift = S.  Edges are emitted in the
// lexicographic order (i, j, Sx, Sy, Sz) so the result is deterministic and identical to the host builder.
//
// matten_graph_prep (one wave per crystal) turns the four arrays that cross PCIe into what the search needs: inverse
// cell in closed form, fractional coordinates, the per-axis bound r_cut |inv[:, k]| and the fp32 / index copies the
// model takes.  One thread per ordered atom PAIR (i, j) of a crystal then walks only the image shifts that can hold a
// neighbour: d . inv[:, k] = S_k + (f_j - f_i)_k and |d . inv[:, k]| <= |d| |inv[:, k]| < r_cut |inv[:, k]|, so
// S_k lies in [-bound_k - df_k, bound_k - df_k] (widened by 1e-6 relative: the bound only prunes, the exact test
// decides) -- ~2 values per axis instead of the 2 reach + 1 of a per-crystal box (5^3 = 125 images per pair for an
// fcc-64 cell, 8 now).  The shifts are walked in lexicographic order twice: a counting pass, then (after an exclusive
// scan of the per-pair counts by the caller) a fill pass that writes the pair's edges contiguously.  Pairs are
// numbered crystal by crystal, i-major, so the scan order IS the canonical edge order.
// The destination-sorted CSR the conv layers walk (destination = edge_index[1] = j, stable in the edge id: what
// matten_csr_build derives from the finished list in seven launches) falls out of the same two passes: the counting
// pass also stores each pair's count at its TRANSPOSED (j-major) pair number, the caller scans those too, and the fill
// pass drops the pair's edge ids / centre atoms at that second offset -- for a fixed j the pairs come i-ascending and a
// pair's edges in shift order, i.e. in ascending edge id.  rowptr[j] is the j-major offset of pair (first atom, j).
// Distances use the host builder's fp64 expression with contraction disabled; the square root is only evaluated for the pairs within 1e-15
// (relative) of the cutoff, which leaves the decision bit-identical.
#include "common.h"

#pragma clang fp contract(off)

namespace {

struct Cry {
    const double* pos;        // [N,3] all crystals concatenated
    const double* cell;       // [B,9] rows = lattice vectors
    const int64_t* ptr;       // [B+1] first atom of each crystal
    const double* frac;       // [N,3] fractional coordinates (matten_graph_prep)
    const double* bound;      // [B,3] r_cut |inv[:, k]|
    const int64_t* pair_ptr;  // [B+1] first pair of each crystal (sum of n^2)
};

__global__ __launch_bounds__(64) void graph_prep_kernel(const double* __restrict__ pos, const double* __restrict__ cell,
                                                        const int64_t* __restrict__ ptr, double r_cut,
                                                        double* __restrict__ frac, double* __restrict__ bound,
                                                        int64_t* __restrict__ batch, float* __restrict__ pos_f32,
                                                        float* __restrict__ cell_f32) {
    const int64_t b = blockIdx.x;
    const double* cl = cell + 9 * b;
    const double ax = cl[0], ay = cl[1], az = cl[2], bx = cl[3], by = cl[4], bz = cl[5], cx = cl[6], cy = cl[7], cz = cl[8];
    // inv = [b x c, c x a, a x b] (as columns) / det
    const double c0x = by * cz - bz * cy, c0y = bz * cx - bx * cz, c0z = bx * cy - by * cx;
    const double c1x = cy * az - cz * ay, c1y = cz * ax - cx * az, c1z = cx * ay - cy * ax;
    const double c2x = ay * bz - az * by, c2y = az * bx - ax * bz, c2z = ax * by - ay * bx;
    const double det = (ax * c0x + ay * c0y) + az * c0z;
    const double i00 = c0x / det, i10 = c0y / det, i20 = c0z / det;   // column 0
    const double i01 = c1x / det, i11 = c1y / det, i21 = c1z / det;
    const double i02 = c2x / det, i12 = c2y / det, i22 = c2z / det;
    if (threadIdx.x < 9) cell_f32[9 * b + threadIdx.x] = (float)cl[threadIdx.x];
    if (threadIdx.x == 0) {
        bound[3 * b] = r_cut * sqrt((i00 * i00 + i10 * i10) + i20 * i20);
        bound[3 * b + 1] = r_cut * sqrt((i01 * i01 + i11 * i11) + i21 * i21);
        bound[3 * b + 2] = r_cut * sqrt((i02 * i02 + i12 * i12) + i22 * i22);
    }
    const int64_t lo = ptr[b], hi = ptr[b + 1];
    for (int64_t n = lo + threadIdx.x; n < hi; n += blockDim.x) {
        const double x = pos[3 * n], y = pos[3 * n + 1], z = pos[3 * n + 2];
        frac[3 * n] = (x * i00 + y * i10) + z * i20;
        frac[3 * n + 1] = (x * i01 + y * i11) + z * i21;
        frac[3 * n + 2] = (x * i02 + y * i12) + z * i22;
        pos_f32[3 * n] = (float)x, pos_f32[3 * n + 1] = (float)y, pos_f32[3 * n + 2] = (float)z;
        batch[n] = b;
    }
}

// image shifts along one axis that can hold a neighbour of the pair: [lo, hi] (empty when lo > hi)
__device__ __forceinline__ void shift_range(double bound, double df, int& lo, int& hi) {
    const double slack = 1e-6 * (1.0 + fabs(df) + bound);
    lo = (int)ceil(-bound - df - slack);
    hi = (int)floor(bound - df + slack);
}

struct CsrOut {                // optional outputs of the fill pass: the destination-sorted view (all or none)
    const int64_t* offsets_t;  // [n_pairs + 1] scan of the j-major counts; offsets_t[0] (any constant) is subtracted
    int32_t* rowptr;           // [N + 1]
    int32_t* src_sorted;       // [E] centre atom i of the edge at each sorted position
    int32_t* perm;             // [E] sorted position -> edge id
    int64_t n_atoms;
};

template <bool FILL>
__global__ __launch_bounds__(256) void neighbor_kernel(Cry c, double r_cut, int32_t* __restrict__ counts,
                                                       int32_t* __restrict__ counts_t,
                                                       const int64_t* __restrict__ offsets,
                                                       int64_t* __restrict__ edge_index, int64_t n_edges,
                                                       float* __restrict__ shifts, float* __restrict__ num_neigh,
                                                       CsrOut csr) {
    const int64_t b = blockIdx.y;
    const int64_t lo = c.ptr[b];
    const int64_t n = c.ptr[b + 1] - lo;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * n) return;
    const int64_t i = lo + t / n, j = lo + t % n;
    const int64_t pair = c.pair_ptr[b] + t;
    const double* cl = c.cell + 9 * b;
    int x0, x1, y0, y1, z0, z1;
    shift_range(c.bound[3 * b], c.frac[3 * j] - c.frac[3 * i], x0, x1);
    shift_range(c.bound[3 * b + 1], c.frac[3 * j + 1] - c.frac[3 * i + 1], y0, y1);
    shift_range(c.bound[3 * b + 2], c.frac[3 * j + 2] - c.frac[3 * i + 2], z0, z1);
    const double pix = c.pos[3 * i], piy = c.pos[3 * i + 1], piz = c.pos[3 * i + 2];
    const double pjx = c.pos[3 * j], pjy = c.pos[3 * j + 1], pjz = c.pos[3 * j + 2];
    const double r2 = r_cut * r_cut;
    const double r2_in = r2 * (1.0 - 1e-15), r2_out = r2 * (1.0 + 1e-15);
    int64_t out = FILL ? offsets[pair] : 0;
    if (FILL && num_neigh && j == lo) num_neigh[i] = (float)(offsets[pair + n] - offsets[pair]);   // atom i's n pairs
    const int64_t pair_t = c.pair_ptr[b] + (j - lo) * n + (i - lo);   // the same pair numbered j-major
    int64_t out_t = 0;
    if (FILL && csr.rowptr) {
        out_t = csr.offsets_t[pair_t] - csr.offsets_t[0];   // (a scan that continues the i-major one starts at n_edges)
        if (i == lo) csr.rowptr[j] = (int32_t)out_t;
        if (pair == 0) csr.rowptr[csr.n_atoms] = (int32_t)n_edges;
    }
    int cnt = 0;
    for (int sx = x0; sx <= x1; ++sx)
        for (int sy = y0; sy <= y1; ++sy) {
            // T = S @ cell : ((sx*c0 + sy*c1) + sz*c2) per component, as the host builder's matmul
            const double ax = (double)sx * cl[0] + (double)sy * cl[3];
            const double ay = (double)sx * cl[1] + (double)sy * cl[4];
            const double az = (double)sx * cl[2] + (double)sy * cl[5];
            for (int sz = z0; sz <= z1; ++sz) {
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
                        if (csr.rowptr) {
                            csr.perm[out_t] = (int32_t)out;
                            csr.src_sorted[out_t] = (int32_t)i;
                            ++out_t;
                        }
                        ++out;
                    }
                    ++cnt;
                }
            }
        }
    if (!FILL) {
        counts[pair] = cnt;
        if (counts_t) counts_t[pair_t] = cnt;
    }
}

// {number of edges, smallest edge count of a crystal} from the scanned pair counts: the one read-back of the builder
__global__ __launch_bounds__(256) void neighbor_summary_kernel(const int64_t* __restrict__ offsets,
                                                               const int64_t* __restrict__ pair_ptr, int64_t n_crystals,
                                                               int64_t* __restrict__ out) {
    __shared__ long long red[256];
    long long m = LLONG_MAX;
    for (int64_t b = threadIdx.x; b < n_crystals; b += blockDim.x)
        m = min(m, (long long)(offsets[pair_ptr[b + 1]] - offsets[pair_ptr[b]]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] = min(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = offsets[pair_ptr[n_crystals]];
        out[1] = n_crystals > 0 ? (int64_t)red[0] : 0;
    }
}

}  // namespace

extern "C" int matten_graph_prep(const double* pos, const double* cell, const int64_t* ptr, int64_t n_crystals,
                                 double r_cut, double* frac, double* bound, int64_t* batch, float* pos_f32,
                                 float* cell_f32, matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_crystals < 0 || !(r_cut > 0.0) || n_crystals >= ((int64_t)1 << 31)) return MATTEN_EINVAL;
    if (n_crystals == 0) return MATTEN_OK;
    if (!pos || !cell || !ptr || !frac || !bound || !batch || !pos_f32 || !cell_f32) return MATTEN_EINVAL;
    graph_prep_kernel<<<(unsigned)n_crystals, 64, 0, stream>>>(pos, cell, ptr, r_cut, frac, bound, batch, pos_f32, cell_f32);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}

extern "C" int matten_neighbor_count(const double* pos, const double* cell, const int64_t* ptr, const double* frac,
                                     const double* bound, const int64_t* pair_ptr, double r_cut, int64_t n_crystals,
                                     int64_t max_atoms, int32_t* counts, int32_t* counts_t, matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_crystals < 0 || max_atoms < 0 || !(r_cut > 0.0) || n_crystals > 65535) return MATTEN_EINVAL;
    if (n_crystals == 0 || max_atoms == 0) return MATTEN_OK;
    if (!pos || !cell || !ptr || !frac || !bound || !pair_ptr || !counts) return MATTEN_EINVAL;
    Cry c{pos, cell, ptr, frac, bound, pair_ptr};
    dim3 grid((unsigned)matten_cdiv(max_atoms * max_atoms, 256), (unsigned)n_crystals);
    neighbor_kernel<false><<<grid, 256, 0, stream>>>(c, r_cut, counts, counts_t, nullptr, nullptr, 0, nullptr, nullptr, CsrOut{});
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}

extern "C" int matten_neighbor_summary(const int64_t* offsets, const int64_t* pair_ptr, int64_t n_crystals,
                                       int64_t* out2, matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_crystals < 0 || !offsets || !pair_ptr || !out2) return MATTEN_EINVAL;
    neighbor_summary_kernel<<<1, 256, 0, stream>>>(offsets, pair_ptr, n_crystals, out2);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}

extern "C" int matten_neighbor_fill(const double* pos, const double* cell, const int64_t* ptr, const double* frac,
                                    const double* bound, const int64_t* pair_ptr, double r_cut, int64_t n_crystals,
                                    int64_t max_atoms, const int64_t* offsets, int64_t n_edges, int64_t* edge_index,
                                    float* edge_cell_shift, float* num_neigh, const int64_t* offsets_t, int64_t n_atoms,
                                    int32_t* rowptr, int32_t* src_sorted, int32_t* perm, matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_crystals < 0 || max_atoms < 0 || n_edges < 0 || !(r_cut > 0.0) || n_crystals > 65535) return MATTEN_EINVAL;
    if (n_crystals == 0 || max_atoms == 0) return MATTEN_OK;
    if (!pos || !cell || !ptr || !frac || !bound || !pair_ptr || !offsets) return MATTEN_EINVAL;
    if (n_edges > 0 && (!edge_index || !edge_cell_shift)) return MATTEN_EINVAL;
    const bool want_csr = offsets_t || rowptr || src_sorted || perm;
    if (want_csr && (!offsets_t || !rowptr || n_atoms < 0 || n_edges >= ((int64_t)1 << 31) ||
                     (n_edges > 0 && (!src_sorted || !perm))))
        return MATTEN_EINVAL;
    Cry c{pos, cell, ptr, frac, bound, pair_ptr};
    CsrOut csr{offsets_t, want_csr ? rowptr : nullptr, src_sorted, perm, n_atoms};
    dim3 grid((unsigned)matten_cdiv(max_atoms * max_atoms, 256), (unsigned)n_crystals);
    neighbor_kernel<true><<<grid, 256, 0, stream>>>(c, r_cut, nullptr, nullptr, offsets, edge_index, n_edges, edge_cell_shift,
                                                    num_neigh, csr);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}
