// Edge geometry + real spherical harmonics + Bessel radial basis, and the species embedding.
//   with_edge_vectors            reference nn/_nequip.py:214-268
//   o3.SphericalHarmonics        reference nn/_nequip.py:167-174   (normalize=True, 'component')
//   soft_one_hot_linspace bessel reference nn/embedding.py:185-199 (cutoff=True, * sqrt(num_basis))
//   SpeciesEmbedding             reference nn/embedding.py:85-110, 230-259
#include "common.h"
#include "sh.h"

namespace {

constexpr int SH_TILE_RS = 36;   // floats per staged harmonics row (32 + 4: 16-byte aligned, rows spread over the banks)

template <int LMAX>
__global__ __launch_bounds__(256) void edge_geom_kernel(const float* __restrict__ pos, const int64_t* __restrict__ edge_index,
                                 const float* __restrict__ shift, const float* __restrict__ cell, int64_t n_cells,
                                 const int64_t* __restrict__ batch, const int32_t* __restrict__ perm, int64_t E,
                                 int64_t N, int n_basis, float r_start, float r_end, float4* __restrict__ geom_sorted,
                                 float* __restrict__ sh_sorted, int sh_stride, float* __restrict__ edge_vectors,
                                 float* __restrict__ edge_lengths, float* __restrict__ edge_attrs,
                                 float* __restrict__ edge_embedding) {
    constexpr int SH = (LMAX + 1) * (LMAX + 1);
    __shared__ __attribute__((aligned(16))) float sh_tile[4 * 64 * SH_TILE_RS];
    const int64_t e_raw = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e_raw - (threadIdx.x & 63) >= E) return;   // whole waves past the end
    // lanes past the end of the last wave stay: the harmonics rows are stored by the wave together.  They recompute
    // the last edge and write nothing of their own.
    const bool live = e_raw < E;
    const int64_t e = live ? e_raw : E - 1;
    int64_t o = perm ? (int64_t)perm[e] : e;
    int64_t i = edge_index[o];      // centre ("source")
    int64_t j = edge_index[E + o];  // neighbour ("target")
    // ids outside the batch are flagged by matten_csr_build; here they only must not fault (the host may read the flag
    // after this kernel has run)
    i = i < 0 ? 0 : (i >= N ? N - 1 : i);
    j = j < 0 ? 0 : (j >= N ? N - 1 : j);
    float vx = pos[3 * j + 0] - pos[3 * i + 0];
    float vy = pos[3 * j + 1] - pos[3 * i + 1];
    float vz = pos[3 * j + 2] - pos[3 * i + 2];
    if (cell) {
        int64_t b = (n_cells > 1 && batch) ? batch[i] : 0;
        b = b < 0 ? 0 : (b >= n_cells ? n_cells - 1 : b);
        const float* c = cell + 9 * b;
        float s0 = shift[3 * o + 0], s1 = shift[3 * o + 1], s2 = shift[3 * o + 2];
        // einsum("ni,nij->nj"): rows of the cell are the lattice vectors
        vx += s0 * c[0] + s1 * c[3] + s2 * c[6];
        vy += s0 * c[1] + s1 * c[4] + s2 * c[7];
        vz += s0 * c[2] + s1 * c[5] + s2 * c[8];
    }
    float len = sqrtf(vx * vx + vy * vy + vz * vz);
    if (live) geom_sorted[e] = make_float4(vx, vy, vz, len);

    float y[SH];
    matten::real_sh<LMAX>(vx, vy, vz, len, y);
    // the whole padded row, zeros included: the caller hands over uninitialised memory
    if (sh_stride == 32) {
        // one 128-byte line per edge.  Written through a wave-private LDS tile so that a store instruction covers 1 KB
        // of consecutive memory (8 whole lines) instead of one 16-byte piece of 64 different lines.
        float* tile = sh_tile + (threadIdx.x >> 6) * (64 * SH_TILE_RS);
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            *reinterpret_cast<float4*>(tile + lane * SH_TILE_RS + 4 * q) =
                make_float4(4 * q < SH ? y[4 * q < SH ? 4 * q : 0] : 0.0f,
                            4 * q + 1 < SH ? y[4 * q + 1 < SH ? 4 * q + 1 : 0] : 0.0f,
                            4 * q + 2 < SH ? y[4 * q + 2 < SH ? 4 * q + 2 : 0] : 0.0f,
                            4 * q + 3 < SH ? y[4 * q + 3 < SH ? 4 * q + 3 : 0] : 0.0f);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        const int64_t e0 = e_raw - lane;                // first edge of this wave
        const int64_t n_here = E - e0 < 64 ? E - e0 : 64;
        float4* dst = reinterpret_cast<float4*>(sh_sorted + e0 * 32);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int f = i * 64 + lane;                // 16-byte piece f of the wave's 8 KB: row f / 8, piece f % 8
            if ((f >> 3) < n_here)
                dst[f] = *reinterpret_cast<const float4*>(tile + (f >> 3) * SH_TILE_RS + 4 * (f & 7));
        }
    } else if (live) {
#pragma unroll
        for (int k = 0; k < SH; ++k) sh_sorted[e * sh_stride + k] = y[k];
        for (int k = SH; k < sh_stride; ++k) sh_sorted[e * sh_stride + k] = 0.0f;
    }
    if (!live) return;

    if (edge_vectors) {
        edge_vectors[3 * o + 0] = vx;
        edge_vectors[3 * o + 1] = vy;
        edge_vectors[3 * o + 2] = vz;
    }
    if (edge_lengths) edge_lengths[o] = len;
    if (edge_attrs) {
#pragma unroll
        for (int k = 0; k < SH; ++k) edge_attrs[o * SH + k] = y[k];
    }
    if (edge_embedding) {
        for (int k = 0; k < n_basis; ++k)
            edge_embedding[o * n_basis + k] = matten::bessel_basis(len, k, n_basis, r_start, r_end);
    }
}

__global__ void species_embed_kernel(const int64_t* __restrict__ Z, int64_t N, const int64_t* __restrict__ lut,
                                     int64_t min_z, int64_t max_z, int64_t S, const float* __restrict__ W,
                                     const float* __restrict__ b, int64_t dim, int64_t* __restrict__ species_index,
                                     int32_t* __restrict__ species_i32, float* __restrict__ feats,
                                     float* __restrict__ attrs, int32_t* err_flag) {
    int64_t n = (int64_t)blockIdx.x * blockDim.y + threadIdx.y;
    if (n >= N) return;
    int64_t z = Z[n];
    int64_t idx = -1;
    if (z < min_z || z > max_z) {
        if (threadIdx.x == 0) atomicOr(err_flag, 2);
    } else {
        idx = lut[z - min_z];
        if (idx < 0 || idx >= S) {
            if (threadIdx.x == 0) atomicOr(err_flag, 4);
            idx = -1;
        }
    }
    if (threadIdx.x == 0) {
        if (species_index) species_index[n] = idx;
        if (species_i32) species_i32[n] = (int32_t)(idx < 0 ? 0 : idx);
    }
    int64_t s = idx < 0 ? 0 : idx;
    for (int64_t d = threadIdx.x; d < dim; d += blockDim.x) feats[n * dim + d] = W[d * S + s] + b[d];
    if (attrs)
        for (int64_t c = threadIdx.x; c < S; c += blockDim.x) attrs[n * S + c] = (c == idx) ? 1.0f : 0.0f;
}

}  // namespace

extern "C" int matten_species_embed(const int64_t* atomic_numbers, int64_t n_nodes, const int64_t* z_to_index,
                                    int64_t min_z, int64_t max_z, int64_t n_species, const float* weight,
                                    const float* bias, int64_t dim, int64_t* species_index, int32_t* species_i32,
                                    float* node_feats, float* node_attrs, int32_t* err_flag,
                                    matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_nodes < 0 || dim <= 0 || n_species <= 0) return MATTEN_EINVAL;
    if (n_nodes == 0) return MATTEN_OK;
    if (!atomic_numbers || !z_to_index || !weight || !bias || !node_feats || !err_flag) return MATTEN_EINVAL;
    dim3 block(16, 16);
    species_embed_kernel<<<(unsigned)matten_cdiv(n_nodes, 16), block, 0, stream>>>(
        atomic_numbers, n_nodes, z_to_index, min_z, max_z, n_species, weight, bias, dim, species_index, species_i32,
        node_feats, node_attrs, err_flag);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}

extern "C" int matten_edge_geom(const float* pos, const int64_t* edge_index, const float* edge_cell_shift,
                                const float* cell, int64_t n_cells, const int64_t* batch, const int32_t* perm,
                                int64_t n_edges, int64_t n_nodes, int lmax, int n_basis, float r_start, float r_end,
                                float* geom_sorted, float* sh_sorted, int sh_stride, float* edge_vectors, float* edge_lengths,
                                float* edge_attrs, float* edge_embedding, matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_edges < 0 || n_nodes < 0 || (n_edges > 0 && n_nodes == 0) || lmax < 0 || lmax > 4 || n_basis < 0 || sh_stride < (lmax + 1) * (lmax + 1)) return MATTEN_EINVAL;
    if (n_edges == 0) return MATTEN_OK;
    if (!pos || !edge_index || !geom_sorted || !sh_sorted) return MATTEN_EINVAL;
    if (cell && !edge_cell_shift) return MATTEN_EINVAL;
    const int T = 256;
    unsigned grid = (unsigned)matten_cdiv(n_edges, T);
#define LAUNCH(L)                                                                                              \
    edge_geom_kernel<L><<<grid, T, 0, stream>>>(pos, edge_index, edge_cell_shift, cell, n_cells, batch, perm,  \
                                                n_edges, n_nodes, n_basis, r_start, r_end, (float4*)geom_sorted, \
                                                sh_sorted, sh_stride, edge_vectors, edge_lengths, edge_attrs, edge_embedding)
    switch (lmax) {
        case 0: LAUNCH(0); break;
        case 1: LAUNCH(1); break;
        case 2: LAUNCH(2); break;
        case 3: LAUNCH(3); break;
        default: LAUNCH(4); break;
    }
#undef LAUNCH
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}
