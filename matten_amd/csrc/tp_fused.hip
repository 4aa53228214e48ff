// Fused radial-weight GEMM + 'uvu' Clebsch-Gordan tensor product + gather + neighbour sum (v4).
// (reference nn/utils.py:246-251,260,263 + nn/conv.py:113-120)
//
// The per-edge radial weights w[E, W] are the one operand of the conv layer that does not fit on
// chip (3.9 GB per layer at 1000 crystals).  This kernel never materialises them: the last layer of
// the radial MLP is evaluated on the matrix cores right where its output is consumed,
//
//      w[e, col] = sum_k h2[e, k] * W2p[k, col]          (h2 = 32 hidden features per edge, 128 B)
//
// Arithmetic of that GEMM.  On gfx950 the fp32 matrix instruction (v_mfma_f32_16x16x4_f32, 33 cycles) runs at the
// fp32 VALU rate and does NOT overlap with fp32 VALU work of any wave of the SIMD (tools/ubench/mfma_chain.hip,
// mfma_valu_coissue.hip: the times add), so every MFMA cycle is taken from the contraction.  The fp16 one
// (v_mfma_f32_16x16x32_f16, 17 cycles for 8x the K) is 16x faster per flop, so both operands are split into two
// fp16 pieces, v = hi + 2^-11 lo with hi = fp16(v), lo = fp16(2^11 (v - hi)): 22+ significant bits, i.e. the fp32
// value to within 2^-24 relative.  Three products (hi hi, hi lo, lo hi; lo lo is below 2^-24) accumulate in fp32:
// the result carries the rounding of an fp32 dot product, at 51 instead of 264 matrix cycles per 16x16 tile.
// fp16 range: the A tile is scaled by a power of two per wave (folded into the output normalisation; the
// contraction is linear in w).  h2 (silu outputs) are kept inside the fp16 range the same way: the hidden-layer kernels
// multiply them by a per-MLP power of two h_scale <= 1 that the host derives from a bound on |h2| (the weights' column
// sums; 1 for any normally scaled MLP) and folds back into a_scale_inv -- a checkpoint with huge radial weights stays exact.
//
// A wave owns one (input block, l2 group, node group) unit exactly like tp_block_kernel (a lane = one
// channel u of one destination node, walking the node's CSR segment).  It proceeds in chunks of
// CH edge slots:
//   MFMA phase  v_mfma_f32_16x16x4_f32, edges of the chunk as the N dimension, the unit's weight
//               columns [u][coupling] as M, K = 32.  A = W2p tile (L2-resident, 121 KB per layer),
//               B = h2 rows.  The D fragment (4 consecutive columns of one edge) goes to a wave-private
//               LDS tile with one 16-byte store.
//   VALU phase  each lane reads its NC weights of its node's edge from LDS and contracts them with
//               x[src] and Y(e) through the literal-coefficient CG code (cg_gen.h).
// Matrix-core and vector work of different waves overlap on the SIMD; HBM traffic per edge drops from
// ~7.7 KB (write + read of w) to ~0.3 KB (h2 + harmonics + indices).
//
// h2s layout: [E, 2, 32] fp16 (128 B per edge): piece 0 = hi, piece 1 = lo; column g*8 + kk of a piece  <->  hidden
// feature pi(kk,g) = 16 (kk>>2) + 4 g + (kk&3): exactly the registers lane group g of the hidden-layer kernel
// holds, and the 8 K-slots lane group g feeds to the MFMA (A rows follow the same pi).
#include <type_traits>

#include "cg_gen.h"
#include "common.h"
#include "sh.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr float SPLIT_LO_SCALE = 2048.0f;          // 2^11
constexpr float SPLIT_LO_INV = 1.0f / 2048.0f;
constexpr float F16_MIN_NORMAL = 6.103515625e-05f;  // 2^-14

// v ~= hi + 2^-11 lo.  fp16 subnormals are zeroed in software (hi: the residual then moves into lo), so the result
// does not depend on whether the matrix unit flushes them.
__device__ __forceinline__ void split_f16(float v, _Float16& hi, _Float16& lo) {
    const float h = fabsf(v) < F16_MIN_NORMAL ? 0.0f : (float)(_Float16)v;
    const float r = (v - h) * SPLIT_LO_SCALE;
    hi = (_Float16)h;
    lo = fabsf(r) < F16_MIN_NORMAL ? (_Float16)0.0f : (_Float16)r;
}

#ifdef MATTEN_ABLATE_NO_GATHER   // timing experiment: every gather reads the destination node's own (cache-resident) row
#define TPF_SRC(v) ((v) >= 0 ? (node < a.n_nodes ? node : 0) : 0)
#else
#define TPF_SRC(v) (v)
#endif
#ifdef MATTEN_ABLATE_NO_XLOAD    // timing experiment: no neighbour-row load at all (the value depends on the index only)
#define TPF_XLD(xp, i, v) (1e-9f * (float)((v) + (i)))
#else
#define TPF_XLD(xp, i, v) ((xp)[i])
#endif
#ifndef TPF_SETPRIO
#define TPF_SETPRIO 3
#endif
// weight columns ([u][coupling]) an entry may have: the plan caps an entry at the largest power-of-two channel count whose
// block fits (plan.py TP_MAX_COLS, checked at load through matten_tp_max_cols); its MFMA A operand stays in registers
#ifndef TPF_MAX_COLS
#define TPF_MAX_COLS 64
#endif
#ifndef TPF_MAX_COLS_L0
#define TPF_MAX_COLS_L0 96   // scalar input blocks (l1 = 0): the lightest kind has registers for a wider entry (16 channels)
#endif
#ifndef TPF_MAX_COLS_L1
#define TPF_MAX_COLS_L1 TPF_MAX_COLS   // vector (l1 = 1) input blocks; 7 couplings with l2 <= 2 (112 columns for 16 channels:
#endif                                 // measured, spills), 5 with l2 = 3, 4 (80 columns)
__host__ __device__ constexpr int cap_channels(int l1, int nc) {
    int cap = 64;
    while (cap > 1 && cap * nc > (l1 == 0 ? TPF_MAX_COLS_L0 : l1 == 1 ? TPF_MAX_COLS_L1 : TPF_MAX_COLS)) cap /= 2;
    return cap;
}
constexpr int TILE_NODES = 64;
constexpr int WAVES_PER_BLOCK = 4;
constexpr int N_XCD = 8;
constexpr int MAXC = matten::GROUP_MAX_COMBOS;
constexpr int HID = 32;

struct GroupEntry {  // 32 x int32, built by matten_amd/plan.py (same record as tp_block.hip)
    int kind;        // l1*GROUP_KIND_STRIDE + group index
    int x_off;       // offset of channel 0 of this entry in the node feature row
    int mul;         // channels in this entry
    int cu_log2;     // lanes per node = 1 << cu_log2 >= mul
    unsigned mask;   // bit c set <=> coupling c of the group exists in this layer
    int w_base;      // first weight column of this entry ([u][c] order)
    int a_tile;      // first 16-column tile of this entry in the pre-split A operand (Args::a_split)
    int n_mt;        // its tile count, ceil(mul * couplings / 16)
    int t_off[MAXC];  // conv-fused kernel (StoreLds): accumulator offset of coupling c in the wave's LDS region;
                      // StoreAgg: 0 = mul_ir output row, else floats between two components (component-major row)
    int out_off[MAXC];
};
static_assert(sizeof(GroupEntry) == 32 * 4, "GroupEntry layout");

struct Args {
    const float* x;
    const _Float16* h2s;  // [E, 2, 32] split hidden features (see header)
    const float* w2p;   // [32, w_pad] last MLP layer, pre-scaled, fused column order
    const _Float16* a_split;   // optional: the same weights as ready-made MFMA A fragments (see matten_hip.h), or NULL
    const float* a_scale_inv;  // [n_entries] with a_split: 1 / the power-of-two scale of the entry's fragments
    const float* sh;
    const int* rowptr;
    const int* src_sorted;
    const float* num_neigh;
    float* agg;
    int d_in, w_pad, sh_stride, d_mid, n_nodes, lds_per_wave;
    float avg_nn;
};

template <int L1, int GI>
__device__ __forceinline__ void run_group(const Args& a, const GroupEntry& ge, float* __restrict__ tile, int node,
                                          int lane, bool valid, int beg, int deg_node, int maxdeg) {
    const int deg = valid ? deg_node : 0;  // edges this lane contracts (idle channel lanes: none)
    using G = matten::Group<L1, GI>;
    constexpr int NC = G::NC;
    float acc[G::NACC];
#pragma unroll
    for (int k = 0; k < G::NACC; ++k) acc[k] = 0.0f;

    const unsigned mask = ge.mask;
    const int cu_log2 = ge.cu_log2;
    const int cu = 1 << cu_log2;
    const int npw = 64 >> cu_log2;               // nodes per wave
    const int ch_log2 = npw >= 16 ? 0 : (4 - (6 - cu_log2));  // CH = max(1, 16 / npw)
    const int CH = 1 << ch_log2;
    const int T = npw >= 16 ? (npw >> 4) : 1;     // N tiles of 16 edges per chunk
    const int ncols = ge.mul * NC;
    const int MT = (ncols + 15) >> 4;
    const int ycol = MT * 16;                    // harmonics of the edge live behind its weight columns
    const int stride = MT * 16 + 32 + 4;         // floats per edge row of the LDS tile (+4: bank spread)

    const int j = lane >> cu_log2;               // consumer role: node slot in the wave, channel
    const int u = lane & (cu - 1);
    const int g = lane >> 4, c = lane & 15;      // MFMA role
    const int xcol = ge.x_off + u * G::D1;
    // A operand (the entry's weight columns, <= 64 by construction of the plan) stays in registers for the
    // whole CSR walk: av[mt][kk] = W2p[pi(kk,g)][w_base + 16*mt + c]
    // MTMAX: the plan caps an entry at the largest power-of-two channel count whose [u][c] block fits 64 columns
    constexpr int CAPC = cap_channels(L1, NC);
    constexpr int MTMAX = (CAPC * NC + 15) / 16;
    f16x8 ah[MTMAX], al[MTMAX];
    float a_scale_inv;
    {
        float av[MTMAX][8];
        float amax = 0.0f;
#pragma unroll
        for (int mt = 0; mt < MTMAX; ++mt) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int k = 16 * (kk >> 2) + 4 * g + (kk & 3);
                av[mt][kk] = (mt < MT) ? a.w2p[(int64_t)k * a.w_pad + ge.w_base + mt * 16 + c] : 0.0f;
                amax = fmaxf(amax, fabsf(av[mt][kk]));
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
        // power-of-two scale that puts the tile's largest magnitude in [2^13, 2^14)
        int e = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 127;
        e = amax > 0.0f ? max(-100, min(100, e)) : 13;
        const float a_scale = __uint_as_float((unsigned)(127 + 13 - e) << 23);
        a_scale_inv = __uint_as_float((unsigned)(127 - 13 + e) << 23);
#pragma unroll
        for (int mt = 0; mt < MTMAX; ++mt) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                _Float16 hi, lo;
                split_f16(av[mt][kk] * a_scale, hi, lo);
                ah[mt][kk] = hi;
                al[mt][kk] = lo;
            }
        }
    }

    // Software pipeline of the neighbour gather: the source index runs two edges ahead and the feature row
    // x[src] one edge ahead of the contraction, across chunk boundaries (the loads are unconditional on a clamped
    // edge index, so nothing but the data dependence orders them against the MFMA phase).
    const int e_last = deg > 0 ? beg + deg - 1 : 0;
    float xn[G::D1];
    int src_nn;
    {
        const int src0 = a.src_sorted[min(beg, e_last)];
        src_nn = a.src_sorted[min(beg + 1, e_last)];
        const float* xp0 = a.x + (int64_t)src0 * a.d_in + xcol;
#pragma unroll
        for (int i = 0; i < G::D1; ++i) xn[i] = xp0[i];
    }
    for (int s0 = 0; s0 < maxdeg; s0 += CH) {
        // ---- MFMA: w[edge n, col] = h2[edge n, :] . W2p[:, col] into the wave's LDS tile (edge n = jn*CH + so) ----
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t < T) {
                const int n = 16 * t + c;
                const int jn = n >> ch_log2, so = n & (CH - 1);
                const int begn = __shfl(beg, jn << cu_log2);
                const int degn = __shfl(deg_node, jn << cu_log2);
                f16x8 bh = {0, 0, 0, 0, 0, 0, 0, 0}, bl = {0, 0, 0, 0, 0, 0, 0, 0};
#ifdef MATTEN_ABLATE_NO_H2LOAD
                bh = f16x8{1, 2, 3, 4, 1, 2, 3, 4};
                bl = bh;
                if (false) {
#else
                if (s0 + so < degn) {
#endif
                    const int64_t en = begn + s0 + so;
                    const f16x8* hp = reinterpret_cast<const f16x8*>(a.h2s + en * (2 * HID) + g * 8);
                    bh = hp[0];
                    bl = hp[HID / 8];
                    // stage the edge's harmonics once per edge (4 lanes x 32 B) instead of once per channel lane
                    const f32x4* yp4 = reinterpret_cast<const f32x4*>(a.sh + en * a.sh_stride + g * 8);
                    f32x4* yd = reinterpret_cast<f32x4*>(tile + (16 * t + c) * stride + ycol + g * 8);
                    yd[0] = yp4[0];
                    yd[1] = yp4[1];
                }
#ifndef MATTEN_ABLATE_NO_MFMA
#pragma unroll
                for (int mt = 0; mt < MTMAX; ++mt) {
                    if (mt < MT) {
                        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                        f32x4 dx = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mt], bh, zero, 0, 0, 0);
                        f32x4 dh = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bh, zero, 0, 0, 0);
                        dx = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bl, dx, 0, 0, 0);
                        *reinterpret_cast<f32x4*>(tile + (16 * t + c) * stride + mt * 16 + 4 * g) =
                            dh + SPLIT_LO_INV * dx;
                    }
                }
#else
                if ((float)bh[0] == 12345.f) tile[c] = (float)bl[0];
#endif
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the tile is written
        __builtin_amdgcn_wave_barrier();
        // ---- VALU: contract this lane's channel for its node's edges of the chunk ----
#ifndef MATTEN_ABLATE_NO_VALU
        for (int so = 0; so < CH; ++so) {
            const int s = s0 + so;
            if (s >= maxdeg) break;  // wave-uniform: no lane has an edge in the remaining slots
            float x[G::D1];
#pragma unroll
            for (int i = 0; i < G::D1; ++i) x[i] = xn[i];
            {   // issue edge s+1's row and edge s+2's index
                const float* xp = a.x + (int64_t)src_nn * a.d_in + xcol;
#pragma unroll
                for (int i = 0; i < G::D1; ++i) xn[i] = xp[i];
                src_nn = a.src_sorted[min(beg + s + 2, e_last)];
            }
            if (s < deg) {
                const float* wp = tile + ((j << ch_log2) + so) * stride + u * NC;
                const float* yp = tile + ((j << ch_log2) + so) * stride + ycol + G::Y0;
                float y[G::NY], w[NC];
#pragma unroll
                for (int jj = 0; jj < G::NY; ++jj) y[jj] = yp[jj];
#pragma unroll
                for (int cc = 0; cc < NC; ++cc) w[cc] = wp[cc];
                G::apply(mask, x, y, w, acc);
            }
        }
#endif
        __builtin_amdgcn_wave_barrier();  // LDS is in order per wave: the next chunk's stores follow these reads
    }
#ifdef MATTEN_ABLATE_NO_STORE
    if (valid && acc[0] == 12345.678f) {
#else
    if (valid) {
#endif
        const float nn = a.avg_nn > 0.0f ? a.avg_nn : a.num_neigh[node];
        const float norm = a_scale_inv / sqrtf(nn);  // undoes the power-of-two scale of the A tile
        float* orow = a.agg + (int64_t)node * a.d_mid;
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) {
            if ((mask >> cc) & 1u) {
                const int d3 = 2 * G::L3[cc] + 1;
                float* op = orow + ge.out_off[cc] + u * d3;
#pragma unroll
                for (int k = 0; k < 2 * matten::CG_LMAX + 1; ++k)
                    if (k < d3) op[k] = acc[G::OFF[cc] + k] * norm;
            }
        }
    }
}


// ---- workgroup-shared staging (units flagged by the host, plan.fused_unit_map) --------------------------------------
// When the four waves of a workgroup contract four entries of the SAME destination nodes (equal lanes per node, one or
// two 16-edge MFMA tiles per chunk) they need the same hidden-feature and harmonics rows.  Each of the 256 threads then
// fetches ONE 16-byte piece per MFMA tile of the chunk's edge rows (hi 64 B | lo 64 B | harmonics 128 B), one chunk
// ahead of its use (4-8 registers in flight instead of 16-32 per wave), and publishes it in a double-buffered LDS
// stage: a quarter of the vector-memory requests per wave, their latency behind a whole chunk of work, and the
// harmonics are no longer copied into every wave's private tile.  One workgroup barrier per chunk.  Workgroups with
// fewer than four entries for their nodes are filled up by the host with loader-only units (run_loader_only).
constexpr int STAGE_ROW = 68;               // floats per staged edge row: 16 hi | 16 lo | 32 harmonics | 4 pad (banks)
constexpr int STAGE_TMAX = 2;               // MFMA tiles (16 edge rows each) per chunk a shared workgroup may have
constexpr int STAGE_FLOATS = 16 * STAGE_TMAX * STAGE_ROW;
constexpr int STAGE_TOTAL_FLOATS = 2 * STAGE_FLOATS;

// w tile rows [edge][col]: D fragments of MTC column tiles, hi.hi + 2^-11 (lo.hi + hi.lo), 16-byte LDS stores
template <int MTC>
__device__ __forceinline__ void mfma_tiles(const f16x8* __restrict__ ah, const f16x8* __restrict__ al, f16x8 bh, f16x8 bl,
                                           float* __restrict__ trow) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 dx[MTC], dh[MTC];
#pragma unroll
    for (int mt = 0; mt < MTC; ++mt) {
        dx[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mt], bh, zero, 0, 0, 0);
        dh[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bh, zero, 0, 0, 0);
    }
#pragma unroll
    for (int mt = 0; mt < MTC; ++mt) dx[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bl, dx[mt], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < MTC; ++mt) *reinterpret_cast<f32x4*>(trow + mt * 16) = dh[mt] + SPLIT_LO_INV * dx[mt];
}

// loader role of a thread: (edge row of the chunk, 16-byte piece of the row) per MFMA tile; fixed for the whole walk
template <int TT>  // MFMA tiles (16 edge rows) per chunk: a compile-time count keeps every load of the loop unconditional
struct StageLoader {
    const char* base;
    const char* safe;
    int64_t row_bytes;
    float* st_w;
    int beg_ld[TT], deg_ld[TT], so_ld[TT];
    int CH;
    f32x4 pf[TT];

    __device__ __forceinline__ void init(const Args& a, float* stage, int cu_log2, int beg, int deg_node) {
        const int npw = 64 >> cu_log2;
        const int ch_log2 = npw >= 16 ? 0 : 4 - (6 - cu_log2);
        CH = 1 << ch_log2;
        const int piece = threadIdx.x & 15;
        base = piece < 8 ? reinterpret_cast<const char*>(a.h2s) + piece * 16
                         : reinterpret_cast<const char*>(a.sh) + (piece - 8) * 16;
        row_bytes = piece < 8 ? (int64_t)(2 * HID * sizeof(_Float16)) : (int64_t)a.sh_stride * 4;
        safe = reinterpret_cast<const char*>(a.w2p);
        st_w = stage + (threadIdx.x >> 4) * STAGE_ROW + piece * 4;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const int n = (int)(threadIdx.x >> 4) + 16 * t;
            const int jn = (n >> ch_log2) & (npw - 1);
            so_ld[t] = n & (CH - 1);
            beg_ld[t] = __shfl(beg, jn << cu_log2);
            deg_ld[t] = __shfl(deg_node, jn << cu_log2);
            pf[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // fetch the pieces of the chunk starting at slot s0 (past the end of a segment: its last edge again, never consumed).
    // The loads are unconditional (rows of an empty segment read the weight table instead): with a branch around a
    // load the compiler cannot count the loads in flight and falls back to s_waitcnt vmcnt(0) at the next gather.
    __device__ __forceinline__ void issue(int s0) {
#ifndef MATTEN_ABLATE_NO_H2LOAD
#pragma unroll
        for (int t = 0; t < TT; ++t)
            pf[t] = *reinterpret_cast<const f32x4*>(
                deg_ld[t] > 0 ? base + (int64_t)(beg_ld[t] + min(s0 + so_ld[t], deg_ld[t] - 1)) * row_bytes : safe);
#endif
    }
    __device__ __forceinline__ void publish(int buf) {
#pragma unroll
        for (int t = 0; t < TT; ++t)
            *reinterpret_cast<f32x4*>(st_w + buf * (16 * TT * STAGE_ROW) + 16 * t * STAGE_ROW) = pf[t];
    }
};

// PAIRED workgroups: a class of entries that leaves only TWO for a workgroup would idle half of its waves as
// loader-only units.  Instead waves 0, 1 take the two entries on node group r and waves 2, 3 the same two entries on node
// group r + 1; the stage then holds 32 rows (rows 0-15: the chunk of group r, 16-31: of group r + 1) and every thread
// fetches one piece of each half.  The two halves' CSR segments are exchanged through LDS once per unit.
constexpr int PAIR_INFO_INTS = 2 * 16 * 2 + 2;   // [group][node][beg, deg] + [group] max degree
struct PairLoader {
    const char* base;
    const char* safe;
    int64_t row_bytes;
    float* st_w;
    int beg_ld[2], deg_ld[2], so_ld;
    int CH, maxdeg;
    f32x4 pf[2];

    __device__ __forceinline__ void init(const Args& a, float* stage, int* info, int cu_log2, int beg, int deg_node,
                                         int my_maxdeg) {
        const int npw = 64 >> cu_log2;                     // <= 16 here
        const int ch_log2 = 4 - (6 - cu_log2);
        CH = 1 << ch_log2;
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        if ((wave & 1) == 0 && (lane & ((1 << cu_log2) - 1)) == 0) {
            info[((wave >> 1) * 16 + (lane >> cu_log2)) * 2] = beg;
            info[((wave >> 1) * 16 + (lane >> cu_log2)) * 2 + 1] = deg_node;
            if (lane == 0) info[2 * 16 * 2 + (wave >> 1)] = my_maxdeg;
        }
        __syncthreads();
        maxdeg = max(info[2 * 16 * 2], info[2 * 16 * 2 + 1]);
        const int piece = threadIdx.x & 15;
        base = piece < 8 ? reinterpret_cast<const char*>(a.h2s) + piece * 16
                         : reinterpret_cast<const char*>(a.sh) + (piece - 8) * 16;
        row_bytes = piece < 8 ? (int64_t)(2 * HID * sizeof(_Float16)) : (int64_t)a.sh_stride * 4;
        safe = reinterpret_cast<const char*>(a.w2p);
        const int n = (int)(threadIdx.x >> 4);
        st_w = stage + n * STAGE_ROW + piece * 4;
        const int jn = (n >> ch_log2) & (npw - 1);
        so_ld = n & (CH - 1);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            beg_ld[t] = info[(t * 16 + jn) * 2];
            deg_ld[t] = info[(t * 16 + jn) * 2 + 1];
            pf[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    __device__ __forceinline__ void issue(int s0) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
            pf[t] = *reinterpret_cast<const f32x4*>(
                deg_ld[t] > 0 ? base + (int64_t)(beg_ld[t] + min(s0 + so_ld, deg_ld[t] - 1)) * row_bytes : safe);
    }
    __device__ __forceinline__ void publish(int buf) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
            *reinterpret_cast<f32x4*>(st_w + buf * (32 * STAGE_ROW) + 16 * t * STAGE_ROW) = pf[t];
    }
};

// loader-only unit of a paired workgroup
__device__ __forceinline__ void run_loader_only_paired(const Args& a, int cu_log2, float* __restrict__ stage, int beg,
                                                       int deg_node, int maxdeg) {
    PairLoader ld;
    ld.init(a, stage, reinterpret_cast<int*>(stage + STAGE_TOTAL_FLOATS), cu_log2, beg, deg_node, maxdeg);
    ld.issue(0);
    ld.publish(0);
    __syncthreads();
    int buf = 0;
    for (int s0 = 0; s0 < ld.maxdeg; s0 += ld.CH, buf ^= 1) {
        ld.issue(s0 + ld.CH);
        ld.publish(buf ^ 1);
        __syncthreads();
    }
}

// a unit that only feeds the stage (same barrier sequence as run_group_shared)
template <int TT>
__device__ __forceinline__ void run_loader_only_t(const Args& a, int cu_log2, float* __restrict__ stage, int beg,
                                                  int deg_node, int maxdeg) {
    StageLoader<TT> ld;
    ld.init(a, stage, cu_log2, beg, deg_node);
    ld.issue(0);
    ld.publish(0);
    __syncthreads();
    int buf = 0;
    for (int s0 = 0; s0 < maxdeg; s0 += ld.CH, buf ^= 1) {
        ld.issue(s0 + ld.CH);
        ld.publish(buf ^ 1);
        __syncthreads();
    }
}
__device__ __forceinline__ void run_loader_only(const Args& a, int cu_log2, float* __restrict__ stage, int beg, int deg_node,
                                                int maxdeg) {
    if ((64 >> cu_log2) > 16) run_loader_only_t<2>(a, cu_log2, stage, beg, deg_node, maxdeg);
    else run_loader_only_t<1>(a, cu_log2, stage, beg, deg_node, maxdeg);
}

// ---- what a unit does with its neighbour sums ------------------------------------------------------------------------
// StoreAgg: one row slice of agg[N, d_mid] per (node, channel)  (the two-kernel conv: lin2 reads agg afterwards)
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int STORE_PASS = 32;   // accumulators per pass of the vector epilogue (32 x 64 lanes = 2048 floats of the wave's tile)

struct StoreAgg {
    float* t;   // the wave's LDS tile (free once the walk is over), or nullptr: scalar stores only
    // Experiment (-DTPF_VECTOR_EPILOGUE, off): the epilogue as the walk leaves it is NACC (25-52) four-byte stores per
    // lane; here four neighbouring channel lanes exchange their accumulators through the LDS tile ([accumulator][lane])
    // and each writes one accumulator of a group for four channels with a 16-byte store (8-byte for two-channel entries).
    // Parity-green and SLOWER: last layer 1.38 vs 1.28 ms with the component-major row, 1.31 with an entry-major row
    // (one contiguous 128-256-byte block per node and instruction).  The stores cost 0.20 of that layer's 1.25 ms
    // (-DMATTEN_ABLATE_NO_STORE), but neither by instruction count nor by coalescing: an ablation that issued 16-byte
    // stores into a fifth of the address range (overlapping rows) took 1.08 ms -- it is the 1.1 GB (1.6 GB at the L2
    // boundary) of distinct bytes per launch that costs, in a kernel whose waves wait on memory 40 % of the time.
    template <class G, int V>
    __device__ __forceinline__ void store_vec(const Args& a, const GroupEntry& ge, const float* __restrict__ acc,
                                              float norm, int node, bool node_ok) const {
        typedef float vec_t __attribute__((ext_vector_type(V)));
        const int lane = threadIdx.x & 63;
        // Which piece a lane writes: the V accumulators of a group x the node's cu channels are cu lanes x V floats.
        // Lane lu of the node takes accumulator lu / (cu / V) of the group and channels V (lu % (cu / V)) .. + V - 1, so that
        // when the group's accumulators are adjacent in memory (entry-major rows) consecutive lanes write consecutive
        // 16-byte pieces: one contiguous 4 cu-float block per node and instruction, which the memory pipeline takes as
        // whole lines (lanes whose addresses interleave are not merged).
        const int cu = 1 << ge.cu_log2, cq = cu / V;
        const int lu = lane & (cu - 1), nbase = lane & ~(cu - 1);
        const int jq = lu / cq, u0 = V * (lu - jq * cq), lbase = nbase + u0;
        float* orow = a.agg + (int64_t)node * a.d_mid + u0;
#pragma unroll
        for (int p0 = 0; p0 < G::NACC; p0 += STORE_PASS) {
            __builtin_amdgcn_wave_barrier();                     // the previous pass's reads are done
#pragma unroll
            for (int i = 0; i < STORE_PASS; ++i)
                if (p0 + i < G::NACC) t[i * 64 + lane] = acc[p0 + i] * norm;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int g0 = 0; g0 < STORE_PASS; g0 += V) {
                if (p0 + g0 < G::NACC) {
                    int off = 0;
                    bool on = false;
#pragma unroll
                    for (int i = 0; i < V; ++i) {
                        const int ai = p0 + g0 + i;              // compile-time after unrolling: so are cc and k below
                        if (ai < G::NACC) {
                            int cc = 0;
#pragma unroll
                            for (int c2 = 1; c2 < G::NC; ++c2)
                                if (ai >= G::OFF[c2]) cc = c2;
                            const int k = ai - G::OFF[cc];
                            if (jq == i) {
                                off = ge.out_off[cc] + k * ge.t_off[cc];
                                on = (ge.mask >> cc) & 1u;
                            }
                        }
                    }
                    const vec_t v = *reinterpret_cast<const vec_t*>(t + (g0 + jq) * 64 + lbase);
#ifdef MATTEN_ABLATE_NO_STORE
                    if (node_ok && on && v[0] == 12345.678f)
#else
                    if (node_ok && on)
#endif
                        *reinterpret_cast<vec_t*>(orow + off) = v;
                }
            }
        }
    }

    template <class G>
    __device__ __forceinline__ void store(const Args& a, const GroupEntry& ge, const float* __restrict__ acc,
                                          float a_scale_inv, int node, int j, int u, bool valid) const {
#ifdef TPF_VECTOR_EPILOGUE
        const int cu = 1 << ge.cu_log2;   // measured slower with the component-major row (see store_vec): off by default
        // wave-uniform: component-major row (the stride of the entry's first coupling says so), every channel lane of a
        // node in use, room for a pass in the tile
        if (t != nullptr && ge.t_off[__builtin_ctz(ge.mask | 0x80000000u) % MAXC] != 0 && ge.mul == cu && cu >= 2 &&
            a.lds_per_wave >= 64 * STORE_PASS) {
            float norm = 0.0f;   // mul == cu: `valid` is a per-node condition here (lanes of nodes past the end store nothing)
            if (valid) norm = a_scale_inv / sqrtf(a.avg_nn > 0.0f ? a.avg_nn : a.num_neigh[node]);
            if (cu >= 4) store_vec<G, 4>(a, ge, acc, norm, node, valid);
            else store_vec<G, 2>(a, ge, acc, norm, node, valid);
            return;
        }
#endif
#ifdef MATTEN_ABLATE_NO_STORE
        if (valid && acc[0] == 12345.678f) {
#else
        if (valid) {
#endif
            const float nn = a.avg_nn > 0.0f ? a.avg_nn : a.num_neigh[node];
            const float norm = a_scale_inv / sqrtf(nn);
            float* orow = a.agg + (int64_t)node * a.d_mid;
#pragma unroll
            for (int cc = 0; cc < G::NC; ++cc) {
                if ((ge.mask >> cc) & 1u) {
                    const int d3 = 2 * G::L3[cc] + 1;
                    // t_off[cc] == 0: the reference's "mul_ir" row, [channel][component].  Otherwise the component-major
                    // row of plan.plan_agg_linear: t_off[cc] floats between components, the channel lanes side by side
                    const int ks = ge.t_off[cc];
                    float* op = orow + ge.out_off[cc] + (ks ? u : u * d3);
                    const int kstep = ks ? ks : 1;
#pragma unroll
                    for (int k = 0; k < 2 * matten::CG_LMAX + 1; ++k)
                        if (k < d3) {
#ifdef TPF_NT_STORES   // agg is written once and read once by another kernel: non-temporal hint (experiment: slower)
                            __builtin_nontemporal_store(acc[G::OFF[cc] + k] * norm, op + k * kstep);
#else
                            op[k * kstep] = acc[G::OFF[cc] + k] * norm;
#endif
                        }
                }
            }
        }
    }
};
// StoreLds: the wave's LDS region, [coupling][node j, component k][channel u (8)] -- the operand layout of the lin2
// stage of tp_lin2_kernel (8 lanes per node).  Every lane writes (idle channels and nodes past the end hold zeros).
struct StoreLds {
    float* t;
    template <class G>
    __device__ __forceinline__ void store(const Args& a, const GroupEntry& ge, const float* __restrict__ acc,
                                          float a_scale_inv, int node, int j, int u, bool valid) const {
        float norm = 0.0f;
        if (valid) norm = a_scale_inv / sqrtf(a.avg_nn > 0.0f ? a.avg_nn : a.num_neigh[node]);
#pragma unroll
        for (int cc = 0; cc < G::NC; ++cc) {
            if ((ge.mask >> cc) & 1u) {
                const int d3 = 2 * G::L3[cc] + 1;
                float* tp = t + 64 * ge.t_off[cc] + j * d3 * 8 + u;
#pragma unroll
                for (int k = 0; k < 2 * matten::CG_LMAX + 1; ++k)
                    if (k < d3) tp[k * 8] = acc[G::OFF[cc] + k] * norm;
            }
        }
    }
};

// the kinds with registers to spare for a second neighbour row in flight (two-slot chunks: 8 lanes per node)
template <int L1, int GI>
struct TwoDeepOk { static constexpr bool value = L1 == 0 || (L1 == 1 && GI == 0); };

// TWO_DEEP is a template parameter, not a run-time flag: with both gather schedules in one instantiation the compiler
// reconciled their register assignments with 50-90 v_mov per CHUNK (60 % of the l1 = 0 kind's vector instructions).
// CMASK != 0: the entry's coupling mask as a compile-time constant (HotMask below); the couplings of a step then form
// one basic block instead of NC uniformly-branched ones
template <int L1, int GI, int TT, bool TWO_DEEP, bool PAIRED, unsigned CMASK, class Epilogue>
__device__ __forceinline__ void run_group_shared(const Args& a, const GroupEntry& ge, float* __restrict__ tile,
                                                 float* __restrict__ stage, int entry, int node, int lane, bool valid,
                                                 int beg, int deg_node, int maxdeg, const Epilogue& epi) {
    static_assert(!PAIRED || TT == 1, "paired workgroups stage 2 x 16 rows");
    const int row0 = PAIRED ? 16 * (int)(threadIdx.x >> 7) : 0;       // this wave's half of a paired stage
    constexpr int STAGE_BUF = (PAIRED ? 32 : 16 * TT) * STAGE_ROW;    // floats per stage buffer
    const int deg = valid ? deg_node : 0;
    using G = matten::Group<L1, GI>;
    constexpr int NC = G::NC;
    float acc[G::NACC];
#pragma unroll
    for (int k = 0; k < G::NACC; ++k) acc[k] = 0.0f;

    const unsigned mask = ge.mask;
    const int cu_log2 = ge.cu_log2;              // >= 1 here: at most 32 nodes per wave
    const int cu = 1 << cu_log2;
    const int npw = 64 >> cu_log2;
    const int ch_log2 = npw >= 16 ? 0 : 4 - (6 - cu_log2);
    const int CH = 1 << ch_log2;
    const int ncols = ge.mul * NC;
    const int MT = (ncols + 15) >> 4;

    const int j = lane >> cu_log2;
    const int u = lane & (cu - 1);
    const int g = lane >> 4, c = lane & 15;
    const int xcol = ge.x_off + u * G::D1;
    constexpr int CAPC = cap_channels(L1, NC);
    constexpr int MTMAX = (CAPC * NC + 15) / 16;
    const int stride = MT * 16 + 4;              // floats per edge row of the wave's weight tile
    f16x8 ah[MTMAX], al[MTMAX];
    float a_scale_inv;
    if (a.a_split) {
        // ready-made fragments from the host (two 16-byte loads per tile instead of 8 scattered loads and ~110
        // conversion instructions per tile and wave: a quarter of a light wave's vector instructions)
        const f16x8* ap = reinterpret_cast<const f16x8*>(a.a_split) + ((int64_t)ge.a_tile * 64 + lane) * 2;
#pragma unroll
        for (int mt = 0; mt < MTMAX; ++mt) {
            const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            ah[mt] = mt < MT ? ap[mt * 128] : z;
            al[mt] = mt < MT ? ap[mt * 128 + 1] : z;
        }
        a_scale_inv = a.a_scale_inv[entry];
    } else
    {
        float av[MTMAX][8];
        float amax = 0.0f;
#pragma unroll
        for (int mt = 0; mt < MTMAX; ++mt) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int k = 16 * (kk >> 2) + 4 * g + (kk & 3);
                av[mt][kk] = (mt < MT) ? a.w2p[(int64_t)k * a.w_pad + ge.w_base + mt * 16 + c] : 0.0f;
                amax = fmaxf(amax, fabsf(av[mt][kk]));
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
        int e = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 127;
        e = amax > 0.0f ? max(-100, min(100, e)) : 13;
        const float a_scale = __uint_as_float((unsigned)(127 + 13 - e) << 23);
        a_scale_inv = __uint_as_float((unsigned)(127 - 13 + e) << 23);
#pragma unroll
        for (int mt = 0; mt < MTMAX; ++mt) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                _Float16 hi, lo;
                split_f16(av[mt][kk] * a_scale, hi, lo);
                ah[mt][kk] = hi;
                al[mt][kk] = lo;
            }
        }
    }

    typename std::conditional<PAIRED, PairLoader, StageLoader<TT>>::type ld;
    if constexpr (PAIRED) {
        ld.init(a, stage, reinterpret_cast<int*>(stage + STAGE_TOTAL_FLOATS), cu_log2, beg, deg_node, maxdeg);
        maxdeg = ld.maxdeg;   // both node groups walk the same number of chunks (one barrier sequence)
    } else {
        ld.init(a, stage, cu_log2, beg, deg_node);
    }
    ld.issue(0);

    // Neighbour gather.  Two-slot chunks (8 nodes per wave: short steps) keep TWO rows in flight, one per slot of the
    // chunk: a row is refilled right after its contraction with the edge two steps on, so a gather has a whole step,
    // the stage hand-over and the next MFMA phase to land, and nothing is copied.  Other chunk shapes (long steps) keep
    // the one-step-ahead pipeline: source index two edges ahead, row one edge ahead.  All loads are unconditional on a
    // clamped edge index.
    const int e_last = deg > 0 ? beg + deg - 1 : 0;
    // (compiled in only for the kinds with registers to spare: the second row buffer costs the heavy kinds spills)
    static_assert(!TWO_DEEP || TwoDeepOk<L1, GI>::value, "two rows in flight only for the light kinds");
    constexpr bool two_deep = TWO_DEEP;   // the caller guarantees CH == 2 (8 lanes per node)
    float xn[G::D1], xb[G::D1];
    int src_nn, src_b = 0;
    {
        const int src0 = a.src_sorted[min(beg, e_last)];
        const int src1 = a.src_sorted[min(beg + 1, e_last)];
        const float* xp0 = a.x + (int64_t)src0 * a.d_in + xcol;
#pragma unroll
        for (int i = 0; i < G::D1; ++i) xn[i] = xp0[i];
        src_nn = src1;
        if (two_deep) {
            const float* xp1 = a.x + (int64_t)src1 * a.d_in + xcol;
#pragma unroll
            for (int i = 0; i < G::D1; ++i) xb[i] = xp1[i];
            src_nn = a.src_sorted[min(beg + 2, e_last)];
            src_b = a.src_sorted[min(beg + 3, e_last)];
        }
    }
    ld.publish(0);
    __syncthreads();
    int buf = 0;
    for (int s0 = 0; s0 < maxdeg; s0 += CH, buf ^= 1) {
        // the short serial head of a chunk (issue the stage loads, LDS -> MFMA -> LDS) runs at raised priority: it is a
        // latency chain, and every cycle another wave's contraction delays it is added to this wave's chunk (-1 %)
        __builtin_amdgcn_s_setprio(TPF_SETPRIO);
        ld.issue(s0 + CH);
        const float* sb = stage + buf * STAGE_BUF + row0 * STAGE_ROW;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            {
                const f16x8 bh = *reinterpret_cast<const f16x8*>(sb + (16 * t + c) * STAGE_ROW + g * 4);
                const f16x8 bl = *reinterpret_cast<const f16x8*>(sb + (16 * t + c) * STAGE_ROW + 16 + g * 4);
#ifndef MATTEN_ABLATE_NO_MFMA
                float* trow = tile + (16 * t + c) * stride + 4 * g;
                // branch-free per tile count: the 3 MTC matrix instructions of a chunk interleave freely
                if (MTMAX == 1 || MT == 1) mfma_tiles<1>(ah, al, bh, bl, trow);
                else if (MTMAX == 2 || MT == 2) mfma_tiles<(MTMAX < 2 ? MTMAX : 2)>(ah, al, bh, bl, trow);
                else if (MTMAX == 3 || MT == 3) mfma_tiles<(MTMAX < 3 ? MTMAX : 3)>(ah, al, bh, bl, trow);
                else if (MTMAX == 4 || MT == 4) mfma_tiles<(MTMAX < 4 ? MTMAX : 4)>(ah, al, bh, bl, trow);
                else if (MTMAX == 5 || MT == 5) mfma_tiles<(MTMAX < 5 ? MTMAX : 5)>(ah, al, bh, bl, trow);
                else if (MTMAX == 6 || MT == 6) mfma_tiles<(MTMAX < 6 ? MTMAX : 6)>(ah, al, bh, bl, trow);
                else mfma_tiles<MTMAX>(ah, al, bh, bl, trow);
#else
                if ((float)bh[0] == 12345.f) tile[c] = (float)bl[0];
#endif
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the weight tile is written
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_setprio(0);
#ifndef MATTEN_ABLATE_NO_VALU
        auto contract = [&](int so, const float* __restrict__ x) {
            const float* wp = tile + ((j << ch_log2) + so) * stride + u * NC;
            const float* yp = sb + ((j << ch_log2) + so) * STAGE_ROW + 32 + G::Y0;
            float y[G::NY], w[NC];
            {   // the harmonics of a staged row are 16-byte aligned: whole ds_read_b128 over [Y0, Y0 + NY)
                constexpr int Q0 = G::Y0 / 4 * 4, NQ = (G::Y0 + G::NY - Q0 + 3) / 4;
                float yq[4 * NQ];
                const f32x4* y4 = reinterpret_cast<const f32x4*>(yp - (G::Y0 - Q0));
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const f32x4 v = y4[q];
                    yq[4 * q] = v[0], yq[4 * q + 1] = v[1], yq[4 * q + 2] = v[2], yq[4 * q + 3] = v[3];
                }
#pragma unroll
                for (int jj = 0; jj < G::NY; ++jj) y[jj] = yq[G::Y0 - Q0 + jj];
            }
#pragma unroll
            for (int cc = 0; cc < NC; ++cc) w[cc] = wp[cc];
            G::apply(CMASK ? CMASK : mask, x, y, w, acc);
        };
        if (two_deep) {
            for (int so = 0; so < CH; so += 2) {   // CH is 2, 4 or 8 here: slot pairs, one row buffer per parity
                const int s = s0 + so;
                if (s < deg) contract(so, xn);
                {
                    const float* xp = a.x + (int64_t)TPF_SRC(src_nn) * a.d_in + xcol;
#pragma unroll
                    for (int i = 0; i < G::D1; ++i) xn[i] = TPF_XLD(xp, i, src_nn);
                    src_nn = a.src_sorted[min(beg + s + 4, e_last)];
                }
                if (s + 1 < deg) contract(so + 1, xb);
                {
                    const float* xp = a.x + (int64_t)TPF_SRC(src_b) * a.d_in + xcol;
#pragma unroll
                    for (int i = 0; i < G::D1; ++i) xb[i] = TPF_XLD(xp, i, src_b);
                    src_b = a.src_sorted[min(beg + s + 5, e_last)];
                }
            }
        } else {
            for (int so = 0; so < CH; ++so) {
                const int s = s0 + so;
                if (s >= maxdeg) break;
                float x[G::D1];
#pragma unroll
                for (int i = 0; i < G::D1; ++i) x[i] = xn[i];
                {
                    const float* xp = a.x + (int64_t)TPF_SRC(src_nn) * a.d_in + xcol;
#pragma unroll
                    for (int i = 0; i < G::D1; ++i) xn[i] = TPF_XLD(xp, i, src_nn);
                    src_nn = a.src_sorted[min(beg + s + 2, e_last)];
                }
                if (s < deg) contract(so, x);
            }
        }
#endif
        ld.publish(buf ^ 1);
#ifdef MATTEN_ABLATE_NO_BARRIER
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();  // timing experiment only: results are wrong
#else
        __syncthreads();  // the next stage is published; every wave is done with this chunk's rows
#endif
    }
    epi.template store<G>(a, ge, acc, a_scale_inv, node, j, u, valid);
}

// Coupling masks worth a specialisation: only the scalar-block kind (all five couplings, or l2 <= 3 when the target has
// no 4o) -- for every other kind the extra instantiations cost the rest of the kernel more than they gain (DESIGN.md
// section 8: the register allocation of this one function is shared by all kinds).
template <int L1, int GI> struct HotMask { static constexpr unsigned M0 = 0, M1 = 0; };
#ifndef TPF_NO_HOT_MASKS
template <> struct HotMask<0, 0> { static constexpr unsigned M0 = 0x1f, M1 = 0xf; };
#endif
#define MATTEN_RGS_ARGS a, ge, tile, stage, (um >> 8) & 0xffff, node, lane, valid, beg, deg, maxdeg, StoreAgg{tile}
#define MATTEN_RGS(L1, GI, TT, TD, P) \
    do { \
        using HM = HotMask<L1, GI>; \
        if (HM::M0 && ge.mask == HM::M0) run_group_shared<L1, GI, TT, TD, P, HM::M0>(MATTEN_RGS_ARGS); \
        else if (HM::M1 && ge.mask == HM::M1) run_group_shared<L1, GI, TT, TD, P, HM::M1>(MATTEN_RGS_ARGS); \
        else run_group_shared<L1, GI, TT, TD, P, 0u>(MATTEN_RGS_ARGS); \
    } while (0)
#define MATTEN_GROUP_CASE_SHARED(L1, GI) \
    case (L1 * matten::GROUP_KIND_STRIDE + GI): \
        if (paired) MATTEN_RGS(L1, GI, 1, false, true); \
        else if (nodes_per_wave > 16) MATTEN_RGS(L1, GI, 2, false, false); \
        else if (TwoDeepOk<L1, GI>::value && nodes_per_wave <= 8 && nodes_per_wave >= TPF_TWO_DEEP_MIN_NPW) MATTEN_RGS(L1, GI, 1, (TwoDeepOk<L1, GI>::value), false); \
        else MATTEN_RGS(L1, GI, 1, false, false); \
        break;

#define MATTEN_GROUP_CASE(L1, GI) \
    case (L1 * matten::GROUP_KIND_STRIDE + GI): run_group<L1, GI>(a, ge, tile, node, lane, valid, beg, deg, maxdeg); break;

#ifndef TPF_MIN_BLOCKS
#define TPF_MIN_BLOCKS 3
#endif
#ifndef TPF_TWO_DEEP_MIN_NPW
#define TPF_TWO_DEEP_MIN_NPW 2   // two neighbour rows in flight for 8, 4 and 2 nodes per wave (2, 4, 8 slots per chunk)
#endif
// experiment switches (tools/fused_kind_ablate.sh): compile the kernel for a subset of the group kinds only
#if defined(TPF_ONLY_LIGHT)
#define TPF_FOR_EACH_GROUP(X) X(0, 0) X(1, 0) X(1, 1)
#elif defined(TPF_ONLY_HEAVY)
#define TPF_FOR_EACH_GROUP(X) X(2, 0) X(2, 1) X(3, 0) X(3, 1) X(4, 0) X(4, 1)
#else
#define TPF_FOR_EACH_GROUP(X) MATTEN_FOR_EACH_GROUP(X)
#endif
__global__ __launch_bounds__(WAVES_PER_BLOCK * 64, TPF_MIN_BLOCKS) void tp_fused_kernel(Args a, const GroupEntry* __restrict__ entries,
                                                                        const int* __restrict__ umap,
                                                                        int n_entries, int units_per_tile,
                                                                        int blocks_per_tile, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int xcd = blockIdx.x % N_XCD;
    const int q = blockIdx.x / N_XCD;
    const int tile_id = (q / blocks_per_tile) * N_XCD + xcd;
    if (tile_id >= n_tiles) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int unit = (q % blocks_per_tile) * WAVES_PER_BLOCK + wave;
    if (unit >= units_per_tile) return;
    const int lane = threadIdx.x & 63;
    float* tile = lds + wave * a.lds_per_wave;

    // unit -> (entry, node group of the tile), host-ordered (plan.fused_unit_map)
    const int um = __builtin_amdgcn_readfirstlane(umap[unit]);
    if (((um >> 8) & 0xffff) >= n_entries) return;
    const GroupEntry& ge = entries[(um >> 8) & 0xffff];
    const int r = um & 255;
    const bool shared_stage = (um >> 24) & 1;  // uniform over the workgroup (host contract)

    const int cu_log2 = ge.cu_log2;
    const int cu = 1 << cu_log2;
    const int nodes_per_wave = 64 >> cu_log2;
    const int g_in_tile = r * nodes_per_wave + (lane >> cu_log2);
    const int u = lane & (cu - 1);
    const int node = tile_id * TILE_NODES + g_in_tile;
    const bool in_range = (g_in_tile < TILE_NODES) && (node < a.n_nodes);
    const bool valid = in_range && (u < ge.mul);
    int beg = 0, deg = 0;
    if (in_range) {  // every lane of a node (also idle channels) knows the segment: the MFMA role needs it
        beg = a.rowptr[node];
        deg = a.rowptr[node + 1] - beg;
    }
    int maxdeg = deg;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) maxdeg = max(maxdeg, __shfl_xor(maxdeg, off));
#ifdef MATTEN_ABLATE_NO_LOOP
    maxdeg = 0;  // timing experiment: prologue (segment, gather, fragment and stage set-up) + epilogue only
#endif
    if (shared_stage) {
        float* stage = lds + WAVES_PER_BLOCK * a.lds_per_wave;
        const bool paired = (um >> 26) & 1;   // uniform over the workgroup: two entries x two node groups (PairLoader)
        if ((um >> 25) & 1) {  // loader-only unit: fills the workgroup up to four waves
            if (paired) run_loader_only_paired(a, cu_log2, stage, beg, deg, maxdeg);
            else run_loader_only(a, cu_log2, stage, beg, deg, maxdeg);
            return;
        }
        switch (ge.kind) {
            TPF_FOR_EACH_GROUP(MATTEN_GROUP_CASE_SHARED)
            default: break;
        }
        return;
    }
    switch (ge.kind) {
        TPF_FOR_EACH_GROUP(MATTEN_GROUP_CASE)
        default: break;
    }
}

// ---- conv-fused kernel: tensor product + neighbour sum + lin2 of the light input blocks ---------------------------------
// (reference nn/conv.py:113-123: tp -> scatter -> / sqrt(avg) -> lin2(., species) + self-connection)
//
// agg[N, d_mid] exists only to carry the neighbour sums from the tensor-product kernel to lin2: 1.07 GB written and
// read back per launch in the last layer, three quarters of it from the l1 <= 1 input blocks (32 and 16 channels).
// Here a workgroup owns LIN2_NODES = 8 destination nodes and walks those blocks' group entries itself, four at a time
// (a ROUND: one entry per wave, 8 lanes per node, the workgroup-shared stage of run_group_shared).  After a round the
// waves leave their sums in LDS (StoreLds) and the workgroup applies lin2 to them on the spot:
//     out[n, io, v, k] += fan^-1/2 sum_{paths p -> io} sum_u W_p[u, species(n), v] * acc_p[n, u, k]
// A SLOT is 8 consecutive (v, k) pairs of one output irrep; the 8 lanes of a node take one pair each and run the
// slot's CHAIN (every (wave, coupling) of the round that feeds the irrep) as 8-channel dot products: one 16-byte
// global load of the node's species' weights (L2 resident: 27 KB per species) + one ds_read_b128 per 4 FMAs.  The
// weights are per NODE, so nodes need no species sorting (the x[src] gathers keep their crystal locality) and a slot
// belongs to one wave: the accumulation order into the output tile is fixed.  The tile starts as the self-connection
// and leaves as out[8, d_out]; the heavy blocks (l1 >= 2: 2-4 channels, a quarter of agg) keep the agg_rest + lin2 route.
// Host tables: plan.plan_conv_fused.
constexpr int LIN2_NODES = 8;
constexpr int LIN2_T_WAVE_FLOATS = 64 * 28;   // == plan.LIN2_T_WAVE_FLOATS
constexpr int LIN2_STAGE_FLOATS = 2 * 16 * STAGE_ROW;
#ifndef LIN2_NB
#define LIN2_NB 8
#endif

struct Lin2Args {
    const int* rounds;        // [n_rounds, 4] entry or -1
    const int* slot_index;    // [n_rounds, 4, 2] (first slot, count) of (round, wave)
    const int4* slots;        // [n_slots, 2] {d3, n_pairs, out_off, pair_base} {magic, first item, item count, 0}
    const int4* items;        // [n_items] {t_off, a_off, n_chunks, a_stride}
    const float* atab;        // [n_species, a_numel]
    const int* species;       // [N]
    const float* add;         // [N, add_ld] or NULL
    float* out;               // [N, d_out]
    int n_rounds, n_slots, n_items, a_numel, n_species, add_ld, d_out, ld, n_groups;
};

#define MATTEN_LIN2_CASE(L1, GI) \
    case (L1 * matten::GROUP_KIND_STRIDE + GI): run_group_shared<L1, GI, 1, TwoDeepOk<L1, GI>::value, false, 0u>(a, ge, tile, stage, e, node, lane, valid, beg, deg, maxdeg, StoreLds{tile}); break;

__global__ __launch_bounds__(WAVES_PER_BLOCK * 64, TPF_MIN_BLOCKS) void tp_lin2_kernel(Args a, Lin2Args la,
                                                                                      const GroupEntry* __restrict__ entries) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // 8 consecutive groups = one 64-node tile, pinned to an XCD like tp_fused_kernel's tiles
    const int xcd = blockIdx.x % N_XCD;
    const int q8 = blockIdx.x / N_XCD;
    const int grp = ((q8 >> 3) * N_XCD + xcd) * 8 + (q8 & 7);
    if (grp >= la.n_groups) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    float* tarea = lds;                                        // [4][LIN2_T_WAVE_FLOATS]: weight tile during a walk, sums after
    float* tile = tarea + wave * LIN2_T_WAVE_FLOATS;
    float* stage = lds + WAVES_PER_BLOCK * LIN2_T_WAVE_FLOATS;  // [2][16][STAGE_ROW]
    float* otile = stage + LIN2_STAGE_FLOATS;                    // [8][ld]
    int4* slots = reinterpret_cast<int4*>(otile + LIN2_NODES * la.ld);  // [n_slots][2], [n_items]: the lin2 work lists
    int4* items = slots + 2 * la.n_slots;
    for (int i = threadIdx.x; i < 2 * la.n_slots + la.n_items; i += WAVES_PER_BLOCK * 64)
        slots[i] = i < 2 * la.n_slots ? la.slots[i] : la.items[i - 2 * la.n_slots];

    const int j = lane >> 3, q = lane & 7;
    const int node = grp * LIN2_NODES + j;
    const bool in_range = node < a.n_nodes;
    int beg = 0, deg = 0;
    if (in_range) {
        beg = a.rowptr[node];
        deg = a.rowptr[node + 1] - beg;
    }
    int maxdeg = deg;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) maxdeg = max(maxdeg, __shfl_xor(maxdeg, off));

    // output tile <- self-connection (or zero)
    for (int i = threadIdx.x; i < LIN2_NODES * la.d_out; i += WAVES_PER_BLOCK * 64) {
        const int jj = i / la.d_out, col = i - jj * la.d_out;
        const int nn = grp * LIN2_NODES + jj;
        otile[jj * la.ld + col] = (la.add && nn < a.n_nodes) ? la.add[(int64_t)nn * la.add_ld + col] : 0.0f;
    }
    int sp = in_range ? la.species[node] : 0;
    sp = min(max(sp, 0), la.n_species - 1);

    for (int r = 0; r < la.n_rounds; ++r) {
        const int e = __builtin_amdgcn_readfirstlane(la.rounds[r * WAVES_PER_BLOCK + wave]);
        if (e < 0) {
            run_loader_only_t<1>(a, 3, stage, beg, deg, maxdeg);
        } else {
            const GroupEntry& ge = entries[e];
            const bool valid = in_range && (q < ge.mul);
            switch (ge.kind) {
                MATTEN_LIN2_CASE(0, 0)
                MATTEN_LIN2_CASE(1, 0)
                MATTEN_LIN2_CASE(1, 1)
                default: break;
            }
        }
        __syncthreads();  // every wave's sums are in LDS (and, in round 0, the output tile is initialised)
        // lin2 of this round.  Per slot the lane's (v, k) pair, weight row and LDS row are set up once; an item is the
        // <= 4 channel chunks of one path (constant strides), all its weight loads in flight before the first is used.
        const int s_beg = __builtin_amdgcn_readfirstlane(la.slot_index[(r * WAVES_PER_BLOCK + wave) * 2]);
        const int s_cnt = __builtin_amdgcn_readfirstlane(la.slot_index[(r * WAVES_PER_BLOCK + wave) * 2 + 1]);
#ifndef MATTEN_ABLATE_NO_EPI
        for (int si = s_beg; si < s_beg + s_cnt; ++si) {
            const int4 r0 = slots[2 * si], r1 = slots[2 * si + 1];
            const int d3 = __builtin_amdgcn_readfirstlane(r0.x), n_pairs = __builtin_amdgcn_readfirstlane(r0.y);
            const int i_beg = __builtin_amdgcn_readfirstlane(r1.y), i_cnt = __builtin_amdgcn_readfirstlane(r1.z);
            const int idx = __builtin_amdgcn_readfirstlane(r0.w) + q;
            const int idc = min(idx, n_pairs - 1);                      // clamped: the loads stay unconditional
            const int v = (idc * __builtin_amdgcn_readfirstlane(r1.x)) >> 16, k = idc - v * d3;
            const unsigned aoff = (unsigned)(sp * la.a_numel + v * 8);  // floats from atab: this node's species row, row v
            const float* trow = tarea + (j * d3 + k) * 8;
            float sum0 = 0.0f, sum1 = 0.0f;
            for (int ii = i_beg; ii < i_beg + i_cnt; ++ii) {
                const int4 it = items[ii];
                const int t_off = __builtin_amdgcn_readfirstlane(it.x), nch = __builtin_amdgcn_readfirstlane(it.z);
                const int astr = __builtin_amdgcn_readfirstlane(it.w);
                const float* ab = la.atab + __builtin_amdgcn_readfirstlane(it.y);   // uniform base + per-lane 32-bit offset
                f32x4 a0[4], a1[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int cc = min(c, nch - 1);
#ifdef LIN2_ABL_NOGL
                    a0[c] = f32x4{1.f, 2.f, 3.f, (float)(cc * astr + aoff)};
                    a1[c] = a0[c];
#else
                    a0[c] = *reinterpret_cast<const f32x4*>(ab + (unsigned)(cc * astr) + aoff);
                    a1[c] = *reinterpret_cast<const f32x4*>(ab + (unsigned)(cc * astr) + aoff + 4);
#endif
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (c < nch) {
#ifdef LIN2_ABL_NOLDS
                        const f32x4 t0 = f32x4{1.f, 2.f, 3.f, (float)(t_off + c)};
                        const f32x4 t1 = t0;
#else
                        const f32x4 t0 = *reinterpret_cast<const f32x4*>(trow + t_off + c * LIN2_T_WAVE_FLOATS);
                        const f32x4 t1 = *reinterpret_cast<const f32x4*>(trow + t_off + c * LIN2_T_WAVE_FLOATS + 4);
#endif
                        sum0 = fmaf(a0[c][0], t0[0], sum0); sum1 = fmaf(a1[c][0], t1[0], sum1);
                        sum0 = fmaf(a0[c][1], t0[1], sum0); sum1 = fmaf(a1[c][1], t1[1], sum1);
                        sum0 = fmaf(a0[c][2], t0[2], sum0); sum1 = fmaf(a1[c][2], t1[2], sum1);
                        sum0 = fmaf(a0[c][3], t0[3], sum0); sum1 = fmaf(a1[c][3], t1[3], sum1);
                    }
                }
            }
#ifdef LIN2_ABL_NOOUT
            if (idx < n_pairs && sum0 + sum1 == 12345.678f) otile[j * la.ld + __builtin_amdgcn_readfirstlane(r0.z) + idx] += sum0 + sum1;
#else
            if (idx < n_pairs) otile[j * la.ld + __builtin_amdgcn_readfirstlane(r0.z) + idx] += sum0 + sum1;
#endif
        }
#endif
        __syncthreads();  // the sums are consumed: the next round may overwrite the regions
    }
    for (int i = threadIdx.x; i < LIN2_NODES * la.d_out; i += WAVES_PER_BLOCK * 64) {
        const int jj = i / la.d_out, col = i - jj * la.d_out;
        const int nn = grp * LIN2_NODES + jj;
        if (nn < a.n_nodes) la.out[(int64_t)nn * la.d_out + col] = otile[jj * la.ld + col];
    }
}

// Hidden layers of the radial MLP: rbf(|v|) -> 32 -> 32, written as the split fp16 form h2s [E,2,32] (see header comment).
constexpr int NT = 4;
// silu on the hardware transcendental units: v_exp_f32 (2^x) + v_rcp_f32, ~1 ulp each, against the ~25-instruction
// expf + IEEE division; the hidden kernel is bound by exactly this arithmetic (64 silu per edge and layer)
__device__ __forceinline__ float silu(float z) {
#ifdef RH_ABLATE_NO_SILU
    return z * 0.5f;
#endif
    return z * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z));
}

template <int KS0>
__global__ __launch_bounds__(256) void radial_hidden_kernel(const float4* __restrict__ geom, int64_t E, int n_basis,
                                                            float r_start, float r_end,
                                                            const float* __restrict__ w0p,
                                                            const float* __restrict__ w1p, _Float16* __restrict__ h2s,
                                                            const float* __restrict__ h_scale) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int64_t e0 = ((int64_t)blockIdx.x * 4 + wave) * (NT * 16);
    if (e0 >= E) return;
    const float hs = h_scale ? *h_scale : 1.0f;
    const float inv_c = 1.0f / (r_end - r_start);
    const float bes_pref = sqrtf(2.0f * inv_c) * sqrtf((float)n_basis);  // soft_one_hot_linspace 'bessel' x sqrt(nb)
    float a1[2][8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        int k = 16 * (kk >> 2) + 4 * g + (kk & 3);
        a1[0][kk] = w1p[k * HID + c];
        a1[1][kk] = w1p[k * HID + 16 + c];
    }
    // all NT edge lengths of the wave are requested up front: the tiles below are long dependent chains (Bessel ->
    // MFMA -> silu -> MFMA -> silu -> split -> store) and would otherwise each start with an exposed load
    float lens[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int64_t e = e0 + nt * 16 + c;
        lens[nt] = geom[e < E ? e : E - 1].w;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int64_t e = e0 + nt * 16 + c;
        const float len = lens[nt];
        const float xr = len - r_start;
        const float t01 = xr * inv_c, inv_xr = __builtin_amdgcn_rcpf(xr);
        f32x4 h0 = {0.f, 0.f, 0.f, 0.f}, h1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < KS0; ++kk) {
            int k = 4 * kk + g;
            // Bessel basis on the transcendental units: sin(pi (k+1) t) = v_sin_f32 of (k+1) t / 2 revolutions, 1/r by
            // v_rcp_f32 (the precise sinf + three IEEE divisions were half of this kernel's instructions)
            float b = 0.0f;
            if (k < n_basis && t01 > 0.0f && t01 < 1.0f)
                b = bes_pref * __builtin_amdgcn_sinf(0.5f * (float)(k + 1) * t01) * inv_xr;
            h0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0p[k * HID + c], b, h0, 0, 0, 0);
            h1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0p[k * HID + 16 + c], b, h1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            h0[r] = silu(h0[r]);
            h1[r] = silu(h1[r]);
        }
        f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
#ifdef RH_ABLATE_NO_L1
        o0 = h0, o1 = h1;
#else
        for (int kk = 0; kk < 8; ++kk) {
            float b = kk < 4 ? h0[kk & 3] : h1[kk & 3];
            o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[0][kk], b, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[1][kk], b, o1, 0, 0, 0);
        }
#endif
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            o0[r] = silu(o0[r]);
            o1[r] = silu(o1[r]);
        }
#ifdef RH_ABLATE_NO_STORE
        if (e < E && o0[0] == 12345.678f) {
#else
        if (e < E) {
#endif
            f16x8 hi, lo;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                _Float16 h, l;
                split_f16(o0[r] * hs, h, l);
                hi[r] = h, lo[r] = l;
                split_f16(o1[r] * hs, h, l);
                hi[4 + r] = h, lo[4 + r] = l;
            }
            f16x8* dst = reinterpret_cast<f16x8*>(h2s + e * (2 * HID) + g * 8);
            dst[0] = hi;
            dst[HID / 8] = lo;
        }
    }
}

// All conv layers' hidden radial features in ONE launch: the edge lengths and the Bessel basis are the same for every
// layer (reference nn/embedding.py:185-203 computes the embedding once per batch), only the MLP weights differ; one
// launch instead of one per layer removes three launch + drain phases of a latency-bound kernel and evaluates the basis once.
constexpr int RH_MAX_LAYERS = 8;
struct RadialLayers {
    const float* w0p[RH_MAX_LAYERS];
    const float* w1p[RH_MAX_LAYERS];
    _Float16* h2s[RH_MAX_LAYERS];
    const float* h_scale[RH_MAX_LAYERS];   // per MLP: device pointer to its power-of-two output scale, or NULL (1)
    int n_layers;
};

template <int KS0>
__global__ __launch_bounds__(256) void radial_hidden_multi_kernel(const float4* __restrict__ geom, int64_t E, int n_basis,
                                                                  float r_start, float r_end, RadialLayers L) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int64_t e0 = ((int64_t)blockIdx.x * 4 + wave) * (NT * 16);
    if (e0 >= E) return;
    const float inv_c = 1.0f / (r_end - r_start);
    const float bes_pref = sqrtf(2.0f * inv_c) * sqrtf((float)n_basis);
    float bes[NT][KS0];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int64_t e = e0 + nt * 16 + c;
        const float len = geom[e < E ? e : E - 1].w;
        const float xr = len - r_start;
        const float t01 = xr * inv_c, inv_xr = __builtin_amdgcn_rcpf(xr);
#pragma unroll
        for (int kk = 0; kk < KS0; ++kk) {
            const int k = 4 * kk + g;
            float b = 0.0f;
            if (k < n_basis && t01 > 0.0f && t01 < 1.0f)
                b = bes_pref * __builtin_amdgcn_sinf(0.5f * (float)(k + 1) * t01) * inv_xr;
            bes[nt][kk] = b;
        }
    }
    for (int l = 0; l < L.n_layers; ++l) {
        const float* __restrict__ w0p = L.w0p[l];
        const float* __restrict__ w1p = L.w1p[l];
        _Float16* __restrict__ h2s = L.h2s[l];
        const float hs = L.h_scale[l] ? *L.h_scale[l] : 1.0f;
        float a0[2][KS0], a1[2][8];
#pragma unroll
        for (int kk = 0; kk < KS0; ++kk) {
            a0[0][kk] = w0p[(4 * kk + g) * HID + c];
            a0[1][kk] = w0p[(4 * kk + g) * HID + 16 + c];
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int k = 16 * (kk >> 2) + 4 * g + (kk & 3);
            a1[0][kk] = w1p[k * HID + c];
            a1[1][kk] = w1p[k * HID + 16 + c];
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int64_t e = e0 + nt * 16 + c;
            f32x4 h0 = {0.f, 0.f, 0.f, 0.f}, h1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KS0; ++kk) {
                h0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[0][kk], bes[nt][kk], h0, 0, 0, 0);
                h1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[1][kk], bes[nt][kk], h1, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                h0[r] = silu(h0[r]);
                h1[r] = silu(h1[r]);
            }
            f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const float b = kk < 4 ? h0[kk & 3] : h1[kk & 3];
                o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[0][kk], b, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[1][kk], b, o1, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                o0[r] = silu(o0[r]);
                o1[r] = silu(o1[r]);
            }
            if (e < E) {
                f16x8 hi, lo;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    _Float16 h, ll;
                    split_f16(o0[r] * hs, h, ll);
                    hi[r] = h, lo[r] = ll;
                    split_f16(o1[r] * hs, h, ll);
                    hi[4 + r] = h, lo[4 + r] = ll;
                }
                f16x8* dst = reinterpret_cast<f16x8*>(h2s + e * (2 * HID) + g * 8);
                dst[0] = hi;
                dst[HID / 8] = lo;
            }
        }
    }
}

}  // namespace

extern "C" int matten_radial_hidden_multi(const float* geom_sorted, int64_t n_edges, int n_basis, float r_start,
                                          float r_end, const float* const* w0p, int nb_pad, const float* const* w1p,
                                          int hidden, uint16_t* const* h2s, const float* const* h_scale, int n_layers,
                                          matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_edges < 0 || hidden != HID || (nb_pad & 3) || nb_pad < n_basis || nb_pad > 16 || n_layers < 1 ||
        n_layers > RH_MAX_LAYERS)
        return MATTEN_EINVAL;
    if (n_edges == 0) return MATTEN_OK;
    if (!geom_sorted || !w0p || !w1p || !h2s) return MATTEN_EINVAL;
    RadialLayers L{};
    L.n_layers = n_layers;
    for (int l = 0; l < n_layers; ++l) {
        if (!w0p[l] || !w1p[l] || !h2s[l]) return MATTEN_EINVAL;
        L.w0p[l] = w0p[l], L.w1p[l] = w1p[l], L.h2s[l] = (_Float16*)h2s[l];
        L.h_scale[l] = h_scale ? h_scale[l] : nullptr;
    }
    unsigned grid = (unsigned)matten_cdiv(n_edges, 4 * NT * 16);
#define LAUNCH(K) \
    radial_hidden_multi_kernel<K><<<grid, 256, 0, stream>>>((const float4*)geom_sorted, n_edges, n_basis, r_start, r_end, L)
    switch (nb_pad >> 2) {
        case 1: LAUNCH(1); break;
        case 2: LAUNCH(2); break;
        case 3: LAUNCH(3); break;
        default: LAUNCH(4); break;
    }
#undef LAUNCH
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}

extern "C" int matten_radial_hidden(const float* geom_sorted, int64_t n_edges, int n_basis, float r_start, float r_end,
                                    const float* w0p, int nb_pad, const float* w1p, int hidden, uint16_t* h2s,
                                    const float* h_scale, matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_edges < 0 || hidden != HID || (nb_pad & 3) || nb_pad < n_basis || nb_pad > 16) return MATTEN_EINVAL;
    if (n_edges == 0) return MATTEN_OK;
    if (!geom_sorted || !w0p || !w1p || !h2s) return MATTEN_EINVAL;
    unsigned grid = (unsigned)matten_cdiv(n_edges, 4 * NT * 16);
#define LAUNCH(K) \
    radial_hidden_kernel<K><<<grid, 256, 0, stream>>>((const float4*)geom_sorted, n_edges, n_basis, r_start, r_end, w0p, w1p, (_Float16*)h2s, h_scale)
    switch (nb_pad >> 2) {
        case 1: LAUNCH(1); break;
        case 2: LAUNCH(2); break;
        case 3: LAUNCH(3); break;
        default: LAUNCH(4); break;
    }
#undef LAUNCH
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}

extern "C" int matten_tp_fused(const float* x, int64_t d_in, const uint16_t* h2s, const float* w2p, int64_t w_pad,
                               const float* sh_sorted, int64_t sh_stride, const int32_t* rowptr,
                               const int32_t* src_sorted, int64_t n_nodes, const int32_t* group_entries,
                               const int32_t* unit_map, int64_t n_entries, int64_t units_per_tile,
                               int64_t lds_floats_per_wave, int64_t d_mid, float avg_num_neighbors,
                               const float* num_neigh, const uint16_t* a_split, const float* a_scale_inv, float* agg,
                               matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_nodes < 0 || d_in <= 0 || w_pad <= 0 || sh_stride < 32 || (sh_stride & 3) || n_entries <= 0 ||
        units_per_tile <= 0 || d_mid <= 0 || lds_floats_per_wave <= 0 || (lds_floats_per_wave & 3))
        return MATTEN_EINVAL;
    if (n_nodes == 0) return MATTEN_OK;
    if (!x || !h2s || !w2p || !sh_sorted || !rowptr || !src_sorted || !group_entries || !unit_map || !agg)
        return MATTEN_EINVAL;
    if (!(avg_num_neighbors > 0.0f) && !num_neigh) return MATTEN_EINVAL;
    const size_t lds = sizeof(float) * ((size_t)lds_floats_per_wave * WAVES_PER_BLOCK + STAGE_TOTAL_FLOATS + PAIR_INFO_INTS);
    if (lds > 64 * 1024) return MATTEN_EINVAL;
    if ((a_split == nullptr) != (a_scale_inv == nullptr)) return MATTEN_EINVAL;
    Args a{x, (const _Float16*)h2s, w2p, (const _Float16*)a_split, a_scale_inv, sh_sorted, rowptr, src_sorted, num_neigh, agg, (int)d_in, (int)w_pad, (int)sh_stride,
           (int)d_mid, (int)n_nodes, (int)lds_floats_per_wave, avg_num_neighbors};
    const int n_tiles = (int)matten_cdiv(n_nodes, TILE_NODES);
    const int blocks_per_tile = (int)matten_cdiv(units_per_tile, WAVES_PER_BLOCK);
    const int64_t grid = matten_cdiv(n_tiles, N_XCD) * N_XCD * (int64_t)blocks_per_tile;
    if (grid >= ((int64_t)1 << 31)) return MATTEN_EINVAL;
    tp_fused_kernel<<<(unsigned)grid, WAVES_PER_BLOCK * 64, lds, stream>>>(
        a, (const GroupEntry*)group_entries, unit_map, (int)n_entries, (int)units_per_tile, blocks_per_tile, n_tiles);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}

extern "C" int matten_tp_max_cols(void) { return TPF_MAX_COLS; }
extern "C" int matten_tp_max_cols_l0(void) { return TPF_MAX_COLS_L0; }
extern "C" int matten_tp_max_cols_l1(void) { return TPF_MAX_COLS_L1; }
extern "C" int matten_tp_lin2_group_nodes(void) { return LIN2_NODES; }
extern "C" int matten_tp_lin2_t_wave_floats(void) { return LIN2_T_WAVE_FLOATS; }

extern "C" int matten_tp_lin2(const float* x, int64_t d_in, const uint16_t* h2s, const float* w2p, int64_t w_pad,
                              const float* sh_sorted, int64_t sh_stride, const int32_t* rowptr,
                              const int32_t* src_sorted, int64_t n_nodes, const int32_t* light_entries,
                              int64_t n_entries, const int32_t* rounds, int64_t n_rounds, const int32_t* slot_index,
                              const int32_t* slots, int64_t n_slots, const int32_t* items, int64_t n_items,
                              const float* atab, int64_t a_numel, int64_t n_species, const int32_t* species,
                              float avg_num_neighbors, const float* num_neigh, const uint16_t* a_split,
                              const float* a_scale_inv, const float* add, int64_t add_ld, int64_t d_out, float* out,
                              matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_nodes < 0 || d_in <= 0 || w_pad <= 0 || sh_stride < 32 || (sh_stride & 3) || n_entries <= 0 || n_rounds <= 0 ||
        n_slots < 0 || n_items < 0 || a_numel <= 0 || (a_numel & 7) || n_species <= 0 || d_out <= 0)
        return MATTEN_EINVAL;
    if (n_nodes == 0) return MATTEN_OK;
    if (!x || !h2s || !w2p || !sh_sorted || !rowptr || !src_sorted || !light_entries || !rounds || !slot_index ||
        !slots || !items || !atab || !species || !a_split || !a_scale_inv || !out)
        return MATTEN_EINVAL;
    if (!(avg_num_neighbors > 0.0f) && !num_neigh) return MATTEN_EINVAL;
    if (add && add_ld < d_out) return MATTEN_EINVAL;
    const int ld = (int)((d_out + 23) / 32 * 32 + 8);   // row stride of the output tile: == 8 mod 32 (bank spread)
    const size_t lds = sizeof(float) * ((size_t)WAVES_PER_BLOCK * LIN2_T_WAVE_FLOATS + LIN2_STAGE_FLOATS +
                                        (size_t)LIN2_NODES * ld + 8 * (size_t)n_slots + 4 * (size_t)n_items);
    if (lds > 64 * 1024) return MATTEN_EINVAL;
    Args a{x, (const _Float16*)h2s, w2p, (const _Float16*)a_split, a_scale_inv, sh_sorted, rowptr, src_sorted, num_neigh,
           nullptr, (int)d_in, (int)w_pad, (int)sh_stride, 0, (int)n_nodes, LIN2_T_WAVE_FLOATS, avg_num_neighbors};
    const int n_groups = (int)matten_cdiv(n_nodes, LIN2_NODES);
    Lin2Args la{rounds, slot_index, (const int4*)slots, (const int4*)items, atab, species, add, out,
                (int)n_rounds, (int)n_slots, (int)n_items, (int)a_numel, (int)n_species, (int)add_ld, (int)d_out, ld, n_groups};
    // groups are numbered tile-major (8 per 64-node tile); the grid covers whole sets of N_XCD tiles
    const int64_t n_tiles = matten_cdiv(n_groups, 8);
    const int64_t grid = matten_cdiv(n_tiles, N_XCD) * N_XCD * 8;
    if (grid >= ((int64_t)1 << 31)) return MATTEN_EINVAL;
    tp_lin2_kernel<<<(unsigned)grid, WAVES_PER_BLOCK * 64, lds, stream>>>(a, la, (const GroupEntry*)light_entries);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}
