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
#include "tp_walk.h"

namespace {

using namespace matten_walk;

// ---- what a unit does with its neighbour sums: one row slice of agg[N, d_mid] per (node, channel) --------------------
// (Measured and removed, docs/LAB_NOTES.md: a vector epilogue -- channel lanes exchanging accumulators through LDS for
// 16-byte stores, +8 % -- and non-temporal stores, +20 %: what costs is the 1.1 GB of distinct bytes per launch.)
// MERGED entries (plan.TP_KIND_MERGED: two adjacent two-channel blocks in one four-lane entry): lanes u = 0, 1 are the
// first block's channels and use this record's offsets, lanes u = 2, 3 the second block's and use the record that follows
// it in the entry array (its own coupling mask there too; the first half's own mask travels in that record's w_base word).
constexpr int KIND_MERGED = 256;   // == plan.TP_KIND_MERGED
struct StoreAgg {
    template <class G>
    __device__ __forceinline__ void store(const Args& a, const GroupEntry& ge, const float* __restrict__ acc,
                                          float a_scale_inv, int node, int j, int u, bool valid) const {
        if (TPF_LAB_NO_STORE ? (valid && acc[0] == 12345.678f) : valid) {
            const float nn = a.avg_nn > 0.0f ? a.avg_nn : a.num_neigh[node];
            const float norm = a_scale_inv / sqrtf(nn);
            float* orow = a.agg + (int64_t)node * a.d_mid;
            const bool merged = (ge.kind & KIND_MERGED) != 0;          // wave-uniform
            const GroupEntry& g2 = (&ge)[merged ? 1 : 0];
            const bool second = merged && u >= 2;
            const unsigned mask = merged ? (second ? g2.mask : (unsigned)g2.w_base) : ge.mask;
            const int uc = second ? u - 2 : u;
#pragma unroll
            for (int cc = 0; cc < G::NC; ++cc) {
                if ((mask >> cc) & 1u) {
                    const int d3 = 2 * G::L3[cc] + 1;
                    // t_off[cc] == 0: the reference's "mul_ir" row, [channel][component].  Otherwise the component-major
                    // row of plan.plan_agg_linear: t_off[cc] floats between components, the channel lanes side by side
                    const int ks = second ? g2.t_off[cc] : ge.t_off[cc];
                    float* op = orow + (second ? g2.out_off[cc] : ge.out_off[cc]) + (ks ? uc : uc * d3);
                    const int kstep = ks ? ks : 1;
#pragma unroll
                    for (int k = 0; k < 2 * matten::CG_LMAX + 1; ++k)
                        if (k < d3) op[k * kstep] = acc[G::OFF[cc] + k] * norm;
                }
            }
        }
    }
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
                if (s0 + so < degn) {
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
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the tile is written
        __builtin_amdgcn_wave_barrier();
        // ---- VALU: contract this lane's channel for its node's edges of the chunk ----
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
        __builtin_amdgcn_wave_barrier();  // LDS is in order per wave: the next chunk's stores follow these reads
    }
    StoreAgg{}.store<G>(a, ge, acc, a_scale_inv, node, j, u, valid);   // (a_scale_inv undoes the power-of-two scale of the A tile)
}

#define MATTEN_RGS_ARGS a, ge, tile, stage, (um >> 8) & 0xffff, tile_id, r, reps, lane, StoreAgg{}
#define MATTEN_RGS(L1, GI, TT, TD, P) \
    do { \
        using HM = HotMask<L1, GI>; \
        if (HM::M0 && ge.mask == HM::M0) run_group_shared<L1, GI, TT, TD, P, HM::M0>(MATTEN_RGS_ARGS); \
        else if (HM::M1 && ge.mask == HM::M1) run_group_shared<L1, GI, TT, TD, P, HM::M1>(MATTEN_RGS_ARGS); \
        else run_group_shared<L1, GI, TT, TD, P, 0u>(MATTEN_RGS_ARGS); \
    } while (0)
#if defined(MATTEN_LAB) && defined(TPF_ONLY_VARIANT)
// tools/isa_mix.py: ONE (kind, walk variant, coupling mask) per object, so that every loop of the disassembly has a name.
//   TPF_ONLY_VARIANT 0 = paired workgroup, 1 = two MFMA tiles per chunk (32 nodes per wave), 2 = two rows in flight, 3 = plain
//   TPF_ONLY_MASK    0 = the run-time mask (uniform branch per coupling), else that mask at compile time
#ifndef TPF_ONLY_MASK
#define TPF_ONLY_MASK 0
#endif
#define MATTEN_RGS_ONLY(L1, GI, TT, TD, P) run_group_shared<L1, GI, TT, TD, P, (unsigned)(TPF_ONLY_MASK)>(MATTEN_RGS_ARGS)
#define MATTEN_GROUP_CASE_SHARED(L1, GI) \
    case (L1 * matten::GROUP_KIND_STRIDE + GI): \
        if (TPF_ONLY_VARIANT == 0) MATTEN_RGS_ONLY(L1, GI, 1, (TPF_CHUNK_DEEP && L1 == 0 ? 2 : 0), true); \
        else if (TPF_ONLY_VARIANT == 1) MATTEN_RGS_ONLY(L1, GI, 2, 0, false); \
        else if (TPF_ONLY_VARIANT == 2) MATTEN_RGS_ONLY(L1, GI, 1, (TPF_CHUNK_DEEP && L1 == 0 ? 2 : TwoDeepOk<L1, GI>::value ? 1 : 0), false); \
        else MATTEN_RGS_ONLY(L1, GI, 1, 0, false); \
        break;
#else
#define MATTEN_GROUP_CASE_SHARED(L1, GI) \
    case (L1 * matten::GROUP_KIND_STRIDE + GI): \
        if (paired && TPF_CHUNK_DEEP && L1 == 0 && nodes_per_wave <= 8 && nodes_per_wave >= 4) MATTEN_RGS(L1, GI, 1, (L1 == 0 ? 2 : 0), true); \
        else if (paired) MATTEN_RGS(L1, GI, 1, 0, true); \
        else if (nodes_per_wave > 16) MATTEN_RGS(L1, GI, 2, 0, false); \
        else if (TPF_CHUNK_DEEP && L1 == 0 && nodes_per_wave <= 8 && nodes_per_wave >= 4) MATTEN_RGS(L1, GI, 1, (L1 == 0 ? 2 : 0), false); \
        else if (TwoDeepOk<L1, GI>::value && nodes_per_wave <= 8 && nodes_per_wave >= TPF_TWO_DEEP_MIN_NPW) MATTEN_RGS(L1, GI, 1, (TwoDeepOk<L1, GI>::value ? 1 : 0), false); \
        else MATTEN_RGS(L1, GI, 1, 0, false); \
        break;
#endif

#define MATTEN_GROUP_CASE(L1, GI) \
    case (L1 * matten::GROUP_KIND_STRIDE + GI): run_group<L1, GI>(a, ge, tile, un.node, lane, un.valid, un.beg, un.deg, un.maxdeg); break;

#ifndef TPF_MIN_BLOCKS
#define TPF_MIN_BLOCKS 3
#endif
#ifndef TPF_CHUNK_DEEP
#define TPF_CHUNK_DEEP 1   // scalar input blocks with 2 or 4 slots per chunk gather a whole chunk ahead (tp_walk.h, TWO_DEEP = 2)
#endif
#ifndef TPF_TWO_DEEP_MIN_NPW
#define TPF_TWO_DEEP_MIN_NPW 2   // two neighbour rows in flight for 8, 4 and 2 nodes per wave (2, 4, 8 slots per chunk)
#endif
// lab builds (tools/fused_kind_ablate.sh, -DMATTEN_LAB): the kernel compiled for a subset of the group kinds only
#if defined(MATTEN_LAB) && defined(TPF_ONLY_L1) && defined(TPF_ONLY_GI)
#define TPF_FOR_EACH_GROUP(X) X(TPF_ONLY_L1, TPF_ONLY_GI)
#elif defined(MATTEN_LAB) && defined(TPF_ONLY_LIGHT)
#define TPF_FOR_EACH_GROUP(X) X(0, 0) X(1, 0) X(1, 1)
#elif defined(MATTEN_LAB) && defined(TPF_ONLY_HEAVY)
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
    const int reps = 1 << ((um >> 27) & 3);    // persistent unit: node groups r, r + 1, ... (paired: r, r + 2, ...) in turn

    const int cu_log2 = ge.cu_log2;
    const int nodes_per_wave = 64 >> cu_log2;
    if (shared_stage) {
        float* stage = lds + WAVES_PER_BLOCK * a.lds_per_wave;
        const bool paired = (um >> 26) & 1;   // uniform over the workgroup: two entries x two node groups (PairLoader)
        if ((um >> 25) & 1) {  // loader-only unit: fills the workgroup up to four waves
            if (paired) run_loader_only_paired(a, ge, stage, tile_id, r, reps, lane);
            else run_loader_only(a, ge, stage, tile_id, r, reps, lane);
            return;
        }
        switch (ge.kind & (KIND_MERGED - 1)) {
            TPF_FOR_EACH_GROUP(MATTEN_GROUP_CASE_SHARED)
            default: break;
        }
        return;
    }
#if !(defined(MATTEN_LAB) && defined(TPF_ONLY_VARIANT))
    const UnitNodes un = unit_nodes(a, ge, tile_id, r, lane);
    switch (ge.kind & (KIND_MERGED - 1)) {
        TPF_FOR_EACH_GROUP(MATTEN_GROUP_CASE)
        default: break;
    }
#endif
}

// Hidden layers of the radial MLP: rbf(|v|) -> 32 -> 32, written as the split fp16 form h2s [E,2,32] (see header comment).
constexpr int NT = 4;
// silu on the hardware transcendental units: v_exp_f32 (2^x) + v_rcp_f32, ~1 ulp each, against the ~25-instruction
// expf + IEEE division; the hidden kernel is bound by exactly this arithmetic (64 silu per edge and layer)
__device__ __forceinline__ float silu(float z) {
#if defined(MATTEN_LAB) && defined(RH_ABLATE_NO_SILU)
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
        for (int kk = 0; kk < 8; ++kk) {
            float b = kk < 4 ? h0[kk & 3] : h1[kk & 3];
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

#if TPF_TRACING
static unsigned* g_tp_trace = nullptr;
static int64_t g_tp_trace_waves = 0, g_tp_trace_skip = 0;
// lab builds only (tools/tp_trace.py): the waves of ONE matten_tp_fused launch -- the one after `skip` further launches --
// write 16 words each into buf (skip < 0: every launch until disarmed with buf = NULL)
extern "C" int matten_lab_tp_trace(unsigned* buf, int64_t n_waves, int64_t skip) {
    g_tp_trace = buf, g_tp_trace_waves = n_waves, g_tp_trace_skip = skip;
    return 0;
}
#endif

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
#if TPF_TRACING
    a.trace = nullptr;
    if (g_tp_trace && grid * WAVES_PER_BLOCK <= g_tp_trace_waves) {
        if (g_tp_trace_skip <= 0) a.trace = g_tp_trace;
        if (g_tp_trace_skip == 0) g_tp_trace = nullptr;
        if (g_tp_trace_skip > 0) --g_tp_trace_skip;
    }
#endif
    tp_fused_kernel<<<(unsigned)grid, WAVES_PER_BLOCK * 64, lds, stream>>>(
        a, (const GroupEntry*)group_entries, unit_map, (int)n_entries, (int)units_per_tile, blocks_per_tile, n_tiles);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}

extern "C" int matten_tp_max_cols(void) { return TPF_MAX_COLS; }
extern "C" int matten_tp_max_cols_l0(void) { return TPF_MAX_COLS_L0; }
extern "C" int matten_tp_max_cols_l1(void) { return TPF_MAX_COLS_L1; }
extern "C" int matten_tp_groups_hash(void) { return matten::GROUPS_HASH; }

