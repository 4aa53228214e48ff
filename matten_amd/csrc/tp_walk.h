// The walk of matten_tp_fused (csrc/tp_fused.hip): a destination node's CSR segment in chunks with a workgroup-shared
// stage.  (Until round 5 also included by the conv-tile kernel -- neighbour sums kept on chip, lin2 per 16-node tile --
// which lost 8-20 % to tp_fused + agg_linear and was removed: docs/LAB_NOTES.md round 4.)  See the header comment of
// tp_fused.hip for the arithmetic (fp16-split matrix products for the last radial layer, literal-coefficient CG code).
//   reference nn/utils.py:246-251,260,263 (radial MLP -> per-edge weights -> uvu tensor product), nn/conv.py:113-120
#pragma once
#include <type_traits>

#include "cg_gen.h"
#include "common.h"
#include "sh.h"

// ---- experiment switches ------------------------------------------------------------------------------------------------
// Timing builds of tools/*.sh (-DMATTEN_LAB -DMATTEN_ABLATE_...: a phase of the kernel compiled out, results wrong).  The
// production object is built without MATTEN_LAB: every switch below is then the constant 0 and its branch is dead code.
#ifndef MATTEN_LAB
#if defined(MATTEN_ABLATE_NO_GATHER) || defined(MATTEN_ABLATE_NO_XLOAD) || defined(MATTEN_ABLATE_NO_H2LOAD) || \
    defined(MATTEN_ABLATE_NO_MFMA) || defined(MATTEN_ABLATE_NO_VALU) || defined(MATTEN_ABLATE_NO_STORE) ||      \
    defined(MATTEN_ABLATE_NO_BARRIER) || defined(MATTEN_ABLATE_NO_LOOP) || defined(MATTEN_ABLATE_NO_EPI)
#error "MATTEN_ABLATE_* switches are honoured by -DMATTEN_LAB builds only (tools/*.sh)"
#endif
#endif
#if defined(MATTEN_LAB) && defined(MATTEN_ABLATE_NO_GATHER)   // every gather reads the destination node's own (cache-resident) row
#define TPF_SRC(v) ((v) >= 0 ? (node < a.n_nodes ? node : 0) : 0)
#else
#define TPF_SRC(v) (v)
#endif
#if defined(MATTEN_LAB) && defined(MATTEN_ABLATE_NO_XLOAD)    // no neighbour-row load at all (the value depends on the index only)
#define TPF_XLD(xp, i, v) (1e-9f * (float)((v) + (i)))
#else
#define TPF_XLD(xp, i, v) ((xp)[i])
#endif
#if defined(MATTEN_LAB) && defined(MATTEN_ABLATE_NO_H2LOAD)
constexpr bool TPF_LAB_NO_H2LOAD = true;
#else
constexpr bool TPF_LAB_NO_H2LOAD = false;
#endif
#if defined(MATTEN_LAB) && defined(MATTEN_ABLATE_NO_MFMA)
constexpr bool TPF_LAB_NO_MFMA = true;
#else
constexpr bool TPF_LAB_NO_MFMA = false;
#endif
#if defined(MATTEN_LAB) && defined(MATTEN_ABLATE_NO_VALU)
constexpr bool TPF_LAB_NO_VALU = true;
#else
constexpr bool TPF_LAB_NO_VALU = false;
#endif
#if defined(MATTEN_LAB) && defined(MATTEN_ABLATE_NO_STORE)
constexpr bool TPF_LAB_NO_STORE = true;
#else
constexpr bool TPF_LAB_NO_STORE = false;
#endif
#if defined(MATTEN_LAB) && defined(MATTEN_ABLATE_NO_BARRIER)
constexpr bool TPF_LAB_NO_BARRIER = true;
#else
constexpr bool TPF_LAB_NO_BARRIER = false;
#endif
#if defined(MATTEN_LAB) && defined(MATTEN_ABLATE_NO_LOOP)
constexpr bool TPF_LAB_NO_LOOP = true;
#else
constexpr bool TPF_LAB_NO_LOOP = false;
#endif

// -DMATTEN_LAB -DTPF_TRACE: every contracting wave of a shared workgroup records where its cycles go (s_memtime at the
// phase boundaries of the chunk loop, summed per wave) into a buffer set through matten_lab_tp_trace (tp_fused.hip)
#if defined(MATTEN_LAB) && defined(TPF_TRACE)
#define TPF_TRACING 1
#else
#define TPF_TRACING 0
#endif

namespace matten_walk {

#if TPF_TRACING
__device__ __forceinline__ unsigned long long tpf_stamp() {
    unsigned long long t;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return t;
}
template <int N>
__device__ __forceinline__ unsigned long long tpf_stamp_after(float (&acc)[N]) {   // ... once the accumulators exist
    unsigned long long t;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)"
                 : "=s"(t), "+v"(acc[0]), "+v"(acc[N / 2]), "+v"(acc[N - 1]) : : "memory");
    return t;
}
#endif

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
#ifndef TPF_BRANCH_FREE_STEPS
#define TPF_BRANCH_FREE_STEPS 0
#endif
#ifndef TPF_EXACT_Y
#define TPF_EXACT_Y 0
#endif
#ifndef TPF_PUBLISH_EARLY
#define TPF_PUBLISH_EARLY 0
#endif
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
#if TPF_TRACING
    unsigned* trace;   // [waves of the launch][16]
#endif
};

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


// y[0..NY) = the staged row's harmonics [Y0, Y0 + NY): 16-byte reads over the aligned middle, 4- / 8-byte reads at the ragged
// ends -- exactly NY registers.  (Whole quads over the covering range left 3-4 dead registers per step that the allocator
// reused for the weights: a write-after-read on a load still in flight, i.e. a second LDS round trip per edge step.)
template <int Y0, int NY>
__device__ __forceinline__ void read_harmonics(const float* __restrict__ row, float (&y)[NY]) {
    constexpr int A = (Y0 + 3) / 4 * 4;                 // first 16-byte aligned column at or after Y0
    constexpr int B = (Y0 + NY) / 4 * 4;                // end of the aligned middle
    if constexpr (A >= B) {
#pragma unroll
        for (int j = 0; j < NY; ++j) y[j] = row[Y0 + j];
    } else {
        constexpr int HEAD = A - Y0, TAIL = Y0 + NY - B;
        if constexpr (HEAD == 1) y[0] = row[Y0];
        if constexpr (HEAD == 2) { const float2 v = *reinterpret_cast<const float2*>(row + Y0); y[0] = v.x, y[1] = v.y; }
        if constexpr (HEAD == 3) {
            y[0] = row[Y0];
            const float2 v = *reinterpret_cast<const float2*>(row + Y0 + 1);
            y[1] = v.x, y[2] = v.y;
        }
#pragma unroll
        for (int q = A; q < B; q += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(row + q);
            y[q - Y0] = v[0], y[q - Y0 + 1] = v[1], y[q - Y0 + 2] = v[2], y[q - Y0 + 3] = v[3];
        }
        if constexpr (TAIL == 1) y[NY - 1] = row[B];
        if constexpr (TAIL == 2) { const float2 v = *reinterpret_cast<const float2*>(row + B); y[NY - 2] = v.x, y[NY - 1] = v.y; }
        if constexpr (TAIL == 3) {
            const float2 v = *reinterpret_cast<const float2*>(row + B);
            y[NY - 3] = v.x, y[NY - 2] = v.y;
            y[NY - 1] = row[B + 2];
        }
    }
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
        if constexpr (TPF_LAB_NO_H2LOAD) return;
#pragma unroll
        for (int t = 0; t < TT; ++t)
            pf[t] = *reinterpret_cast<const f32x4*>(
                deg_ld[t] > 0 ? base + (int64_t)(beg_ld[t] + min(s0 + so_ld[t], deg_ld[t] - 1)) * row_bytes : safe);
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
#if TPF_TRACING
    const unsigned long long tr_in = tpf_stamp();
    unsigned tr_mfma = 0, tr_con = 0, tr_pub = 0, tr_bar = 0, tr_chunks = 0;
#endif
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
    // the weight block holds the columns of ALL couplings of the group, [u][c] (absent ones: zero columns)
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
#if TPF_TRACING
    const unsigned long long tr_loop = tpf_stamp();
    unsigned long long tr_a = tr_loop;
#endif
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
                if constexpr (TPF_LAB_NO_MFMA) {
                    if ((float)bh[0] == 12345.f) tile[c] = (float)bl[0];
                    continue;
                }
                float* trow = tile + (16 * t + c) * stride + 4 * g;
                // branch-free per tile count: the 3 MTC matrix instructions of a chunk interleave freely
                if (MTMAX == 1 || MT == 1) mfma_tiles<1>(ah, al, bh, bl, trow);
                else if (MTMAX == 2 || MT == 2) mfma_tiles<(MTMAX < 2 ? MTMAX : 2)>(ah, al, bh, bl, trow);
                else if (MTMAX == 3 || MT == 3) mfma_tiles<(MTMAX < 3 ? MTMAX : 3)>(ah, al, bh, bl, trow);
                else if (MTMAX == 4 || MT == 4) mfma_tiles<(MTMAX < 4 ? MTMAX : 4)>(ah, al, bh, bl, trow);
                else if (MTMAX == 5 || MT == 5) mfma_tiles<(MTMAX < 5 ? MTMAX : 5)>(ah, al, bh, bl, trow);
                else if (MTMAX == 6 || MT == 6) mfma_tiles<(MTMAX < 6 ? MTMAX : 6)>(ah, al, bh, bl, trow);
                else mfma_tiles<MTMAX>(ah, al, bh, bl, trow);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the weight tile is written
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_setprio(0);
#if TPF_TRACING
        const unsigned long long tr_b = tpf_stamp();
#endif
        auto contract = [&](int so, const float* __restrict__ x) {
            const float* wp = tile + ((j << ch_log2) + so) * stride + u * NC;
            float y[G::NY], w[NC];
            if constexpr (TPF_EXACT_Y) {
                read_harmonics<G::Y0, G::NY>(sb + ((j << ch_log2) + so) * STAGE_ROW + 32, y);
            } else {   // the harmonics of a staged row are 16-byte aligned: whole ds_read_b128 over [Y0, Y0 + NY)
                const float* yp = sb + ((j << ch_log2) + so) * STAGE_ROW + 32 + G::Y0;
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
        if constexpr (TPF_LAB_NO_VALU) {
        } else if (two_deep && TPF_BRANCH_FREE_STEPS) {
            // Branch-free slot pairs.  A step of a light kind is ~30-200 vector instructions behind ~8 LDS reads (harmonics
            // row, weights): inside `if (s < deg)` regions the reads of step s + 1 cannot start before step s has finished
            // and every step pays the LDS round trip (the l1 = 0 kind: 735 cycles per step for 30 instructions,
            // tools/tp_trace.py).  Here the reads are unconditional (every slot of a chunk has a staged row: past the end of
            // a segment its last edge again) and double-buffered one slot ahead; a slot past the segment's end contracts
            // x = 0 instead of being skipped (w, Y finite: + 0 to every accumulator).
            auto fetch = [&](int so, float (&y)[G::NY], float (&w)[NC]) {
                const float* wp = tile + ((j << ch_log2) + so) * stride + u * NC;
                const float* yp = sb + ((j << ch_log2) + so) * STAGE_ROW + 32 + G::Y0;
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
#pragma unroll
                for (int cc = 0; cc < NC; ++cc) w[cc] = wp[cc];
            };
            auto step = [&](bool on, const float (&x)[G::D1], const float (&y)[G::NY], const float (&w)[NC]) {
                float xe[G::D1];
#pragma unroll
                for (int i = 0; i < G::D1; ++i) xe[i] = on ? x[i] : 0.0f;
                G::apply(CMASK ? CMASK : mask, xe, y, w, acc);
            };
            float ya[G::NY], wa[NC], yb[G::NY], wb[NC];
            fetch(0, ya, wa);
            for (int so = 0; so < CH; so += 2) {   // CH is 2, 4 or 8 here: slot pairs, one row buffer per parity
                const int s = s0 + so;
                fetch(so + 1, yb, wb);
                step(s < deg, xn, ya, wa);
                {
                    const float* xp = a.x + (int64_t)TPF_SRC(src_nn) * a.d_in + xcol;
#pragma unroll
                    for (int i = 0; i < G::D1; ++i) xn[i] = TPF_XLD(xp, i, src_nn);
                    src_nn = a.src_sorted[min(beg + s + 4, e_last)];
                }
                fetch(min(so + 2, CH - 1), ya, wa);     // (the last pair re-reads its own slot: never used)
                step(s + 1 < deg, xb, yb, wb);
                {
                    const float* xp = a.x + (int64_t)TPF_SRC(src_b) * a.d_in + xcol;
#pragma unroll
                    for (int i = 0; i < G::D1; ++i) xb[i] = TPF_XLD(xp, i, src_b);
                    src_b = a.src_sorted[min(beg + s + 5, e_last)];
                }
            }
        } else if (two_deep) {
            auto refill_n = [&](int s) {
                const float* xp = a.x + (int64_t)TPF_SRC(src_nn) * a.d_in + xcol;
#pragma unroll
                for (int i = 0; i < G::D1; ++i) xn[i] = TPF_XLD(xp, i, src_nn);
                src_nn = a.src_sorted[min(beg + s + 4, e_last)];
            };
            auto refill_b = [&](int s) {
                const float* xp = a.x + (int64_t)TPF_SRC(src_b) * a.d_in + xcol;
#pragma unroll
                for (int i = 0; i < G::D1; ++i) xb[i] = TPF_XLD(xp, i, src_b);
                src_b = a.src_sorted[min(beg + s + 5, e_last)];
            };
            for (int so = 0; so < CH; so += 2) {   // CH is 2, 4 or 8 here: slot pairs, one row buffer per parity
                const int s = s0 + so;
                const bool last = TPF_PUBLISH_EARLY && so + 2 >= CH;
                if (s < deg) contract(so, xn);
                if (!last) refill_n(s);
                if (s + 1 < deg) contract(so + 1, xb);
                if (!last) refill_b(s);
            }
            if (TPF_PUBLISH_EARLY) {
                // the stage rows requested at the top of the chunk go to LDS BEFORE the chunk's last gathers are issued: the
                // publish waits on the vector-memory counter, which retires in order -- behind freshly issued gathers it
                // waited for THEIR round trip too (400-500 cycles per chunk of the l1 <= 1 kinds, tools/tp_trace.py)
                ld.publish(buf ^ 1);
                refill_n(s0 + CH - 2);
                refill_b(s0 + CH - 2);
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
#if TPF_TRACING
        const unsigned long long tr_c = tpf_stamp_after(acc);
#endif
        if (!(two_deep && TPF_PUBLISH_EARLY && !TPF_BRANCH_FREE_STEPS)) ld.publish(buf ^ 1);
#if TPF_TRACING
        const unsigned long long tr_d = tpf_stamp();
#endif
        if constexpr (TPF_LAB_NO_BARRIER) {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        } else {
            __syncthreads();  // the next stage is published; every wave is done with this chunk's rows
        }
#if TPF_TRACING
        const unsigned long long tr_e = tpf_stamp();
        tr_mfma += (unsigned)(tr_b - tr_a), tr_con += (unsigned)(tr_c - tr_b), tr_pub += (unsigned)(tr_d - tr_c);
        tr_bar += (unsigned)(tr_e - tr_d), tr_chunks += 1;
        tr_a = tr_e;
#endif
    }
#if TPF_TRACING
    const unsigned long long tr_end = tpf_stamp();
#endif
    epi.template store<G>(a, ge, acc, a_scale_inv, node, j, u, valid);
#if TPF_TRACING
    if (a.trace && lane == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long tr_out = tpf_stamp();
        unsigned* tr = a.trace + ((size_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6)) * 16;
        tr[0] = 1u, tr[1] = (unsigned)ge.kind, tr[2] = (unsigned)cu_log2 | (PAIRED ? 256u : 0u) | ((unsigned)TT << 12) | ((unsigned)MT << 16);
        tr[3] = tr_chunks, tr[4] = (unsigned)(tr_loop - tr_in), tr[5] = tr_mfma, tr[6] = tr_con, tr[7] = tr_pub, tr[8] = tr_bar;
        tr[9] = (unsigned)(tr_out - tr_end), tr[10] = (unsigned)(tr_out - tr_in), tr[11] = (unsigned)ge.mask;
        tr[12] = (unsigned)(tr_in & 0xffffffffu), tr[13] = (unsigned)(tr_in >> 32);
    }
#endif
}

// Coupling masks worth a specialisation: only the scalar-block kind (all five couplings, or l2 <= 3 when the target has
// no 4o) -- for every other kind the extra instantiations cost the rest of the kernel more than they gain (DESIGN.md
// round 2: the register allocation of this one function is shared by all kinds).
template <int L1, int GI> struct HotMask { static constexpr unsigned M0 = 0, M1 = 0; };
#ifndef TPF_NO_HOT_MASKS
template <> struct HotMask<0, 0> { static constexpr unsigned M0 = 0x1f, M1 = 0xf; };
#endif

}  // namespace matten_walk
