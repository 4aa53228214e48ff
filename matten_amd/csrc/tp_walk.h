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
    defined(MATTEN_ABLATE_NO_BARRIER) || defined(MATTEN_ABLATE_NO_LOOP) || defined(MATTEN_ABLATE_NO_EPI) || defined(MATTEN_ABLATE_ONE_YQUAD)
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
// low word of s_memtime (phase lengths are differences of nearby stamps; 64-bit stamps that get spilled trip a gfx950 backend
// check: "Subtarget requires even aligned vector registers" on the scratch reload of an odd register pair)
__device__ __forceinline__ unsigned tpf_stamp() {
    unsigned long long t;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return (unsigned)t;
}
__device__ __forceinline__ unsigned tpf_stamp_hi() {
    unsigned long long t;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return (unsigned)(t >> 32);
}
template <int N>
__device__ __forceinline__ unsigned tpf_stamp_after(float (&acc)[N]) {   // ... once the accumulators exist
    unsigned long long t;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)"
                 : "=s"(t), "+v"(acc[0]), "+v"(acc[N / 2]), "+v"(acc[N - 1]) : : "memory");
    return (unsigned)t;
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
#ifndef TPF_W_FIRST
#define TPF_W_FIRST 1    // a step's weights are requested before its harmonics (one LDS round trip per step instead of two: -0.4 %)
#endif
#ifndef TPF_W_FIRST2
#define TPF_W_FIRST2 1   // the same in the two-slot pass of the vector input blocks (-0.7 %)
#endif
#ifndef TPF_PAIR_SUM
#define TPF_PAIR_SUM 1   // vector input blocks (l1 = 1): two edge slots per pass, pair products summed before the coefficients
#endif
#ifndef TPF_EXACT_Y0
#define TPF_EXACT_Y0 0
#endif

#ifndef TPF_REP_PREFETCH
#define TPF_REP_PREFETCH 1   // persistent units: the next node group's loads go out in front of this group's agg stores
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
#if TPF_TRACING
    unsigned* trace;   // [waves of the launch][16]
#endif
};


// the destination nodes of a unit: node group r of its tile -> (node of this lane, does the lane contract, its CSR segment,
// the longest segment of the wave).  Every lane of a node (idle channel lanes too) knows the segment: the MFMA role needs it.
struct UnitNodes {
    int node, beg, deg, maxdeg;
    bool valid;
};
struct RawNodes {   // the same before its two row-pointer loads are consumed (a persistent unit requests its NEXT group's early)
    int node, b0, b1;
    bool valid;
};
__device__ __forceinline__ RawNodes unit_nodes_issue(const Args& a, const GroupEntry& ge, int tile_id, int r, int lane) {
    const int cu_log2 = ge.cu_log2;
    const int nodes_per_wave = 64 >> cu_log2;
    const int g_in_tile = r * nodes_per_wave + (lane >> cu_log2);
    const int u = lane & ((1 << cu_log2) - 1);
    RawNodes n;
    n.node = tile_id * TILE_NODES + g_in_tile;
    const bool in_range = (g_in_tile < TILE_NODES) && (n.node < a.n_nodes);
    n.valid = in_range && (u < ge.mul);
    n.b0 = 0, n.b1 = 0;
    if (in_range) {
        n.b0 = a.rowptr[n.node];
        n.b1 = a.rowptr[n.node + 1];
    }
    return n;
}
__device__ __forceinline__ UnitNodes unit_nodes_finish(const RawNodes& raw) {
    UnitNodes n;
    n.node = raw.node, n.valid = raw.valid, n.beg = raw.b0, n.deg = raw.b1 - raw.b0;
    int maxdeg = n.deg;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) maxdeg = max(maxdeg, __shfl_xor(maxdeg, off));
    if constexpr (TPF_LAB_NO_LOOP) maxdeg = 0;  // timing build: prologue (segment, gather, fragment and stage set-up) + epilogue only
    n.maxdeg = maxdeg;
    return n;
}
__device__ __forceinline__ UnitNodes unit_nodes(const Args& a, const GroupEntry& ge, int tile_id, int r, int lane) {
    return unit_nodes_finish(unit_nodes_issue(a, ge, tile_id, r, lane));
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

// loader-only unit of a paired workgroup (same barrier sequence as run_group_shared<PAIRED>, node group after node group)
__device__ __forceinline__ void run_loader_only_paired(const Args& a, const GroupEntry& ge, float* __restrict__ stage,
                                                       int tile_id, int r0, int reps, int lane) {
    for (int rep = 0; rep < reps; ++rep) {
        const UnitNodes un = unit_nodes(a, ge, tile_id, r0 + 2 * rep, lane);
        PairLoader ld;
        ld.init(a, stage, reinterpret_cast<int*>(stage + STAGE_TOTAL_FLOATS), ge.cu_log2, un.beg, un.deg, un.maxdeg);
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
}

// a unit that only feeds the stage (same barrier sequence as run_group_shared)
template <int TT>
__device__ __forceinline__ void run_loader_only_t(const Args& a, const GroupEntry& ge, float* __restrict__ stage, int tile_id,
                                                  int r0, int reps, int lane) {
    for (int rep = 0; rep < reps; ++rep) {
        const UnitNodes un = unit_nodes(a, ge, tile_id, r0 + rep, lane);
        StageLoader<TT> ld;
        ld.init(a, stage, ge.cu_log2, un.beg, un.deg);
        ld.issue(0);
        ld.publish(0);
        __syncthreads();
        int buf = 0;
        for (int s0 = 0; s0 < un.maxdeg; s0 += ld.CH, buf ^= 1) {
            ld.issue(s0 + ld.CH);
            ld.publish(buf ^ 1);
            __syncthreads();
        }
    }
}
__device__ __forceinline__ void run_loader_only(const Args& a, const GroupEntry& ge, float* __restrict__ stage, int tile_id,
                                                int r0, int reps, int lane) {
    if ((64 >> ge.cu_log2) > 16) run_loader_only_t<2>(a, ge, stage, tile_id, r0, reps, lane);
    else run_loader_only_t<1>(a, ge, stage, tile_id, r0, reps, lane);
}

// the kinds with registers to spare for a second neighbour row in flight (two-slot chunks: 8 lanes per node)
template <int L1, int GI>
struct TwoDeepOk { static constexpr bool value = L1 == 0 || (L1 == 1 && (GI == 0 || TPF_PAIR_SUM)); };

// TWO_DEEP is a template parameter, not a run-time flag: with both gather schedules in one instantiation the compiler
// reconciled their register assignments with 50-90 v_mov per CHUNK (60 % of the l1 = 0 kind's vector instructions).
// CMASK != 0: the entry's coupling mask as a compile-time constant (HotMask below); the couplings of a step then form
// one basic block instead of NC uniformly-branched ones
// PERSISTENT units: a wave walks `reps` node groups of its tile one after the other (r0, r0 + 1, ...; paired workgroups:
// r0, r0 + 2, ...) -- the A fragments are fetched once, and the agg stores of one node group drain behind the walk of the
// next instead of holding the wave's slot until they are acknowledged (s_endpgm waits for them: 9-13 % of a wave's life,
// tools/tp_trace.py).  The host picks reps per lanes-per-node class (plan.fused_unit_map).
// TWO_DEEP = 2 (scalar input blocks, 2 or 4 edge slots per chunk): the gather runs a WHOLE CHUNK ahead -- at the top of chunk c the
// rows of chunk c + 1 and the source indices of chunk c + 2 are requested, four of each per lane.  A scalar block's edge step is
// ~45 vector instructions (~200 cycles of issue): a row requested two steps ahead was asked for ~1300 cycles before its use, less
// than a gather's latency under load, and the wave waited on every step (phase trace, round 5: 2100-3200 cycles of contraction
// per four-slot chunk for 150-190 instructions).  The two row buffers swap roles by a two-fold unrolled chunk loop, not by moves:
// a move out of a register with a load in flight would wait for it.
template <int L1, int GI, int TT, int TWO_DEEP, bool PAIRED, unsigned CMASK, class Epilogue>
__device__ __forceinline__ void run_group_shared(const Args& a, const GroupEntry& ge, float* __restrict__ tile,
                                                 float* __restrict__ stage, int entry, int tile_id, int r0, int reps, int lane,
                                                 const Epilogue& epi) {
    static_assert(!PAIRED || TT == 1, "paired workgroups stage 2 x 16 rows");
#if TPF_TRACING
    const unsigned tr_in_hi = tpf_stamp_hi();
    const unsigned tr_in = tpf_stamp();     // (covers every node group of a persistent unit)
    unsigned tr_mfma = 0, tr_con = 0, tr_pub = 0, tr_bar = 0, tr_chunks = 0, tr_pro = 0, tr_epi = 0;
    unsigned tr_pro_pre = 0, tr_epi_sg = 0;   // of the prologue: up to its barrier; of the epilogue: the next group's start_group
#endif
    const int row0 = PAIRED ? 16 * (int)(threadIdx.x >> 7) : 0;       // this wave's half of a paired stage
    constexpr int STAGE_BUF = (PAIRED ? 32 : 16 * TT) * STAGE_ROW;    // floats per stage buffer
    using G = matten::Group<L1, GI>;
    constexpr int NC = G::NC;

    const unsigned mask = ge.mask;
    const int cu_log2 = ge.cu_log2;              // >= 1 here: at most 32 nodes per wave
    const int cu = 1 << cu_log2;
    const int npw = 64 >> cu_log2;
    const int ch_log2 = npw >= 16 ? 0 : 4 - (6 - cu_log2);
    const int CH = 1 << ch_log2;
    // the weight block holds the columns of ALL couplings of the group, [u][c] (absent ones: zero columns)
    const int ncols = ge.mul * NC;
#if defined(MATTEN_LAB) && defined(TPF_ONLY_MT)
    const int MT = TPF_ONLY_MT;   // tools/isa_mix.py: the tile count of the one entry kind in the object, at compile time
#else
    const int MT = (ncols + 15) >> 4;
#endif

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

    // Per node group: (i) start_group -- everything of the group's prologue that is a LOAD (segment bounds, the first chunk's
    // stage rows, the first neighbour rows) -- and (ii) the walk + the agg stores.  A persistent unit runs (i) of group
    // rep + 1 BEFORE the stores of group rep: vector-memory operations retire in order, so loads issued behind ~100 stores
    // would wait for every store's acknowledgement, while loads issued in front of them land under the stores' shadow.
    // The row pointers of group rep + 1 are requested at the top of group rep's walk.
    typename std::conditional<PAIRED, PairLoader, StageLoader<TT>>::type ld;
    static_assert(!TWO_DEEP || TwoDeepOk<L1, GI>::value, "two rows in flight only for the light kinds");
    static_assert(TWO_DEEP != 2 || G::D1 == 1, "a chunk of rows in flight: scalar input blocks");
    constexpr bool two_deep = TWO_DEEP == 1;   // the caller guarantees an even number of slots per chunk
    constexpr bool chunk_deep = TWO_DEEP == 2; // the caller guarantees 2 or 4 slots per chunk
    constexpr int XQ = 4;                      // rows / indices a lane keeps per chunk in chunk_deep mode
    float xqa[XQ], xqb[XQ];
    int sq[XQ];
    int node, beg, deg, maxdeg, e_last;
    bool valid;
    float xn[G::D1], xb[G::D1];
    int src_nn, src_b = 0;
    auto start_group = [&](const UnitNodes& un) {
        node = un.node, beg = un.beg, valid = un.valid, maxdeg = un.maxdeg;
        deg = valid ? un.deg : 0;
        if constexpr (PAIRED) {
            ld.init(a, stage, reinterpret_cast<int*>(stage + STAGE_TOTAL_FLOATS), cu_log2, un.beg, un.deg, un.maxdeg);
            maxdeg = ld.maxdeg;   // both node groups walk the same number of chunks (one barrier sequence)
        } else {
            ld.init(a, stage, cu_log2, un.beg, un.deg);
        }
        ld.issue(0);
        // Neighbour gather.  Two-slot chunks (8 nodes per wave: short steps) keep TWO rows in flight, one per slot of the
        // chunk: a row is refilled right after its contraction with the edge two steps on, so a gather has a whole step,
        // the stage hand-over and the next MFMA phase to land, and nothing is copied.  Other chunk shapes (long steps) keep
        // the one-step-ahead pipeline: source index two edges ahead, row one edge ahead.  All loads are unconditional on a
        // clamped edge index.
        // (the second row buffer is compiled in only for the kinds with registers to spare: it costs the heavy kinds spills)
        e_last = deg > 0 ? beg + deg - 1 : 0;
        if constexpr (chunk_deep) {
            int s0q[XQ];
#pragma unroll
            for (int i = 0; i < XQ; ++i) s0q[i] = a.src_sorted[min(beg + i, e_last)];
#pragma unroll
            for (int i = 0; i < XQ; ++i) sq[i] = a.src_sorted[min(beg + CH + i, e_last)];
#pragma unroll
            for (int i = 0; i < XQ; ++i) {
                const float* xp = a.x + (int64_t)TPF_SRC(s0q[i]) * a.d_in + xcol;
                xqa[i] = TPF_XLD(xp, 0, s0q[i]);
            }
            return;
        }
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
    };
    constexpr int RSTEP = PAIRED ? 2 : 1;
    start_group(unit_nodes(a, ge, tile_id, r0, lane));
    for (int rep = 0; rep < reps; ++rep) {
#if TPF_TRACING
    const unsigned tr_rep = rep == 0 ? tr_in : tpf_stamp();
#endif
    const bool more = rep + 1 < reps;
    const RawNodes raw_next = unit_nodes_issue(a, ge, tile_id, r0 + RSTEP * (rep + (more ? 1 : 0)), lane);
    float acc[G::NACC];
#pragma unroll
    for (int k = 0; k < G::NACC; ++k) acc[k] = 0.0f;
    ld.publish(0);
#if TPF_TRACING
    tr_pro_pre += (unsigned)(tpf_stamp() - tr_rep);
#endif
    __syncthreads();
#if TPF_TRACING
    const unsigned tr_loop = tpf_stamp();
    unsigned tr_a = tr_loop;
    tr_pro += (unsigned)(tr_loop - tr_rep);
#endif
    auto chunk_body = [&](const int s0, const int buf, float (&xcur)[XQ], float (&xnext)[XQ]) {
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
        const unsigned tr_b = tpf_stamp();
#endif
        auto contract = [&](int so, const float* __restrict__ x) {
            const float* wp = tile + ((j << ch_log2) + so) * stride + u * NC;
            const float* yp = sb + ((j << ch_log2) + so) * STAGE_ROW + 32 + G::Y0;
            float y[G::NY], w[NC];
            if constexpr (TPF_W_FIRST) {
                // the weights are requested BEFORE the harmonics: whole quads over [Y0, Y0 + NY) leave up to four dead registers,
                // and weights requested afterwards land IN them -- a write-after-read on a load in flight, i.e. a second LDS
                // round trip per edge step (seen in the ISA of the l1 = 0 kind, docs/LAB_NOTES.md round 5)
#pragma unroll
                for (int cc = 0; cc < NC; ++cc) w[cc] = wp[cc];
                asm volatile("" ::: "memory");
            }
            if constexpr (TPF_EXACT_Y0 && L1 == 0 && G::Y0 == 0 && G::NY % 4 == 1) {
                // scalar blocks: NY = 25 = six quads + one float read on its own -- whole quads leave three dead registers
                // that the allocator reuses for the weights: a write-after-read on a load in flight, a second LDS round trip
                y[G::NY - 1] = yp[G::NY - 1];
#pragma unroll
                for (int q = 0; q < G::NY / 4; ++q) {
                    const f32x4 v = reinterpret_cast<const f32x4*>(yp)[q];
                    y[4 * q] = v[0], y[4 * q + 1] = v[1], y[4 * q + 2] = v[2], y[4 * q + 3] = v[3];
                }
            } else
            {   // the harmonics of a staged row are 16-byte aligned: whole ds_read_b128 over [Y0, Y0 + NY)
                constexpr int Q0 = G::Y0 / 4 * 4, NQ = (G::Y0 + G::NY - Q0 + 3) / 4;
                float yq[4 * NQ];
                const f32x4* y4 = reinterpret_cast<const f32x4*>(yp - (G::Y0 - Q0));
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
#if defined(MATTEN_LAB) && defined(MATTEN_ABLATE_ONE_YQUAD)   // timing build: ONE LDS read of harmonics per edge step (wrong results)
                    const f32x4 v0 = y4[0];
                    const f32x4 v = q == 0 ? v0 : v0 * (float)(q + 1);
#else
                    const f32x4 v = y4[q];
#endif
                    yq[4 * q] = v[0], yq[4 * q + 1] = v[1], yq[4 * q + 2] = v[2], yq[4 * q + 3] = v[3];
                }
#pragma unroll
                for (int jj = 0; jj < G::NY; ++jj) y[jj] = yq[G::Y0 - Q0 + jj];
            }
            if constexpr (!TPF_W_FIRST) {
#pragma unroll
                for (int cc = 0; cc < NC; ++cc) w[cc] = wp[cc];
            }
            G::apply(CMASK ? CMASK : mask, x, y, w, acc);
        };
        if constexpr (TPF_LAB_NO_VALU) {
        } else if constexpr (TWO_DEEP && TPF_PAIR_SUM && G::HAS_APPLY2) {
            // Vector input blocks: the two edge slots of a pair in ONE pass (cg_gen.h CG2: the pair products of both edges
            // are added before they meet the coupling coefficients, -22 / -25 % vector instructions per pair).  The harmonics
            // and weights of both slots are read unconditionally (every slot of a chunk has a staged row: past a segment's
            // end its last edge again); a slot past the end contracts x = 0 -- an exact + 0 to every accumulator.
            auto fetch = [&](int so, float (&y)[G::NY], float (&w)[NC]) {
                const float* wp = tile + ((j << ch_log2) + so) * stride + u * NC;
                const float* yp = sb + ((j << ch_log2) + so) * STAGE_ROW + 32 + G::Y0;
                if constexpr (TPF_W_FIRST2) {   // weights before harmonics: see `contract` below
#pragma unroll
                    for (int cc = 0; cc < NC; ++cc) w[cc] = wp[cc];
                    asm volatile("" ::: "memory");
                }
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
                if constexpr (!TPF_W_FIRST2) {
#pragma unroll
                    for (int cc = 0; cc < NC; ++cc) w[cc] = wp[cc];
                }
            };
            for (int so = 0; so < CH; so += 2) {
                const int s = s0 + so;
                float ya[G::NY], wa[NC], yb[G::NY], wb[NC], xa[G::D1], xc[G::D1];
                fetch(so, ya, wa);
                fetch(so + 1, yb, wb);
#pragma unroll
                for (int i = 0; i < G::D1; ++i) xa[i] = s < deg ? xn[i] : 0.0f, xc[i] = s + 1 < deg ? xb[i] : 0.0f;
                {
                    const float* xp = a.x + (int64_t)TPF_SRC(src_nn) * a.d_in + xcol;
#pragma unroll
                    for (int i = 0; i < G::D1; ++i) xn[i] = TPF_XLD(xp, i, src_nn);
                    src_nn = a.src_sorted[min(beg + s + 4, e_last)];
                }
                {
                    const float* xp = a.x + (int64_t)TPF_SRC(src_b) * a.d_in + xcol;
#pragma unroll
                    for (int i = 0; i < G::D1; ++i) xb[i] = TPF_XLD(xp, i, src_b);
                    src_b = a.src_sorted[min(beg + s + 5, e_last)];
                }
                G::apply2(CMASK ? CMASK : mask, xa, ya, wa, xc, yb, wb, acc);
            }
        } else if constexpr (chunk_deep) {
            // rows of the NEXT chunk (their indices arrived a chunk ago), then the indices of the chunk after it
#pragma unroll
            for (int i = 0; i < XQ; ++i) {
                const float* xp = a.x + (int64_t)TPF_SRC(sq[i]) * a.d_in + xcol;
                xnext[i] = TPF_XLD(xp, 0, sq[i]);
            }
#pragma unroll
            for (int i = 0; i < XQ; ++i) sq[i] = a.src_sorted[min(beg + s0 + 2 * CH + i, e_last)];
#pragma unroll
            for (int so = 0; so < XQ; ++so) {
                if (so < CH && s0 + so < deg) contract(so, xcur + so);
            }
        } else if (two_deep) {
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
#if TPF_TRACING
        const unsigned tr_c = tpf_stamp_after(acc);
#endif
        ld.publish(buf ^ 1);
#if TPF_TRACING
        const unsigned tr_d = tpf_stamp();
#endif
        if constexpr (TPF_LAB_NO_BARRIER) {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        } else {
            __syncthreads();  // the next stage is published; every wave is done with this chunk's rows
        }
#if TPF_TRACING
        const unsigned tr_e = tpf_stamp();
        tr_mfma += (unsigned)(tr_b - tr_a), tr_con += (unsigned)(tr_c - tr_b), tr_pub += (unsigned)(tr_d - tr_c);
        tr_bar += (unsigned)(tr_e - tr_d), tr_chunks += 1;
        tr_a = tr_e;
#endif
    };
    if constexpr (chunk_deep) {
        for (int s0 = 0; s0 < maxdeg; s0 += 2 * CH) {
            chunk_body(s0, 0, xqa, xqb);
            if (s0 + CH < maxdeg) chunk_body(s0 + CH, 1, xqb, xqa);   // (wave-uniform; an odd tail leaves the next group's rows to start_group: buffer a)
        }
    } else {
        int buf = 0;
        for (int s0 = 0; s0 < maxdeg; s0 += CH, buf ^= 1) chunk_body(s0, buf, xqa, xqb);
    }
#if TPF_TRACING
    const unsigned tr_end = tpf_stamp();
#endif
    const int node_done = node;
    const bool valid_done = valid;
    if (TPF_REP_PREFETCH && more) start_group(unit_nodes_finish(raw_next));
#if TPF_TRACING
    tr_epi_sg += (unsigned)(tpf_stamp() - tr_end);
#endif
    epi.template store<G>(a, ge, acc, a_scale_inv, node_done, j, u, valid_done);
    if (!TPF_REP_PREFETCH && more) start_group(unit_nodes_finish(raw_next));
#if TPF_TRACING
    if (a.trace && lane == 0) {
        // the LAST node group's stores are waited for (as s_endpgm does); the earlier ones drain behind the next group's walk
        if (rep + 1 == reps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned tr_out = tpf_stamp();
        tr_epi += (unsigned)(tr_out - tr_end);
        if (rep + 1 == reps) {
            unsigned* tr = a.trace + ((size_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6)) * 16;
            tr[0] = 1u, tr[1] = (unsigned)ge.kind, tr[2] = (unsigned)cu_log2 | (PAIRED ? 256u : 0u) | ((unsigned)reps << 9) | ((unsigned)TT << 12) | ((unsigned)MT << 16);
            tr[3] = tr_chunks, tr[4] = tr_pro, tr[5] = tr_mfma, tr[6] = tr_con, tr[7] = tr_pub, tr[8] = tr_bar;
            tr[9] = tr_epi, tr[10] = (unsigned)(tr_out - tr_in), tr[11] = (unsigned)ge.mask;
            tr[12] = tr_in, tr[13] = tr_in_hi, tr[14] = tr_pro_pre, tr[15] = tr_epi_sg;
        }
    }
#endif
    }   // rep
}

// Coupling masks worth a specialisation: only the scalar-block kind (all five couplings, or l2 <= 3 when the target has
// no 4o) -- for every other kind the extra instantiations cost the rest of the kernel more than they gain (DESIGN.md
// round 2: the register allocation of this one function is shared by all kinds).
template <int L1, int GI> struct HotMask { static constexpr unsigned M0 = 0, M1 = 0; };
#ifndef TPF_NO_HOT_MASKS
template <> struct HotMask<0, 0> { static constexpr unsigned M0 = 0x1f, M1 = 0xf; };
#endif
#ifndef TPF_HOT_ALL
#define TPF_HOT_ALL 1   // the coupling masks of the paper model's full layers at compile time too: one basic block per edge step instead of
#endif                  // one uniform branch per coupling (round 6, same-box A/B: 3.818 -> 3.776 ms per forward, layer 2 0.70 -> 0.67)
#if TPF_HOT_ALL
template <> struct HotMask<1, 0> { static constexpr unsigned M0 = 0x7f, M1 = 0; };
template <> struct HotMask<1, 1> { static constexpr unsigned M0 = 0x1b, M1 = 0xf; };
template <> struct HotMask<2, 0> { static constexpr unsigned M0 = 0x7ff, M1 = 0x6ff; };
template <> struct HotMask<2, 1> { static constexpr unsigned M0 = 0x1d, M1 = 0xf; };
template <> struct HotMask<3, 0> { static constexpr unsigned M0 = 0x3ff, M1 = 0x37f; };
template <> struct HotMask<3, 1> { static constexpr unsigned M0 = 0x7f, M1 = 0x3f; };
template <> struct HotMask<4, 0> { static constexpr unsigned M0 = 0xfb, M1 = 0; };
template <> struct HotMask<4, 1> { static constexpr unsigned M0 = 0x7d, M1 = 0; };
#endif
#ifndef TPF_HOT_ALT
#define TPF_HOT_ALT 1
#endif
#if TPF_HOT_ALT
// the alternative groups (plan._ALT_GROUPS: the even-l3 couplings of an odd- / even-parity block) are taken by blocks that hold
// ALL their couplings: the full mask at compile time (last conv layer 0.706 -> 0.68 ms with the first of them)
template <> struct HotMask<0, 1> { static constexpr unsigned M0 = 0x7, M1 = 0; };    // (the lmax-2 lists: 0,1  1,4  2,4)
template <> struct HotMask<1, 4> { static constexpr unsigned M0 = 0x3f, M1 = 0; };
template <> struct HotMask<2, 4> { static constexpr unsigned M0 = 0x3f, M1 = 0; };
template <> struct HotMask<1, 2> { static constexpr unsigned M0 = 0xf, M1 = 0; };
template <> struct HotMask<1, 3> { static constexpr unsigned M0 = 0x3, M1 = 0; };
template <> struct HotMask<2, 2> { static constexpr unsigned M0 = 0x7, M1 = 0; };
template <> struct HotMask<2, 3> { static constexpr unsigned M0 = 0x3f, M1 = 0; };
template <> struct HotMask<3, 2> { static constexpr unsigned M0 = 0x1f, M1 = 0; };
template <> struct HotMask<3, 3> { static constexpr unsigned M0 = 0xf, M1 = 0; };
template <> struct HotMask<4, 2> { static constexpr unsigned M0 = 0x7, M1 = 0; };
template <> struct HotMask<4, 3> { static constexpr unsigned M0 = 0x3f, M1 = 0; };
#endif

}  // namespace matten_walk
