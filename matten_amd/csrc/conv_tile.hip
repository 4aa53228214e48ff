// Conv layer on 16-node single-species tiles: tensor product + neighbour sum + lin2 + self-connection in one kernel, the
// neighbour sums agg[N, d_mid] never leave the chip.
//   reference nn/conv.py:113-123      msg = tp(x[src], edge_attrs, edge_embedding); agg = scatter(msg, dst) / sqrt(avg)
//                                     out = lin2(agg, species) + self_connection
//
// Why: agg is written by matten_tp_fused and read back by matten_agg_linear -- ~1 GB each way per layer at 64 000 nodes,
// ~70 % of the forward's HBM traffic.  Two earlier fused attempts lost because lin2's weights are indexed by
// the DESTINATION species: 27-41 KB of weights per node against 12 KB of agg saved.  Here a workgroup owns a TILE of 16
// destination nodes of ONE species (matten_species_tiles below: blocks of ~32 crystals, every (block, species) run padded
// to 16), so lin2 is a matrix product per tile with the species' weights as ready A fragments, and the x[src] gathers of
// a tile stay inside a ~2 MB L2 footprint.
//
// The workgroup (4 waves) walks the layer's group entries in ROUNDS: four entries of one class on one node group of the
// tile, i.e. the workgroup-shared walk of matten_tp_fused (tp_walk.h: same staging, same fp16-split matrix products for
// the last radial layer, same literal-coefficient contraction, same per-node summation order).  After a round the waves
// park their sums in LDS register by register -- dump[reg][lane], conflict-free stores, in passes of DUMP_REGS registers
// cut at coupling boundaries so that the dump fits the LDS the walk has just released -- and lin2 is applied there:
//     out[n, io, v, k] += sum_u A_p[u, v] * acc_e[n, u, (c, k)]            p = path (entry e, coupling c) -> io
// as v_mfma_f32_16x16x4_f32: M = 16 output channels v, K = 4 channels u per step, N = 16 (node, component) columns.
// Lane (g, col) reads the KS = lanes-per-node / 4 channels g*KS.. of its column from the dump with ONE ds_read (they are
// neighbouring lanes of one register), the A fragments come from a per-species table with one coalesced load per (piece,
// output tile).  A UNIT (output irrep, 16-channel tile, <= 4 column tiles) belongs to one wave of the round (balanced by
// the host, plan_conv.py): it sums every piece of the round that ends in that irrep in registers and adds the result to
// the tile's output rows in LDS.  Fixed order everywhere, no atomics: a node's result does not depend on its tile mates
// (matrix columns are independent), so tiles may be composed in any order.
// The rows start as the self-connection and leave as lin2's output; with the Gate tables (cmeta) the e3nn Gate and the
// eval-mode BatchNorm that follow the conv (reference nn/conv.py:209-213) are applied while the rows are written.
#include <atomic>

#include "tp_walk.h"

namespace {

using namespace matten_walk;

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int CT_NODES = 16;      // == plan_conv.TILE_NODES
constexpr int CT_DUMP_RS = 68;    // == plan_conv.DUMP_RS
constexpr int CT_DUMP_REGS = 28;  // == plan_conv.DUMP_REGS
constexpr int CT_MAX_NT = 4;      // == plan_conv.UNIT_MAX_NT
#ifndef CT_MIN_BLOCKS
#define CT_MIN_BLOCKS 3
#endif
#if defined(MATTEN_LAB) && defined(CT_ABLATE_NO_LIN2)    // timing build: the lin2 phases compiled out (barriers stay)
constexpr bool CT_LAB_NO_LIN2 = true;
#else
constexpr bool CT_LAB_NO_LIN2 = false;
#endif
#if defined(MATTEN_LAB) && defined(CT_ABLATE_NO_ALOAD)   // timing build: the A fragments are constants (no global load in the lin2 phase)
constexpr bool CT_LAB_NO_ALOAD = true;
#else
constexpr bool CT_LAB_NO_ALOAD = false;
#endif
#if defined(MATTEN_LAB) && defined(CT_ABLATE_NO_DUMP)    // timing build: the accumulators are not parked either
constexpr bool CT_LAB_NO_DUMP = true;
#else
constexpr bool CT_LAB_NO_DUMP = false;
#endif

struct CArgs {
    const int* tile_nodes;     // [n_slots, 16] node ids, -1 = padding
    const int* tile_species;   // [n_slots] species of the tile, -1 = empty slot
    const int4* quads;         // [n_quads, 2] {e0, e1, e2, e3} {class lanes per node (log2), passes, node groups, wave_units base}
    const int2* rounds;        // [n_rounds] {quad, node group}: the order the tile walks them (rounds over the same rows adjacent)
    const int* frag_recs;      // [n_frag]   A offset / 64 | first register << 14 | dump wave << 19 | lanes per node (log2) << 21 | last of unit << 24
    const int2* unit_recs;     // [n_unit]   {out col | d3 << 12 | valid v << 16 | nt0 << 21 | n_nt << 25, log2 nodes per wave | class lanes (log2) << 4}
    const int2* phase_recs;    // [n_phase]  {first fragment | count << 16, first unit} of (quad, pass, wave) at base + 4 pass + wave
    const float* atab;         // [n_species, a_stride]
    const float* add;          // [N, add_ld] self-connection, or NULL
    float* out;                // [N, out_gld]
    const int4* cmeta;         // Gate: per ACTIVATED column {source column, gate column or -1, act | gate act << 8, bn index | mean index << 16}, or NULL
    const float* act_cst;
    const float* bn_scale;     // [d_act] or NULL
    const float* bn_shift;
    int n_rounds, a_stride, n_slots, slots_per_block, add_ld, out_gld, d_out, d_act, out_ld, walk_floats, n_frag, n_unit, n_phase;
};

// ---- compile-time pass layout of a group kind: couplings in order, a new pass when the next one does not fit ----------
template <class G>
struct DumpLayout {
    int pass[G::NC];
    int rel[G::NC];
    int n_pass;
    constexpr DumpLayout() : pass{}, rel{}, n_pass(1) {
        int used = 0, p = 0;
        for (int c = 0; c < G::NC; ++c) {
            const int d3 = 2 * G::L3[c] + 1;
            if (used + d3 > CT_DUMP_REGS) {
                ++p;
                used = 0;
            }
            pass[c] = p;
            rel[c] = used;
            used += d3;
        }
        n_pass = p + 1;
    }
};

// The lin2 phase is a real function (one copy of its code, its own register allocation: inlined into every group kind it
// doubled the kernel and moved the walk's registers); its pointers carry their address spaces explicitly, because a
// generic pointer argument would turn every LDS access into a flat instruction that queues behind the gathers.
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) const float lds_cf32;
typedef __attribute__((address_space(3))) const int lds_ci32;
typedef __attribute__((address_space(1))) const float glb_cf32;
typedef float lds_f32x4 __attribute__((ext_vector_type(4)));
struct Lin2Ctx {
    glb_cf32* atab_sp;      // the tile's species row of the A table (global)
    lds_cf32* dump_all;     // the four waves' dump regions
    lds_f32* otile;         // [16][out_ld]
    lds_ci32* frag_l;       // LDS copies of the packed work lists: fragments, units (2 ints each), phases (2 ints each)
    lds_ci32* unit_l;
    lds_ci32* phase_l;
    int out_ld, wu_base, n_pass, r, class_cu_log2;
};

__device__ __forceinline__ float ct_act(int code, float v) {
    switch (code) {
        case 1: return v / (1.0f + expf(-v));                          // silu
        case 2: return tanhf(v);                                       // tanh
        case 3: return 1.0f / (1.0f + expf(-v));                       // sigmoid
        case 4: return (v > 20.0f ? v : log1pf(expf(v))) - 0.6931471805599453f;  // shifted softplus
        case 5: return fabsf(v);                                       // abs
        default: return v;
    }
}

// lin2 of one pass of a round: this wave's fragments, unit by unit (see the header comment).  KSV = channels per lane group
// and contraction step of the round's class (4 / 2 / 1 for 16 / 8 / <= 4 lanes per node).
// Latency is the whole cost of this phase (the matrix work is ~1 % of the tile's time): under the load of three
// workgroups' gathers a dependent global load takes microseconds.  So the work lists are read from LDS (packed records,
// copied once per workgroup), and the A fragments of up to CT_MAXF pieces are requested TOGETHER -- one coalesced load
// each from the L2-resident table -- before any of them is used: one memory round trip per batch instead of one per piece.
#ifndef CT_MAXF_N
#define CT_MAXF_N 8
#endif
constexpr int CT_MAXF = CT_MAXF_N;
template <int KSV>
__device__ __forceinline__ void lin2_phase_t(const Lin2Ctx& cx, int pass) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, c = lane & 15;
    const int ph_x = __builtin_amdgcn_readfirstlane(cx.phase_l[2 * (cx.wu_base + pass * WAVES_PER_BLOCK + wave)]);
    const int f_beg = ph_x & 0xffff, f_cnt = (ph_x >> 16) & 0xffff;
    int ui = __builtin_amdgcn_readfirstlane(cx.phase_l[2 * (cx.wu_base + pass * WAVES_PER_BLOCK + wave) + 1]);
    // state of the unit being accumulated (persists across batches)
    int jn[CT_MAX_NT], kk[CT_MAX_NT], boff[CT_MAX_NT];
    f32x4 D[CT_MAX_NT];
    int col0 = 0, d3 = 1, vcount = 0, n_nt = 0, npw_log2 = 0;
    bool fresh = true;
    for (int fb = 0; fb < f_cnt; fb += CT_MAXF) {
        const int nb = min(CT_MAXF, f_cnt - fb);
        int rec[CT_MAXF];
        float av[CT_MAXF][KSV];
#pragma unroll
        for (int b = 0; b < CT_MAXF; ++b) {
            rec[b] = __builtin_amdgcn_readfirstlane(cx.frag_l[f_beg + fb + min(b, nb - 1)]);
            glb_cf32* ap = cx.atab_sp + (rec[b] & 0x3fff) * 64 + lane * KSV;
            if constexpr (CT_LAB_NO_ALOAD) {
#pragma unroll
                for (int t = 0; t < KSV; ++t) av[b][t] = 1e-3f * (float)(rec[b] & 0xff);
            } else if constexpr (KSV == 4) {
                const f32x4 v = *reinterpret_cast<__attribute__((address_space(1))) const f32x4*>(ap);
                av[b][0] = v[0], av[b][1] = v[1], av[b][2] = v[2], av[b][3] = v[3];
            } else if constexpr (KSV == 2) {
                const f32x2 v = *reinterpret_cast<__attribute__((address_space(1))) const f32x2*>(ap);
                av[b][0] = v[0], av[b][1] = v[1];
            } else {
                av[b][0] = ap[0];
            }
        }
#pragma unroll
        for (int b = 0; b < CT_MAXF; ++b) {
            if (b < nb) {
                if (fresh) {   // first fragment of a unit: its columns
                    const int ux = __builtin_amdgcn_readfirstlane(cx.unit_l[2 * ui]);
                    const int uy = __builtin_amdgcn_readfirstlane(cx.unit_l[2 * ui + 1]);
                    col0 = ux & 0xfff, d3 = (ux >> 12) & 15, vcount = (ux >> 16) & 31, n_nt = (ux >> 25) & 7;
                    const int nt0 = (ux >> 21) & 15, class_cu_log2 = uy >> 4;
                    npw_log2 = uy & 15;
#pragma unroll
                    for (int nt = 0; nt < CT_MAX_NT; ++nt) {
                        const int col = (nt0 + min(nt, n_nt - 1)) * 16 + c;
                        jn[nt] = col & ((1 << npw_log2) - 1);
                        kk[nt] = col >> npw_log2;
                        // float offset of this lane's B operand inside a piece's dump block (columns past d3: clamped, discarded)
                        boff[nt] = min(kk[nt], d3 - 1) * CT_DUMP_RS + g * KSV + (KSV > 1 ? (jn[nt] << class_cu_log2) : 0);
                        D[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                    fresh = false;
                }
                const int r = rec[b];
                lds_cf32* db = cx.dump_all + ((r >> 19) & 3) * (CT_DUMP_REGS * CT_DUMP_RS) + ((r >> 14) & 31) * CT_DUMP_RS;
                const int cul = (r >> 21) & 7;
#pragma unroll
                for (int nt = 0; nt < CT_MAX_NT; ++nt) {
                    if (nt < n_nt) {
                        // KSV >= 2: the class has one lanes-per-node value (folded into boff); KSV == 1: 4 or 2 lanes
                        const int o = KSV == 1 ? boff[nt] + (jn[nt] << cul) : boff[nt];
                        float bv[KSV];
                        if constexpr (KSV == 4) {
                            const f32x4 v = *reinterpret_cast<__attribute__((address_space(3))) const f32x4*>(db + o);
                            bv[0] = v[0], bv[1] = v[1], bv[2] = v[2], bv[3] = v[3];
                        } else if constexpr (KSV == 2) {
                            const f32x2 v = *reinterpret_cast<__attribute__((address_space(3))) const f32x2*>(db + o);
                            bv[0] = v[0], bv[1] = v[1];
                        } else {
                            bv[0] = db[o];
                        }
#pragma unroll
                        for (int t = 0; t < KSV; ++t) D[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[b][t], bv[t], D[nt], 0, 0, 0);
                    }
                }
                if ((r >> 24) & 1) {   // last fragment of the unit: D[nt][i] = out channel 4 g + i of column (jn, kk) -> the tile's rows
#pragma unroll
                    for (int nt = 0; nt < CT_MAX_NT; ++nt) {
                        if (nt < n_nt && kk[nt] < d3) {
                            lds_f32* op = cx.otile + ((cx.r << npw_log2) + jn[nt]) * cx.out_ld + col0 + kk[nt] + 4 * g * d3;
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (4 * g + i < vcount) op[i * d3] += D[nt][i];
                        }
                    }
                    ++ui;
                    fresh = true;
                }
            }
        }
    }
}

__device__ __noinline__ void lin2_phase(Lin2Ctx cx, int pass) {
    if constexpr (CT_LAB_NO_LIN2) return;
    if (cx.class_cu_log2 >= 4) lin2_phase_t<4>(cx, pass);
    else if (cx.class_cu_log2 == 3) lin2_phase_t<2>(cx, pass);
    else lin2_phase_t<1>(cx, pass);
}

// what a wave does with its neighbour sums: park them for lin2, pass by pass
struct StoreDump {
    float* dump;            // this wave's region [CT_DUMP_REGS][CT_DUMP_RS]
    const Lin2Ctx* cx;
    template <class G>
    __device__ __forceinline__ void store(const Args& a, const GroupEntry& ge, const float* __restrict__ acc,
                                          float a_scale_inv, int node, int j, int u, bool valid) const {
        constexpr DumpLayout<G> L{};
        const int lane = threadIdx.x & 63;
        float norm = 0.0f;   // idle channel lanes and padding nodes park zeros: the matrix products read every lane
        if (valid) norm = a_scale_inv / sqrtf(a.avg_nn > 0.0f ? a.avg_nn : a.num_neigh[node]);
        // the round's pass count (a round mate may have more passes than this kind: its pieces may be this wave's units)
        for (int p = 0; p < cx->n_pass; ++p) {
#pragma unroll
            for (int cc = 0; cc < G::NC; ++cc) {
                if (L.pass[cc] == p && ((ge.mask >> cc) & 1u) && (!CT_LAB_NO_DUMP || acc[0] == 12345.678f)) {      // wave-uniform
                    const int d3 = 2 * G::L3[cc] + 1;
#pragma unroll
                    for (int k = 0; k < 2 * matten::CG_LMAX + 1; ++k)
                        if (k < d3) dump[(L.rel[cc] + k) * CT_DUMP_RS + lane] = acc[G::OFF[cc] + k] * norm;
                }
            }
            __syncthreads();          // every wave's sums of this pass are parked
            lin2_phase(*cx, p);
            __syncthreads();          // ... and consumed: the next pass / the next round's walk may overwrite them
        }
    }
};

#define CT_RGS(L1, GI, TD) \
    run_group_shared<L1, GI, 1, TD, false, 0u>(a, ge, tile, stage, e, node, lane, valid, beg, deg, maxdeg, StoreDump{dump, &cx})
#define CT_GROUP_CASE(L1, GI) \
    case (L1 * matten::GROUP_KIND_STRIDE + GI): \
        if (TwoDeepOk<L1, GI>::value && nodes_per_wave <= 8 && nodes_per_wave >= 2) CT_RGS(L1, GI, (TwoDeepOk<L1, GI>::value)); \
        else CT_RGS(L1, GI, false); \
        break;

__global__ __launch_bounds__(WAVES_PER_BLOCK * 64, CT_MIN_BLOCKS) void conv_tile_kernel(Args a, CArgs ca,
                                                                                       const GroupEntry* __restrict__ entries) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // tile slots are numbered block-major; the slots of one block of crystals run on ONE XCD (its L2 holds their x rows)
    const int xcd = blockIdx.x % N_XCD;
    const int q = blockIdx.x / N_XCD;
    const int blk = (q / ca.slots_per_block) * N_XCD + xcd;
    const int slot = blk * ca.slots_per_block + q % ca.slots_per_block;
    if (slot >= ca.n_slots) return;
    const int sp = __builtin_amdgcn_readfirstlane(ca.tile_species[slot]);
    if (sp < 0) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    // LDS: [walk area: 4 weight tiles + stage | overlaid by the 4 dump regions] [output rows 16 x out_ld] [node ids 16]
    float* tile = lds + wave * a.lds_per_wave;
    float* stage = lds + WAVES_PER_BLOCK * a.lds_per_wave;
    float* dump = lds + wave * (CT_DUMP_REGS * CT_DUMP_RS);
    float* otile = lds + ca.walk_floats;
    int* nid = reinterpret_cast<int*>(otile + CT_NODES * ca.out_ld);
    int* frag_l = nid + 32;                      // packed work lists of the lin2 phases (read many times per tile)
    int* unit_l = frag_l + ca.n_frag;
    int* phase_l = unit_l + 2 * ca.n_unit;
    if (threadIdx.x < CT_NODES) nid[threadIdx.x] = ca.tile_nodes[slot * CT_NODES + threadIdx.x];
    for (int i = threadIdx.x; i < ca.n_frag; i += WAVES_PER_BLOCK * 64) frag_l[i] = ca.frag_recs[i];
    for (int i = threadIdx.x; i < 2 * ca.n_unit; i += WAVES_PER_BLOCK * 64) unit_l[i] = reinterpret_cast<const int*>(ca.unit_recs)[i];
    for (int i = threadIdx.x; i < 2 * ca.n_phase; i += WAVES_PER_BLOCK * 64) phase_l[i] = reinterpret_cast<const int*>(ca.phase_recs)[i];
    __syncthreads();
    for (int row = wave; row < CT_NODES; row += WAVES_PER_BLOCK) {      // a wave per row, lanes along the columns
        const int n = nid[row];
        const float* arow = (ca.add && n >= 0) ? ca.add + (int64_t)n * ca.add_ld : nullptr;
        for (int col = lane; col < ca.d_out; col += 64) otile[row * ca.out_ld + col] = arow ? arow[col] : 0.0f;
    }
    Lin2Ctx cx{(glb_cf32*)(ca.atab + (int64_t)sp * ca.a_stride), (lds_cf32*)lds, (lds_f32*)otile, (lds_ci32*)frag_l,
               (lds_ci32*)unit_l, (lds_ci32*)phase_l, ca.out_ld, 0, 0, 0, 0};
    __syncthreads();

    for (int ri = 0; ri < ca.n_rounds; ++ri) {
        const int2 rd = ca.rounds[ri];
        const int qi = __builtin_amdgcn_readfirstlane(rd.x), r = __builtin_amdgcn_readfirstlane(rd.y);
        const int4 q0 = ca.quads[2 * qi], q1 = ca.quads[2 * qi + 1];
        const int e = __builtin_amdgcn_readfirstlane(wave == 0 ? q0.x : wave == 1 ? q0.y : wave == 2 ? q0.z : q0.w);
        const int class_cu_log2 = __builtin_amdgcn_readfirstlane(q1.x);
        cx.n_pass = __builtin_amdgcn_readfirstlane(q1.y);
        cx.wu_base = __builtin_amdgcn_readfirstlane(q1.w);
        cx.r = r;
        cx.class_cu_log2 = class_cu_log2;
        // a loader-only wave takes the geometry of the class (its rows are the class's rows)
        const int cu_log2 = e >= 0 ? entries[e].cu_log2 : class_cu_log2;
        const int cu = 1 << cu_log2;
        const int nodes_per_wave = 64 >> cu_log2;
        const int g_in_tile = r * nodes_per_wave + (lane >> cu_log2);
        const int u = lane & (cu - 1);
        const int node = g_in_tile < CT_NODES ? nid[g_in_tile] : -1;
        const bool in_range = node >= 0;
        int beg = 0, deg = 0;
        if (in_range) {
            beg = a.rowptr[node];
            deg = a.rowptr[node + 1] - beg;
        }
        int maxdeg = deg;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) maxdeg = max(maxdeg, __shfl_xor(maxdeg, off));
        if (e < 0) {
            run_loader_only_t<1>(a, cu_log2, stage, beg, deg, maxdeg);
            for (int p = 0; p < cx.n_pass; ++p) {
                __syncthreads();
                lin2_phase(cx, p);
                __syncthreads();
            }
        } else {
            const GroupEntry& ge = entries[e];
            const bool valid = in_range && (u < ge.mul);
            switch (ge.kind) {
                MATTEN_FOR_EACH_GROUP(CT_GROUP_CASE)
                default: break;
            }
        }
    }
    // the rows leave: plain lin2 output, or Gate (+ eval BatchNorm) applied on the way.  A wave per row, lanes along the
    // columns: whole 256-byte runs per store instruction
    for (int row = wave; row < CT_NODES; row += WAVES_PER_BLOCK) {
        const int n = nid[row];
        if (n < 0) continue;
        const float* orow = otile + row * ca.out_ld;
        float* grow = ca.out + (int64_t)n * ca.out_gld;
        if (ca.cmeta) {
            for (int col = lane; col < ca.d_act; col += 64) {
                const int4 m = ca.cmeta[col];
                const int act = m.z & 255, gact = (m.z >> 8) & 255;
                float v = orow[m.x];
                if (m.y < 0) {
                    if (act) v = ct_act(act, v) * ca.act_cst[act];
                } else {
                    float gv = orow[m.y];
                    if (gact) gv = ct_act(gact, gv) * ca.act_cst[gact];
                    v *= gv;
                }
                if (ca.bn_scale) v = fmaf(v, ca.bn_scale[col], ca.bn_shift[col]);
                grow[col] = v;
            }
        } else {
            for (int col = lane; col < ca.d_out; col += 64) grow[col] = orow[col];
        }
    }
}

// ---- tiles: per block of `block_nodes` consecutive nodes, the nodes grouped by species and cut into 16-node tiles -------
// One workgroup per block.  Slot layout: block b owns slots [b * slots_per_block, (b + 1) * slots_per_block), the tiles of
// its species in species order from the front, unused slots marked -1.  Inside a species the order follows the atomics
// (a node's result does not depend on its place in a tile).
constexpr int ST_THREADS = 256;
constexpr int ST_MAX_SPECIES = 1024;

__global__ __launch_bounds__(ST_THREADS) void species_tiles_kernel(const int* __restrict__ species, int n_nodes, int n_species,
                                                                   int block_nodes, int slots_per_block,
                                                                   int* __restrict__ tile_nodes, int* __restrict__ tile_species) {
    __shared__ int cnt[ST_MAX_SPECIES], first[ST_MAX_SPECIES], cur[ST_MAX_SPECIES];
    const int b = blockIdx.x;
    const int lo = b * block_nodes, hi = min(n_nodes, lo + block_nodes);
    for (int s = threadIdx.x; s < n_species; s += ST_THREADS) cnt[s] = 0, cur[s] = 0;
    for (int i = threadIdx.x; i < slots_per_block; i += ST_THREADS) tile_species[b * slots_per_block + i] = -1;
    for (int i = threadIdx.x; i < slots_per_block * CT_NODES; i += ST_THREADS)
        tile_nodes[(int64_t)b * slots_per_block * CT_NODES + i] = -1;
    __syncthreads();
    for (int n = lo + threadIdx.x; n < hi; n += ST_THREADS) atomicAdd(&cnt[min(max(species[n], 0), n_species - 1)], 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int s = 0; s < n_species; ++s) {
            first[s] = t;
            const int nt = (cnt[s] + CT_NODES - 1) / CT_NODES;
            for (int k = 0; k < nt; ++k) tile_species[b * slots_per_block + t + k] = s;
            t += nt;
        }
    }
    __syncthreads();
    for (int n = lo + threadIdx.x; n < hi; n += ST_THREADS) {
        const int s = min(max(species[n], 0), n_species - 1);
        const int pos = atomicAdd(&cur[s], 1);
        tile_nodes[((int64_t)b * slots_per_block + first[s]) * CT_NODES + pos] = n;
    }
}

}  // namespace

extern "C" int matten_conv_tile_nodes(void) { return CT_NODES; }
extern "C" int matten_conv_tile_dump_regs(void) { return CT_DUMP_REGS; }
extern "C" int matten_conv_tile_dump_stride(void) { return CT_DUMP_RS; }

extern "C" int64_t matten_species_tiles_slots_per_block(int64_t block_nodes, int64_t n_species) {
    return block_nodes / CT_NODES + n_species;
}

extern "C" int matten_species_tiles(const int32_t* species, int64_t n_nodes, int64_t n_species, int64_t block_nodes,
                                    int32_t* tile_nodes, int32_t* tile_species, matten_stream_t stream_) {
    if (n_nodes < 0 || n_species <= 0 || n_species > ST_MAX_SPECIES || block_nodes <= 0 || (block_nodes % CT_NODES))
        return MATTEN_EINVAL;
    if (n_nodes == 0) return MATTEN_OK;
    if (!species || !tile_nodes || !tile_species) return MATTEN_EINVAL;
    const int64_t n_blocks = matten_cdiv(n_nodes, block_nodes);
    const int64_t spb = matten_species_tiles_slots_per_block(block_nodes, n_species);
    if (n_blocks * spb * CT_NODES >= ((int64_t)1 << 31)) return MATTEN_EINVAL;
    species_tiles_kernel<<<(unsigned)n_blocks, ST_THREADS, 0, (hipStream_t)stream_>>>(
        species, (int)n_nodes, (int)n_species, (int)block_nodes, (int)spb, tile_nodes, tile_species);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}

extern "C" int matten_conv_tile(const float* x, int64_t d_in, const uint16_t* h2s, const float* w2p, int64_t w_pad,
                                const float* sh_sorted, int64_t sh_stride, const int32_t* rowptr,
                                const int32_t* src_sorted, int64_t n_nodes, const int32_t* entries, int64_t n_entries,
                                int64_t lds_floats_per_wave, const uint16_t* a_split, const float* a_scale_inv,
                                float avg_num_neighbors, const float* num_neigh, const int32_t* tile_nodes,
                                const int32_t* tile_species, int64_t n_slots, int64_t slots_per_block,
                                const int32_t* quads, int64_t n_quads, const int32_t* rounds, int64_t n_rounds,
                                const int32_t* frag_recs, int64_t n_frag, const int32_t* unit_recs, int64_t n_unit,
                                const int32_t* phase_recs, int64_t n_phase, const float* atab, int64_t a_stride, const float* add,
                                int64_t add_ld, int64_t d_out, const int32_t* cmeta, const float* act_cst,
                                const float* bn_scale, const float* bn_shift, int64_t d_act, float* out, int64_t out_ld,
                                matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_nodes < 0 || d_in <= 0 || w_pad <= 0 || sh_stride < 32 || (sh_stride & 3) || n_entries <= 0 ||
        lds_floats_per_wave <= 0 || (lds_floats_per_wave & 3) || n_slots < 0 || slots_per_block <= 0 || n_quads <= 0 ||
        n_rounds <= 0 || n_frag <= 0 || n_unit <= 0 || n_phase <= 0 ||
        a_stride <= 0 || (a_stride & 3) || d_out <= 0 || out_ld <= 0)
        return MATTEN_EINVAL;
    if (n_nodes == 0 || n_slots == 0) return MATTEN_OK;
    if (!x || !h2s || !w2p || !sh_sorted || !rowptr || !src_sorted || !entries || !a_split || !a_scale_inv || !tile_nodes ||
        !tile_species || !quads || !rounds || !frag_recs || !unit_recs || !phase_recs || !atab || !out)
        return MATTEN_EINVAL;
    if (!(avg_num_neighbors > 0.0f) && !num_neigh) return MATTEN_EINVAL;
    if (add && add_ld < d_out) return MATTEN_EINVAL;
    if (cmeta && (!act_cst || d_act <= 0 || out_ld < d_act)) return MATTEN_EINVAL;
    if (!cmeta && out_ld < d_out) return MATTEN_EINVAL;
    if ((bn_scale == nullptr) != (bn_shift == nullptr)) return MATTEN_EINVAL;
    const int lds_out_ld = (int)((d_out + 3) / 4 * 4 + 4);
    const int64_t walk = (int64_t)WAVES_PER_BLOCK * lds_floats_per_wave + 2 * 16 * STAGE_ROW;
    const int64_t dump = (int64_t)WAVES_PER_BLOCK * CT_DUMP_REGS * CT_DUMP_RS;
    const int64_t walk_floats = walk > dump ? walk : dump;
    const size_t lds = sizeof(float) * (size_t)(walk_floats + (int64_t)CT_NODES * lds_out_ld + 32 + n_frag + 2 * n_unit + 2 * n_phase);
    if (lds > 64 * 1024) return MATTEN_EINVAL;
    Args a{x, (const _Float16*)h2s, w2p, (const _Float16*)a_split, a_scale_inv, sh_sorted, rowptr, src_sorted, num_neigh,
           nullptr, (int)d_in, (int)w_pad, (int)sh_stride, 0, (int)n_nodes, (int)lds_floats_per_wave, avg_num_neighbors};
    CArgs ca{tile_nodes, tile_species, (const int4*)quads, (const int2*)rounds, frag_recs, (const int2*)unit_recs,
             (const int2*)phase_recs, atab, add, out, (const int4*)cmeta, act_cst, bn_scale, bn_shift, (int)n_rounds, (int)a_stride,
             (int)n_slots,
             (int)slots_per_block, (int)add_ld, (int)out_ld, (int)d_out, (int)d_act, lds_out_ld, (int)walk_floats,
             (int)n_frag, (int)n_unit, (int)n_phase};
    const int64_t n_blocks = matten_cdiv(n_slots, slots_per_block);
    const int64_t grid = matten_cdiv(n_blocks, N_XCD) * N_XCD * slots_per_block;
    if (grid >= ((int64_t)1 << 31)) return MATTEN_EINVAL;
    conv_tile_kernel<<<(unsigned)grid, WAVES_PER_BLOCK * 64, lds, stream>>>(a, ca, (const GroupEntry*)entries);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}
