// lin2 of a conv layer as a pure row stream: out[n] = add[n] + lin2(agg[n], species(n))
//   reference nn/conv.py:84-86,123: FullyConnectedTensorProduct(irreps_mid, one_hot(species), conv_layer_irreps)
//
// The neighbour sums agg[N, ld] arrive in the component-major layout of plan.plan_agg_linear: one region per output
// irrep io, [component k][channel slot], every 16-float CHUNK of a row being the contraction slice of exactly one
// (io, k).  For that (io, k) the op is
//      out[n, o_off + v*d3 + k] = sum_slot W_s[slot, v] * agg[n, region(io) + k*Kpad + slot]
// i.e. rows = the N dimension of v_mfma_f32_16x16x4_f32 (a wave owns 16 rows of ONE species), output channels v = M,
// channel slots = the contraction.  Lane (g = lane >> 4, c = lane & 15) fetches the 16 bytes
// agg[row_c][16 f + 4 g .. + 3] of chunk f -- one instruction = the f-th 64-byte piece of each of the 16 rows, whole
// lines only -- and those four floats ARE its B operands of the chunk's four matrix steps (step s contracts slots
// 16 t + 4 g + s over g): no transposition, no LDS on the streamed side.  The species' weights sit in LDS as ready A
// fragments, [t][mt][g][c][s] so that one ds_read_b128 feeds the four steps of a column tile.  The row is walked in
// host-listed BLOCKS of <= 4 chunks of one (io, k); the loads of block j + 1 (always 4 per lane, so the compiler's
// vmcnt counts stay exact) are in flight while block j is multiplied: 4-8 KB per wave, 24 waves per CU.
// Results leave through a wave-private LDS stage, one 128-byte line per row and table row (see the kernel).
// Bound: HBM (each 17 KB row is read once; ~1100 matrix instructions per 16 rows, a quarter of the stream time).
#include <atomic>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef AL_WAVES_PER_WG
#define AL_WAVES_PER_WG 8
#endif
constexpr int AL_WAVES = AL_WAVES_PER_WG;   // waves per workgroup: they share one copy of the species weights in LDS
constexpr int AL_ROWS = 16;      // rows per wave (the N dimension of the matrix instruction)
#ifndef AL_BLK_CHUNKS
#define AL_BLK_CHUNKS 4
#endif
#ifndef AL_MIN_BLOCKS
#define AL_MIN_BLOCKS 3
#endif
#ifndef AL_GATE_PREFETCH
#define AL_GATE_PREFETCH 1   // the Gate's column records are requested when a table row opens, not when it is flushed
#endif
constexpr int AL_BLK = AL_BLK_CHUNKS;  // chunks per block = 16-byte loads per lane issued together; two blocks in flight
constexpr int AL_MAX_MT = 2;     // 16-channel output tiles per table row: wider irreps are split by the host into rows
                                 // of <= 32 output channels that re-read the same chunks (few: wide irreps are the
                                 // scalars, a few chunks per row); rows of <= 32 channels whose channels x components
                                 // exceed the 32-float stage are cut by component range instead: nothing is re-read.  Small on purpose: accumulators + addend = 16 registers,
                                 // ~80 in all, six waves per SIMD -- the stream is hidden by occupancy, not by depth

constexpr int AL_STAGE_RS = 33;   // row stride of the wave's output stage (odd: the 16 rows hit 16 banks)

struct AggIo {   // plan.AggLinearPlan.io_table: a table row = channels [v0, v0 + mo) x components [k0, k0 + kk) of one output irrep
    int chunk0, T, K, packed, a_off, out_off, mo, k0;   // packed = d3 | n_mt << 8 | cw << 16 | kk << 24; out_off = column of (v0, k = 0)
};
struct AggBlk {  // plan.AggLinearPlan.blocks: <= AL_BLK consecutive chunks of one (io, k) unit
    int chunk;   // first chunk of the block in the row
    int info;    // n chunks | first-of-unit << 8 | last-of-unit << 9 | last unit of its table row << 10 | k << 12 | io << 20
    int t0;      // chunk index inside the unit (slot base = 16 t0)
    int pad;
};

struct Args {
    const float* agg;
    const int* order;     // [n_rows] rows sorted by species, or NULL (one species)
    const int* seg;       // [n_species + 1]
    const float* wtab;    // [n_species, w_stride] A fragments
    const AggIo* io;
    const AggBlk* blk;
    const float* add;     // [n_rows, add_ld] or NULL
    float* out;           // [n_rows, d_out]
    int64_t ld;           // agg row stride (floats)
    int n_species, w_stride, n_io, n_blk, add_ld, d_out, n_rows;
    // Gate (+ eval BatchNorm) applied in the epilogue (agg_linear_kernel<true>): out then holds the ACTIVATED row
    const int4* cmeta;      // [d_out of lin2] per conv-output column {type | act << 8, column in the activated row, gate lane, gate set}
    const float* act_cst;   // normalize2mom constants by activation code
    const float* bn_scale;  // [out_ld] weight / sqrt(running_var + eps) per activated column, or NULL
    const float* bn_shift;  // [out_ld] bias - running_mean * scale on 0e columns, 0 elsewhere
    int out_ld;             // row stride of out
};

__device__ __forceinline__ float al_act(int code, float v) {
    switch (code) {
        case 1: return v / (1.0f + expf(-v));                          // silu
        case 2: return tanhf(v);                                       // tanh
        case 3: return 1.0f / (1.0f + expf(-v));                       // sigmoid
        case 4: return (v > 20.0f ? v : log1pf(expf(v))) - 0.6931471805599453f;  // shifted softplus
        case 5: return fabsf(v);                                       // abs
        default: return v;
    }
}
constexpr int AL_GATE_SETS = 3;   // table rows that may hold gate scalars (32 each): up to 96 gated channels per layer

#ifdef AL_WAVES_PER_EU
#define AL_OCC __attribute__((amdgpu_waves_per_eu(AL_WAVES_PER_EU, AL_WAVES_PER_EU)))
#else
#define AL_OCC
#endif
// GATE: the e3nn Gate (reference nn/utils.py:134-140) and the eval-mode BatchNorm that follow lin2 are applied while a
// table row leaves the wave (reference nn/conv.py:209-213: conv -> Gate -> BatchNorm).  The gate scalars are lin2 outputs
// of earlier table rows of the SAME 16 nodes: their activated values stay in the registers of the lanes that flushed
// them (8 rows per lane and table row) and reach the gated channels' lanes through ds_bpermute -- no memory, no LDS
// footprint.  Saves the conv output's write + read (150 MB per layer) and the gate kernel's launch.
template <bool GATE>
__global__ __launch_bounds__(AL_WAVES * 64, AL_MIN_BLOCKS) AL_OCC void agg_linear_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    AggIo* io_l = reinterpret_cast<AggIo*>(lds + a.w_stride);
    AggBlk* blk_l = reinterpret_cast<AggBlk*>(io_l + a.n_io);
    float* stage_all = reinterpret_cast<float*>(blk_l + a.n_blk);   // [AL_WAVES][16 rows][AL_STAGE_RS] + [AL_WAVES][16] row ids
    // ---- which species, which AL_WAVES x 16 rows of it: workgroups are numbered species-major.  The per-species offsets
    // are fetched by all threads at once and scanned from LDS by one: a chain of dependent global loads (one per
    // species) cost a small batch with 73 species more than its whole stream.
    int s = 0, blk = blockIdx.x, beg = 0, end = a.n_rows;
    if (a.order) {
        int* sseg = reinterpret_cast<int*>(stage_all);                 // free until the walk starts
        const int cap = AL_WAVES * (16 * AL_STAGE_RS + 16) - 4;
        if (a.n_species + 1 <= cap) {
            for (int i = threadIdx.x; i <= a.n_species; i += AL_WAVES * 64) sseg[i] = a.seg[i];
            __syncthreads();
            if (threadIdx.x == 0) {
                int b = blk, sp = 0;
                for (; sp < a.n_species; ++sp) {
                    const int nb = (sseg[sp + 1] - sseg[sp] + AL_WAVES * AL_ROWS - 1) / (AL_WAVES * AL_ROWS);
                    if (b < nb) break;
                    b -= nb;
                }
                sseg[cap] = sp;
                sseg[cap + 1] = b;
            }
            __syncthreads();
            s = sseg[cap];
            blk = sseg[cap + 1];
            if (s < a.n_species) beg = sseg[s], end = sseg[s + 1];
            __syncthreads();                                           // the region becomes the output stage again
        } else {
            for (; s < a.n_species; ++s) {
                beg = a.seg[s];
                end = a.seg[s + 1];
                const int nb = (end - beg + AL_WAVES * AL_ROWS - 1) / (AL_WAVES * AL_ROWS);
                if (blk < nb) break;
                blk -= nb;
            }
        }
    } else if (blk >= (a.n_rows + AL_WAVES * AL_ROWS - 1) / (AL_WAVES * AL_ROWS)) {
        s = a.n_species;
    }
    if (s >= a.n_species) return;
    {   // the species' A fragments and the two work tables -> LDS
        const f32x4* src = reinterpret_cast<const f32x4*>(a.wtab + (int64_t)s * a.w_stride);
        f32x4* dst = reinterpret_cast<f32x4*>(lds);
        for (int i = threadIdx.x; i < a.w_stride / 4; i += AL_WAVES * 64) dst[i] = src[i];
        const int* isrc = reinterpret_cast<const int*>(a.io);
        int* idst = reinterpret_cast<int*>(io_l);
        for (int i = threadIdx.x; i < a.n_io * 8; i += AL_WAVES * 64) idst[i] = isrc[i];
        const int* bsrc = reinterpret_cast<const int*>(a.blk);
        int* bdst = reinterpret_cast<int*>(blk_l);
        for (int i = threadIdx.x; i < a.n_blk * 4; i += AL_WAVES * 64) bdst[i] = bsrc[i];
    }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, c = lane & 15;
    const int r0 = beg + (blk * AL_WAVES + wave) * AL_ROWS;
    if (r0 >= end) return;
    const int ri = min(r0 + c, end - 1);                    // past the end: the last row again (never stored)
    const bool row_ok = r0 + c < end;
    const int64_t row = a.order ? a.order[ri] : ri;
    const float* rp = a.agg + row * a.ld + 4 * g;
    const bool has_add = a.add != nullptr;
    // Output path.  A lane's D fragment is 4 channels of ONE row: stored directly that is 64 four-byte pieces per
    // instruction, and 44 M such write requests per launch cost more than the whole input stream (measured: 0.39 vs
    // 0.25 ms).  Instead the units of a table row (<= 32 output floats per row, one 128-byte line) are parked in a
    // wave-private LDS stage [16 rows][33] and flushed when the table row is complete: 8 instructions, each writing two
    // rows' whole lines; the addend arrives the same way (8 loads per table row, requested when the row opens).
    float* stage = stage_all + wave * (16 * AL_STAGE_RS);
    int* rowid = reinterpret_cast<int*>(stage_all + AL_WAVES * 16 * AL_STAGE_RS) + wave * 16;
    if (g == 0) rowid[c] = row_ok ? (int)row : -1;
    const int fcol = lane & 31, frow = lane >> 5;   // flush role: column of the table row's slice, row parity

    // (io, k) unit state, wave-uniform
    int K = 0, d3 = 0, n_mt = 0, cw = 0, a_off = 0, out_off = 0, mo = 0, k0 = 0, kk = 1;
    auto load_io = [&](int ii) {
        const AggIo r = io_l[ii];
        K = __builtin_amdgcn_readfirstlane(r.K);
        d3 = __builtin_amdgcn_readfirstlane(r.packed & 255);
        n_mt = __builtin_amdgcn_readfirstlane((r.packed >> 8) & 255);
        cw = __builtin_amdgcn_readfirstlane((r.packed >> 16) & 255);
        a_off = __builtin_amdgcn_readfirstlane(r.a_off);
        out_off = __builtin_amdgcn_readfirstlane(r.out_off);
        mo = __builtin_amdgcn_readfirstlane(r.mo);
        k0 = __builtin_amdgcn_readfirstlane(r.k0);
        kk = __builtin_amdgcn_readfirstlane((r.packed >> 24) & 255);
    };
    f32x4 acc[AL_MAX_MT];
    float addv[8];
    auto zero_acc = [&]() {
#pragma unroll
        for (int mt = 0; mt < AL_MAX_MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // a table row opens: request its slice of the addend rows and -- GATE -- this lane's column record with the constants it
    // points at (used by flush_row, many blocks later: requested there they were two dependent global loads in front of
    // every flush)
    int4 cm_row = {0, 0, 0, 0};
    float cst_row = 1.0f, bsc_row = 1.0f, bsh_row = 0.0f;
    int ocol = 0;   // the conv-output column this lane flushes: stage position v * kk + (k - k0) -> out_off + v * d3 + k
    auto open_row = [&]() {
        const int w = mo * kk;
        const int fc = min(fcol, w - 1), fv = fc / kk;
        ocol = out_off + fv * d3 + k0 + (fc - fv * kk);
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int rid = max(rowid[2 * p + frow], 0);
            addv[p] = has_add ? a.add[(int64_t)rid * a.add_ld + ocol] : 0.0f;
        }
        if constexpr (GATE && AL_GATE_PREFETCH) {
            cm_row = a.cmeta[ocol];
            const int type = cm_row.x & 255, code = (cm_row.x >> 8) & 255;
            cst_row = code ? a.act_cst[code] : 1.0f;
            bsc_row = 1.0f, bsh_row = 0.0f;
            if (a.bn_scale && (type == 1 || type == 3)) bsc_row = a.bn_scale[cm_row.y], bsh_row = a.bn_shift[cm_row.y];
        }
    };
    // a unit (component k) is complete: park its 16 x mo values
    auto close_unit = [&](int k) {
#pragma unroll
        for (int mt = 0; mt < AL_MAX_MT; ++mt) {
            if (mt < n_mt) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int v = 16 * mt + 4 * g + i;
                    if (v < mo) stage[c * AL_STAGE_RS + v * kk + (k - k0)] = acc[mt][i];
                }
            }
        }
    };
    float gate_reg[GATE ? AL_GATE_SETS : 1][8];
    auto flush_row = [&]() {
        const int w = mo * kk;
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the stage is written (LDS is in order per wave)
        __builtin_amdgcn_wave_barrier();
        if constexpr (GATE) {
            // per column of the conv output: what happens to it.  type 0: dropped (an irrep the Gate does not take), 1:
            // activated scalar, 2: gate scalar (kept in gate_reg[set] of THIS lane), 3: gated component
            const int4 cm = AL_GATE_PREFETCH ? cm_row : a.cmeta[ocol];
            const int type = cm.x & 255, code = (cm.x >> 8) & 255;
            float cst = cst_row, bsc = bsc_row, bsh = bsh_row;
            if (!AL_GATE_PREFETCH) {
                cst = code ? a.act_cst[code] : 1.0f;
                if (a.bn_scale && (type == 1 || type == 3)) bsc = a.bn_scale[cm.y], bsh = a.bn_shift[cm.y];
            }
            const int src_lane = (cm.z & 31) | (frow << 5);
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int r = 2 * p + frow;
                const int rid = rowid[r];
                float v = stage[r * AL_STAGE_RS + min(fcol, w - 1)] + addv[p];
                // every lane takes part in the exchanges; the set is per lane
                float gv = __shfl(gate_reg[0][p], src_lane);
#pragma unroll
                for (int st = 1; st < AL_GATE_SETS; ++st) {
                    const float o = __shfl(gate_reg[st][p], src_lane);
                    gv = cm.w == st ? o : gv;
                }
                if (type == 2) {
                    const float act = (code ? al_act(code, v) : v) * cst;
#pragma unroll
                    for (int st = 0; st < AL_GATE_SETS; ++st)
                        if (cm.w == st && fcol < w) gate_reg[st][p] = act;
                } else if (fcol < w && rid >= 0 && type != 0) {
                    if (type == 1) v = (code ? al_act(code, v) : v) * cst;
                    else v = v * gv;
                    a.out[(int64_t)rid * a.out_ld + cm.y] = fmaf(v, bsc, bsh);
                }
            }
            __builtin_amdgcn_wave_barrier();
            return;
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int r = 2 * p + frow;
            const int rid = rowid[r];
#ifdef AL_ABL_NO_STORE
            if (fcol < w && rid >= 0 && addv[p] == 12345.678f)
#elif defined(AL_ABL_SMALL_OUT)   // timing experiment: same store instructions, all into 64 cache-resident rows
            if (fcol < w && rid >= 0) a.out[(int64_t)(rid & 63) * a.d_out + ocol] = stage[r * AL_STAGE_RS + fcol] + addv[p];
            if (false)
#else
            if (fcol < w && rid >= 0)
#endif
                a.out[(int64_t)rid * a.d_out + ocol] = stage[r * AL_STAGE_RS + fcol] + addv[p];
        }
        __builtin_amdgcn_wave_barrier();      // the next table row's units overwrite the stage after these reads
    };
    auto fetch = [&](int j, f32x4* __restrict__ buf) {   // the block's chunks, always AL_BLK loads (short blocks repeat their last chunk)
        const AggBlk d = blk_l[j];
        const int c0 = __builtin_amdgcn_readfirstlane(d.chunk);
        const int n = __builtin_amdgcn_readfirstlane(d.info & 255);
#pragma unroll
        for (int i = 0; i < AL_BLK; ++i) {
#ifdef AL_NT_LOADS   // the row is read exactly once: non-temporal hint (experiment, tools/agg_ab.sh)
            buf[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(rp + 16 * (c0 + min(i, n - 1))));
#else
            buf[i] = *reinterpret_cast<const f32x4*>(rp + 16 * (c0 + min(i, n - 1)));
#endif
        }
    };
    // one block: request the next one, multiply this one, close / open units at its end
    auto step = [&](int j, const f32x4* __restrict__ cur, f32x4* __restrict__ nxt) {
        fetch(min(j + 1, a.n_blk - 1), nxt);
        const AggBlk d = blk_l[j];
        const int info = __builtin_amdgcn_readfirstlane(d.info);
        const int n = info & 255, k = (info >> 12) & 255;
        const int t0 = __builtin_amdgcn_readfirstlane(d.t0);
        const int cc = min(c, cw - 1);
#pragma unroll
        for (int i = 0; i < AL_BLK; ++i) {
            if (i < n) {
                f32x4 b = cur[i];
#ifdef AL_ABL_NO_LDS
                const f32x4 afix = {1.f, 2.f, 3.f, (float)i};
#endif
                const int t = t0 + i;
                // slots past the irrep's last channel hold whatever was there: select, do not multiply
                const int slot0 = 16 * t + 4 * g;
#pragma unroll
                for (int sst = 0; sst < 4; ++sst) b[sst] = slot0 + sst < K ? b[sst] : 0.0f;
                const float* ab = lds + a_off + ((t * n_mt * 4 + g) * cw + cc) * 4;
#pragma unroll
                for (int mt = 0; mt < AL_MAX_MT; ++mt) {
                    if (mt < n_mt) {
#ifdef AL_ABL_NO_LDS
                        f32x4 a4 = afix;
#else
                        f32x4 a4 = *reinterpret_cast<const f32x4*>(ab + mt * 4 * cw * 4);
#endif
                        if (c >= cw) a4 = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef AL_ABL_NO_MFMA
                        acc[mt][0] += a4[0] * b[0] + a4[1] * b[1] + a4[2] * b[2] + a4[3] * b[3];   // timing experiment
#else
#pragma unroll
                        for (int sst = 0; sst < 4; ++sst)
                            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[sst], b[sst], acc[mt], 0, 0, 0);
#endif
                    }
                }
            }
        }
        if ((info >> 9) & 1) {   // last block of its unit
            close_unit(k);
            zero_acc();
            if ((info >> 10) & 1) {   // ... and of its table row
                flush_row();
                if (j + 1 < a.n_blk) {
                    load_io((__builtin_amdgcn_readfirstlane(blk_l[j + 1].info) >> 20) & 4095);
                    open_row();
                }
            }
        }
    };
#pragma unroll
    for (int st = 0; st < (GATE ? AL_GATE_SETS : 1); ++st)
#pragma unroll
        for (int p = 0; p < 8; ++p) gate_reg[st][p] = 0.0f;
    f32x4 bufa[AL_BLK], bufb[AL_BLK];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();   // rowid is visible to the wave
    load_io(0);
    zero_acc();
    open_row();
    fetch(0, bufa);
    for (int j = 0; j < a.n_blk; j += 2) {
        step(j, bufa, bufb);
        if (j + 1 < a.n_blk) step(j + 1, bufb, bufa);
    }
}

}  // namespace

constexpr size_t AL_MAX_LDS = 80 * 1024;   // two workgroups per CU
extern "C" size_t matten_agg_linear_lds_bytes(int64_t w_stride, int64_t n_io, int64_t n_blocks) {
    return sizeof(float) * (size_t)w_stride + sizeof(AggIo) * (size_t)n_io + sizeof(AggBlk) * (size_t)n_blocks +
           sizeof(float) * AL_WAVES * (16 * AL_STAGE_RS + 16);
}
extern "C" size_t matten_agg_linear_max_lds_bytes(void) { return AL_MAX_LDS; }
extern "C" int matten_agg_linear_max_mt(void) { return AL_MAX_MT; }   // (a table row is also at most 32 output floats wide)
extern "C" int matten_agg_linear_block_chunks(void) { return AL_BLK; }

static int agg_linear_launch(const float* agg, int64_t ld, const int32_t* order, const int32_t* seg, int64_t n_species,
                             const float* wtab, int64_t w_stride, const int32_t* io_table, int64_t n_io,
                             const int32_t* blocks, int64_t n_blocks, const float* add, int64_t add_ld, int64_t d_out,
                             int64_t n_rows, float* out, const int32_t* cmeta, const float* act_cst, const float* bn_scale,
                             const float* bn_shift, int64_t out_ld, hipStream_t stream) {
    if (n_rows < 0 || ld <= 0 || (ld & 3) || n_species <= 0 || w_stride <= 0 || (w_stride & 3) || n_io <= 0 ||
        n_blocks <= 0 || d_out <= 0 || out_ld <= 0)
        return MATTEN_EINVAL;
    if (n_rows == 0) return MATTEN_OK;
    if (!agg || !wtab || !io_table || !blocks || !out) return MATTEN_EINVAL;
    if ((order == nullptr) != (seg == nullptr)) return MATTEN_EINVAL;
    if (!order && n_species != 1) return MATTEN_EINVAL;
    if (add && add_ld < d_out) return MATTEN_EINVAL;
    if (cmeta && !act_cst) return MATTEN_EINVAL;
    if ((bn_scale == nullptr) != (bn_shift == nullptr)) return MATTEN_EINVAL;
    const size_t lds = matten_agg_linear_lds_bytes(w_stride, n_io, n_blocks);
    if (lds > AL_MAX_LDS) return MATTEN_EINVAL;
    if (lds > 64 * 1024) {   // gfx950: 160 KB of LDS per CU, a workgroup may take more than the default 64 KB on request
        // the attribute is per DEVICE: one bit per device ordinal, set after the call succeeded (two threads racing here
        // both make the idempotent call)
        static std::atomic<uint64_t> raised{0};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0) return MATTEN_ELAUNCH;
        const uint64_t bit = dev < 64 ? (uint64_t)1 << dev : 0;
        if (!bit || !(raised.load(std::memory_order_acquire) & bit)) {
            if (hipFuncSetAttribute((const void*)agg_linear_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)AL_MAX_LDS) != hipSuccess ||
                hipFuncSetAttribute((const void*)agg_linear_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)AL_MAX_LDS) != hipSuccess)
                return MATTEN_ELAUNCH;
            raised.fetch_or(bit, std::memory_order_release);
        }
    }
    // species-major workgroups of AL_WAVES x 16 rows: at most one partly filled workgroup per species
    const int64_t grid = matten_cdiv(n_rows, AL_WAVES * AL_ROWS) + n_species;
    if (grid >= ((int64_t)1 << 31) || n_rows >= ((int64_t)1 << 31)) return MATTEN_EINVAL;
    Args a{agg, order, seg, wtab, (const AggIo*)io_table, (const AggBlk*)blocks, add, out, ld, (int)n_species,
           (int)w_stride, (int)n_io, (int)n_blocks, (int)add_ld, (int)d_out, (int)n_rows, (const int4*)cmeta, act_cst,
           bn_scale, bn_shift, (int)out_ld};
    if (cmeta) agg_linear_kernel<true><<<(unsigned)grid, AL_WAVES * 64, lds, stream>>>(a);
    else agg_linear_kernel<false><<<(unsigned)grid, AL_WAVES * 64, lds, stream>>>(a);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}

extern "C" int matten_agg_linear(const float* agg, int64_t ld, const int32_t* order, const int32_t* seg,
                                 int64_t n_species, const float* wtab, int64_t w_stride, const int32_t* io_table,
                                 int64_t n_io, const int32_t* blocks, int64_t n_blocks, const float* add,
                                 int64_t add_ld, int64_t d_out, int64_t n_rows, float* out, matten_stream_t stream_) {
    return agg_linear_launch(agg, ld, order, seg, n_species, wtab, w_stride, io_table, n_io, blocks, n_blocks, add, add_ld,
                             d_out, n_rows, out, nullptr, nullptr, nullptr, nullptr, d_out, (hipStream_t)stream_);
}

extern "C" int matten_agg_linear_gate_sets(void) { return AL_GATE_SETS; }

// matten_agg_linear followed by the layer's Gate and eval-mode BatchNorm in the same launch (see agg_linear_kernel<true>)
extern "C" int matten_agg_linear_gate(const float* agg, int64_t ld, const int32_t* order, const int32_t* seg,
                                      int64_t n_species, const float* wtab, int64_t w_stride, const int32_t* io_table,
                                      int64_t n_io, const int32_t* blocks, int64_t n_blocks, const float* add,
                                      int64_t add_ld, int64_t d_out, int64_t n_rows, const int32_t* cmeta,
                                      const float* act_cst, const float* bn_scale, const float* bn_shift, int64_t d_act,
                                      float* out, matten_stream_t stream_) {
    if (!cmeta || d_act <= 0) return MATTEN_EINVAL;
    return agg_linear_launch(agg, ld, order, seg, n_species, wtab, w_stride, io_table, n_io, blocks, n_blocks, add, add_ld,
                             d_out, n_rows, out, cmeta, act_cst, bn_scale, bn_shift, d_act, (hipStream_t)stream_);
}
