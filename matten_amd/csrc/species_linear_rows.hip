// Species-indexed per-irrep linear for SHORT rows: the 16 rows of a tile stay resident in LDS.
//   e3nn FullyConnectedTensorProduct(x, one_hot)  reference nn/conv.py:59-61,77-79 (lin1, self-connection)
//   e3nn o3.Linear (order == NULL)                reference nn/nodewise.py:111-117, tfn_scalar_tensor.py:49-51
//
// matten_species_linear streams rows of up to 16.7 KB through a chunk pipeline whose cursor logic costs ~600
// instructions per (irrep block, chunk) and wave; for the node-feature sized rows (1 KB: lin1 and the self-connection,
// lin2 of the first layer, the read-out) that walk, not memory, was the time (133 us for 270 MB).  Here a workgroup
// owns 16 rows of ONE species: the rows (coalesced) and the species' packed weights are copied to LDS once, then the
// four waves split the (irrep block, 16-channel output tile) items round-robin.  An item is a plain K loop of
// v_mfma_f32_16x16x4_f32 with rows as N, output channels as M, input channels as the contraction, both operands
// read from LDS; the addend and the result go straight from / to memory in the D-fragment layout (4 d contiguous
// floats per lane).  Same arithmetic and summation order per output element as matten_species_linear.
#include "common.h"

namespace {

struct LinSeg {  // 8 x int32: one irrep block (matten_amd/plan.py:_plan_linear_like)
    int x_off, d, mul_in, w_off, mo, o_off, pad0, pad1;
};
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) f4u {  // 16 bytes at 4-byte alignment
    float v[4];
};
constexpr int ROWS = 16;
constexpr int WAVES = 4;
#ifndef SLR_TPB
#define SLR_TPB 2
#endif
constexpr int TPB = SLR_TPB;  // 16-row tiles per workgroup: the weight fill and the species look-up are paid once for 32 rows
                              // (2 and 4 measure the same, 3 and 6 are 20 % slower: tools/slr_tpb_ab.sh)

// accumulate-in-place MFMA through inline asm with explicit wait states, see species_linear.hip
__device__ __forceinline__ void mfma_16x16x4(f32x4& acc, float a, float b) {
#ifdef SLR_ABLATE_NO_MFMA
    acc[0] += a * b;
    return;
#endif
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 15" ::: "memory"); }

template <int D, bool HAS_ADD>
__device__ __forceinline__ void run_item(const LinSeg& L, int vt, const float* __restrict__ xs, int xs_stride,
                                         const float* __restrict__ ws, const float* __restrict__ add, int add_ld,
                                         float* __restrict__ out, int d_out, int node, bool row_ok, int g, int c) {
    f32x4 acc[D];
    const int vb = 16 * vt + 4 * g;  // this lane's first output channel of the D fragment
    // HAS_ADD is a template parameter: with a run-time test the compiler must assume addend loads in flight and puts
    // s_waitcnt vmcnt(0) in front of the first matrix instruction -- which on gfx950 also waits for the previous
    // item's STORES to be acknowledged (62 % of the waves' cycles)
    if (HAS_ADD && row_ok && vb + 4 <= L.mo) {
        const float* ap = add + (int64_t)node * add_ld + L.o_off + vb * D;
        float o[4 * D];
#pragma unroll
        for (int q = 0; q < D; ++q) {
            const f4u t = *reinterpret_cast<const f4u*>(ap + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[4 * q + e] = t.v[e];
        }
#pragma unroll
        for (int m = 0; m < D; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[m][r] = o[r * D + m];
    } else {
#pragma unroll
        for (int m = 0; m < D; ++m) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[m][r] = (HAS_ADD && row_ok && vb + r < L.mo) ? add[(int64_t)node * add_ld + L.o_off + (vb + r) * D + m] : 0.0f;
        }
    }
    const int va = 16 * vt + c;      // the A operand's output channel
    const float* xrow = xs + c * xs_stride + L.x_off;
    // two contraction steps per pass: the LDS operands of the second are in flight while the first one's matrix
    // instructions issue (a step past the block's last channel multiplies by a zero A column)
    for (int k0 = 0; k0 < L.mul_in; k0 += 8) {
        const int u0 = k0 + g, u1 = k0 + 4 + g;
        const bool ok0 = u0 < L.mul_in, ok1 = u1 < L.mul_in;
        const float a0 = (ok0 && va < L.mo) ? ws[L.w_off + u0 * L.mo + va] : 0.0f;
        const float a1 = (ok1 && va < L.mo) ? ws[L.w_off + u1 * L.mo + va] : 0.0f;
        float b0[D], b1[D];
#pragma unroll
        for (int m = 0; m < D; ++m) b0[m] = ok0 ? xrow[u0 * D + m] : 0.0f;
#pragma unroll
        for (int m = 0; m < D; ++m) b1[m] = ok1 ? xrow[u1 * D + m] : 0.0f;
#pragma unroll
        for (int m = 0; m < D; ++m) mfma_16x16x4(acc[m], a0, b0[m]);
        if (k0 + 4 < L.mul_in) {  // uniform
#pragma unroll
            for (int m = 0; m < D; ++m) mfma_16x16x4(acc[m], a1, b1[m]);
        }
    }
    mfma_drain();
#ifdef SLR_ABLATE_NO_STORE
    if (row_ok && acc[0][0] == 12345.678f) {
#else
    if (row_ok) {
#endif
        // a lane's fragment is 4 channels x D components = 4 D contiguous floats of the output row: D 16-byte stores
        // (4-byte aligned) instead of 4 D scattered dwords, which kept the texture addresser busy for the whole kernel
        float* op = out + (int64_t)node * d_out + L.o_off + vb * D;
        if (vb + 4 <= L.mo) {
            float o[4 * D];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int m = 0; m < D; ++m) o[r * D + m] = acc[m][r];
#pragma unroll
            for (int q = 0; q < D; ++q) {
                f4u t;
#pragma unroll
                for (int e = 0; e < 4; ++e) t.v[e] = o[4 * q + e];
                *reinterpret_cast<f4u*>(op + 4 * q) = t;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (vb + r < L.mo) {
#pragma unroll
                    for (int m = 0; m < D; ++m) op[r * D + m] = acc[m][r];
                }
            }
        }
    }
}

__global__ __launch_bounds__(WAVES * 64, 3) void species_linear_rows_kernel(
    const float* __restrict__ x, int d_in, const int32_t* __restrict__ order, const int32_t* __restrict__ seg,
    int n_species, const float* __restrict__ wp, int w_stride, const LinSeg* __restrict__ segs, int n_segs, int d_out,
    const float* __restrict__ add, int add_ld, int n_rows, float* __restrict__ out, int xs_stride) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;                          // [16][xs_stride]
    float* ws = lds + ROWS * xs_stride;       // [w_stride]
    // ---- block -> (species, 16 rows) ----
    int b = blockIdx.x, s = 0, lo = 0, hi = 0;
    if (seg) {
        // (the LDS is not in use yet: its first words serve as the look-up's scratch)
        if (!matten_block_species<WAVES * 64>(seg, n_species, TPB * ROWS, b, reinterpret_cast<int*>(lds), s, lo, hi)) return;
    } else {
        lo = b * TPB * ROWS;
        hi = min(n_rows, lo + TPB * ROWS);
        if (lo >= hi) return;
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    // ---- rows and weights to LDS ----
    // Row copy, software-pipelined over the workgroup's tiles: the loads of tile t + 1 are issued before tile t's
    // items are multiplied and land in registers meanwhile (16 per thread for rows of up to 256 floats; wider rows
    // take further sweeps at copy time), so only the first tile's HBM latency is exposed.
    constexpr int RW = ROWS / WAVES;
    float pre[RW][4];
    auto fetch = [&](int lo_t, int col0, float (&v)[RW][4]) {
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            const int r = wave * RW + rr;
            const bool in = lo_t + r < hi;
            const int nd = in ? (order ? order[lo_t + r] : lo_t + r) : 0;
            const float* xr = x + (int64_t)nd * d_in;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = col0 + 64 * q;
                v[rr][q] = (in && col < d_in) ? xr[col] : 0.0f;
            }
        }
    };
    auto put = [&](int col0, const float (&v)[RW][4]) {
#pragma unroll
        for (int rr = 0; rr < RW; ++rr)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = col0 + 64 * q;
                if (col < xs_stride) xs[(wave * RW + rr) * xs_stride + col] = v[rr][q];
            }
    };
    fetch(lo, lane, pre);   // the first tile's rows travel while the weights are copied
    // Every thread issues its loads in batches before it stores (the loops have run-time bounds: left to itself the
    // compiler waits for each load, and a workgroup spent ~25 us copying 40 KB).
    const float* wsp = wp + (int64_t)s * w_stride;
    if (((w_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(wsp) & 15) == 0)) {
        const f32x4* w4 = reinterpret_cast<const f32x4*>(wsp);
        f32x4* d4 = reinterpret_cast<f32x4*>(ws);   // ws is 16-byte aligned: ROWS * xs_stride is a multiple of 4
        const int n4 = w_stride >> 2;
        for (int i0 = threadIdx.x; i0 < n4; i0 += 4 * WAVES * 64) {
            f32x4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + q * WAVES * 64;
                v[q] = i < n4 ? w4[i] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + q * WAVES * 64;
                if (i < n4) d4[i] = v[q];
            }
        }
    } else {
        for (int i0 = threadIdx.x; i0 < w_stride; i0 += 8 * WAVES * 64) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = i0 + q * WAVES * 64;
                v[q] = i < w_stride ? wsp[i] : 0.0f;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = i0 + q * WAVES * 64;
                if (i < w_stride) ws[i] = v[q];
            }
        }
    }
    LinSeg* sl = reinterpret_cast<LinSeg*>(ws + ((w_stride + 3) & ~3));  // the segment table, read per item below
    for (int i = threadIdx.x; i < n_segs * 8; i += WAVES * 64)
        reinterpret_cast<int*>(sl)[i] = reinterpret_cast<const int*>(segs)[i];
    const int g = lane >> 4, c = lane & 15;
    for (int tile = 0; tile < TPB && lo < hi; ++tile, lo += ROWS) {
        if (tile) __syncthreads();  // everyone is done with the previous tile's rows
        put(lane, pre);
        for (int col0 = lane + 4 * 64; col0 < xs_stride; col0 += 4 * 64) {
            float v[RW][4];
            fetch(lo, col0, v);
            put(col0, v);
        }
        if (tile + 1 < TPB && lo + ROWS < hi) fetch(lo + ROWS, lane, pre);   // in flight during this tile's items
        __syncthreads();
        const bool row_ok = lo + c < hi;
        const int node = row_ok ? (order ? order[lo + c] : lo + c) : 0;
        // ---- (irrep block, output tile) items, round-robin over the waves ----
        int item = 0;
        for (int sg = 0; sg < n_segs; ++sg) {
            const int n_vt = (sl[sg].mo + 15) >> 4;
            for (int vt = 0; vt < n_vt; ++vt, ++item) {
                if ((item & (WAVES - 1)) != wave) continue;
#ifdef SLR_ABLATE_NO_ITEMS
                if (n_segs >= 0) continue;
#endif
                const LinSeg L = sl[sg];
#define SLR_ITEM(DD)                                                                                              \
    if (add) run_item<DD, true>(L, vt, xs, xs_stride, ws, add, add_ld, out, d_out, node, row_ok, g, c);           \
    else run_item<DD, false>(L, vt, xs, xs_stride, ws, add, add_ld, out, d_out, node, row_ok, g, c);              \
    break;
                switch (L.d) {
                    case 1: SLR_ITEM(1)
                    case 3: SLR_ITEM(3)
                    case 5: SLR_ITEM(5)
                    case 7: SLR_ITEM(7)
                    case 9: SLR_ITEM(9)
                    default: break;  // 2l+1 > 9: rejected by the host plan
                }
#undef SLR_ITEM
            }
        }
    }
}

}  // namespace

extern "C" int matten_species_linear_rows(const float* x, int64_t d_in, const int32_t* order, const int32_t* seg,
                                          int64_t n_species, const float* wp, int64_t w_stride, const int32_t* segs,
                                          int64_t n_segs, int64_t d_out, const float* add, int64_t add_ld,
                                          int64_t n_rows, float* out, matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_rows < 0 || d_in <= 0 || d_out <= 0 || n_species <= 0 || w_stride < 0 || n_segs < 0 ||
        n_rows >= ((int64_t)1 << 31))
        return MATTEN_EINVAL;
    if (n_rows == 0 || n_segs == 0) return MATTEN_OK;
    if (!x || !wp || !out || !segs) return MATTEN_EINVAL;
    if ((order == nullptr) != (seg == nullptr)) return MATTEN_EINVAL;
    if (!order && n_species != 1) return MATTEN_EINVAL;
    if (add && add_ld < d_out) return MATTEN_EINVAL;
    if (!add) add_ld = d_out;
    const int xs_stride = (int)(d_in | 1);  // odd row stride: the 16 rows of a column fall into 16 different LDS banks
    size_t lds = sizeof(float) * ((size_t)ROWS * xs_stride + (size_t)((w_stride + 3) & ~3) + 8 * (size_t)n_segs);
    lds = lds < sizeof(int) * (WAVES * 64 + 4) ? sizeof(int) * (WAVES * 64 + 4) : lds;   // (the species look-up's scratch)
    if (lds > 64 * 1024) return MATTEN_EINVAL;  // rows or weight table too large for this variant: use matten_species_linear
    const int64_t blocks = matten_cdiv(n_rows, TPB * ROWS) + (order ? n_species : 0);
    species_linear_rows_kernel<<<(unsigned)blocks, WAVES * 64, lds, stream>>>(
        x, (int)d_in, order, seg, (int)n_species, wp, (int)w_stride, (const LinSeg*)segs, (int)n_segs, (int)d_out, add,
        (int)add_ld, (int)n_rows, out, xs_stride);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}
