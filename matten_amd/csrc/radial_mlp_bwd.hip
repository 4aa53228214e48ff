// Adjoint of the radial MLP (training step): given dL/dw[E, w_ld] from the tensor-product adjoint, the gradients of the
// three bias-free layers of e3nn FullyConnectedNet([nb, 32, 32, W], silu)  (reference nn/utils.py:246-251,260; the
// reference gets them from autograd through three torch.mm).
//
// In the packed form the forward kernel (radial_mlp.hip) evaluates:
//      b = bessel(|v|) [nb]   z1 = b W0p   h1 = silu(z1)   z2 = h1 W1p   h2 = silu(z2)   w = h2 W2p
// (the 1/sqrt(fan_in) and normalize2mom factors live in W0p, W1p, W2p; the host maps the packed gradients back).
//      dh2 = dw W2p^T          dW2p = h2^T dw
//      dz2 = dh2 * silu'(z2)   dW1p = h1^T dz2     dh1 = dz2 W1p^T
//      dz1 = dh1 * silu'(z1)   dW0p = b^T dz1
// Layers of 128 to 512 weight columns run ONE kernel that reads dw once (radial_mlp_bwd_fused below); narrower / wider
// ones two kernels, both on the fp32 matrix cores (v_mfma_f32_16x16x4_f32), both reading dw once:
//   radial_mlp_bwd_edges   a wave owns tiles of 16 edges (edges = the N dimension, like the forward kernel): recomputes
//                          z1, h1, z2, h2 in registers, contracts dw with W2p over the weight columns (16 bytes per lane
//                          and row: whole 64-byte pieces of 16 rows per load, the four floats being the B operands of
//                          four matrix steps), chains dz2 -> dh1 -> dz1 in registers (the D fragment of one layer is the B
//                          operand of the next, as in the forward kernel), writes h2 for the second kernel, and sums the
//                          two small weight gradients over its edges (operands transposed through a wave-private LDS
//                          tile: there the contraction runs over the edges)
//   radial_mlp_bwd_w2      dW2p[k, q] = sum_e h2[e, k] dw[e, q]: a wave owns 16 weight columns and an edge range
// Both write PARTIAL sums (one slice per wave / per edge range) that the host adds up in a fixed order: no atomics,
// bit-reproducible.  dw may be fp32 or bf16 (the opt-in bf16 storage of the per-edge tensors).
#include <algorithm>

#include <hip/hip_bf16.h>

#include "common.h"
#include "sh.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int HID = 32;
constexpr int BW_WAVES = 4;
#ifndef BW_WAVES_TARGET
#define BW_WAVES_TARGET 2048   // waves' worth of tiles before a wave takes more than one
#endif
#ifndef BW_TILES_MAX_N
#define BW_TILES_MAX_N 8
#endif
constexpr int BW_TILES_MAX = BW_TILES_MAX_N;      // 16-edge tiles per wave of radial_mlp_bwd_edges: fewer on small graphs (see bw_tiles)
constexpr int TS = 17;               // row stride of the LDS transpose tiles (16 edges + 1: bank spread)
constexpr int W2_RANGE_MAX = 4096;   // edges per partial sum of radial_mlp_bwd_w2: fewer on small graphs (see w2_range)

// Small graphs (a training batch of 32 crystals has ~4 k edges) need the parallelism more than the short partial lists:
// a wave takes one tile / a range is 256 edges until there are ~2000 waves' worth of work.
inline int bw_tiles(int64_t n_edges) {
    const int64_t t = n_edges / (16 * BW_WAVES_TARGET);
    return (int)(t < 1 ? 1 : (t > BW_TILES_MAX ? BW_TILES_MAX : t));
}
// enough (column group, edge range) workgroups to fill the chip (~1024): a launch with 48 weight columns has ONE column
// group, and 72 ranges of 4096 edges each left 184 CUs idle while three waves per workgroup walked 256 dependent steps
// (0.31 ms for the first conv layer's 48 columns at 293 k edges; LAB_NOTES round 4)
inline int w2_range(int64_t n_edges, int64_t w_pad) {
    const int64_t groups = (w_pad + 16 * BW_WAVES - 1) / (16 * BW_WAVES);
    int64_t r = (n_edges * groups / 1024 + 15) / 16 * 16;
    return (int)(r < 256 ? 256 : (r > W2_RANGE_MAX ? W2_RANGE_MAX : r));
}

__device__ __forceinline__ float silu(float z) { return z / (1.0f + expf(-z)); }
__device__ __forceinline__ float dsilu(float z) {
    const float s = 1.0f / (1.0f + expf(-z));
    return s * (1.0f + z * (1.0f - s));
}

template <bool BF16>
__device__ __forceinline__ f32x4 load4(const void* base, int64_t idx) {   // 4 consecutive elements at element index idx
    if constexpr (BF16) {
        const uint2 raw = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + idx);
        return f32x4{__uint_as_float(raw.x << 16), __uint_as_float(raw.x & 0xffff0000u), __uint_as_float(raw.y << 16),
                     __uint_as_float(raw.y & 0xffff0000u)};
    } else {
        return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + idx);
    }
}
template <bool BF16>
__device__ __forceinline__ float load1(const void* base, int64_t idx) {
    if constexpr (BF16) return __uint_as_float((uint32_t)reinterpret_cast<const uint16_t*>(base)[idx] << 16);
    else return reinterpret_cast<const float*>(base)[idx];
}

// part_small[wave slice][nb_pad*32 (dW0p) + 32*32 (dW1p)]
template <int KS0, bool BF16>
__global__ __launch_bounds__(BW_WAVES * 64) void radial_mlp_bwd_edges(
    const float4* __restrict__ geom, int64_t E, int n_basis, float r_start, float r_end, const float* __restrict__ w0p,
    const float* __restrict__ w1p, const float* __restrict__ w2p, int w_pad, int w_cols, const void* __restrict__ dw,
    int64_t dw_ld, float* __restrict__ h2_out, float* __restrict__ part_small, int tiles_per_wave, float s0, float s1) {
    __shared__ float lds[BW_WAVES][(3 * HID + 16) * TS];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int64_t slice = (int64_t)blockIdx.x * BW_WAVES + wave;
    const int64_t e_first = slice * (tiles_per_wave * 16);
    float* t_h1 = lds[wave];                 // [32][TS]  h1[k][e]
    float* t_dz2 = t_h1 + HID * TS;          // [32][TS]
    float* t_dz1 = t_dz2 + HID * TS;         // [32][TS]
    float* t_b = t_dz1 + HID * TS;           // [16][TS]  bessel[k0][e]
    // weights as matrix operands, fixed for the whole walk
    float a0[2][KS0], a1[2][8], a1t[2][8];
#pragma unroll
    for (int kk = 0; kk < KS0; ++kk) {
        a0[0][kk] = w0p[(4 * kk + g) * HID + c];
        a0[1][kk] = w0p[(4 * kk + g) * HID + 16 + c];
    }
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const int k = 16 * (kk >> 2) + 4 * g + (kk & 3);    // pi(kk, g): the feature the lane's D register (kk) holds
        a1[0][kk] = w1p[k * HID + c];                        // forward:  A[m = out c][K = in k]
        a1[1][kk] = w1p[k * HID + 16 + c];
        a1t[0][kk] = w1p[c * HID + k];                       // adjoint:  A[m = in c][K = out k]
        a1t[1][kk] = w1p[(16 + c) * HID + k];
    }
    f32x4 g_w1[2][2], g_w0[2];   // dW1p[m-tile over in][n-tile over out], dW0p[n-tile over hidden] (rows = basis index)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        g_w0[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2; ++j) g_w1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int tile = 0; tile < tiles_per_wave; ++tile) {
        const int64_t e0 = e_first + tile * 16;
        if (e0 >= E) break;
        const int64_t e = e0 + c;
        const bool e_ok = e < E;
        const int64_t ec = e_ok ? e : E - 1;
        const float len = geom[ec].w;
        // ---- forward recomputation (same operand order as radial_mlp_kernel) ----
        float bes[KS0];
        f32x4 z1[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, z2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int kk = 0; kk < KS0; ++kk) {
            const int k = 4 * kk + g;
            bes[kk] = (k < n_basis && e_ok) ? matten::bessel_basis(len, k, n_basis, r_start, r_end) : 0.0f;
            z1[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[0][kk], bes[kk], z1[0], 0, 0, 0);
            z1[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[1][kk], bes[kk], z1[1], 0, 0, 0);
        }
        f32x4 h1[2], h2[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) h1[t][r] = silu(z1[t][r]);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const float b = h1[kk >> 2][kk & 3];
            z2[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[0][kk], b, z2[0], 0, 0, 0);
            z2[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[1][kk], b, z2[1], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) h2[t][r] = silu(z2[t][r]);
        if (e_ok) {   // h2[e, 16 t + 4 g + r]: four consecutive floats per tile
            *reinterpret_cast<f32x4*>(h2_out + e * HID + 4 * g) = h2[0];
            *reinterpret_cast<f32x4*>(h2_out + e * HID + 16 + 4 * g) = h2[1];
        }
        // ---- dh2[k, e] = sum_q W2p[k, q] dw[e, q]: rows k = M, edges = N, columns q contracted 16 per load ----
        f32x4 dh2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        for (int q0 = 0; q0 < w_pad; q0 += 16) {
            const int q = q0 + 4 * g;
            f32x4 b4 = load4<BF16>(dw, ec * dw_ld + q);
            const f32x4 wa = *reinterpret_cast<const f32x4*>(w2p + (int64_t)c * w_pad + q);
            const f32x4 wb = *reinterpret_cast<const f32x4*>(w2p + (int64_t)(16 + c) * w_pad + q);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float b = (q + s < w_cols && e_ok) ? b4[s] : 0.0f;   // pad columns of dw are never written: select
                dh2[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s], b, dh2[0], 0, 0, 0);
                dh2[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[s], b, dh2[1], 0, 0, 0);
            }
        }
        // ---- dz2, dh1 = W1p dz2 (contraction over the out index: the D registers are the B operand), dz1 ----
        f32x4 dz2[2], dh1[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, dz1[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) dz2[t][r] = dh2[t][r] * dsilu(z2[t][r]);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const float b = dz2[kk >> 2][kk & 3];
            dh1[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1t[0][kk], b, dh1[0], 0, 0, 0);
            dh1[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1t[1][kk], b, dh1[1], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) dz1[t][r] = dh1[t][r] * dsilu(z1[t][r]);
        // ---- the two small weight gradients: contraction over the tile's 16 edges, operands through LDS ----
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 16 * t + 4 * g + r;
                t_h1[k * TS + c] = h1[t][r];
                t_dz2[k * TS + c] = dz2[t][r];
                t_dz1[k * TS + c] = dz1[t][r];
            }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) t_b[(4 * kk + g) * TS + c] = kk < KS0 ? bes[kk < KS0 ? kk : 0] : 0.0f;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < 4; ++s) {   // K step s: edges 4 s + g
            const int ee = 4 * s + g;
            const float h1a = t_h1[c * TS + ee], h1b = t_h1[(16 + c) * TS + ee];
            const float d2a = t_dz2[c * TS + ee], d2b = t_dz2[(16 + c) * TS + ee];
            const float d1a = t_dz1[c * TS + ee], d1b = t_dz1[(16 + c) * TS + ee];
            const float bb = t_b[c * TS + ee];
            g_w1[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(h1a, d2a, g_w1[0][0], 0, 0, 0);
            g_w1[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(h1a, d2b, g_w1[0][1], 0, 0, 0);
            g_w1[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(h1b, d2a, g_w1[1][0], 0, 0, 0);
            g_w1[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(h1b, d2b, g_w1[1][1], 0, 0, 0);
            g_w0[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bb, d1a, g_w0[0], 0, 0, 0);
            g_w0[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bb, d1b, g_w0[1], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // ---- this wave's partial sums: D[row 4 g + r][col c] ----
    const int nb_pad = 4 * KS0;
    float* out = part_small + slice * (int64_t)(nb_pad * HID + HID * HID);
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k0 = 4 * g + r;
            if (k0 < nb_pad) out[k0 * HID + 16 * tn + c] = g_w0[tn][r] * s0;
        }
    float* out1 = out + nb_pad * HID;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 4; ++r) out1[(16 * tm + 4 * g + r) * HID + 16 * tn + c] = g_w1[tm][tn][r] * s1;
}

// ---- both halves in ONE kernel, dw read ONCE (w_pad <= 16 * 4 * FU_MAXCH) ---------------------------------------------
// radial_mlp_bwd_edges + radial_mlp_bwd_w2 stream dw[E, w_pad] twice and pass h2[E, 32] through memory.  Here a workgroup
// works in ROUNDS of four 16-edge tiles (tile t = wave t's tile of the round, the same edge assignment as above):
//   phase 0  wave t recomputes z1, h1, z2, h2 of tile t and publishes h2 to LDS ([tile][edge][k]: the A operand of dW2p)
//   phase A  the 16-column chunks of dw are dealt round-robin to the waves; for each of its chunks and each of the four
//            tiles a wave loads the 16 x 16 piece of dw once (16 bytes per lane, whole 64-byte pieces of 16 rows) and uses
//            it twice: as the B operand of dh2 += W2p dw^T (edges = N, as in radial_mlp_bwd_edges) and, transposed through
//            a wave-private LDS tile, of dW2p += h2^T dw (edges = the contraction).  The wave's dW2p columns stay in
//            registers for the whole walk (8 accumulator registers per chunk).
//   phase B  the four waves' partial dh2 of tile t are added in wave order by wave t (LDS), which then runs the rest of
//            the chain for its tile exactly as radial_mlp_bwd_edges does.
// Partial sums: part_small per wave (as above), part_w2 per WORKGROUP; fixed order, no atomics.
constexpr int FU_MAXCH = 8;                 // 16-column chunks per wave at most: w_pad <= 512
constexpr int FU_HS = HID + 1;              // row stride of the published h2 tiles
constexpr int FU_TT = 20;                   // row stride of the dw transpose tile (16-byte aligned rows)
constexpr int FU_POOL = BW_WAVES * 64 * 8;  // floats per wave of the shared pool: partial dh2 [tile][lane][8] / the phase-B tiles
static_assert((3 * HID + 16) * TS <= FU_POOL, "the phase-B transposes live in the wave's share of the pool");
template <int KS0, bool BF16, int MAXCH>
__global__ __launch_bounds__(BW_WAVES * 64, 2) void radial_mlp_bwd_fused(
    const float4* __restrict__ geom, int64_t E, int n_basis, float r_start, float r_end, const float* __restrict__ w0p,
    const float* __restrict__ w1p, const float* __restrict__ w2p, int w_pad, int w_cols, const void* __restrict__ dw,
    int64_t dw_ld, float* __restrict__ part_small, float* __restrict__ part_w2, int tiles_per_wave, float s0, float s1,
    float s2) {
    // pool[wave]: the wave's partial dh2 [tile][lane][m, r] during phase A, its transposes in phase B (barrier between)
    __shared__ __attribute__((aligned(16))) float pool[BW_WAVES][FU_POOL];
    __shared__ float h2s[BW_WAVES][16 * FU_HS];                               // [tile][edge][k]
    __shared__ __attribute__((aligned(16))) float tts[BW_WAVES][16 * FU_TT];   // dw piece [edge][column] (per wave)
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int64_t slice = (int64_t)blockIdx.x * BW_WAVES + wave;
    float* t_h1 = pool[wave];
    float* t_dz2 = t_h1 + HID * TS;
    float* t_dz1 = t_dz2 + HID * TS;
    float* t_b = t_dz1 + HID * TS;
    float* tt = tts[wave];
    // the small layers' weights as matrix operands: from LDS at every use (kept in registers they cost the kernel its
    // second wave per SIMD)
    __shared__ float w0s[16 * HID], w1s[HID * HID];
    for (int i = threadIdx.x; i < 4 * KS0 * HID; i += BW_WAVES * 64) w0s[i] = w0p[i];
    for (int i = threadIdx.x; i < HID * HID; i += BW_WAVES * 64) w1s[i] = w1p[i];
    __syncthreads();
    f32x4 g_w1[2][2], g_w0[2], g_w2[MAXCH][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        g_w0[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2; ++j) g_w1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < MAXCH; ++j) g_w2[j][0] = g_w2[j][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int n_chunks = w_pad >> 4;
    const int64_t wg_first = (int64_t)blockIdx.x * BW_WAVES * tiles_per_wave * 16;   // first edge of the workgroup
    for (int round = 0; round < tiles_per_wave; ++round) {
        if (wg_first + (int64_t)round * 16 >= E) break;   // uniform: not even wave 0 has a tile in this round
        // ---- phase 0: forward of this wave's tile (z1, h1, z2 are recomputed in phase B: held across phase A they cost
        // the registers that keep a second wave per SIMD) ----
        const int64_t e0 = (slice * tiles_per_wave + round) * 16;
        const int64_t e = e0 + c;
        const bool e_ok = e < E;
        const float len = geom[e_ok ? e : E - 1].w;
        auto forward_tile = [&](float (&bes)[KS0], f32x4 (&z1)[2], f32x4 (&h1)[2], f32x4 (&z2)[2]) {
            z1[0] = z1[1] = z2[0] = z2[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KS0; ++kk) {
                const int k = 4 * kk + g;
                bes[kk] = (k < n_basis && e_ok) ? matten::bessel_basis(len, k, n_basis, r_start, r_end) : 0.0f;
                z1[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0s[k * HID + c], bes[kk], z1[0], 0, 0, 0);
                z1[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0s[k * HID + 16 + c], bes[kk], z1[1], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) h1[t][r] = silu(z1[t][r]);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const float b = h1[kk >> 2][kk & 3];
                const int kf = 16 * (kk >> 2) + 4 * g + (kk & 3);   // pi(kk, g): the feature the lane's D register (kk) holds
                z2[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1s[kf * HID + c], b, z2[0], 0, 0, 0);
                z2[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1s[kf * HID + 16 + c], b, z2[1], 0, 0, 0);
            }
        };
        // wide layers (more than 4 chunks per wave) recompute z1, h1, z2 in phase B: held across phase A they would spill;
        // narrow layers keep them (their phase A is short, the second evaluation would show: 0.20 vs 0.18 ms at 128 columns)
        constexpr bool RECOMPUTE = MAXCH > 4;
        float bes[KS0];
        f32x4 z1[2], h1[2], z2[2];
        {
            float bes0[KS0];
            f32x4 z1a[2], h1a[2], z2a[2];
            if constexpr (RECOMPUTE) forward_tile(bes0, z1a, h1a, z2a);
            else forward_tile(bes, z1, h1, z2);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)   // (0 for an edge past the end: its Bessel row is 0)
                    h2s[wave][c * FU_HS + 16 * t + 4 * g + r] = silu(RECOMPUTE ? z2a[t][r] : z2[t][r]);
        }
        __syncthreads();
        // ---- phase A: this wave's column chunks over the round's four tiles ----
        f32x4 pdh[BW_WAVES][2];
#pragma unroll
        for (int t = 0; t < BW_WAVES; ++t) pdh[t][0] = pdh[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        int64_t erow[BW_WAVES];
        bool eok[BW_WAVES];
#pragma unroll
        for (int t = 0; t < BW_WAVES; ++t) {
            const int64_t et = (((int64_t)blockIdx.x * BW_WAVES + t) * tiles_per_wave + round) * 16 + c;
            eok[t] = et < E;
            erow[t] = (eok[t] ? et : E - 1) * dw_ld;
        }
#pragma unroll
        for (int jj = 0; jj < MAXCH; ++jj) {
            const int j = jj * BW_WAVES + wave;
            if (j < n_chunks) {   // uniform over the wave
                const int q = 16 * j + 4 * g;
                f32x4 b4[BW_WAVES];
#pragma unroll
                for (int t = 0; t < BW_WAVES; ++t) b4[t] = load4<BF16>(dw, erow[t] + q);
                const f32x4 wa = *reinterpret_cast<const f32x4*>(w2p + (int64_t)c * w_pad + q);
                const f32x4 wb = *reinterpret_cast<const f32x4*>(w2p + (int64_t)(16 + c) * w_pad + q);
#pragma unroll
                for (int t = 0; t < BW_WAVES; ++t) {
                    f32x4 b = b4[t];
#pragma unroll
                    for (int sst = 0; sst < 4; ++sst) b[sst] = (q + sst < w_cols && eok[t]) ? b[sst] : 0.0f;   // pad columns of dw are never written: select
#pragma unroll
                    for (int sst = 0; sst < 4; ++sst) {
                        pdh[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[sst], b[sst], pdh[t][0], 0, 0, 0);
                        pdh[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[sst], b[sst], pdh[t][1], 0, 0, 0);
                    }
                    // the same piece with the edges as the contraction: [edge c][column 4 g + s] -> [edge 4 s + g][column c]
                    *reinterpret_cast<f32x4*>(tt + c * FU_TT + 4 * g) = b;
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_wave_barrier();
                    float bt[4];
#pragma unroll
                    for (int sst = 0; sst < 4; ++sst) bt[sst] = tt[(4 * sst + g) * FU_TT + c];
                    __builtin_amdgcn_wave_barrier();   // (the next piece's store follows these reads in the wave's LDS order)
#pragma unroll
                    for (int sst = 0; sst < 4; ++sst) {
                        // A operand of dW2p: h2[tile][edge 4 s + g][k = c + 16 m]
                        const float ha = h2s[t][(4 * sst + g) * FU_HS + c], hb = h2s[t][(4 * sst + g) * FU_HS + 16 + c];
                        g_w2[jj][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ha, bt[sst], g_w2[jj][0], 0, 0, 0);
                        g_w2[jj][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(hb, bt[sst], g_w2[jj][1], 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < BW_WAVES; ++t) {
            *reinterpret_cast<f32x4*>(&pool[wave][(t * 64 + lane) * 8]) = pdh[t][0];
            *reinterpret_cast<f32x4*>(&pool[wave][(t * 64 + lane) * 8 + 4]) = pdh[t][1];
        }
        __syncthreads();
        // ---- phase B: dh2 of this wave's tile (waves added in order), then the chain of radial_mlp_bwd_edges ----
        f32x4 dh2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int w = 0; w < BW_WAVES; ++w) {
            dh2[0] += *reinterpret_cast<const f32x4*>(&pool[w][(wave * 64 + lane) * 8]);
            dh2[1] += *reinterpret_cast<const f32x4*>(&pool[w][(wave * 64 + lane) * 8 + 4]);
        }
        __syncthreads();   // every wave has its sums: the pool now takes the transposes
        if constexpr (RECOMPUTE) forward_tile(bes, z1, h1, z2);
        f32x4 dz2[2], dh1[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, dz1[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) dz2[t][r] = dh2[t][r] * dsilu(z2[t][r]);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const float b = dz2[kk >> 2][kk & 3];
            const int kf = 16 * (kk >> 2) + 4 * g + (kk & 3);
            dh1[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1s[c * HID + kf], b, dh1[0], 0, 0, 0);
            dh1[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1s[(16 + c) * HID + kf], b, dh1[1], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) dz1[t][r] = dh1[t][r] * dsilu(z1[t][r]);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 16 * t + 4 * g + r;
                t_h1[k * TS + c] = h1[t][r];
                t_dz2[k * TS + c] = dz2[t][r];
                t_dz1[k * TS + c] = dz1[t][r];
            }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) t_b[(4 * kk + g) * TS + c] = kk < KS0 ? bes[kk < KS0 ? kk : 0] : 0.0f;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int sst = 0; sst < 4; ++sst) {
            const int ee = 4 * sst + g;
            const float h1a = t_h1[c * TS + ee], h1b = t_h1[(16 + c) * TS + ee];
            const float d2a = t_dz2[c * TS + ee], d2b = t_dz2[(16 + c) * TS + ee];
            const float d1a = t_dz1[c * TS + ee], d1b = t_dz1[(16 + c) * TS + ee];
            const float bb = t_b[c * TS + ee];
            g_w1[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(h1a, d2a, g_w1[0][0], 0, 0, 0);
            g_w1[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(h1a, d2b, g_w1[0][1], 0, 0, 0);
            g_w1[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(h1b, d2a, g_w1[1][0], 0, 0, 0);
            g_w1[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(h1b, d2b, g_w1[1][1], 0, 0, 0);
            g_w0[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bb, d1a, g_w0[0], 0, 0, 0);
            g_w0[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bb, d1b, g_w0[1], 0, 0, 0);
        }
        __syncthreads();   // the next round overwrites h2s / dh2p
    }
    const int nb_pad = 4 * KS0;
    float* out = part_small + slice * (int64_t)(nb_pad * HID + HID * HID);
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k0 = 4 * g + r;
            if (k0 < nb_pad) out[k0 * HID + 16 * tn + c] = g_w0[tn][r] * s0;
        }
    float* out1 = out + nb_pad * HID;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 4; ++r) out1[(16 * tm + 4 * g + r) * HID + 16 * tn + c] = g_w1[tm][tn][r] * s1;
    // dW2p[k = 16 m + 4 g + r][column 16 j + c] of this workgroup
    float* out2 = part_w2 + (int64_t)blockIdx.x * HID * w_pad;
#pragma unroll
    for (int jj = 0; jj < MAXCH; ++jj) {
        const int j = jj * BW_WAVES + wave;
        if (j < n_chunks) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) out2[(int64_t)(16 * m + 4 * g + r) * w_pad + 16 * j + c] = g_w2[jj][m][r] * s2;
        }
    }
}

// part_w2[range][32][w_pad]; grid = (ceil(w_pad / 64), n_ranges), wave w of a block owns columns [64 bx + 16 w, +16)
template <bool BF16>
__global__ __launch_bounds__(BW_WAVES * 64) void radial_mlp_bwd_w2(const float* __restrict__ h2, const void* __restrict__ dw,
                                                                  int64_t dw_ld, int64_t E, int w_pad,
                                                                  float* __restrict__ part_w2, int range, float s2) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int q0 = (blockIdx.x * BW_WAVES + wave) * 16;
    if (q0 >= w_pad) return;
    const int64_t e_beg = (int64_t)blockIdx.y * range;
    const int64_t e_end = min(E, e_beg + range);
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int64_t e0 = e_beg; e0 < e_end; e0 += 16) {
        float a[4][2], b[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {   // K step s: edge e0 + 4 s + g; A[m = k][K = edge] = h2[edge][k], B[K = edge][n = q]
            const int64_t e = e0 + 4 * s + g;
            const bool ok = e < e_end;
            const int64_t ec = ok ? e : e_end - 1;
            a[s][0] = ok ? h2[ec * HID + c] : 0.0f;
            a[s][1] = ok ? h2[ec * HID + 16 + c] : 0.0f;
            b[s] = ok ? load1<BF16>(dw, ec * dw_ld + q0 + c) : 0.0f;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][0], b[s], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][1], b[s], acc[1], 0, 0, 0);
        }
    }
    float* out = part_w2 + (int64_t)blockIdx.y * HID * w_pad;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(int64_t)(16 * t + 4 * g + r) * w_pad + q0 + c] = acc[t][r] * s2;
}

// out[i] = sum over the n partial rows part[s][i] in a FIXED order: a workgroup owns RED_COLS columns, its RED_GROUPS thread
// groups take every RED_GROUPS-th row each (sequentially), and group 0 adds the group sums in group order.  grid.y = 0: the
// small gradients, 1: dW2.  (16 columns x 64 groups: with ~1000 partial rows -- radial_mlp_bwd_w2 cuts the edges into as
// many ranges as fill the chip -- a thread adds 16 rows instead of 64: the kernel is a chain of dependent loads.)
constexpr int RED_COLS = 16, RED_GROUPS = 64;
__global__ __launch_bounds__(RED_COLS * RED_GROUPS) void radial_mlp_bwd_reduce(const float* __restrict__ part_small, int64_t n_small,
                                                              int small_len, const float* __restrict__ part_w2,
                                                              int64_t n_rng, int w2_len, float* __restrict__ out_small,
                                                              float* __restrict__ out_w2) {
    __shared__ float red[RED_GROUPS][RED_COLS];
    const float* part = blockIdx.y ? part_w2 : part_small;
    const int64_t n = blockIdx.y ? n_rng : n_small;
    const int len = blockIdx.y ? w2_len : small_len;
    float* out = blockIdx.y ? out_w2 : out_small;
    const int lc = threadIdx.x % RED_COLS, grp = threadIdx.x / RED_COLS;
    const int col = blockIdx.x * RED_COLS + lc;
    if (blockIdx.x * RED_COLS >= len) return;
    float v = 0.0f;
    if (col < len)
        for (int64_t s = grp; s < n; s += RED_GROUPS) v += part[s * len + col];
    red[grp][lc] = v;
    __syncthreads();
    if (grp == 0 && col < len) {
        float t = red[0][lc];
#pragma unroll 8
        for (int gq = 1; gq < RED_GROUPS; ++gq) t += red[gq][lc];
        out[col] = t;
    }
}

}  // namespace

extern "C" int64_t matten_radial_mlp_bwd_small_slices(int64_t n_edges) {
    return matten_cdiv(matten_cdiv(n_edges, bw_tiles(n_edges) * 16), BW_WAVES) * BW_WAVES;
}
// one kernel (dw read once) where the wave's share of dW2p fits its registers
#ifndef BW_FUSED
#define BW_FUSED 1
#endif
#ifndef BW_FUSED_MIN_W
#define BW_FUSED_MIN_W 128   // narrower layers: too few column chunks for four waves (48 columns: 0.16 vs 0.13 ms at 293 k edges)
#endif
static inline bool bw_fused(int64_t w_pad) { return BW_FUSED && w_pad >= BW_FUSED_MIN_W && w_pad <= 16 * BW_WAVES * FU_MAXCH; }
// partial rows of dW2p: one per workgroup of the fused kernel, else one per edge range of radial_mlp_bwd_w2
extern "C" int64_t matten_radial_mlp_bwd_w2_ranges(int64_t n_edges, int64_t w_pad) {
    if (bw_fused(w_pad)) return matten_radial_mlp_bwd_small_slices(n_edges) / BW_WAVES;
    return matten_cdiv(n_edges, w2_range(n_edges, w_pad));
}

extern "C" int matten_radial_mlp_bwd(const float* geom_sorted, int64_t n_edges, int n_basis, float r_start, float r_end,
                                     const float* w0p, int nb_pad, const float* w1p, const float* w2p, int hidden,
                                     int w_pad, int w_cols, const void* dw, int64_t dw_ld, int dw_is_bf16,
                                     float* h2_scratch, float* part_small, float* part_w2, float scale0, float scale1,
                                     float scale2, float* grad_small, float* grad_w2, matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_edges < 0 || hidden != HID || (w_pad & 15) || w_pad <= 0 || w_cols <= 0 || w_cols > w_pad || (nb_pad & 3) ||
        nb_pad < n_basis || nb_pad > 16 || dw_ld < w_pad || (dw_ld & 3))
        return MATTEN_EINVAL;
    if (n_edges == 0) return MATTEN_OK;
    if (!geom_sorted || !w0p || !w1p || !w2p || !dw || !h2_scratch || !part_small || !part_w2) return MATTEN_EINVAL;
    const unsigned grid1 = (unsigned)(matten_radial_mlp_bwd_small_slices(n_edges) / BW_WAVES);
    const bool fused = bw_fused(w_pad);
#define LAUNCH_FUSED_M(K, B, M)                                                                                       \
    radial_mlp_bwd_fused<K, B, M><<<grid1, BW_WAVES * 64, 0, stream>>>((const float4*)geom_sorted, n_edges, n_basis,  \
                                                                        r_start, r_end, w0p, w1p, w2p, w_pad, w_cols,  \
                                                                        dw, dw_ld, part_small, part_w2,                \
                                                                        bw_tiles(n_edges), scale0, scale1, scale2)
#define LAUNCH_FUSED(K, B)                                                          \
    do {                                                                            \
        if (w_pad <= 16 * BW_WAVES * 4) LAUNCH_FUSED_M(K, B, 4);                    \
        else if (w_pad <= 16 * BW_WAVES * 6) LAUNCH_FUSED_M(K, B, 6);               \
        else LAUNCH_FUSED_M(K, B, FU_MAXCH);                                        \
    } while (0)
#define LAUNCH(K, B) if (fused) LAUNCH_FUSED(K, B); else                                                                                                  \
    radial_mlp_bwd_edges<K, B><<<grid1, BW_WAVES * 64, 0, stream>>>((const float4*)geom_sorted, n_edges, n_basis,     \
                                                                     r_start, r_end, w0p, w1p, w2p, w_pad, w_cols, dw, \
                                                                     dw_ld, h2_scratch, part_small, bw_tiles(n_edges), scale0, scale1)
#define LAUNCH_K(B)                     \
    switch (nb_pad >> 2) {              \
        case 1: LAUNCH(1, B); break;    \
        case 2: LAUNCH(2, B); break;    \
        case 3: LAUNCH(3, B); break;    \
        default: LAUNCH(4, B); break;   \
    }
    if (dw_is_bf16) { LAUNCH_K(true) } else { LAUNCH_K(false) }
#undef LAUNCH_K
#undef LAUNCH
#undef LAUNCH_FUSED
#undef LAUNCH_FUSED_M
    MATTEN_LAUNCH_CHECK();
    dim3 grid2((unsigned)matten_cdiv(w_pad, 16 * BW_WAVES), (unsigned)matten_radial_mlp_bwd_w2_ranges(n_edges, w_pad));
    if (fused) {
    } else if (dw_is_bf16)
        radial_mlp_bwd_w2<true><<<grid2, BW_WAVES * 64, 0, stream>>>(h2_scratch, dw, dw_ld, n_edges, w_pad, part_w2,
                                                                     w2_range(n_edges, w_pad), scale2);
    else
        radial_mlp_bwd_w2<false><<<grid2, BW_WAVES * 64, 0, stream>>>(h2_scratch, dw, dw_ld, n_edges, w_pad, part_w2,
                                                                      w2_range(n_edges, w_pad), scale2);
    MATTEN_LAUNCH_CHECK();
    if (grad_small || grad_w2) {   // the final, ordered sums in the same call (both or neither)
        if (!grad_small || !grad_w2) return MATTEN_EINVAL;
        const int small_len = nb_pad * HID + HID * HID, w2_len = HID * w_pad;
        radial_mlp_bwd_reduce<<<dim3((unsigned)matten_cdiv(std::max(small_len, w2_len), RED_COLS), 2), RED_COLS * RED_GROUPS, 0, stream>>>(
            part_small, matten_radial_mlp_bwd_small_slices(n_edges), small_len, part_w2,
            matten_radial_mlp_bwd_w2_ranges(n_edges, w_pad), w2_len, grad_small, grad_w2);
        MATTEN_LAUNCH_CHECK();
    }
    return MATTEN_OK;
}

// (w0 [nb,32], w1 [32,32], w2 [32,W] raw parameters) -> the packed operands of matten_radial_mlp / _bwd in ONE launch:
// w0p [nb_pad,32] = s0 w0 (zero rows past nb), w1p = s1 w1, w2p [32,w_pad] = s2 w2 (zero columns past W).
namespace {
__global__ void radial_pack_kernel(const float* __restrict__ w0, const float* __restrict__ w1, const float* __restrict__ w2,
                                   int nb, int nb_pad, int W, int w_pad, float s0, float s1, float s2,
                                   float* __restrict__ w0p, float* __restrict__ w1p, float* __restrict__ w2p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n0 = nb_pad * HID, n1 = HID * HID, n2 = HID * w_pad;
    if (i < n0) {
        w0p[i] = (i / HID) < nb ? w0[i] * s0 : 0.0f;
    } else if (i < n0 + n1) {
        w1p[i - n0] = w1[i - n0] * s1;
    } else if (i < n0 + n1 + n2) {
        const int j = i - n0 - n1, k = j / w_pad, q = j - k * w_pad;
        w2p[j] = q < W ? w2[k * W + q] * s2 : 0.0f;
    }
}
}  // namespace

extern "C" int matten_radial_pack(const float* w0, const float* w1, const float* w2, int n_basis, int nb_pad, int w_cols,
                                  int w_pad, float scale0, float scale1, float scale2, float* w0p, float* w1p, float* w2p,
                                  matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_basis <= 0 || nb_pad < n_basis || w_cols <= 0 || w_pad < w_cols) return MATTEN_EINVAL;
    if (!w0 || !w1 || !w2 || !w0p || !w1p || !w2p) return MATTEN_EINVAL;
    const int n = nb_pad * HID + HID * HID + HID * w_pad;
    radial_pack_kernel<<<(unsigned)matten_cdiv(n, 256), 256, 0, stream>>>(w0, w1, w2, n_basis, nb_pad, w_cols, w_pad, scale0,
                                                                         scale1, scale2, w0p, w1p, w2p);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}

// ---- operands of matten_tp_fused derived from the RAW radial layers by kernels (training on the production kernel: the
// parameters change every step, the inference path's cached library-op derivations would run every step) --------------
namespace {

// as radial_pack_kernel, the last layer's columns gathered through `cols` (fused column order; -1 = structural zero)
__global__ void radial_pack_cols_kernel(const float* __restrict__ w0, const float* __restrict__ w1,
                                        const float* __restrict__ w2, int nb, int nb_pad, int W,
                                        const int64_t* __restrict__ cols, int n_cols, int w_pad, float s0, float s1, float s2,
                                        float* __restrict__ w0p, float* __restrict__ w1p, float* __restrict__ w2p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n0 = nb_pad * HID, n1 = HID * HID, n2 = HID * w_pad;
    if (i < n0) {
        w0p[i] = (i / HID) < nb ? w0[i] * s0 : 0.0f;
    } else if (i < n0 + n1) {
        w1p[i - n0] = w1[i - n0] * s1;
    } else if (i < n0 + n1 + n2) {
        const int j = i - n0 - n1, k = j / w_pad, q = j - k * w_pad;
        const int64_t src = q < n_cols ? cols[q] : -1;
        w2p[j] = src >= 0 ? w2[(int64_t)k * W + src] * s2 : 0.0f;
    }
}

// [s, 1 / s]: the power of two the hidden features are multiplied by so that |s h2| < 2^15 for every edge length
// (nn/utils.py RadialMLP._fp16_scale, same bound: |h2| <= sqrt(2/c) sqrt(nb) pi / c * n0 * n1 with the column-sum norms
// n0 = max_col sum_k |W0[k,col]| (k+1) / sqrt(nb), n1 = max_col sum_k |W1[k,col]| act_cst / sqrt(h))
__global__ void radial_h_scale_kernel(const float* __restrict__ w0, const float* __restrict__ w1, int nb, float c,
                                      float act_cst, float* __restrict__ out) {
    __shared__ float red[2][HID];
    const int col = threadIdx.x;   // 32 threads
    float a0 = 0.0f, a1 = 0.0f;
    for (int k = 0; k < nb; ++k) a0 += fabsf(w0[k * HID + col]) * (float)(k + 1);
    for (int k = 0; k < HID; ++k) a1 += fabsf(w1[k * HID + col]);
    red[0][col] = a0 / sqrtf((float)nb);
    red[1][col] = a1 * (act_cst / sqrtf((float)HID));
    __syncthreads();
    if (col == 0) {
        float n0 = 0.0f, n1 = 0.0f;
        for (int i = 0; i < HID; ++i) n0 = fmaxf(n0, red[0][i]), n1 = fmaxf(n1, red[1][i]);
        const float bound = sqrtf(2.0f / c) * sqrtf((float)nb) * 3.141592653589793f / c * n0 * n1;
        float s = 1.0f;
        if (bound > 32768.0f) s = exp2f(floorf(log2f(16384.0f / fmaxf(bound, 1e-30f))));
        if (!(s > 0.0f) || isinf(s) || isnan(s)) s = exp2f(-100.0f);
        out[0] = s;
        out[1] = 1.0f / s;
    }
}

// group entries' weight blocks as the fp16 hi / lo MFMA fragments matten_tp_fused consumes (same values as
// ops.split_a_tiles): per entry a power-of-two scale that puts its largest magnitude in [2^13, 2^14).
// frag[(a_tile + mt) * 64 + lane][16 halves: hi(kk 0..7) | lo(kk 0..7)], lane = 16 g + c:
//   value = W2p[pi(kk, g)][w_base + 16 mt + c] * scale,  pi(kk, g) = 16 (kk >> 2) + 4 g + (kk & 3)
// scale_inv[entry] = 2^-(scale exponent) * h_scale_inv
__global__ void split_a_tiles_kernel(const float* __restrict__ w2p, int w_pad, const int32_t* __restrict__ entries,
                                     const float* __restrict__ h_scale, _Float16* __restrict__ frag,
                                     float* __restrict__ scale_inv) {
    __shared__ float red[256];
    const int e = blockIdx.x;
    const int32_t* ge = entries + (int64_t)e * 32;
    const int w_base = ge[5], a_tile = ge[6], n_mt = ge[7];
    const int ncol = 16 * n_mt;
    float amax = 0.0f;
    for (int i = threadIdx.x; i < HID * ncol; i += blockDim.x) {
        const int k = i / ncol, q = i - k * ncol;
        if (w_base + q < w_pad) amax = fmaxf(amax, fabsf(w2p[(int64_t)k * w_pad + w_base + q]));   // (a tile may reach past the last column)
    }
    red[threadIdx.x] = amax;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    amax = red[0];
    int ex = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 127;
    ex = amax > 0.0f ? max(-100, min(100, ex)) : 13;
    const float sc = __uint_as_float((unsigned)(127 + 13 - ex) << 23);
    if (threadIdx.x == 0) scale_inv[e] = __uint_as_float((unsigned)(127 - 13 + ex) << 23) * (h_scale ? h_scale[1] : 1.0f);
    const float tiny = 6.103515625e-05f;
    for (int i = threadIdx.x; i < n_mt * 64; i += blockDim.x) {
        const int mt = i >> 6, lane = i & 63, g = lane >> 4, c = lane & 15;
        _Float16* dst = frag + ((int64_t)(a_tile + mt) * 64 + lane) * 16;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int k = 16 * (kk >> 2) + 4 * g + (kk & 3);
            const int col = w_base + 16 * mt + c;
            const float v = col < w_pad ? w2p[(int64_t)k * w_pad + col] * sc : 0.0f;
            const float hi = fabsf(v) < tiny ? 0.0f : (float)(_Float16)v;
            float lo = (v - hi) * 2048.0f;
            lo = fabsf(lo) < tiny ? 0.0f : lo;
            dst[kk] = (_Float16)hi;
            dst[8 + kk] = (_Float16)lo;
        }
    }
}

}  // namespace

extern "C" int matten_radial_pack_cols(const float* w0, const float* w1, const float* w2, int n_basis, int nb_pad, int w_cols,
                                       const int64_t* cols, int n_cols, int w_pad, float scale0, float scale1, float scale2,
                                       float* w0p, float* w1p, float* w2p, matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_basis <= 0 || nb_pad < n_basis || w_cols <= 0 || n_cols <= 0 || w_pad < n_cols) return MATTEN_EINVAL;
    if (!w0 || !w1 || !w2 || !cols || !w0p || !w1p || !w2p) return MATTEN_EINVAL;
    const int n = nb_pad * HID + HID * HID + HID * w_pad;
    radial_pack_cols_kernel<<<(unsigned)matten_cdiv(n, 256), 256, 0, stream>>>(w0, w1, w2, n_basis, nb_pad, w_cols, cols,
                                                                              n_cols, w_pad, scale0, scale1, scale2, w0p,
                                                                              w1p, w2p);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}

extern "C" int matten_radial_h_scale(const float* w0, const float* w1, int n_basis, float r_start, float r_end, float act_cst,
                                     float* out2, matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_basis <= 0 || !(r_end > r_start) || !w0 || !w1 || !out2) return MATTEN_EINVAL;
    radial_h_scale_kernel<<<1, HID, 0, stream>>>(w0, w1, n_basis, r_end - r_start, act_cst, out2);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}

extern "C" int matten_split_a_tiles(const float* w2p, int64_t w_pad, const int32_t* group_entries, int64_t n_entries,
                                    const float* h_scale, uint16_t* frag, float* scale_inv, matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_entries <= 0 || w_pad <= 0 || !w2p || !group_entries || !frag || !scale_inv) return MATTEN_EINVAL;
    split_a_tiles_kernel<<<(unsigned)n_entries, 256, 0, stream>>>(w2p, (int)w_pad, group_entries, h_scale, (_Float16*)frag,
                                                                 scale_inv);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}
