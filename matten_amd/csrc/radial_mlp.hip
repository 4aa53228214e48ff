// Radial MLP on the fp32 matrix cores: e3nn FullyConnectedNet([nb, 32, 32, W], silu)
// (reference nn/utils.py:246-251,260; the only MFMA-shaped op on the path).
//
// Formulation: out^T = W2^T . silu(W1^T . silu(W0^T . rbf^T)) with EDGES as the MFMA N dimension.
// v_mfma_f32_16x16x4_f32: lane l = (g = l>>4, c = l&15)
//     A[m=c][k=g]   (weights, transposed)       B[k=g][n=c] (activations, n = edge)
//     D[row = 4g + r][col = c], r = 0..3          (row = output feature, col = edge)
// so after a layer, lane (g, c) holds features {16t + 4g + r} of edge c.  The next layer's
// contraction index may be enumerated in any order as long as A and B agree, so k-step kk
// (0..7) of lane group g is *defined* as feature pi(kk, g) = 16 (kk>>2) + 4g + (kk&3): exactly
// the register the lane already holds.  The whole 3-layer chain therefore runs in registers,
// with no LDS transposes, and the final D fragment is 4 consecutive features of one edge
// => one 16-byte store per lane per tile.
#include "common.h"
#include "sh.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// N tiles (16 edges each) per wave: 4 on large graphs (weights fetched once per 64 edges), 1 on small ones (a training
// batch of 32 crystals has ~4 k edges: parallelism matters more than the amortisation)
constexpr int WAVES = 4;     // waves per workgroup
constexpr int HID = 32;      // hidden width (2 M-tiles)

__device__ __forceinline__ float silu(float z) { return z / (1.0f + expf(-z)); }

template <int KS0, int NT>  // number of k-steps of the first layer: nb_pad / 4; edge tiles per wave
__global__ __launch_bounds__(WAVES * 64) void radial_mlp_kernel(
    const float4* __restrict__ geom, int64_t E, int n_basis, float r_start, float r_end,
    const float* __restrict__ w0p, const float* __restrict__ w1p, const float* __restrict__ w2p, int w_pad,
    void* __restrict__ w_edge, int out_bf16) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int64_t e0 = ((int64_t)blockIdx.x * WAVES + wave) * (NT * 16);
    if (e0 >= E) return;

    // ---- layer 0: B = rbf^T, computed per lane for its own edge and its own k = 4*kk + g ----
    f32x4 h[NT][2];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        int64_t e = e0 + nt * 16 + c;
        float len = geom[e < E ? e : E - 1].w;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < KS0; ++kk) {
            int k = 4 * kk + g;
            float b = (k < n_basis) ? matten::bessel_basis(len, k, n_basis, r_start, r_end) : 0.0f;
            float a0 = w0p[k * HID + c];
            float a1 = w0p[k * HID + 16 + c];
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, acc1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc0[r] = silu(acc0[r]);
            acc1[r] = silu(acc1[r]);
        }
        h[nt][0] = acc0;
        h[nt][1] = acc1;
    }

    // ---- layer 1: 32 -> 32 ----
    {
        float a[2][8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            int k = 16 * (kk >> 2) + 4 * g + (kk & 3);
            a[0][kk] = w1p[k * HID + c];
            a[1][kk] = w1p[k * HID + 16 + c];
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                float b = h[nt][kk >> 2][kk & 3];
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0][kk], b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1][kk], b, acc1, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc0[r] = silu(acc0[r]);
                acc1[r] = silu(acc1[r]);
            }
            h[nt][0] = acc0;
            h[nt][1] = acc1;
        }
    }

    // ---- layer 2: 32 -> w_pad, streamed tile by tile ----
    const int n_mt = w_pad >> 4;
    for (int mt = 0; mt < n_mt; ++mt) {
        float a[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            int k = 16 * (kk >> 2) + 4 * g + (kk & 3);
            a[kk] = w2p[(int64_t)k * w_pad + mt * 16 + c];
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk], h[nt][kk >> 2][kk & 3], acc, 0, 0, 0);
            int64_t e = e0 + nt * 16 + c;
            if (e < E) {
                const int64_t o = e * w_pad + mt * 16 + 4 * g;
                if (out_bf16) {   // opt-in bf16 storage of the per-edge weights (training): 8 bytes per lane
                    uint2 pk;
                    pk.x = (uint32_t)matten_f32_to_bf16(acc[0]) | ((uint32_t)matten_f32_to_bf16(acc[1]) << 16);
                    pk.y = (uint32_t)matten_f32_to_bf16(acc[2]) | ((uint32_t)matten_f32_to_bf16(acc[3]) << 16);
                    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(w_edge) + o) = pk;
                } else {
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(w_edge) + o) = acc;
                }
            }
        }
    }
}

}  // namespace

extern "C" int matten_radial_mlp(const float* geom_sorted, int64_t n_edges, int n_basis, float r_start, float r_end,
                                 const float* w0p, int nb_pad, const float* w1p, const float* w2p, int hidden,
                                 int w_pad, float act_cst, void* w_edge, int out_is_bf16, matten_stream_t stream_) {
    (void)act_cst;  // folded into w1p / w2p by the host-side prepack
    hipStream_t stream = (hipStream_t)stream_;
    if (n_edges < 0 || hidden != HID || (w_pad & 15) || w_pad <= 0 || (nb_pad & 3) || nb_pad < n_basis || nb_pad > 16)
        return MATTEN_EINVAL;
    if (n_edges == 0) return MATTEN_OK;
    if (!geom_sorted || !w0p || !w1p || !w2p || !w_edge) return MATTEN_EINVAL;
#define LAUNCH(K, NTT)                                                                                               \
    radial_mlp_kernel<K, NTT><<<(unsigned)matten_cdiv(n_edges, WAVES * NTT * 16), WAVES * 64, 0, stream>>>(           \
        (const float4*)geom_sorted, n_edges, n_basis, r_start, r_end, w0p, w1p, w2p, w_pad, w_edge, out_is_bf16)
#define LAUNCH_K(NTT)                     \
    switch (nb_pad >> 2) {                \
        case 1: LAUNCH(1, NTT); break;    \
        case 2: LAUNCH(2, NTT); break;    \
        case 3: LAUNCH(3, NTT); break;    \
        default: LAUNCH(4, NTT); break;   \
    }
    if (n_edges >= 64 * 1024) { LAUNCH_K(4) } else { LAUNCH_K(1) }
#undef LAUNCH_K
#undef LAUNCH
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}
