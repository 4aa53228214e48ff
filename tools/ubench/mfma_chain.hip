// Issue behaviour of one wave's MFMA stream on gfx950: cycles per MFMA as a function of the number of independent
// accumulator chains (NCH) and of the waves per SIMD, for v_mfma_f32_16x16x4_f32 and v_mfma_f32_16x16x32_bf16;
// and the same with independent fp32 FMAs interleaved (co-execution inside a wave).
// hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize mfma_chain.hip -o mfma_chain && ./mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int ITERS = 2000;

template <int NCH, bool BF16, int NFMA>
__global__ __launch_bounds__(1024) void k(const float* in, float* out) {
    float a = in[threadIdx.x], b = in[1024 + threadIdx.x];
    bf16x8 ah, bh;
#pragma unroll
    for (int i = 0; i < 8; ++i) { ah[i] = (__bf16)(a + i); bh[i] = (__bf16)(b - i); }
    f32x4 d[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) d[c] = f32x4{0, 0, 0, 0};
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < 8 / NCH; ++r) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (BF16) d[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, d[c], 0, 0, 0);
                else d[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d[c], 0, 0, 0);
#pragma unroll
                for (int f = 0; f < NFMA; ++f) v[f] = __builtin_fmaf(v[f], a, b);
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) s += d[c][0] + d[c][1] + d[c][2] + d[c][3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
}

template <int NCH, bool BF16, int NFMA>
void run(const float* in, float* out, int waves_per_simd) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int threads = 256 * waves_per_simd;
    k<NCH, BF16, NFMA><<<256, threads>>>(in, out);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<NCH, BF16, NFMA><<<256, threads>>>(in, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    const double cyc = ms * 1e-3 * 2.4e9 / (ITERS * 8.0 * waves_per_simd);
    printf("%s chains %d fma/mfma %d waves/SIMD %d: %.3f ms, %.1f cycles per MFMA (+%d FMA) per SIMD\n", BF16 ? "bf16 16x16x32" : "f32  16x16x4 ",
           NCH, NFMA, waves_per_simd, ms, cyc, NFMA);
}

int main() {
    float *in, *out;
    hipMalloc(&in, 2048 * 4);
    hipMalloc(&out, 256 * 1024 * 4);
    hipMemset(in, 0, 2048 * 4);
    for (int w = 1; w <= 4; w *= 2) {
        run<1, false, 0>(in, out, w); run<2, false, 0>(in, out, w); run<4, false, 0>(in, out, w); run<8, false, 0>(in, out, w);
        run<1, true, 0>(in, out, w);  run<2, true, 0>(in, out, w);  run<4, true, 0>(in, out, w);  run<8, true, 0>(in, out, w);
    }
    for (int w = 1; w <= 4; w *= 2) {
        run<8, false, 4>(in, out, w); run<8, false, 8>(in, out, w);
        run<8, true, 2>(in, out, w);  run<8, true, 4>(in, out, w); run<8, true, 8>(in, out, w);
        run<2, true, 4>(in, out, w);
    }
    return 0;
}
