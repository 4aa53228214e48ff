// Do v_mfma_f32_16x16x4_f32 and fp32 VALU work overlap on a gfx950 SIMD, (a) from different waves, (b) inside one wave?
// One 512-thread workgroup per CU (LDS-limited), i.e. two waves per SIMD: waves 0-3 and 4-7 share SIMDs 0-3.
//   mode 0: all waves MFMA chains          mode 1: all waves FMA chains
//   mode 2: waves 0-3 MFMA, waves 4-7 FMA  mode 3: every wave both, interleaved in program order (1 MFMA : 8 FMA)
//   mode 4: every wave both, phase after phase (8 MFMA, then 64 FMA)
// hipcc --offload-arch=gfx950 -O3 mfma_valu_coissue.hip -o mfma_valu_coissue && ./mfma_valu_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#ifdef USE_BF16  // XDL op: v_mfma_f32_16x16x32_bf16 (4 passes) instead of the fp32 one (8 passes)
#define MFMA(A, B, D) D = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A##h, B##h, D, 0, 0, 0)
#else
#define MFMA(A, B, D) D = __builtin_amdgcn_mfma_f32_16x16x4f32(A, B, D, 0, 0, 0)
#endif
constexpr int ITERS = 4000;

__device__ __forceinline__ void fma8(float (&v)[8], float a, float b) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], a, b);
}

template <int MODE>
__global__ __launch_bounds__(512) void k(const float* in, float* out) {
    extern __shared__ float lds[];
    // role by the wave's position on ITS SIMD (HW_ID bits 5:4), not by the wave index: the dispatcher's wave -> SIMD
    // placement is not documented
    __shared__ int cnt[4];
    if (threadIdx.x < 4) cnt[threadIdx.x] = 0;
    __syncthreads();
    int slot = 0;
    const int simd = (__builtin_amdgcn_s_getreg((4 << 0) | (4 << 6) | ((2 - 1) << 11)));  // hwreg(HW_REG_HW_ID, 4, 2)
    if ((threadIdx.x & 63) == 0) slot = atomicAdd(&cnt[simd], 1);
    slot = __builtin_amdgcn_readfirstlane(slot);
    const int wave = (slot & 1) ? 4 : 0;  // first wave on a SIMD: MFMA role, second: FMA role (mode 2)
    if (MODE == 2 && blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[256 * 512 + (threadIdx.x >> 6)] = simd * 16 + slot;
    float a = in[threadIdx.x], b = in[512 + threadIdx.x];
    bf16x8 ah, bh;
#pragma unroll
    for (int i = 0; i < 8; ++i) { ah[i] = (__bf16)(a + i); bh[i] = (__bf16)(b - i); }
    f32x4 d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0};
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    const bool do_mfma = MODE == 0 || (MODE == 2 && wave < 4) || MODE >= 3;
    const bool do_fma = MODE == 1 || (MODE == 2 && wave >= 4) || MODE >= 3;
    for (int it = 0; it < ITERS; ++it) {
        if (MODE == 3) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                MFMA(a, b, d0);
                fma8(v, a, b);
                MFMA(b, a, d1);
                fma8(v, a, b);
            }
        } else {
            if (do_mfma) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    MFMA(a, b, d0);
                    MFMA(b, a, d1);
                }
            }
            if (do_fma) {
#pragma unroll
                for (int r = 0; r < 8; ++r) fma8(v, a, b);
            }
        }
    }
    float s = d0[0] + d0[1] + d0[2] + d0[3] + d1[0] + d1[1] + d1[2] + d1[3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (s == 12345.f) lds[threadIdx.x] = s;
}

template <int MODE>
float run(const float* in, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t lds = 100 * 1024;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k<MODE><<<256, 512, lds>>>(in, out);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<MODE><<<256, 512, lds>>>(in, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main() {
    float *in, *out;
    hipMalloc(&in, 1024 * 4);
    hipMalloc(&out, 256 * 512 * 4 + 64);
    hipMemset(in, 0, 1024 * 4);
    const float t0 = run<0>(in, out), t1 = run<1>(in, out), t2 = run<2>(in, out), t3 = run<3>(in, out), t4 = run<4>(in, out);
    // per wave and iteration: 8 MFMA (8 x 32 cycles of matrix pipe) and 64 FMA (64 x 4 cycles of VALU)
    printf("mode0 all-MFMA %.3f ms  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", t0, t0 * 1e-3 * 2.4e9 / (ITERS * 8 * 2));
    printf("mode1 all-FMA  %.3f ms  (%.2f cycles/FMA/SIMD)\n", t1, t1 * 1e-3 * 2.4e9 / (ITERS * 64 * 2));
    printf("mode2 wave-specialised (half the MFMA + half the FMA work of modes 0/1): %.3f ms; sum of halves %.3f, max %.3f\n", t2,
           (t0 + t1) / 2, (t0 > t1 ? t0 : t1) / 2);
    printf("mode3 interleaved in one wave: %.3f ms; sum %.3f, max %.3f\n", t3, t0 + t1, t0 > t1 ? t0 : t1);
    printf("mode4 phases in one wave (2 waves/SIMD): %.3f ms; sum %.3f, max %.3f\n", t4, t0 + t1, t0 > t1 ? t0 : t1);
    float map[8];
    hipMemcpy(map, out + 256 * 512, 32, hipMemcpyDeviceToHost);
    printf("wave -> (SIMD, arrival slot) of workgroup 0:");
    for (int w = 0; w < 8; ++w) printf(" w%d:(%d,%d)", w, (int)map[w] / 16, (int)map[w] % 16);
    printf("\n");
    return 0;
}
