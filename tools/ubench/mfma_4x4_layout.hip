// Which lane / register holds what for v_mfma_f32_4x4x1_16B_f32 (16 independent 4x4 outer products, K = 1)?
// Hypothesis checked here: A_b[i] in lane 4 b + i, B_b[j] in lane 4 b + j, D_b[i][j] in register i of lane 4 b + j.
// Also times the instruction (dependent chain vs 4 independent accumulators).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void layout(float* out) {
    const int lane = threadIdx.x;
    const int b = lane >> 2, r = lane & 3;
    const float a = 100.0f * b + 10.0f * (r + 1);   // A_b[i = r]
    const float bb = 1.0f + 0.001f * b + 0.01f * r;   // B_b[j = r]
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, bb, d, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = d[i];
}

__global__ void rate(float* out, int iters, int mode) {
    const int lane = threadIdx.x & 63;
    float a = 1.0f + lane, b = 0.5f;
    f32x4 d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0, d3 = d0;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (mode == 0) {
            d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d0, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d0, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d0, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d0, 0, 0, 0);
        } else {
            d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d2, 0, 0, 0);
            d3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d3, 0, 0, 0);
        }
    }
    long long t1 = clock64();
    f32x4 s = d0 + d1 + d2 + d3;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (4.0f * iters);
    if (s[0] == 12345.f) out[1] = s[1];
}

int main() {
    float* d; hipMalloc(&d, 64 * 4 * 4);
    layout<<<1, 64>>>(d);
    float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int i = 0; i < 4; ++i) {
            const int b = lane >> 2, j = lane & 3;
            const float want = (100.0f * b + 10.0f * (i + 1)) * (1.0f + 0.001f * b + 0.01f * j);
            if (fabsf(h[lane * 4 + i] - want) > 1e-3f * fabsf(want)) ++bad;
        }
    printf("layout hypothesis (A_b[i]: lane 4b+i; B_b[j]: lane 4b+j; D_b[i][j]: reg i, lane 4b+j): %s (%d mismatches)\n",
           bad ? "WRONG" : "confirmed", bad);
    if (bad) for (int lane = 0; lane < 8; ++lane) printf("lane %d: %g %g %g %g\n", lane, h[lane*4], h[lane*4+1], h[lane*4+2], h[lane*4+3]);
    for (int mode = 0; mode < 2; ++mode) {
        rate<<<1, 64>>>(d, 4096, mode); hipDeviceSynchronize();
        rate<<<1, 64>>>(d, 4096, mode);
        hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("%s: %.2f clock64 ticks per instruction (1 wave)\n", mode ? "4 independent accumulators" : "dependent chain", h[0]);
    }
    return 0;
}
