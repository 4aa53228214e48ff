// Issue rate of fp32 VALU on a gfx950 SIMD: cycles per wave64 v_fma_f32 / v_pk_fma_f32 with W waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITERS = 20000;
template <int PK>
__global__ void k(const float* in, float* out, long long* cyc) {
    float a = in[threadIdx.x & 63], b = in[(threadIdx.x + 1) & 63];
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = in[(threadIdx.x + i) & 63];
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
        if (PK) {
#pragma unroll
            for (int i = 0; i < 16; i += 2)
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*reinterpret_cast<f2*>(&v[i])) : "v"(f2{a, a}), "v"(f2{b, b}));
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    float *in, *out; long long* cyc;
    hipMalloc(&in, 256); hipMalloc(&out, 4 * 1024 * 1024); hipMalloc(&cyc, 8 * 4096);
    hipMemset(in, 0, 256);
    for (int pk = 0; pk < 2; ++pk)
        for (int wps : {1, 2, 4}) {   // waves per SIMD: one block of 256*wps threads per CU
            int threads = 256 * wps, blocks = 256;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (pk) k<1><<<blocks, threads>>>(in, out, cyc); else k<0><<<blocks, threads>>>(in, out, cyc);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            double instr = (double)ITERS * (pk ? 8 : 16);
            printf("%s waves/SIMD %d: %.3f ms, counter ticks per instr per wave %.2f, ns per instr per SIMD %.3f, lane-FMA/s chip %.1f T\n",
                   pk ? "v_pk_fma_f32" : "v_fma_f32   ", wps, ms, c / instr, ms * 1e6 / (instr * wps),
                   instr * wps * 1024 * 64 * (pk ? 2 : 1) / (ms * 1e-3) / 1e12);
        }
    return 0;
}
