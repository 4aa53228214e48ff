// Issue rate of fp32 VALU on a gfx950 SIMD by encoding, for W waves per SIMD: cycles per wave64 instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITERS = 40000;
// 16 independent accumulators v[0..15]; operands a (v), b (v), s (sgpr)
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
template <int MODE>
__global__ void k(const float* in, float* out, long long* cyc) {
    float a = in[threadIdx.x & 63], b = in[(threadIdx.x + 1) & 63];
    float s = in[blockIdx.x & 63];
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = in[(threadIdx.x + i) & 63];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 pv[8], pa = {a, b}, pb = {b, a};
    for (int i = 0; i < 8; ++i) pv[i] = f2{v[i], v[i + 8]};
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
#define FMA3(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[i]) : "v"(a), "v"(b));
#define FMAC(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
#define FMACL(i) asm volatile("v_fmac_f32 %0, 0x3f9e0419, %1" : "+v"(v[i]) : "v"(a));
#define FMACS(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i]) : "s"(s), "v"(a));
#define MUL2(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(v[i]) : "v"(a));
#define FMA3L(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[i]) : "s"(s), "v"(b));
// packed fp32 (two lanes' worth of FMAs per instruction): 8 independent 64-bit accumulators
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pv[(i) & 7]) : "v"(pa), "v"(pb));
#define PKMUL(i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(pv[(i) & 7]) : "v"(pa));
// scalar fma whose multiplier is a literal, interleaved 1:1 with LDS reads of the previous result's neighbour (not dependent)
#define FMADPP(i) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(a), "v"(b));
        if (MODE == 0) { REP16(FMA3) }
        if (MODE == 1) { REP16(FMAC) }
        if (MODE == 2) { REP16(FMACL) }
        if (MODE == 3) { REP16(FMACS) }
        if (MODE == 4) { REP16(MUL2) }
        if (MODE == 5) { REP16(FMA3L) }
        if (MODE == 6) { REP16(PKFMA) }
        if (MODE == 7) { REP16(PKMUL) }
        if (MODE == 8) { REP16(FMADPP) }
    }
    long long t1 = __builtin_readcyclecounter();
    float r = 0; for (int i = 0; i < 16; ++i) r += v[i];
    for (int i = 0; i < 8; ++i) r += pv[i][0] + pv[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* name, const float* in, float* out, long long* cyc) {
    printf("%-34s", name);
    for (int wps : {1, 2, 3, 4, 6, 8}) {
        int threads = 256, blocks = 256 * wps;   // wps blocks of 4 waves per CU -> wps waves per SIMD
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            k<MODE><<<blocks, threads>>>(in, out, cyc);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        double instr = (double)ITERS * 16;
        // per-SIMD cycles per instruction from the wave's own counter: wave time / instr / waves sharing the SIMD
        // wall clock: all waves of the launch / 1024 SIMDs, at 2.4 GHz nominal; wave: its own cycle counter
        printf("  W=%d: %.2f ns/SIMD-instr (wave %.2f cyc)", wps, ms * 1e6 / (instr * wps), (double)c / instr);
    }
    printf("\n");
}
int main() {
    float *in, *out; long long* cyc;
    (void)hipMalloc(&in, 256); (void)hipMalloc(&out, 4 * 256 * 8 * 256); (void)hipMalloc(&cyc, 8 * 4096);
    (void)hipMemset(in, 0, 256);
    printf("wall-clock ns per wave64 instruction per SIMD (1 / issue rate), and the cycles one wave sees per instruction, vs waves per SIMD\n");
    run<0>("v_fma_f32 v,v,v (VOP3, 8 B)", in, out, cyc);
    run<1>("v_fmac_f32 v,v (VOP2, 4 B)", in, out, cyc);
    run<2>("v_fmac_f32 literal,v (8 B)", in, out, cyc);
    run<3>("v_fmac_f32 s,v (VOP2, 4 B)", in, out, cyc);
    run<4>("v_mul_f32 v,v (VOP2, 4 B)", in, out, cyc);
    run<5>("v_fma_f32 s,v,v (VOP3, 8 B)", in, out, cyc);
    run<6>("v_pk_fma_f32 v,v,v (2 FMA/lane)", in, out, cyc);
    run<7>("v_pk_mul_f32 v,v", in, out, cyc);
    run<8>("v_fmac_f32_dpp quad_perm", in, out, cyc);
    return 0;
}
