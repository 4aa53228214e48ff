#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <class T> __device__ __forceinline__ void cg_202(const T* xw, const float* y, T* acc) {
    { const T p = xw[0] * y[0];
      acc[0] += 1.000000000e+00f * p;
    }
    { const T p = xw[1] * y[0];
      acc[1] += 1.000000000e+00f * p;
    }
    { const T p = xw[2] * y[0];
      acc[2] += 1.000000000e+00f * p;
    }
    { const T p = xw[3] * y[0];
      acc[3] += 1.000000000e+00f * p;
    }
    { const T p = xw[4] * y[0];
      acc[4] += 1.000000000e+00f * p;
    }
}
template <class T> __device__ __forceinline__ void cg_211(const T* xw, const float* y, T* acc) {
    { const T p = xw[0] * y[0];
      acc[2] += 5.477225575e-01f * p;
    }
    { const T p = xw[0] * y[2];
      acc[0] += 5.477225575e-01f * p;
    }
    { const T p = xw[1] * y[0];
      acc[1] += 5.477225575e-01f * p;
    }
    { const T p = xw[1] * y[1];
      acc[0] += 5.477225575e-01f * p;
    }
    { const T p = xw[2] * y[0];
      acc[0] += -3.162277660e-01f * p;
    }
    { const T p = xw[2] * y[1];
      acc[1] += 6.324555320e-01f * p;
    }
    { const T p = xw[2] * y[2];
      acc[2] += -3.162277660e-01f * p;
    }
    { const T p = xw[3] * y[1];
      acc[2] += 5.477225575e-01f * p;
    }
    { const T p = xw[3] * y[2];
      acc[1] += 5.477225575e-01f * p;
    }
    { const T p = xw[4] * y[0];
      acc[0] += -5.477225575e-01f * p;
    }
    { const T p = xw[4] * y[2];
      acc[2] += 5.477225575e-01f * p;
    }
}
template <class T> __device__ __forceinline__ void cg_212(const T* xw, const float* y, T* acc) {
    { const T p = xw[0] * y[0];
      acc[1] += 4.082482905e-01f * p;
    }
    { const T p = xw[0] * y[1];
      acc[4] += 8.164965809e-01f * p;
    }
    { const T p = xw[0] * y[2];
      acc[3] += -4.082482905e-01f * p;
    }
    { const T p = xw[1] * y[0];
      acc[0] += -4.082482905e-01f * p;
    }
    { const T p = xw[1] * y[1];
      acc[3] += 4.082482905e-01f * p;
    }
    { const T p = xw[1] * y[2];
      acc[2] += -7.071067812e-01f * p;
      acc[4] += -4.082482905e-01f * p;
    }
    { const T p = xw[2] * y[0];
      acc[3] += -7.071067812e-01f * p;
    }
    { const T p = xw[2] * y[2];
      acc[1] += 7.071067812e-01f * p;
    }
    { const T p = xw[3] * y[0];
      acc[2] += 7.071067812e-01f * p;
      acc[4] += -4.082482905e-01f * p;
    }
    { const T p = xw[3] * y[1];
      acc[1] += -4.082482905e-01f * p;
    }
    { const T p = xw[3] * y[2];
      acc[0] += 4.082482905e-01f * p;
    }
    { const T p = xw[4] * y[0];
      acc[3] += 4.082482905e-01f * p;
    }
    { const T p = xw[4] * y[1];
      acc[0] += -8.164965809e-01f * p;
    }
    { const T p = xw[4] * y[2];
      acc[1] += 4.082482905e-01f * p;
    }
}
template <class T> __device__ __forceinline__ void cg_213(const T* xw, const float* y, T* acc) {
    { const T p = xw[0] * y[0];
      acc[4] += -1.825741858e-01f * p;
      acc[6] += -7.071067812e-01f * p;
    }
    { const T p = xw[0] * y[1];
      acc[1] += 5.773502692e-01f * p;
    }
    { const T p = xw[0] * y[2];
      acc[0] += 7.071067812e-01f * p;
      acc[2] += -1.825741858e-01f * p;
    }
    { const T p = xw[1] * y[0];
      acc[3] += -4.472135955e-01f * p;
      acc[5] += -5.773502692e-01f * p;
    }
    { const T p = xw[1] * y[1];
      acc[2] += 7.302967433e-01f * p;
    }
    { const T p = xw[1] * y[2];
      acc[1] += 5.773502692e-01f * p;
    }
    { const T p = xw[2] * y[0];
      acc[2] += 6.324555320e-01f * p;
    }
    { const T p = xw[2] * y[1];
      acc[3] += 7.745966692e-01f * p;
    }
    { const T p = xw[2] * y[2];
      acc[4] += 6.324555320e-01f * p;
    }
    { const T p = xw[3] * y[0];
      acc[1] += 5.773502692e-01f * p;
    }
    { const T p = xw[3] * y[1];
      acc[4] += 7.302967433e-01f * p;
    }
    { const T p = xw[3] * y[2];
      acc[3] += -4.472135955e-01f * p;
      acc[5] += 5.773502692e-01f * p;
    }
    { const T p = xw[4] * y[0];
      acc[0] += 7.071067812e-01f * p;
      acc[2] += 1.825741858e-01f * p;
    }
    { const T p = xw[4] * y[1];
      acc[5] += 5.773502692e-01f * p;
    }
    { const T p = xw[4] * y[2];
      acc[4] += -1.825741858e-01f * p;
      acc[6] += 7.071067812e-01f * p;
    }
}
template <class T> __device__ __forceinline__ void cg_220(const T* xw, const float* y, T* acc) {
    { const T p = xw[0] * y[0];
      acc[0] += 4.472135955e-01f * p;
    }
    { const T p = xw[1] * y[1];
      acc[0] += 4.472135955e-01f * p;
    }
    { const T p = xw[2] * y[2];
      acc[0] += 4.472135955e-01f * p;
    }
    { const T p = xw[3] * y[3];
      acc[0] += 4.472135955e-01f * p;
    }
    { const T p = xw[4] * y[4];
      acc[0] += 4.472135955e-01f * p;
    }
}
template <class T> __device__ __forceinline__ void cg_221(const T* xw, const float* y, T* acc) {
    { const T p = xw[0] * y[1];
      acc[0] += -3.162277660e-01f * p;
    }
    { const T p = xw[0] * y[3];
      acc[2] += 3.162277660e-01f * p;
    }
    { const T p = xw[0] * y[4];
      acc[1] += -6.324555320e-01f * p;
    }
    { const T p = xw[1] * y[0];
      acc[0] += 3.162277660e-01f * p;
    }
    { const T p = xw[1] * y[2];
      acc[2] += 5.477225575e-01f * p;
    }
    { const T p = xw[1] * y[3];
      acc[1] += -3.162277660e-01f * p;
    }
    { const T p = xw[1] * y[4];
      acc[2] += 3.162277660e-01f * p;
    }
    { const T p = xw[2] * y[1];
      acc[2] += -5.477225575e-01f * p;
    }
    { const T p = xw[2] * y[3];
      acc[0] += 5.477225575e-01f * p;
    }
    { const T p = xw[3] * y[0];
      acc[2] += -3.162277660e-01f * p;
    }
    { const T p = xw[3] * y[1];
      acc[1] += 3.162277660e-01f * p;
    }
    { const T p = xw[3] * y[2];
      acc[0] += -5.477225575e-01f * p;
    }
    { const T p = xw[3] * y[4];
      acc[0] += 3.162277660e-01f * p;
    }
    { const T p = xw[4] * y[0];
      acc[1] += 6.324555320e-01f * p;
    }
    { const T p = xw[4] * y[1];
      acc[2] += -3.162277660e-01f * p;
    }
    { const T p = xw[4] * y[3];
      acc[0] += -3.162277660e-01f * p;
    }
}
template <class T> __device__ __forceinline__ void cg_222(const T* xw, const float* y, T* acc) {
    { const T p = xw[0] * y[0];
      acc[2] += -5.345224838e-01f * p;
    }
    { const T p = xw[0] * y[1];
      acc[3] += 4.629100499e-01f * p;
    }
    { const T p = xw[0] * y[2];
      acc[0] += -5.345224838e-01f * p;
    }
    { const T p = xw[0] * y[3];
      acc[1] += 4.629100499e-01f * p;
    }
    { const T p = xw[1] * y[0];
      acc[3] += 4.629100499e-01f * p;
    }
    { const T p = xw[1] * y[1];
      acc[2] += 2.672612419e-01f * p;
      acc[4] += -4.629100499e-01f * p;
    }
    { const T p = xw[1] * y[2];
      acc[1] += 2.672612419e-01f * p;
    }
    { const T p = xw[1] * y[3];
      acc[0] += 4.629100499e-01f * p;
    }
    { const T p = xw[1] * y[4];
      acc[1] += -4.629100499e-01f * p;
    }
    { const T p = xw[2] * y[0];
      acc[0] += -5.345224838e-01f * p;
    }
    { const T p = xw[2] * y[1];
      acc[1] += 2.672612419e-01f * p;
    }
    { const T p = xw[2] * y[2];
      acc[2] += 5.345224838e-01f * p;
    }
    { const T p = xw[2] * y[3];
      acc[3] += 2.672612419e-01f * p;
    }
    { const T p = xw[2] * y[4];
      acc[4] += -5.345224838e-01f * p;
    }
    { const T p = xw[3] * y[0];
      acc[1] += 4.629100499e-01f * p;
    }
    { const T p = xw[3] * y[1];
      acc[0] += 4.629100499e-01f * p;
    }
    { const T p = xw[3] * y[2];
      acc[3] += 2.672612419e-01f * p;
    }
    { const T p = xw[3] * y[3];
      acc[2] += 2.672612419e-01f * p;
      acc[4] += 4.629100499e-01f * p;
    }
    { const T p = xw[3] * y[4];
      acc[3] += 4.629100499e-01f * p;
    }
    { const T p = xw[4] * y[1];
      acc[1] += -4.629100499e-01f * p;
    }
    { const T p = xw[4] * y[2];
      acc[4] += -5.345224838e-01f * p;
    }
    { const T p = xw[4] * y[3];
      acc[3] += 4.629100499e-01f * p;
    }
    { const T p = xw[4] * y[4];
      acc[2] += -5.345224838e-01f * p;
    }
}
template <class T> __device__ __forceinline__ void cg_223(const T* xw, const float* y, T* acc) {
    { const T p = xw[0] * y[1];
      acc[0] += 5.000000000e-01f * p;
      acc[2] += 3.872983346e-01f * p;
    }
    { const T p = xw[0] * y[2];
      acc[5] += 7.071067812e-01f * p;
    }
    { const T p = xw[0] * y[3];
      acc[4] += -3.872983346e-01f * p;
      acc[6] += 5.000000000e-01f * p;
    }
    { const T p = xw[0] * y[4];
      acc[3] += 3.162277660e-01f * p;
    }
    { const T p = xw[1] * y[0];
      acc[0] += -5.000000000e-01f * p;
      acc[2] += -3.872983346e-01f * p;
    }
    { const T p = xw[1] * y[2];
      acc[4] += 4.472135955e-01f * p;
    }
    { const T p = xw[1] * y[3];
      acc[3] += -6.324555320e-01f * p;
    }
    { const T p = xw[1] * y[4];
      acc[4] += -3.872983346e-01f * p;
      acc[6] += -5.000000000e-01f * p;
    }
    { const T p = xw[2] * y[0];
      acc[5] += -7.071067812e-01f * p;
    }
    { const T p = xw[2] * y[1];
      acc[4] += -4.472135955e-01f * p;
    }
    { const T p = xw[2] * y[3];
      acc[2] += 4.472135955e-01f * p;
    }
    { const T p = xw[2] * y[4];
      acc[1] += 7.071067812e-01f * p;
    }
    { const T p = xw[3] * y[0];
      acc[4] += 3.872983346e-01f * p;
      acc[6] += -5.000000000e-01f * p;
    }
    { const T p = xw[3] * y[1];
      acc[3] += 6.324555320e-01f * p;
    }
    { const T p = xw[3] * y[2];
      acc[2] += -4.472135955e-01f * p;
    }
    { const T p = xw[3] * y[4];
      acc[0] += 5.000000000e-01f * p;
      acc[2] += -3.872983346e-01f * p;
    }
    { const T p = xw[4] * y[0];
      acc[3] += -3.162277660e-01f * p;
    }
    { const T p = xw[4] * y[1];
      acc[4] += 3.872983346e-01f * p;
      acc[6] += 5.000000000e-01f * p;
    }
    { const T p = xw[4] * y[2];
      acc[1] += -7.071067812e-01f * p;
    }
    { const T p = xw[4] * y[3];
      acc[0] += -5.000000000e-01f * p;
      acc[2] += 3.872983346e-01f * p;
    }
}
template <class T> __device__ __forceinline__ void cg_224(const T* xw, const float* y, T* acc) {
    { const T p = xw[0] * y[0];
      acc[4] += 1.195228609e-01f * p;
      acc[8] += -7.071067812e-01f * p;
    }
    { const T p = xw[0] * y[1];
      acc[5] += -1.889822365e-01f * p;
      acc[7] += -5.000000000e-01f * p;
    }
    { const T p = xw[0] * y[2];
      acc[2] += 4.629100499e-01f * p;
    }
    { const T p = xw[0] * y[3];
      acc[1] += 5.000000000e-01f * p;
      acc[3] += -1.889822365e-01f * p;
    }
    { const T p = xw[0] * y[4];
      acc[0] += 7.071067812e-01f * p;
    }
    { const T p = xw[1] * y[0];
      acc[5] += -1.889822365e-01f * p;
      acc[7] += -5.000000000e-01f * p;
    }
    { const T p = xw[1] * y[1];
      acc[4] += -4.780914437e-01f * p;
      acc[6] += -5.345224838e-01f * p;
    }
    { const T p = xw[1] * y[2];
      acc[3] += 6.546536707e-01f * p;
    }
    { const T p = xw[1] * y[3];
      acc[2] += 5.345224838e-01f * p;
    }
    { const T p = xw[1] * y[4];
      acc[1] += 5.000000000e-01f * p;
      acc[3] += 1.889822365e-01f * p;
    }
    { const T p = xw[2] * y[0];
      acc[2] += 4.629100499e-01f * p;
    }
    { const T p = xw[2] * y[1];
      acc[3] += 6.546536707e-01f * p;
    }
    { const T p = xw[2] * y[2];
      acc[4] += 7.171371656e-01f * p;
    }
    { const T p = xw[2] * y[3];
      acc[5] += 6.546536707e-01f * p;
    }
    { const T p = xw[2] * y[4];
      acc[6] += 4.629100499e-01f * p;
    }
    { const T p = xw[3] * y[0];
      acc[1] += 5.000000000e-01f * p;
      acc[3] += -1.889822365e-01f * p;
    }
    { const T p = xw[3] * y[1];
      acc[2] += 5.345224838e-01f * p;
    }
    { const T p = xw[3] * y[2];
      acc[5] += 6.546536707e-01f * p;
    }
    { const T p = xw[3] * y[3];
      acc[4] += -4.780914437e-01f * p;
      acc[6] += 5.345224838e-01f * p;
    }
    { const T p = xw[3] * y[4];
      acc[5] += -1.889822365e-01f * p;
      acc[7] += 5.000000000e-01f * p;
    }
    { const T p = xw[4] * y[0];
      acc[0] += 7.071067812e-01f * p;
    }
    { const T p = xw[4] * y[1];
      acc[1] += 5.000000000e-01f * p;
      acc[3] += 1.889822365e-01f * p;
    }
    { const T p = xw[4] * y[2];
      acc[6] += 4.629100499e-01f * p;
    }
    { const T p = xw[4] * y[3];
      acc[5] += -1.889822365e-01f * p;
      acc[7] += 5.000000000e-01f * p;
    }
    { const T p = xw[4] * y[4];
      acc[4] += 1.195228609e-01f * p;
      acc[8] += 7.071067812e-01f * p;
    }
}
constexpr int NC=9, NACC=45, D1=5, NY=9, OPS=343;
template <class T> __device__ __forceinline__ void apply(const T* x, const float* y, const T* w, T* acc) {
    { T xw[5]; for (int i = 0; i < 5; ++i) xw[i] = w[0] * x[i]; cg_202<T>(xw, y + 0, acc + 0); }
    { T xw[5]; for (int i = 0; i < 5; ++i) xw[i] = w[1] * x[i]; cg_211<T>(xw, y + 1, acc + 5); }
    { T xw[5]; for (int i = 0; i < 5; ++i) xw[i] = w[2] * x[i]; cg_212<T>(xw, y + 1, acc + 8); }
    { T xw[5]; for (int i = 0; i < 5; ++i) xw[i] = w[3] * x[i]; cg_213<T>(xw, y + 1, acc + 13); }
    { T xw[5]; for (int i = 0; i < 5; ++i) xw[i] = w[4] * x[i]; cg_220<T>(xw, y + 4, acc + 20); }
    { T xw[5]; for (int i = 0; i < 5; ++i) xw[i] = w[5] * x[i]; cg_221<T>(xw, y + 4, acc + 21); }
    { T xw[5]; for (int i = 0; i < 5; ++i) xw[i] = w[6] * x[i]; cg_222<T>(xw, y + 4, acc + 24); }
    { T xw[5]; for (int i = 0; i < 5; ++i) xw[i] = w[7] * x[i]; cg_223<T>(xw, y + 4, acc + 29); }
    { T xw[5]; for (int i = 0; i < 5; ++i) xw[i] = w[8] * x[i]; cg_224<T>(xw, y + 4, acc + 36); }
}

template <class T, int MINB> __global__ __launch_bounds__(256, MINB) void k(const float* in, float* out, int iters) {
    const int l = blockIdx.x * 256 + threadIdx.x;
    T x[D1], w[NC], acc[NACC]; float y[NY];
    for (int i = 0; i < D1; ++i) x[i] = T(in[(l + i) & 1023]);
    for (int i = 0; i < NC; ++i) w[i] = T(in[(l + 7 * i) & 1023]);
    for (int i = 0; i < NY; ++i) y[i] = in[(3 * i + (l >> 6)) & 1023];
    for (int i = 0; i < NACC; ++i) acc[i] = T(0.f);
    for (int it = 0; it < iters; ++it) {
        apply<T>(x, y, w, acc);
        for (int i = 0; i < D1; ++i) x[i] += T(1e-3f);   // new "edge"
        y[it % NY] += 1e-3f;
    }
    T s = T(0.f);
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[l] = sizeof(T) == 4 ? *(float*)&s : ((float*)&s)[0] + ((float*)&s)[1];
}
int main() {
    float *in, *out; hipMalloc(&in, 4096); hipMalloc(&out, 1 << 22);
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 0.001f * (i % 97) - 0.03f;
    hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000;
    auto run = [&](auto kern, int blocks, int chan_per_lane, const char* name) {
        kern<<<blocks, 256>>>(in, out, 10); hipDeviceSynchronize();
        hipEventRecord(a); kern<<<blocks, 256>>>(in, out, iters); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        double chan_edges = (double)blocks * 256 * chan_per_lane * iters;
        printf("%-28s %8.3f ms  %.2f G channel-edges/s  (%.1f TFLOP/s of the %d-op schedule)\n", name, ms, chan_edges / ms / 1e6,
               chan_edges * OPS * 2 / ms / 1e9, OPS);
    };
    const int full = 256 * 4 * 3;   // 3 waves per SIMD on 256 CUs (blocks of 4 waves)
    run(k<float, 3>, full, 1, "scalar, 3 waves/SIMD");
    run(k<f2, 3>, full / 2, 2, "packed x2, 1.5 waves/SIMD eq.");
    run(k<f2, 3>, full, 2, "packed x2, 3 waves/SIMD");
    run(k<f2, 2>, 256 * 4 * 2, 2, "packed x2, 2 waves/SIMD");
    run(k<float, 4>, 256 * 4 * 4, 1, "scalar, 4 waves/SIMD");
    return 0;
}
