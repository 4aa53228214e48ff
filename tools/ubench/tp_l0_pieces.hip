// Which piece of the l1=0 group costs what?  One entry: 32 channels, 5 couplings, 2 nodes per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int DEG = 18, W = 944, DIN = 246, SH = 32;

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ w, const float* __restrict__ sh,
                                         const float* __restrict__ x, const int* __restrict__ rowptr,
                                         const int* __restrict__ src, int n_nodes, float* out) {
    int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    int lane = threadIdx.x & 63;
    int node = wave * 2 + (lane >> 5);
    int u = lane & 31;
    if (node >= n_nodes) return;
    int beg = MODE >= 4 ? rowptr[node] : node * DEG;
    int deg = MODE >= 4 ? rowptr[node + 1] - beg : DEG;
    float acc[25];
#pragma unroll
    for (int k2 = 0; k2 < 25; ++k2) acc[k2] = 0.f;
    for (int s = 0; s < deg; ++s) {
        int e = beg + s;
        const float* row = w + (size_t)e * W;
        float wv[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) wv[c] = row[c * 32 + u];
        float y[25];
        if (MODE >= 1) {
#pragma unroll
            for (int j = 0; j < 25; ++j) y[j] = sh[(size_t)e * SH + j];
        } else {
#pragma unroll
            for (int j = 0; j < 25; ++j) y[j] = 1.f;
        }
        float xv = 1.f;
        if (MODE >= 2) { int sidx = src[e]; xv = x[(size_t)sidx * DIN + u]; }
        if (MODE >= 3) {
            const int l0[6] = {0, 1, 4, 9, 16, 25};
#pragma unroll
            for (int l = 0; l < 5; ++l) {
                float xw = xv * wv[l];
#pragma unroll
                for (int j = l0[l]; j < l0[l + 1]; ++j) acc[j] = fmaf(xw, y[j], acc[j]);
            }
        } else {
            float t = xv;
#pragma unroll
            for (int c = 0; c < 5; ++c) t += wv[c];
#pragma unroll
            for (int j = 0; j < 25; ++j) t += y[j];
            acc[0] += t;
        }
    }
    if (MODE >= 3) {
#pragma unroll
        for (int j = 0; j < 25; ++j) out[((size_t)node * 32 + u) * 25 + j] = acc[j];
    } else out[((size_t)node * 32 + u) * 25] = acc[0];
}
int main() {
    const int N = 64000, E = N * DEG;
    float *w, *sh, *x, *out; int *rowptr, *src;
    CK(hipMalloc(&w, (size_t)E * W * 4)); CK(hipMalloc(&sh, (size_t)E * SH * 4)); CK(hipMalloc(&x, (size_t)N * DIN * 4));
    CK(hipMalloc(&out, (size_t)N * 32 * 25 * 4)); CK(hipMalloc(&rowptr, (N + 1) * 4)); CK(hipMalloc(&src, (size_t)E * 4));
    std::vector<float> hw((size_t)E * 64); std::mt19937 g(1); std::uniform_real_distribution<float> d(-1, 1);
    // random (non-zero) data for honest clocks: fill w by repeating a random chunk
    for (auto& v : hw) v = d(g);
    for (size_t off = 0; off < (size_t)E * W; off += hw.size()) {
        size_t n = std::min(hw.size(), (size_t)E * W - off);
        CK(hipMemcpy(w + off, hw.data(), n * 4, hipMemcpyHostToDevice));
    }
    CK(hipMemcpy(sh, hw.data(), (size_t)E * SH * 4 <= hw.size() * 4 ? (size_t)E * SH * 4 : hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(x, hw.data(), (size_t)N * DIN * 4 <= hw.size()*4 ? (size_t)N * DIN * 4 : hw.size()*4, hipMemcpyHostToDevice));
    std::vector<int> hr(N + 1), hs(E);
    for (int i = 0; i <= N; ++i) hr[i] = i * DEG;
    for (int e = 0; e < E; ++e) { int n = e / DEG; int c = n / 64; hs[e] = c * 64 + (int)(g() % 64); }
    CK(hipMemcpy(rowptr, hr.data(), (N + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(src, hs.data(), (size_t)E * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    int grid = (N / 2 * 64 + 255) / 256;
    auto time = [&](auto launch, const char* name) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(a); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
        printf("%-44s %.3f ms\n", name, ms);
    };
    time([&] { k<0><<<grid, 256>>>(w, sh, x, rowptr, src, N, out); }, "w only");
    time([&] { k<1><<<grid, 256>>>(w, sh, x, rowptr, src, N, out); }, "w + y");
    time([&] { k<2><<<grid, 256>>>(w, sh, x, rowptr, src, N, out); }, "w + y + src->x gather");
    time([&] { k<3><<<grid, 256>>>(w, sh, x, rowptr, src, N, out); }, "w + y + x + 25 FMA + 25-float store");
    time([&] { k<4><<<grid, 256>>>(w, sh, x, rowptr, src, N, out); }, "... + rowptr");
    return 0;
}
