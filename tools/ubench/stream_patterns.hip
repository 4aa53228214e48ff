// Micro-benchmark: how fast can a wave-per-node-group kernel stream the per-edge weight rows?
// Patterns mimic tp_block_kernel's reads of w_edge[E, W] (W=944) for the l1=0 group (640 B per edge).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int DEG = 18;
// lanes: 2 nodes x 32 channels; each lane reads NC floats per edge
template <int NC, int MODE, int UNROLL>
__global__ __launch_bounds__(256) void k_rows(const float* __restrict__ w, int w_pad, int n_nodes, float* out) {
    int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    int lane = threadIdx.x & 63;
    int node = wave * 2 + (lane >> 5);
    int u = lane & 31;
    if (node >= n_nodes) return;
    float acc = 0.f;
    const float* base = w + (size_t)node * DEG * w_pad;
#pragma unroll UNROLL
    for (int s = 0; s < DEG; ++s) {
        const float* row = base + (size_t)s * w_pad;
        if (MODE == 0) {          // [u][c]: per-lane contiguous NC floats
#pragma unroll
            for (int c = 0; c < NC; ++c) acc += row[u * NC + c];
        } else {                  // [c][u]: per-c coalesced dword loads
#pragma unroll
            for (int c = 0; c < NC; ++c) acc += row[c * 32 + u];
        }
    }
    out[(size_t)node * 32 + u] = acc;
}
// one wave per edge row chunk: fully coalesced float4 streaming of the same bytes
__global__ __launch_bounds__(256) void k_flat(const float4* __restrict__ w, size_t n4, float* out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    for (; i < n4; i += (size_t)gridDim.x * 256) { float4 v = w[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.f) out[0] = acc;
}
int main() {
    const int N = 64000, E = N * DEG, W = 944;
    float *w, *out;
    CK(hipMalloc(&w, (size_t)E * W * 4)); CK(hipMalloc(&out, (size_t)N * 32 * 4));
    CK(hipMemset(w, 0, (size_t)E * W * 4));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto time = [&](auto launch, const char* name, double bytes) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(a); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
        printf("%-40s %.3f ms  %.2f TB/s\n", name, ms, bytes / ms / 1e9);
    };
    int grid = (N / 2 * 64 + 255) / 256;
    double bytes5 = (double)E * 160 * 4;
    time([&] { k_rows<5, 0, 1><<<grid, 256>>>(w, W, N, out); }, "rows [u][c] NC=5 unroll1", bytes5);
    time([&] { k_rows<5, 1, 1><<<grid, 256>>>(w, W, N, out); }, "rows [c][u] NC=5 unroll1", bytes5);
    time([&] { k_rows<5, 1, 3><<<grid, 256>>>(w, W, N, out); }, "rows [c][u] NC=5 unroll3", bytes5);
    time([&] { k_rows<5, 1, 18><<<grid, 256>>>(w, W, N, out); }, "rows [c][u] NC=5 unroll18", bytes5);
    time([&] { k_rows<5, 0, 18><<<grid, 256>>>(w, W, N, out); }, "rows [u][c] NC=5 unroll18", bytes5);
    size_t n4 = (size_t)E * W / 4;
    time([&] { k_flat<<<256 * 16, 256>>>((const float4*)w, n4, out); }, "flat float4 whole array", (double)E * W * 4);
    return 0;
}
