// Micro-benchmark: throughput of no-return fp32 global atomic adds (global_atomic_add_f32, executed in L2) against
// plain stores, in the access pattern a fused "lin2 in the tensor-product epilogue" would have: 40 workgroups per
// 64-node tile (tile pinned to an XCD as in tp_fused_kernel), every wave instruction adds SEG contiguous floats to
// 64/SEG different node rows of its tile's out[64, D] block (75 KB: L2 resident while the tile's workgroups run).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int D = 292, TILE = 64, WG_PER_TILE = 40, N_XCD = 8;

template <int MODE, int SEG>
__global__ __launch_bounds__(256) void k(float* __restrict__ out, int n_tiles, int iters) {
    const int xcd = blockIdx.x % N_XCD, q = blockIdx.x / N_XCD;
    const int tile = (q / WG_PER_TILE) * N_XCD + xcd;
    if (tile >= n_tiles) return;
    const int wg = q % WG_PER_TILE;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s = lane / SEG, within = lane % SEG;
    float* base = out + (size_t)tile * TILE * D;
    for (int i = 0; i < iters; ++i) {
        const int node = (s * (SEG >= 8 ? 1 : 1) + wave * 7 + i * 3 + wg) % TILE;
        const int col = ((wg * 13 + i * 5 + wave) * SEG) % (D - SEG) + within;
        float* p = base + node * D + col;
        const float v = 1.0f + lane;
        if (MODE == 0) *p = v;
        else unsafeAtomicAdd(p, v);
    }
}

int main() {
    const int N = 64000, n_tiles = N / TILE;
    float* out;
    CK(hipMalloc(&out, (size_t)N * D * 4));
    CK(hipMemset(out, 0, (size_t)N * D * 4));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int grid = (n_tiles + N_XCD - 1) / N_XCD * N_XCD * WG_PER_TILE;
    const int iters = 24;  // 160 waves x 24 instr x 64 lanes = 3840 lane-ops per node: the last conv layer's agg size
    auto time = [&](auto launch, const char* name) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(a); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
        const double ops = (double)grid * 4 * iters * 64;
        printf("%-34s %.3f ms  %.1f G lane-ops/s  (%.2f GB of payload at %.2f TB/s)\n", name, ms, ops / ms / 1e6,
               ops * 4 / 1e9, ops * 4 / ms / 1e9);
    };
#define RUN(M, S) time([&] { k<M, S><<<grid, 256>>>(out, n_tiles, iters); }, #M " (0=store,1=atomic) seg " #S)
    RUN(0, 64); RUN(1, 64); RUN(0, 16); RUN(1, 16); RUN(0, 8); RUN(1, 8); RUN(0, 4); RUN(1, 4);
    RUN(0, 2); RUN(1, 2); RUN(0, 1); RUN(1, 1);
    return 0;
}
