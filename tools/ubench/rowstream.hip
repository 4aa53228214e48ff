// Micro-benchmark: streaming 16 rows per wave (species_linear's access pattern) -- how does the achieved HBM rate
// depend on how many rows one load instruction touches (R rows x 1024/R contiguous bytes each)?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) f4u { float v[4]; };

// R rows per instruction; lane l: row = l / (64/R) within the group of R rows, 16-byte piece = l % (64/R)
template <int R, int U>
__global__ __launch_bounds__(256) void k_tile(const float* __restrict__ x, int S, int n_rows, float* out) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int r0 = wave * 16;
    if (r0 >= n_rows) return;
    constexpr int LPR = 64 / R;            // lanes per row
    constexpr int CB = LPR * 4;            // contiguous floats per row per instruction
    const int lr = lane / LPR, lp = lane % LPR;
    float acc = 0.f;
    const int n_steps = S / CB;            // tail ignored
    for (int grp = 0; grp < 16 / R; ++grp) {
        const float* p = x + (size_t)(r0 + grp * R + lr) * S + lp * 4;
        for (int st = 0; st < n_steps; st += U) {
            f4u v[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (st + u < n_steps) v[u] = *reinterpret_cast<const f4u*>(p + (size_t)(st + u) * CB);
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (st + u < n_steps) acc += v[u].v[0] + v[u].v[1] + v[u].v[2] + v[u].v[3];
        }
    }
    if (acc == 12345.f) out[0] = acc;
}
// same as R=16 but the 16 rows advance together (all rows of the tile at the same column): species_linear today
template <int U>
__global__ __launch_bounds__(256) void k_tile16_lockstep(const float* __restrict__ x, int S, int n_rows, float* out) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int r0 = wave * 16;
    if (r0 >= n_rows) return;
    const int g = lane >> 4, c = lane & 15;
    const float* p = x + (size_t)(r0 + c) * S + 4 * g;
    float acc = 0.f;
    const int n_steps = S / 16;
    for (int st = 0; st < n_steps; st += U) {
        f4u v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (st + u < n_steps) v[u] = *reinterpret_cast<const f4u*>(p + (size_t)(st + u) * 16);
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (st + u < n_steps) acc += v[u].v[0] + v[u].v[1] + v[u].v[2] + v[u].v[3];
    }
    if (acc == 12345.f) out[0] = acc;
}
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ f4 buffer_load_x4(i32x4 srsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
// lockstep pattern through buffer (SRD) loads: per-lane voffset, scalar soffset, as species_linear issues them
template <int U, int WAVES_PER_SIMD>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void k_tile16_buffer(const float* __restrict__ x, int S, int n_rows, float* out) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int r0 = wave * 16;
    if (r0 >= n_rows) return;
    const int g = lane >> 4, c = lane & 15;
    const uint64_t bp = (uint64_t)(x + (size_t)r0 * S);
    const i32x4 rsrc = {(int)(uint32_t)bp, (int)((uint32_t)(bp >> 32) & 0xffffu), 16 * S * 4, 0x00020000};
    const int voff = (c * S + 4 * g) * 4;
    float acc = 0.f;
    const int n_steps = S / 16;
    for (int st = 0; st < n_steps; st += U) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = buffer_load_x4(rsrc, voff, (st + u) * 64, 0);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u][0] + v[u][1] + v[u][2] + v[u][3];
    }
    if (acc == 12345.f) out[0] = acc;
}
// species_linear's d > 1 pattern: lane (g, c) reads the 4d contiguous floats of row c at (16 st + 4g) d, as d 16-byte loads:
// one instruction touches 4 pieces per row at stride 16 d bytes
template <int D, int KS>
__global__ __launch_bounds__(256) void k_tile16_strided(const float* __restrict__ x, int S, int n_rows, float* out) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int r0 = wave * 16;
    if (r0 >= n_rows) return;
    const int g = lane >> 4, c = lane & 15;
    const uint64_t bp = (uint64_t)(x + (size_t)r0 * S);
    const i32x4 rsrc = {(int)(uint32_t)bp, (int)((uint32_t)(bp >> 32) & 0xffffu), 16 * S * 4, 0x00020000};
    const int voff = (c * S + 4 * g * D) * 4;
    float acc = 0.f;
    const int n_steps = S / (16 * D);
    for (int st = 0; st < n_steps; st += KS) {
        f4 v[KS * D];
#pragma unroll
        for (int r = 0; r < KS; ++r)
#pragma unroll
            for (int q = 0; q < D; ++q) v[r * D + q] = buffer_load_x4(rsrc, voff, ((st + r) * 16 * D + 4 * q) * 4, 0);
#pragma unroll
        for (int u = 0; u < KS * D; ++u) acc += v[u][0] + v[u][1] + v[u][2] + v[u][3];
    }
    if (acc == 12345.f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_flat(const f4* __restrict__ w, size_t n4, float* out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    for (; i < n4; i += (size_t)gridDim.x * 256) { f4 v = w[i]; acc += v[0] + v[1] + v[2] + v[3]; }
    if (acc == 12345.f) out[0] = acc;
}
int main() {
    const int N = 64000;
    float *x, *out;
    for (int S : {4170}) {
        CK(hipMalloc(&x, (size_t)N * S * 4)); CK(hipMalloc(&out, 1024));
        CK(hipMemset(x, 0, (size_t)N * S * 4));
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        const double bytes = (double)N * S * 4;
        auto time = [&](auto launch, const char* name) {
            launch(); hipDeviceSynchronize();
            hipEventRecord(a); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
            printf("S=%d %-44s %.3f ms  %.2f TB/s\n", S, name, ms, bytes / ms / 1e9);
        };
        const int grid = (N / 16 + 3) / 4;
        time([&] { k_tile16_lockstep<4><<<grid, 256>>>(x, S, N, out); }, "16 rows lockstep, 64 B/row/instr, U=4");
        time([&] { k_tile16_lockstep<10><<<grid, 256>>>(x, S, N, out); }, "16 rows lockstep, 64 B/row/instr, U=10");
        time([&] { k_tile16_lockstep<20><<<grid, 256>>>(x, S, N, out); }, "16 rows lockstep, 64 B/row/instr, U=20");
        time([&] { k_tile16_buffer<10, 8><<<grid, 256>>>(x, S, N, out); }, "16 rows lockstep, buffer loads, U=10");
        time([&] { k_tile16_buffer<10, 2><<<grid, 256>>>(x, S, N, out); }, "16 rows lockstep, buffer loads, U=10, 2 waves/SIMD");
        time([&] { k_tile16_buffer<20, 2><<<grid, 256>>>(x, S, N, out); }, "16 rows lockstep, buffer loads, U=20, 2 waves/SIMD");
        time([&] { k_tile16_strided<9, 1><<<grid, 256>>>(x, S, N, out); }, "16 rows, d=9 strided pieces, 9 loads/step");
        time([&] { k_tile16_strided<9, 2><<<grid, 256>>>(x, S, N, out); }, "16 rows, d=9 strided pieces, 18 loads/2 steps");
        time([&] { k_tile16_strided<5, 2><<<grid, 256>>>(x, S, N, out); }, "16 rows, d=5 strided pieces, 10 loads/2 steps");
        time([&] { k_tile16_strided<3, 3><<<grid, 256>>>(x, S, N, out); }, "16 rows, d=3 strided pieces, 9 loads/3 steps");
        time([&] { k_tile<8, 10><<<grid, 256>>>(x, S, N, out); }, "8 rows x 128 B per instr, U=10");
        time([&] { k_tile<4, 10><<<grid, 256>>>(x, S, N, out); }, "4 rows x 256 B per instr, U=10");
        time([&] { k_tile<2, 10><<<grid, 256>>>(x, S, N, out); }, "2 rows x 512 B per instr, U=10");
        time([&] { k_tile<1, 10><<<grid, 256>>>(x, S, N, out); }, "1 row x 1 KB per instr, U=10");
        time([&] { k_tile<1, 4><<<grid, 256>>>(x, S, N, out); }, "1 row x 1 KB per instr, U=4");
        time([&] { k_flat<<<256 * 16, 256>>>((const f4*)x, (size_t)N * S / 4, out); }, "flat float4 grid-stride");
        hipFree(x); hipFree(out);
    }
    return 0;
}
