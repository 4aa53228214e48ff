// Calibration of rocprofv3's WRITE_SIZE / FETCH_SIZE on gfx950 for the access shapes of the conv layer's neighbour-sum row
// (agg[N, ld], component-major: tp_fused writes it in channel pieces, agg_linear streams it), against KNOWN byte counts.
//   hipcc --offload-arch=gfx950 -O3 store_shapes.hip -o store_shapes
//   rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out_w -o p -- ./store_shapes
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out_f -o p -- ./store_shapes     (tools/store_calib.sh)
// Every kernel below moves exactly BYTES bytes of distinct addresses (printed per kernel); the counter / BYTES ratio is the
// calibration factor for that shape.
//
// Store shapes (what a wave's store instruction looks like in tp_fused's epilogue, tp_fused.hip StoreAgg): a lane owns channel
// u of node j and stores its accumulator k at row[j] + region + k * kstep + u: the `cu` channel lanes of a node are
// contiguous (a PIECE of 4 cu bytes), the 64 / cu nodes of the wave are `ld` floats apart.
//   st_piece<CU>: pieces of 4 * CU bytes (CU = 1, 2, 4, 8, 16), 9 components kstep = 32 floats apart, rows of ld floats,
//                 regions laid side by side until the row is full: every byte of the row written exactly once per node
//   st_line:      64 lanes x 16 bytes = 8 whole 128-byte lines per instruction (the coalesced reference)
// Fetch shapes:
//   ld_stream:    16 bytes per lane, 1 KB contiguous per instruction (the documented x2 case)
//   ld_rows:      agg_linear's operand fetch: lane (g, c) reads the 16-byte piece g of row c's current 64-byte chunk,
//                 16 rows (ld floats apart) x 64 bytes per instruction, consecutive instructions 64 bytes further
//   ld_gather4:   tp_fused's neighbour gather: 4 bytes per lane, 8 lanes contiguous (32 bytes) from 8 random rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int LD = 4352;          // floats per row (the last conv layer's component-major row)
constexpr int N = 65536;          // rows: 1.14 GB, well past the 256 MB memory-side cache
constexpr int KSTEP = 32, NCOMP = 9;

// FAR = false: one wave writes every slot of its nodes (the pieces of a 128-byte line follow each other closely);
// FAR = true: blockIdx.y picks the slot -- the pieces of a line come from workgroups that run far apart in time (all node
// groups of slot 0 first, then slot 1, ...): the two ends between which tp_fused's tile-major entry order lies
template <int CU, bool FAR>
__global__ __launch_bounds__(256) void st_piece(float* __restrict__ out, int regions) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int npw = 64 / CU;
    const int64_t node = ((int64_t)blockIdx.x * 4 + wave) * npw + lane / CU;
    const int u = lane % CU;
    if (node >= N) return;
    float* row = out + node * LD;
    // a region = NCOMP components x KSTEP channel slots; an entry covers slots [s0, s0 + CU) of every region
    for (int r = 0; r < regions; ++r)
        for (int s0 = FAR ? blockIdx.y * CU : 0; s0 < (FAR ? blockIdx.y * CU + CU : KSTEP); s0 += CU)
#pragma unroll
            for (int k = 0; k < NCOMP; ++k) row[r * (NCOMP * KSTEP) + k * KSTEP + s0 + u] = (float)(k + u);
}

__global__ __launch_bounds__(256) void st_line(f32x4* __restrict__ out, int64_t n16) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = f32x4{1.f, 2.f, 3.f, 4.f};
}

__global__ __launch_bounds__(256) void ld_stream(const f32x4* __restrict__ in, int64_t n16, float* __restrict__ sink) {
    f32x4 a = {0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) a += in[i];
    if (a[0] + a[1] + a[2] + a[3] == 12345.f) sink[0] = 1.f;
}

__global__ __launch_bounds__(256) void ld_rows(const float* __restrict__ in, int used, float* __restrict__ sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
    const int64_t row = ((int64_t)blockIdx.x * 4 + wave) * 16 + c;
    if (row >= N) return;
    const f32x4* p = reinterpret_cast<const f32x4*>(in + row * LD) + g;
    f32x4 a = {0, 0, 0, 0};
    for (int ch = 0; ch < used / 16; ++ch) a += p[ch * 4];
    if (a[0] + a[1] + a[2] + a[3] == 12345.f) sink[0] = 1.f;
}

__global__ __launch_bounds__(256) void ld_gather4(const float* __restrict__ in, const int* __restrict__ rows, int reps,
                                                  float* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    float a = 0.f;
    for (int r = 0; r < reps; ++r) {
        const int row = rows[(w * reps + r) * 8 + (lane >> 3)];
        a += in[(int64_t)row * 256 + (r % 32) * 8 + (lane & 7)];
    }
    if (a == 12345.f) sink[0] = 1.f;
}

int main() {
    float *buf, *sink; int* rows;
    const size_t bytes = (size_t)N * LD * 4;
    (void)hipMalloc(&buf, bytes); (void)hipMalloc(&sink, 64);
    (void)hipMemset(buf, 0, bytes);
    const int regions = LD / (NCOMP * KSTEP);      // 15 regions of 288 floats = 4320 of the 4352 floats of a row
    const double st_bytes = (double)N * regions * NCOMP * KSTEP * 4;
    printf("st_piece<CU>: %.0f bytes each (N = %d rows x %d floats)\n", st_bytes, N, regions * NCOMP * KSTEP);
#define RUN_ST(CU) st_piece<CU, false><<<(N / (64 / CU) + 3) / 4, 256>>>(buf, regions); (void)hipDeviceSynchronize(); \
    st_piece<CU, true><<<dim3((N / (64 / CU) + 3) / 4, KSTEP / CU), 256>>>(buf, regions); (void)hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) { RUN_ST(1) RUN_ST(2) RUN_ST(4) RUN_ST(8) RUN_ST(16) }
    printf("st_line: %.0f bytes\n", (double)bytes);
    for (int rep = 0; rep < 2; ++rep) { st_line<<<4096, 256>>>((f32x4*)buf, bytes / 16); (void)hipDeviceSynchronize(); }
    printf("ld_stream: %.0f bytes\n", (double)bytes);
    for (int rep = 0; rep < 2; ++rep) { ld_stream<<<4096, 256>>>((const f32x4*)buf, bytes / 16, sink); (void)hipDeviceSynchronize(); }
    const int used = 4320;
    printf("ld_rows: %.0f bytes\n", (double)N * used * 4);
    for (int rep = 0; rep < 2; ++rep) { ld_rows<<<N / 64, 256>>>(buf, used, sink); (void)hipDeviceSynchronize(); }
    // gather: rows of 1 KB (256 floats) of a 64 MB table (64 k rows: L2-resident per XCD only in part), 32-byte pieces
    const int waves = 65536, reps = 64;
    std::vector<int> h((size_t)waves * reps * 8);
    uint32_t s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (int)((s >> 8) % 65536u); }
    (void)hipMalloc(&rows, h.size() * 4);
    (void)hipMemcpy(rows, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    printf("ld_gather4: %.0f bytes requested in 32-byte pieces (table 64 MB)\n", (double)waves * reps * 64 * 4);
    for (int rep = 0; rep < 2; ++rep) { ld_gather4<<<waves / 4, 256>>>(buf, rows, reps, sink); (void)hipDeviceSynchronize(); }
    return 0;
}
