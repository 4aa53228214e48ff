// Workgroup start-up rate of a gfx950 as a function of the workgroup's resources: empty workgroups (one store per wave
// at most) of 256 threads with V vector registers per lane and L bytes of LDS, G workgroups per launch.
//   hipcc --offload-arch=gfx950 -O3 dispatch_rate.hip -o /tmp/dispatch_rate && /tmp/dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int VGPRS>
__global__ __launch_bounds__(256) void k(float* out, int n) {
    extern __shared__ float lds[];
    // claim VGPRS registers: the highest one is named in an asm clobber
    if constexpr (VGPRS > 128) asm volatile("v_mov_b32 v167, 0" ::: "v167");
    else if constexpr (VGPRS > 64) asm volatile("v_mov_b32 v119, 0" ::: "v119");
    else if constexpr (VGPRS > 32) asm volatile("v_mov_b32 v39, 0" ::: "v39");
    if (n < 0) { lds[threadIdx.x] = 1.0f; out[blockIdx.x] = lds[(threadIdx.x + 1) & 255]; }   // never: keeps LDS / out alive
}
template <int VGPRS> void run(float* out, int lds) {
    hipFuncSetAttribute((const void*)k<VGPRS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int grid : {1100, 8000, 39000}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 3; ++i) k<VGPRS><<<grid, 256, lds>>>(out, 0);
        hipEventRecord(e0);
        const int reps = 20;
        for (int i = 0; i < reps; ++i) k<VGPRS><<<grid, 256, lds>>>(out, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("  vgprs %3d lds %5d B grid %5d: %8.2f us per launch, %6.2f ns per workgroup\n", VGPRS, lds, grid,
               ms / reps * 1e3, ms / reps * 1e6 / grid);
    }
}
int main() {
    float* out; hipMalloc(&out, 1 << 20);
    for (int lds : {0, 4096, 36 * 1024, 52 * 1024}) {
        run<32>(out, lds); run<40>(out, lds); run<120>(out, lds); run<168>(out, lds);
    }
    return 0;
}
