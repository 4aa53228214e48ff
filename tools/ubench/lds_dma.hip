// global_load_lds_dwordx4 (LDS-DMA) on gfx950: does it accept a global address that is only 4-byte aligned, and
// where does lane l's data land?  (LDS address = M0 base + 16 * lane.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* __restrict__ src, float* out, int shift) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int l = threadIdx.x;
#ifdef USE_BUILTIN
    __builtin_amdgcn_global_load_lds(src + shift + 4 * ((l * 7) % 64), lds, 16, 0, 0);   // clang (ROCm 7.2) leaves M0 unset here
#else
    const float* gp = src + shift + 4 * ((l * 7) % 64);
    const unsigned lds_off = (unsigned)(size_t)lds;   // byte offset of the destination tile in LDS
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_off), "v"(gp) : "memory", "m0");
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const f4 v = *reinterpret_cast<f4*>(lds + 4 * l);
    out[4 * l] = v[0]; out[4 * l + 1] = v[1]; out[4 * l + 2] = v[2]; out[4 * l + 3] = v[3];
}
int main() {
    float h[512], o[256];
    for (int i = 0; i < 512; ++i) h[i] = (float)i;
    float *d, *dout; hipMalloc(&d, sizeof h); hipMalloc(&dout, sizeof o);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    for (int shift = 0; shift < 4; ++shift) {
        k<<<1, 64, 4096>>>(d, dout, shift);
        hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) bad += o[4 * l + e] != (float)(shift + 4 * ((l * 7) % 64) + e);
        if (shift == 0) {   // where did lane l's four floats (values 4*((7l)%64) + e) land?  dump = LDS floats 0..255
            for (int l = 0; l < 3; ++l) for (int e = 0; e < 4; ++e) {
                float want = (float)(4 * ((l * 7) % 64) + e); int pos = -1;
                for (int i = 0; i < 256; ++i) if (o[i] == want) pos = i;
                printf("lane %d elem %d (value %g) found at LDS float %d\n", l, e, want, pos);
            }
        }
        printf("source offset %d floats (%2d B alignment): %d of 256 wrong; lane 1 got %g %g %g %g\n", shift, (shift * 4) % 16 ? 4 * (shift & -shift) : 16, bad, o[4], o[5], o[6], o[7]);
    }
    return 0;
}
