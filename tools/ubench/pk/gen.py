# Generates the packed-vs-scalar CG micro-benchmark: python gen.py <l1> <l2_lo> <l2_hi> > gXY.hip (v_pk_fma_f32 gives no speed-up on MI355X)
import math, sys
sys.path.insert(0, "/root/repo")
from matten_amd.o3 import wigner_3j
LMAX = 4
def lit(c): return f"{c:.9e}f"
def emit(l1, l2, l3, out):
    C = wigner_3j(l1, l2, l3) * math.sqrt(2 * l3 + 1)
    d1, d2, d3 = 2*l1+1, 2*l2+1, 2*l3+1
    nz = [(i, j, k, float(C[i, j, k])) for i in range(d1) for j in range(d2) for k in range(d3) if abs(C[i, j, k]) > 1e-12]
    pairs = sorted({(i, j) for i, j, _, _ in nz})
    out.append(f"template <class T> __device__ __forceinline__ void cg_{l1}{l2}{l3}(const T* xw, const float* y, T* acc) {{")
    for (i, j) in pairs:
        terms = [(k, c) for (ii, jj, k, c) in nz if ii == i and jj == j]
        out.append(f"    {{ const T p = xw[{i}] * y[{j}];")
        for k, c in terms:
            out.append(f"      acc[{k}] += {lit(c)} * p;")
        out.append("    }")
    out.append("}")
    return len(pairs) + len(nz)
l1, lo, hi = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
combos = [(l2, l3) for l2 in range(lo, hi + 1) for l3 in range(abs(l1 - l2), min(LMAX, l1 + l2) + 1)]
out = ["#include <hip/hip_runtime.h>", "#include <cstdio>", "typedef float f2 __attribute__((ext_vector_type(2)));"]
ops = 0
for (l2, l3) in combos: ops += emit(l1, l2, l3, out)
d1 = 2*l1+1; y0 = lo*lo; ny = (hi+1)**2 - y0
offs=[]; o=0
for (_, l3) in combos: offs.append(o); o += 2*l3+1
out.append(f"constexpr int NC={len(combos)}, NACC={o}, D1={d1}, NY={ny}, OPS={ops + len(combos)*d1};")
out.append("template <class T> __device__ __forceinline__ void apply(const T* x, const float* y, const T* w, T* acc) {")
for c, ((l2, l3), off) in enumerate(zip(combos, offs)):
    out.append(f"    {{ T xw[{d1}]; for (int i = 0; i < {d1}; ++i) xw[i] = w[{c}] * x[i]; cg_{l1}{l2}{l3}<T>(xw, y + {l2*l2 - y0}, acc + {off}); }}")
out.append("}")
out.append(r'''
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
}''')
print("\n".join(out))
