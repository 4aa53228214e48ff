// Real spherical harmonics (e3nn convention: polar axis y, m=-l..l, 'component' normalisation,
// input normalised first) and the Bessel radial basis, as closed forms for l <= 4.
// Spec: SURVEY.md Appendix A.1 / A.9; checked against the oracle's CG recursion in tests.
#pragma once
#include <hip/hip_runtime.h>

namespace matten {

// y[(LMAX+1)^2]; (vx,vy,vz) raw edge vector, len = |v|.
template <int LMAX>
__device__ __forceinline__ void real_sh(float vx, float vy, float vz, float len, float* __restrict__ y) {
    // torch.nn.functional.normalize: v / max(|v|, 1e-12)
    float inv = 1.0f / fmaxf(len, 1e-12f);
    float x = vx * inv, yy = vy * inv, z = vz * inv;
    y[0] = 1.0f;
    if constexpr (LMAX >= 1) {
        const float s3 = 1.7320508075688772f;
        y[1] = s3 * x;
        y[2] = s3 * yy;
        y[3] = s3 * z;
    }
    float x2 = x * x, y2 = yy * yy, z2 = z * z;
    if constexpr (LMAX >= 2) {
        const float s15 = 3.872983346207417f;   // sqrt(5)*sqrt(3)
        const float s5 = 2.23606797749979f;     // sqrt(5)
        y[4] = s15 * x * z;
        y[5] = s15 * x * yy;
        y[6] = s5 * (y2 - 0.5f * (x2 + z2));
        y[7] = s15 * yy * z;
        y[8] = (0.5f * s15) * (z2 - x2);
    }
    if constexpr (LMAX >= 3) {
        const float s7 = 2.6457513110645907f;
        const float c0 = s7 * 0.7905694150420949f;   // sqrt(5/8)
        const float c1 = s7 * 3.872983346207417f;    // sqrt(15)
        const float c2 = s7 * 0.6123724356957945f;   // sqrt(3/8)
        y[9] = c0 * x * (3.0f * z2 - x2);
        y[10] = c1 * x * yy * z;
        y[11] = c2 * x * (4.0f * y2 - x2 - z2);
        y[12] = (0.5f * s7) * yy * (2.0f * y2 - 3.0f * x2 - 3.0f * z2);
        y[13] = c2 * z * (4.0f * y2 - x2 - z2);
        y[14] = (0.5f * c1) * yy * (z2 - x2);
        y[15] = c0 * z * (z2 - 3.0f * x2);
    }
    if constexpr (LMAX >= 4) {
        // standard real SH table evaluated at (xs, ys, zs) = (z, x, y), times 3 = sqrt(9)
        float xs = z, ys = x, zs = yy;
        float xs2 = z2, ys2 = x2, zs2 = y2;
        const float a0 = 3.0f * 2.958039891549808f;    // sqrt(35)/2
        const float a1 = 3.0f * 2.0916500663351889f;   // sqrt(35/8)
        const float a2 = 3.0f * 1.118033988749895f;    // sqrt(5)/2
        const float a3 = 3.0f * 0.7905694150420949f;   // sqrt(5/8)
        const float a6 = 3.0f * 0.5590169943749475f;   // sqrt(5)/4
        const float a8 = 3.0f * 0.739509972887452f;    // sqrt(35)/8
        y[16] = a0 * xs * ys * (xs2 - ys2);
        y[17] = a1 * ys * zs * (3.0f * xs2 - ys2);
        y[18] = a2 * xs * ys * (7.0f * zs2 - 1.0f);
        y[19] = a3 * ys * zs * (7.0f * zs2 - 3.0f);
        y[20] = (3.0f / 8.0f) * (35.0f * zs2 * zs2 - 30.0f * zs2 + 3.0f);
        y[21] = a3 * xs * zs * (7.0f * zs2 - 3.0f);
        y[22] = a6 * (xs2 - ys2) * (7.0f * zs2 - 1.0f);
        y[23] = a1 * xs * zs * (xs2 - 3.0f * ys2);
        y[24] = a8 * (xs2 * xs2 - 6.0f * xs2 * ys2 + ys2 * ys2);
    }
}

// k-th (0-based) Bessel basis function of soft_one_hot_linspace(..., basis="bessel", cutoff=True) * sqrt(nb)
__device__ __forceinline__ float bessel_basis(float len, int k, int n_basis, float r_start, float r_end) {
    float xr = len - r_start;
    float c = r_end - r_start;
    float v = sqrtf(2.0f / c) * sinf((float)(k + 1) * 3.14159265358979323846f * xr / c) / xr;
    bool in = (xr / c < 1.0f) && (0.0f < xr);
    return in ? v * sqrtf((float)n_basis) : 0.0f;
}

}  // namespace matten
