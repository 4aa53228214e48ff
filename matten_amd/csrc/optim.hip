// Adam over ONE flat fp32 parameter buffer (the reference configures torch.optim.Adam,
// scripts/configs/materials_tensor.yaml:103-107: lr 0.01, weight_decay 1e-5 = L2 term added to the gradient).
//
// The model's ~60 parameter tensors are views into one buffer (matten_amd/optim.py), so are their gradients: a step is
// one elementwise launch over 3.5 M floats (56 MB of traffic: p, g read, m, v read and written, p written), and
// zeroing the gradients one memset.  The step count lives on the device (hipGraph capture: nothing on the host changes
// between replays); the bias corrections are computed per thread from it.
//   g' = g + wd p;  m = b1 m + (1 - b1) g';  v = b2 v + (1 - b2) g'^2
//   p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)          (torch.optim.Adam, amsgrad = False)
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, const float* __restrict__ step, float lr,
                                                   float b1, float b2, float eps, float wd) {
    const float t = step[0];                       // already incremented by the caller for this step
    const float c1 = 1.0f - powf(b1, t), c2s = sqrtf(1.0f - powf(b2, t));
    const float step_size = lr / c1;
    const int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 >= n) return;
    if (i4 + 4 <= n) {
        f32x4 pp = *reinterpret_cast<f32x4*>(p + i4), mm = *reinterpret_cast<f32x4*>(m + i4),
              vv = *reinterpret_cast<f32x4*>(v + i4);
        const f32x4 gg = *reinterpret_cast<const f32x4*>(g + i4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = gg[k] + wd * pp[k];
            mm[k] = b1 * mm[k] + (1.0f - b1) * gk;
            vv[k] = b2 * vv[k] + (1.0f - b2) * gk * gk;
            pp[k] -= step_size * mm[k] / (sqrtf(vv[k]) / c2s + eps);
        }
        *reinterpret_cast<f32x4*>(p + i4) = pp;
        *reinterpret_cast<f32x4*>(m + i4) = mm;
        *reinterpret_cast<f32x4*>(v + i4) = vv;
    } else {
        for (int64_t i = i4; i < n; ++i) {
            const float gk = g[i] + wd * p[i];
            m[i] = b1 * m[i] + (1.0f - b1) * gk;
            v[i] = b2 * v[i] + (1.0f - b2) * gk * gk;
            p[i] -= step_size * m[i] / (sqrtf(v[i]) / c2s + eps);
        }
    }
}

}  // namespace

extern "C" int matten_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                const float* step, float lr, float beta1, float beta2, float eps, float weight_decay,
                                matten_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0 || !(lr >= 0.0f) || !(beta1 >= 0.0f && beta1 < 1.0f) || !(beta2 >= 0.0f && beta2 < 1.0f) || !(eps >= 0.0f))
        return MATTEN_EINVAL;
    if (n == 0) return MATTEN_OK;
    if (!params || !grads || !exp_avg || !exp_avg_sq || !step) return MATTEN_EINVAL;
    if ((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grads) | reinterpret_cast<uintptr_t>(exp_avg) |
         reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15)
        return MATTEN_EINVAL;
    const int64_t blocks = matten_cdiv(matten_cdiv(n, 4), 256);
    if (blocks >= ((int64_t)1 << 31)) return MATTEN_EINVAL;
    adam_kernel<<<(unsigned)blocks, 256, 0, stream>>>(params, grads, exp_avg, exp_avg_sq, n, step, lr, beta1, beta2, eps,
                                                      weight_decay);
    MATTEN_LAUNCH_CHECK();
    return MATTEN_OK;
}
