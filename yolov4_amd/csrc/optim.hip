// Fused Adam step: replaces torch.optim.Adam(groups, lr, betas=(0.9, 0.999), eps=1e-8) as built
// by the reference (yolo/optim/optimizers/adam.py:14-15, build.py:18-35).  One sweep over (p, g, m, v): 16 B read
// + 12 B written per parameter, HBM-bound.  Same operation order as torch's single-tensor Adam:
//   m = lerp(m, g, 1-b1); v = b2*v + (1-b2)*g*g; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long long n,
                                                   float lr_over_bc1, float b1, float b2, float eps,
                                                   float inv_sqrt_bc2, float wd, float gscale) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float gi = g[i] * gscale;
        const float pi = p[i];
        if (wd != 0.f) gi = gi + wd * pi;                       // L2 weight decay folded into the gradient
        const float mi = m[i] + (1.0f - b1) * (gi - m[i]);      // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;     // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - lr_over_bc1 * (mi / denom);
    }
}

}  // namespace

extern "C" int y4_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n,
                                float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                float grad_scale, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq) return Y4_ERR_NULL;
    if (n <= 0 || step < 1) return Y4_ERR_SHAPE;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, y4_stream(stream), param, grad, exp_avg,
                       exp_avg_sq, n, (float)((double)lr / bc1), beta1, beta2, eps, (float)(1.0 / sqrt(bc2)),
                       weight_decay, grad_scale);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}
