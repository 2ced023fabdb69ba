// Fused optimizer steps: replace torch.optim.Adam(groups, lr, betas=(0.9, 0.999), eps=1e-8) and
// torch.optim.SGD(groups, lr, momentum, weight_decay) as built by the reference
// (yolo/optim/optimizers/adam.py:14-15, sgd.py:14-15, build.py:18-35).
//
// Adam: one sweep over (p, g, m, v): 16 B read + 12 B written per parameter, HBM-bound.  Same operation order as
// torch's single-tensor Adam:
//   m = lerp(m, g, 1-b1); v = b2*v + (1-b2)*g*g; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// SGD: buf = g (first step) | momentum*buf + g; p -= lr*buf, with g += wd*p first (dampening 0, no nesterov).
//
// Multi-tensor form: the host keeps a device table of chunks {p, g, m, v, n, hyper row}; ONE launch walks all
// parameters of the model (327 tensors, 64.9 M elements: ~1000 chunks of 64 Ki elements), each block taking whole
// chunks with 16-B accesses.  Per-group scalars (lr/bias corrections/weight decay) come in a small by-value table.
#include "common.h"

namespace {

struct Hyper { float a, b, c, d; };          // Adam: lr/bc1, 1/sqrt(bc2), wd, -   SGD: lr, momentum, wd, first(0/1)
constexpr int MAX_HYPER = 16;
struct HyperTable { Hyper h[MAX_HYPER]; };

struct Chunk {                                // 48 bytes, mirrored by the host as 6 x int64
    float* p; const float* g; float* m; float* v; long long n; long long hyper;
};

__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, float lr_over_bc1, float b1, float b2,
                                          float eps, float inv_sqrt_bc2, float wd, float gscale) {
    float gi = g * gscale;
    if (wd != 0.f) gi = gi + wd * p;                          // L2 weight decay folded into the gradient
    const float mi = m + (1.0f - b1) * (gi - m);              // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = b2 * v + (1.0f - b2) * gi * gi;          // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    m = mi;
    v = vi;
    p = p - lr_over_bc1 * (mi / denom);
}

__device__ __forceinline__ void sgd_elem(float& p, float g, float& buf, float lr, float mom, float wd, bool first,
                                         float gscale) {
    float gi = g * gscale;
    if (wd != 0.f) gi = gi + wd * p;
    if (mom != 0.f) {
        const float b = first ? gi : mom * buf + gi;
        buf = b;
        gi = b;
    }
    p = p - lr * gi;
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long long n,
                                                   float lr_over_bc1, float b1, float b2, float eps,
                                                   float inv_sqrt_bc2, float wd, float gscale) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float pi = p[i], mi = m[i], vi = v[i];
        adam_elem(pi, g[i], mi, vi, lr_over_bc1, b1, b2, eps, inv_sqrt_bc2, wd, gscale);
        m[i] = mi; v[i] = vi; p[i] = pi;
    }
}

template <bool ADAM>
__global__ __launch_bounds__(256) void multi_tensor_kernel(const Chunk* __restrict__ chunks, int nchunks, const HyperTable ht,
                                                           float b1, float b2, float eps, float gscale) {
    for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const Chunk ck = chunks[c];
        const Hyper h = ht.h[ck.hyper];
        const bool vec = ((reinterpret_cast<uintptr_t>(ck.p) | reinterpret_cast<uintptr_t>(ck.g) |
                           reinterpret_cast<uintptr_t>(ck.m) | (ADAM ? reinterpret_cast<uintptr_t>(ck.v) : 0)) & 15) == 0;
        long long i0 = 0;
        if (vec) {
            const long long n4 = ck.n >> 2;
            f32x4* p4 = reinterpret_cast<f32x4*>(ck.p);
            const f32x4* g4 = reinterpret_cast<const f32x4*>(ck.g);
            f32x4* m4 = reinterpret_cast<f32x4*>(ck.m);
            f32x4* v4 = reinterpret_cast<f32x4*>(ck.v);
            for (long long i = threadIdx.x; i < n4; i += 256) {
                f32x4 pv = p4[i], mv = m4[i], vv;
                const f32x4 gv = g4[i];
                if (ADAM) vv = v4[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float pe = pv[e], me = mv[e], ve = ADAM ? vv[e] : 0.f;
                    if (ADAM) adam_elem(pe, gv[e], me, ve, h.a, b1, b2, eps, h.b, h.c, gscale);
                    else sgd_elem(pe, gv[e], me, h.a, h.b, h.c, h.d != 0.f, gscale);
                    pv[e] = pe; mv[e] = me;
                    if (ADAM) vv[e] = ve;
                }
                p4[i] = pv;
                if (ADAM || h.b != 0.f) m4[i] = mv;
                if (ADAM) v4[i] = vv;
            }
            i0 = n4 << 2;
        }
        for (long long i = i0 + threadIdx.x; i < ck.n; i += 256) {
            float pi = ck.p[i], mi = ck.m[i], vi = ADAM ? ck.v[i] : 0.f;
            if (ADAM) adam_elem(pi, ck.g[i], mi, vi, h.a, b1, b2, eps, h.b, h.c, gscale);
            else sgd_elem(pi, ck.g[i], mi, h.a, h.b, h.c, h.d != 0.f, gscale);
            ck.p[i] = pi;
            if (ADAM || h.b != 0.f) ck.m[i] = mi;
            if (ADAM) ck.v[i] = vi;
        }
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

extern "C" int y4_adam_hyper_f32(float lr, float beta1, float beta2, float weight_decay, int step, float* out4_host) {
    if (!out4_host) return Y4_ERR_NULL;
    if (step < 1) return Y4_ERR_SHAPE;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    out4_host[0] = (float)((double)lr / bc1);
    out4_host[1] = (float)(1.0 / sqrt(bc2));
    out4_host[2] = weight_decay;
    out4_host[3] = 0.f;
    return Y4_OK;
}

static int multi_launch(bool adam, const void* chunks_dev, int nchunks, const float* hyper_host, int nhyper,
                        float b1, float b2, float eps, float grad_scale, void* stream) {
    if (!chunks_dev || !hyper_host) return Y4_ERR_NULL;
    if (nchunks <= 0 || nhyper <= 0 || nhyper > MAX_HYPER) return Y4_ERR_SHAPE;
    if (reinterpret_cast<uintptr_t>(chunks_dev) & 7) return Y4_ERR_SHAPE;
    HyperTable ht{};
    for (int i = 0; i < nhyper; ++i)
        ht.h[i] = Hyper{hyper_host[4 * i], hyper_host[4 * i + 1], hyper_host[4 * i + 2], hyper_host[4 * i + 3]};
    const int grid = nchunks < 2048 ? nchunks : 2048;
    const Chunk* ck = static_cast<const Chunk*>(chunks_dev);
    if (adam)
        hipLaunchKernelGGL(multi_tensor_kernel<true>, dim3(grid), dim3(256), 0, y4_stream(stream), ck, nchunks, ht, b1, b2,
                           eps, grad_scale);
    else
        hipLaunchKernelGGL(multi_tensor_kernel<false>, dim3(grid), dim3(256), 0, y4_stream(stream), ck, nchunks, ht, b1, b2,
                           eps, grad_scale);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

extern "C" int y4_adam_multi_step_f32(const void* chunks_dev, int nchunks, const float* hyper_host, int nhyper,
                                      float beta1, float beta2, float eps, float grad_scale, void* stream) {
    return multi_launch(true, chunks_dev, nchunks, hyper_host, nhyper, beta1, beta2, eps, grad_scale, stream);
}

extern "C" int y4_sgd_multi_step_f32(const void* chunks_dev, int nchunks, const float* hyper_host, int nhyper,
                                     float grad_scale, void* stream) {
    return multi_launch(false, chunks_dev, nchunks, hyper_host, nhyper, 0.f, 0.f, 0.f, grad_scale, stream);
}
