// Direct (one thread per output element) convolution kernels for the shapes the implicit-GEMM kernels do not take:
// Cin not a multiple of 32 (other than the 3-channel stem), kernel sizes other than 1 / 3, strides other than 1 / 2.
// ConvBNAct (darknet/darknet.py:25-36) accepts any such shape; YOLOv4 itself never builds one, so these kernels are
// written for correctness -- plain fp32 fma chains in k order, NHWC with pitch -- not for speed.  Also here: the input
// gradient of the stem conv (Cin = 3), which training never needs (the network input requires no gradient) but which the
// reference's autograd would deliver on request.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void generic_fwd_kernel(const float* __restrict__ x, long long ldx, const float* __restrict__ w,
                                                          float* __restrict__ y, long long ldy, int B, int H, int W, int Cin, int Cout,
                                                          int k, int stride, int pad, int Ho, int Wo, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int act, const float* __restrict__ res,
                                                          long long ldr) {
    const long long total = (long long)B * Ho * Wo * Cout;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i % Cout);
        const long long m = i / Cout;
        const int wo = (int)(m % Wo), ho = (int)((m / Wo) % Ho);
        const long long b = m / ((long long)Wo * Ho);
        float acc = 0.f;
        for (int r = 0; r < k; ++r) {
            const int hi = ho * stride - pad + r;
            if ((unsigned)hi >= (unsigned)H) continue;
            for (int q = 0; q < k; ++q) {
                const int wi = wo * stride - pad + q;
                if ((unsigned)wi >= (unsigned)W) continue;
                const float* xp = x + ((b * H + hi) * W + wi) * ldx;
                const float* wp = w + (((long long)n * k + r) * k + q) * Cin;
                for (int c = 0; c < Cin; ++c) acc = fmaf(xp[c], wp[c], acc);
            }
        }
        float v = acc * (scale ? scale[n] : 1.0f) + (shift ? shift[n] : 0.0f);
        v = y4_act(v, act);
        if (res) v += res[m * ldr + n];
        y[m * ldy + n] = v;
    }
}

// dx[b,h,w,c] = sum_{r,q,n} dy[b,(h+pad-r)/s,(w+pad-q)/s,n] w[n,r,q,c]   (only where the division is exact)
__global__ __launch_bounds__(256) void generic_dgrad_kernel(const float* __restrict__ dy, long long lddy, const float* __restrict__ w,
                                                            float* __restrict__ dx, long long lddx, int B, int H, int W, int Cin,
                                                            int Cout, int k, int stride, int pad, int Ho, int Wo,
                                                            const float* __restrict__ res, long long ldr) {
    const long long total = (long long)B * H * W * Cin;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cin);
        const long long p = i / Cin;
        const int wi = (int)(p % W), hi = (int)((p / W) % H);
        const long long b = p / ((long long)W * H);
        float acc = 0.f;
        for (int r = 0; r < k; ++r) {
            const int th = hi + pad - r;
            if (th < 0 || th % stride) continue;
            const int ho = th / stride;
            if (ho >= Ho) continue;
            for (int q = 0; q < k; ++q) {
                const int tw = wi + pad - q;
                if (tw < 0 || tw % stride) continue;
                const int wo = tw / stride;
                if (wo >= Wo) continue;
                const float* dp = dy + ((b * Ho + ho) * Wo + wo) * lddy;
                for (int n = 0; n < Cout; ++n) acc = fmaf(dp[n], w[(((long long)n * k + r) * k + q) * Cin + c], acc);
            }
        }
        if (res) acc += res[p * ldr + c];
        dx[p * lddx + c] = acc;
    }
}

// dw[n,r,q,c] = sum_{b,ho,wo} dy[b,ho,wo,n] x[b,ho*s-pad+r,wo*s-pad+q,c]: one block per filter element, fixed-order
// block reduction (deterministic)
__global__ __launch_bounds__(256) void generic_wgrad_kernel(const float* __restrict__ x, long long ldx, const float* __restrict__ dy,
                                                            long long lddy, float* __restrict__ dw, int B, int H, int W, int Cin,
                                                            int Cout, int k, int stride, int pad, int Ho, int Wo) {
    __shared__ float red[256];
    const long long e = blockIdx.x;                        // ((n k + r) k + q) Cin + c
    const int c = (int)(e % Cin);
    const int q = (int)((e / Cin) % k), r = (int)((e / ((long long)Cin * k)) % k);
    const int n = (int)(e / ((long long)Cin * k * k));
    const long long M = (long long)B * Ho * Wo;
    float acc = 0.f;
    for (long long m = threadIdx.x; m < M; m += 256) {
        const int wo = (int)(m % Wo), ho = (int)((m / Wo) % Ho);
        const long long b = m / ((long long)Wo * Ho);
        const int hi = ho * stride - pad + r, wi = wo * stride - pad + q;
        if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W)
            acc = fmaf(dy[m * lddy + n], x[((b * H + hi) * W + wi) * ldx + c], acc);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) dw[e] = red[0];
}

// stem: dx[b,c,h,w] (any element strides) = sum_{r,q,n} dy[b,h+1-r,w+1-q,n] w[n,r,q,c],  c < 3
__global__ __launch_bounds__(256) void stem_dgrad_kernel(const float* __restrict__ dy, long long lddy, const float* __restrict__ w,
                                                         float* __restrict__ dx, long long sxb, long long sxc, long long sxh, long long sxw,
                                                         int B, int H, int W, int Cout) {
    __shared__ float ws[32 * 27];
    for (int i = threadIdx.x; i < Cout * 27; i += 256) ws[i] = w[i];
    __syncthreads();
    const long long total = (long long)B * H * W;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < total; p += (long long)gridDim.x * blockDim.x) {
        const int wi = (int)(p % W), hi = (int)((p / W) % H);
        const long long b = p / ((long long)W * H);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int r = 0; r < 3; ++r) {
            const int ho = hi + 1 - r;
            if ((unsigned)ho >= (unsigned)H) continue;
            for (int q = 0; q < 3; ++q) {
                const int wo = wi + 1 - q;
                if ((unsigned)wo >= (unsigned)W) continue;
                const float* dp = dy + ((b * H + ho) * W + wo) * lddy;
                for (int n = 0; n < Cout; ++n) {
                    const float g = dp[n];
                    const float* wp = ws + n * 27 + (r * 3 + q) * 3;
                    a0 = fmaf(g, wp[0], a0); a1 = fmaf(g, wp[1], a1); a2 = fmaf(g, wp[2], a2);
                }
            }
        }
        float* o = dx + b * sxb + hi * sxh + wi * sxw;
        o[0] = a0; o[sxc] = a1; o[2 * sxc] = a2;
    }
}

inline int grid_of(long long total) { long long b = (total + 255) / 256; return (int)(b < 1 ? 1 : (b > 65535 ? 65535 : b)); }

}  // namespace

extern "C" {

int y4_conv2d_generic_fwd_f32(const float* x, int ldx, const float* w, float* y, int ldy,
                              int B, int H, int W, int Cin, int Cout, int k, int stride,
                              const float* scale, const float* shift, int act, const float* residual, int ldr, void* stream) {
    if (!x || !w || !y) return Y4_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || k <= 0 || !(k & 1) || stride <= 0 || ldx < Cin || ldy < Cout ||
        (residual && ldr < Cout)) return Y4_ERR_SHAPE;
    const int pad = (k - 1) / 2;
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return Y4_ERR_SHAPE;
    hipLaunchKernelGGL(generic_fwd_kernel, dim3(grid_of((long long)B * Ho * Wo * Cout)), dim3(256), 0, y4_stream(stream), x,
                       (long long)ldx, w, y, (long long)ldy, B, H, W, Cin, Cout, k, stride, pad, Ho, Wo, scale, shift, act, residual,
                       (long long)ldr);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

int y4_conv2d_generic_dgrad_f32(const float* dy, int lddy, const float* w, float* dx, int lddx,
                                int B, int H, int W, int Cin, int Cout, int k, int stride,
                                const float* residual, int ldr, void* stream) {
    if (!dy || !w || !dx) return Y4_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || k <= 0 || !(k & 1) || stride <= 0 || lddy < Cout || lddx < Cin ||
        (residual && ldr < Cin)) return Y4_ERR_SHAPE;
    const int pad = (k - 1) / 2;
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    hipLaunchKernelGGL(generic_dgrad_kernel, dim3(grid_of((long long)B * H * W * Cin)), dim3(256), 0, y4_stream(stream), dy,
                       (long long)lddy, w, dx, (long long)lddx, B, H, W, Cin, Cout, k, stride, pad, Ho, Wo, residual, (long long)ldr);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

int y4_conv2d_generic_wgrad_f32(const float* x, int ldx, const float* dy, int lddy, float* dw,
                                int B, int H, int W, int Cin, int Cout, int k, int stride, void* stream) {
    if (!x || !dy || !dw) return Y4_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || k <= 0 || !(k & 1) || stride <= 0 || ldx < Cin || lddy < Cout)
        return Y4_ERR_SHAPE;
    const long long ne = (long long)Cout * k * k * Cin;
    if (ne >= (1ll << 31)) return Y4_ERR_SHAPE;
    const int pad = (k - 1) / 2;
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    hipLaunchKernelGGL(generic_wgrad_kernel, dim3((unsigned)ne), dim3(256), 0, y4_stream(stream), x, (long long)ldx, dy,
                       (long long)lddy, dw, B, H, W, Cin, Cout, k, stride, pad, Ho, Wo);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

int y4_conv2d_stem_dgrad_f32(const float* dy, int lddy, const float* w, float* dx, long long sxb, long long sxc, long long sxh,
                             long long sxw, int B, int H, int W, int Cout, void* stream) {
    if (!dy || !w || !dx) return Y4_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cout > 32 || lddy < Cout) return Y4_ERR_SHAPE;
    hipLaunchKernelGGL(stem_dgrad_kernel, dim3(grid_of((long long)B * H * W)), dim3(256), 0, y4_stream(stream), dy, (long long)lddy, w,
                       dx, sxb, sxc, sxh, sxw, B, H, W, Cout);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

}  // extern "C"
