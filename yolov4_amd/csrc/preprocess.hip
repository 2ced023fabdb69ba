// Eval input pipeline: resize-to-SxS + BGR->RGB + /255 + HWC->CHW in one pass.
// Restates the arithmetic of cv2.resize(..., INTER_LINEAR) on 8-bit images (OpenCV imgproc resize.cpp:
// 11-bit fixed-point coefficients, horizontal pass in int32, vertical pass ((b*(S>>4))>>16, +2, >>2), and the
// exact-2x-downscale special case that OpenCV routes to the 2x2 box average), which is what the reference
// calls at yolo/data/transform.py:173-174; channel flip :437; /255 and permute(2,0,1) :461.
// HBM-bound byte work: one thread per output pixel, 3 channels, source rows stay in L2.
#include "common.h"

namespace {

__device__ __forceinline__ int sat_short(float v) {
    int r = __float2int_rn(v);                      // cvRound: round half to even
    return r < -32768 ? -32768 : (r > 32767 ? 32767 : r);
}

__global__ void __launch_bounds__(256)
preprocess_kernel(const unsigned char* __restrict__ src, int sh, int sw, long long pitch, int swap_rb,
                  float* __restrict__ dst, long long dsc, long long dsh, long long dsw, int S,
                  double scale_x, double scale_y, int area2x) {
    const int dx = blockIdx.x * 256 + threadIdx.x;
    const int dy = blockIdx.y;
    if (dx >= S) return;
    int v[3];
    if (area2x) {
        const unsigned char* r0 = src + (long long)(2 * dy) * pitch + (long long)(2 * dx) * 3;
        const unsigned char* r1 = r0 + pitch;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = (r0[c] + r0[3 + c] + r1[c] + r1[3 + c] + 2) >> 2;
    } else {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= (float)sx;
        if (sx < 0) { fx = 0.f; sx = 0; }
        if (sx >= sw - 1) { fx = 0.f; sx = sw - 1; }
        const int a0 = sat_short((1.f - fx) * 2048.f), a1 = sat_short(fx * 2048.f);
        const int sx1 = sx + 1 < sw ? sx + 1 : sw - 1;
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= (float)sy;
        const int b0 = sat_short((1.f - fy) * 2048.f), b1 = sat_short(fy * 2048.f);
        const int y0 = sy < 0 ? 0 : (sy > sh - 1 ? sh - 1 : sy);
        const int y1 = sy + 1 < 0 ? 0 : (sy + 1 > sh - 1 ? sh - 1 : sy + 1);
        const unsigned char* r0 = src + (long long)y0 * pitch;
        const unsigned char* r1 = src + (long long)y1 * pitch;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int h0 = r0[sx * 3 + c] * a0 + r0[sx1 * 3 + c] * a1;
            const int h1 = r1[sx * 3 + c] * a0 + r1[sx1 * 3 + c] * a1;
            int r = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
            v[c] = r < 0 ? 0 : (r > 255 ? 255 : r);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int co = swap_rb ? 2 - c : c;
        dst[co * dsc + (long long)dy * dsh + (long long)dx * dsw] = __fdiv_rn((float)v[c], 255.f);
    }
}

}  // namespace

extern "C" int y4_preprocess_u8_f32(const void* src, int src_h, int src_w, long long src_pitch_bytes, int swap_rb,
                                    float* dst, long long dst_sc, long long dst_sh, long long dst_sw, int S,
                                    void* stream) {
    if (!src || !dst) return Y4_ERR_NULL;
    if (src_h < 1 || src_w < 1 || S < 1 || S > 65535 || src_pitch_bytes < (long long)src_w * 3) return Y4_ERR_SHAPE;
    // OpenCV: inv_scale = dsize/ssize (double); scale = 1./inv_scale
    const double scale_x = 1.0 / ((double)S / (double)src_w), scale_y = 1.0 / ((double)S / (double)src_h);
    const int area2x = (src_w == 2 * S && src_h == 2 * S) ? 1 : 0;
    dim3 grid((S + 255) / 256, S);
    hipLaunchKernelGGL(preprocess_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned char*)src, src_h,
                       src_w, src_pitch_bytes, swap_rb, dst, dst_sc, dst_sh, dst_sw, S, scale_x, scale_y, area2x);
    return hipGetLastError() == hipSuccess ? Y4_OK : Y4_ERR_LAUNCH;
}
