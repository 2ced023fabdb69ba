// Shared device helpers for libyolov4_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/yolov4_amd.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define Y4_CHECK_LAUNCH()                                   \
    do {                                                    \
        if (hipGetLastError() != hipSuccess) return Y4_ERR_LAUNCH; \
    } while (0)

static inline hipStream_t y4_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: one flag word per launcher
// instantiation, bit d = "set on device d" (devices >= 64: set on every launch).  Two threads racing through their first
// launch both set the attribute, which is idempotent.
#include <atomic>
struct Y4DynLds {
    std::atomic<unsigned long long> done{0ull};
    bool ensure(const void* kernel, size_t bytes) {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess) return false;
        if (d >= 0 && d < 64 && ((done.load(std::memory_order_acquire) >> d) & 1ull)) return true;
        if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
        if (d >= 0 && d < 64) done.fetch_or(1ull << d, std::memory_order_release);
        return true;
    }
};

// darknet/darknet.py:14-20: x * tanh(softplus(x)).  With n = e^x:
// tanh(log(1+n)) = ((1+n)^2-1)/((1+n)^2+1) = n(n+2)/(n(n+2)+2); softplus threshold 20 of
// torch (returns x above it) coincides with the ratio being exactly 1.0f there.
// One v_exp_f32 + one v_rcp_f32 (both ~1 ulp): these kernels are HBM-bound only if the
// activation costs a handful of VALU ops per element.
__device__ __forceinline__ float y4_mish(float x) {
    const float n = __expf(fminf(x, 20.0f));
    const float w = n * (n + 2.0f);
    return x * (w * __frcp_rn(w + 2.0f));
}
// d/dx [x * t(x)], t = tanh(softplus(x)):  t + x * (1 - t^2) * sigmoid(x).  With w = n(n + 2), d = w + 2:
// t = w / d, 1 - t^2 = 4 (w + 1) / d^2 = 4 (n + 1)^2 / d^2, sigmoid = n / (n + 1)  =>  (1 - t^2) sigmoid = 4 n (n + 1) / d^2,
// so the derivative is (w + 4 x n (n + 1) / d) / d: ONE v_rcp_f32 beside the v_exp_f32 (no cancellation: every term but x
// is positive).  The BatchNorm backward sweeps evaluate this twice per element and run close to the VALU rate with it.
__device__ __forceinline__ float y4_mish_grad(float x) {
    const float n = __expf(fminf(x, 20.0f));
    const float w = n * (n + 2.0f);
    const float rd = __frcp_rn(w + 2.0f);
    const float q = fmaf(n, n, n);                         // n (n + 1)
    return fmaf(4.0f * (x * q), rd, w) * rd;
}
__device__ __forceinline__ float y4_act(float x, int act) {
    switch (act) {
        case Y4_ACT_LEAKY: return x > 0.0f ? x : 0.1f * x;
        case Y4_ACT_MISH: return y4_mish(x);
        case Y4_ACT_RELU: return fmaxf(x, 0.0f);
        default: return x;
    }
}
__device__ __forceinline__ float y4_act_grad(float x, int act) {
    switch (act) {
        case Y4_ACT_LEAKY: return x > 0.0f ? 1.0f : 0.1f;
        case Y4_ACT_MISH: return y4_mish_grad(x);
        case Y4_ACT_RELU: return x > 0.0f ? 1.0f : 0.0f;
        default: return 1.0f;
    }
}

// the same with the activation a compile-time constant (the BatchNorm sweeps: no branch per element, full unrolling)
template <int ACT> __device__ __forceinline__ float y4_act_t(float x) {
    if constexpr (ACT == Y4_ACT_LEAKY) return x > 0.0f ? x : 0.1f * x;
    else if constexpr (ACT == Y4_ACT_MISH) return y4_mish(x);
    else if constexpr (ACT == Y4_ACT_RELU) return fmaxf(x, 0.0f);
    else return x;
}
template <int ACT> __device__ __forceinline__ float y4_act_grad_t(float x) {
    if constexpr (ACT == Y4_ACT_LEAKY) return x > 0.0f ? 1.0f : 0.1f;
    else if constexpr (ACT == Y4_ACT_MISH) return y4_mish_grad(x);
    else if constexpr (ACT == Y4_ACT_RELU) return x > 0.0f ? 1.0f : 0.0f;
    else return 1.0f;
}

// XCD-aware block remap (MI355X: 8 XCDs, blocks dealt round-robin, private L2 each): give every
// XCD a contiguous chunk of the logical tile sequence so neighbouring tiles (shared halo rows /
// shared filter panels) hit the same L2.  Bijective for any nwg (guide §5 "XCD swizzle").
__device__ __forceinline__ int y4_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// Raw buffer loads (SRD in SGPRs, 32-bit byte offsets): an offset >= num_bytes returns zeros in
// hardware, which replaces both the branch around a masked load and the select on its result.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t y4_make_rsrc(const void* base, unsigned num_bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)num_bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 y4_buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff_bytes, unsigned soff_bytes) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff_bytes, (int)soff_bytes, 0);
    return __builtin_bit_cast(f32x4, v);
}
