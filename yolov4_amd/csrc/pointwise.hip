// HBM-bound kernels around the convolutions: BatchNorm statistics / apply / backward fused with
// the activation (and the ResBlock skip), channel concat copies, gradient fan-in adds, SPP max
// pools, nearest x2 upsample.  All NHWC fp32 with an explicit pixel pitch, 16-B vector access
// (C % 4 == 0, pitch % 4 == 0, 16-B aligned bases), one channel-vector per thread so the
// per-channel constants live in registers for the whole row sweep.
#include "common.h"

namespace {

constexpr int PW_THREADS = 256;
constexpr int PW_UNROLL = 4;               // rows in flight per thread in the streaming sweeps
// the activation as a template argument of the BatchNorm sweeps (A_ inside CALL); unknown ids take the linear instance,
// as y4_act() does
#define Y4_ACT_SWITCH(ACTV, CALL)                                              \
    switch (ACTV) {                                                            \
        case Y4_ACT_MISH: { constexpr int A_ = Y4_ACT_MISH; CALL; } break;     \
        case Y4_ACT_LEAKY: { constexpr int A_ = Y4_ACT_LEAKY; CALL; } break;   \
        case Y4_ACT_RELU: { constexpr int A_ = Y4_ACT_RELU; CALL; } break;     \
        default: { constexpr int A_ = Y4_ACT_LINEAR; CALL; } break;            \
    }

struct RowMap {            // thread -> (channel vector, row group)
    int tpr;               // threads per row (channel vectors handled concurrently)
    int rpb;               // rows per block iteration
};
static inline RowMap row_map(int C) {
    RowMap r;
    int c4 = C / 4;
    r.tpr = c4 < PW_THREADS ? c4 : PW_THREADS;
    // tpr must divide 256: C/4 in {8,16,32,64,128,256,...}; otherwise fall back to the largest
    // power of two <= c4 and loop the remainder
    int p = 1;
    while (p * 2 <= r.tpr) p *= 2;
    r.tpr = p;
    r.rpb = PW_THREADS / r.tpr;
    return r;
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
// four channels c0 .. c0 + 3 of the row starting at `row` (a float pointer: pitch in fp32 elements); ybf: the row holds plain
// bf16 values in its FIRST HALF (conv mode 'bf16': the plane conv kernels write their result that way, conv_planes.hip)
__device__ __forceinline__ f32x4 ld4y(const float* row, int c0, int ybf) {
    if (!ybf) return ld4(row + c0);
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const u2 b = *reinterpret_cast<const u2*>(reinterpret_cast<const unsigned char*>(row) + c0 * 2);
    f32x4 v;
    v[0] = __uint_as_float(b[0] << 16); v[1] = __uint_as_float(b[0] & 0xffff0000u);
    v[2] = __uint_as_float(b[1] << 16); v[3] = __uint_as_float(b[1] & 0xffff0000u);
    return v;
}

// ---------------------------------------------------------------- BN statistics
// Each thread sums <= ROWS_PER_THREAD rows in fp32, the block folds row groups through LDS and
// issues one fp64 atomic per channel: var = E[y^2] - E[y]^2 is formed in fp64.
constexpr int STAT_ROWS_PER_THREAD = 128;     // upper bound; the launcher lowers it when M is small
static inline int stat_rows(long long M, int rpb) {
    // aim at >= ~2048 blocks (8 per CU) so the sweep runs at HBM rate also on the 19x19 / 38x38 maps
    long long r = M / ((long long)rpb * 4096);
    if (r < 4) r = 4;
    if (r > STAT_ROWS_PER_THREAD) r = STAT_ROWS_PER_THREAD;
    return (int)r;
}

__global__ __launch_bounds__(PW_THREADS) void bn_stats_kernel(const float* __restrict__ y, long long ldy,
                                                              long long M, int C, int tpr, int rpb, int nrows,
                                                              float* __restrict__ part /* [blocks][2][C] */) {
    __shared__ float red[2][PW_THREADS][4];
    const int tid = threadIdx.x;
    const int cv = tid % tpr, rg = tid / tpr;
    const long long row0 = (long long)blockIdx.x * rpb * nrows;
    for (int cb = 0; cb < C; cb += tpr * 4) {           // uniform trip count: barriers inside
        const int c0 = cb + cv * 4;
        const bool cok = c0 < C;
        f32x4 s = {0, 0, 0, 0}, ss = {0, 0, 0, 0};
#pragma unroll 8
        for (int i = 0; i < nrows; ++i) {
            const long long m = row0 + (long long)i * rpb + rg;
            if (m < M && cok) {
                const f32x4 v = ld4(y + m * ldy + c0);
                s += v;
                ss += v * v;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[0][tid][e] = s[e]; red[1][tid][e] = ss[e]; }
        __syncthreads();
        if (rg == 0 && cok) {
            f32x4 ds = {0, 0, 0, 0}, dss = {0, 0, 0, 0};
            for (int g = 0; g < rpb; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ds[e] += red[0][g * tpr + cv][e];
                    dss[e] += red[1][g * tpr + cv][e];
                }
            float* pp = part + (long long)blockIdx.x * 2 * C;
            st4(pp + c0, ds);
            st4(pp + C + c0, dss);
        }
        __syncthreads();
    }
}

// stage 2 of every per-channel reduction: partial rows [nparts][2][C] (fp32, each a sum over a few
// thousand pixels at most) -> bacc[gridDim][2][C] in fp64, one row per block (<= FOLD_BLOCKS); the finalize
// kernels add those rows in index order, so the whole reduction is deterministic and needs no memset.
// Threads are laid out [row group][column] so that narrow layers (2C < 256) still use the whole block, and
// every thread keeps 8 independent loads in flight (the loop is latency-bound otherwise).
constexpr int FOLD_BLOCKS = 256;
__global__ __launch_bounds__(PW_THREADS) void fold_partials_kernel(const float* __restrict__ part, long long nparts,
                                                                   int C2, int cw, double* __restrict__ bacc) {
    __shared__ double red[PW_THREADS];
    const long long per = (nparts + gridDim.x - 1) / gridDim.x;
    const long long p0 = (long long)blockIdx.x * per;
    long long p1 = p0 + per;
    if (p1 > nparts) p1 = nparts;
    const int rg_n = PW_THREADS / cw;                      // row groups (cw = power of two <= 256)
    const int col = threadIdx.x % cw, rg = threadIdx.x / cw;
    for (int c0 = 0; c0 < C2; c0 += cw) {
        const int c = c0 + col;
        double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (c < C2) {
            long long p = p0 + rg;
            for (; p + 7ll * rg_n < p1; p += 8ll * rg_n) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = part[(p + (long long)u * rg_n) * C2 + c];
#pragma unroll
                for (int u = 0; u < 8; ++u) s[u] += (double)v[u];
            }
            for (; p < p1; p += rg_n) s[0] += (double)part[p * C2 + c];
        }
        red[threadIdx.x] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
        __syncthreads();
        if (rg == 0 && c < C2) {
            double t = red[col];
            for (int g = 1; g < rg_n; ++g) t += red[g * cw + col];
            bacc[(long long)blockIdx.x * C2 + c] = t;
        }
        __syncthreads();
    }
}

// sums column c of the <= FOLD_BLOCKS fp64 rows: 8 threads per column (independent loads), combined in LDS
__device__ __forceinline__ void fold_rows32(const double* __restrict__ bacc, int nb, int C, int c0, double (&red)[2][8][32],
                                            double& s, double& ss) {
    const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;           // 256 threads = 8 groups x 32 channels
    const int c = c0 + cl;
    double a = 0, b2 = 0;
    if (c < C) {
        // 8 rows (16 loads) in flight per trip: one row per trip made this a chain of nb / 8 dependent memory round trips
        // (12 us for nb = 256, measured); the adds stay in row order, so the sums are unchanged
        int r = g;
        for (; r + 56 < nb; r += 64) {
            double va[8], vb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                va[u] = bacc[(long long)(r + 8 * u) * 2 * C + c];
                vb[u] = bacc[(long long)(r + 8 * u) * 2 * C + C + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { a += va[u]; b2 += vb[u]; }
        }
        for (; r < nb; r += 8) { a += bacc[(long long)r * 2 * C + c]; b2 += bacc[(long long)r * 2 * C + C + c]; }
    }
    red[0][g][cl] = a; red[1][g][cl] = b2;
    __syncthreads();
    s = 0; ss = 0;
    if (g == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { s += red[0][k][cl]; ss += red[1][k][cl]; }
    }
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ bacc, int nb, long long M, int C, float eps,
                                                          float momentum, float* __restrict__ mean, float* __restrict__ invstd,
                                                          float* __restrict__ rmean, float* __restrict__ rvar, long long* nbt) {
    __shared__ double red[2][8][32];
    double s, ss;
    fold_rows32(bacc, nb, C, blockIdx.x * 32, red, s, ss);
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
    const int c = blockIdx.x * 32 + (threadIdx.x & 31);
    if ((threadIdx.x >> 5) != 0 || c >= C) return;
    const double mu = s / (double)M;
    double var = ss / (double)M - mu * mu;
    if (var < 0) var = 0;
    mean[c] = (float)mu;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean) rmean[c] = (1.0f - momentum) * rmean[c] + momentum * (float)mu;
    if (rvar) {
        const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
        rvar[c] = (1.0f - momentum) * rvar[c] + momentum * (float)unb;
    }
}

// optional by-product of the sweeps that PRODUCE a conv operand: bit pattern of max|finite output| folded into *out
// with an integer atomicMax (order independent); conv mode 3 ("f16x2") scales its operands by it
__device__ __forceinline__ void amax_track(unsigned& m, const f32x4 o) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned b = __float_as_uint(o[e]) & 0x7fffffffu;
        if (b < 0x7f800000u && b > m) m = b;
    }
}
__device__ __forceinline__ void amax_commit(unsigned m, unsigned* out) {
    // waves folded through LDS, ONE atomic per block, and only when it would raise the word (it only grows; a stale
    // read merely costs an atomic).  Per-wave atomics on one address serialised: +150 us per BatchNorm sweep, measured.
    __shared__ unsigned wmax[PW_THREADS / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)m, off, 64);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned v = wmax[0];
#pragma unroll
        for (int w = 1; w < PW_THREADS / 64; ++w) v = wmax[w] > v ? wmax[w] : v;
        if (v > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, v);
    }
}

// Upper bound of max|z|, z = act(gamma xhat + beta) (+ residual), WITHOUT a pass over the data: a sample of M values with
// mean mu and (biased) variance var has max|y - mu| <= sqrt((M - 1) var), so |xhat| <= sqrt(M - 1); |act(v)| <= |v| for
// linear / leaky / relu / mish.  The bound is loose by up to ~2^9 (sqrt(M) at M = 370 000), which costs the f16x2 split
// nothing but headroom: elements keep their 22 significant bits down to 2^-29 of the BOUND, i.e. 2^-20 of the true
// maximum (conv_f16x2.hip, header) -- activations of interest sit within 2^-12 of it.  One block, C <= a few thousand.
__global__ __launch_bounds__(256) void bn_planes_bound_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, int C,
                                                              float sqrt_m1, const unsigned* __restrict__ res_amax,
                                                              unsigned* __restrict__ out, const unsigned* __restrict__ floor_word = nullptr) {
    __shared__ float red[256];
    float m = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) m = fmaxf(m, fabsf(gamma[c]) * sqrt_m1 + fabsf(beta[c]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float b = red[0] * 1.0001f;
        if (res_amax) b += __uint_as_float(*res_amax) * 1.0001f;
        unsigned w = __float_as_uint(b);                   // a NaN / Inf parameter poisons the bound -> scale 1 (f16x2_scale_exp)
        // floor_word: the bound of ANOTHER producer writing into the same pre-split tensor (a concat buffer): the larger bit
        // pattern wins -- the larger value for finite bounds, and a poisoned (Inf / NaN) word stays poisoned
        if (floor_word && *floor_word > w) w = *floor_word;
        *out = w;
    }
}


// Plane output of a sweep whose threads hold 4 consecutive channels each: the two fp16 pieces of the f16x2 split leave as
// WHOLE 16-B chunks -- lanes 2k / 2k + 1 (channels 8j..8j+3 / 8j+4..8j+7 of one pixel) swap halves through DPP, the even
// lane stores the 16 B of hi, the odd lane the 16 B of lo -- so 8 lanes write one complete 128-B line [64 B hi | 64 B lo]
// with one store instruction each, instead of two 8-B stores per lane into half lines.
// Stores go through a buffer resource (uniform base of the trip) + the thread's loop-invariant byte offset `voff` (row in the
// trip + planes_col_off) + a uniform `soff` (row group of the trip).
// HAZARD (gfx950, ROCm 7.2 LLVM): `soff` is ADDED INTO THE VECTOR OFFSET of every store, never passed as the instruction's
// scalar offset.  A buffer_store_dwordx4 whose soffset is an SGPR, followed IMMEDIATELY by a VALU write of its first data
// register (here the `v_and_b32 v, 0x7fffffff, v` of amax_track on the value just stored), stored the OVERWRITTEN value
// under load -- blocks of the grid's second wave wrote |x| for negative x in the first of their four channels.  The
// compiler's hazard recognizer inserts the two wait states this needs only when soffset is NOT a register
// (GCNHazardRecognizer::createsVALUHazard assumes the SGPR form is safe); with an immediate soffset it emits `s_nop 1`.
// Loads keep the scalar offset (their results are tracked by vmcnt).
__device__ __forceinline__ unsigned st_off(unsigned voff, unsigned soff) { return voff + soff; }
// Cache policy of the sweeps' LOADS: nt (streaming) -- every operand is read for the last time in a long while, and the loads
// stop evicting what the neighbouring conv kernels want: 3-10 % per sweep on the 76 x 76 ... 304 x 304 maps, measured in situ;
// the sweeps over the 19 x 19 maps, whose operands do sit in the 256-MB Infinity Cache, run 3-9 % slower by themselves, yet
// the STEP is fastest with nt on them too (bn_loads_nt below); the 128-B rows of the 32-channel stem are 3-6 % slower and keep
// the default.  nt STORES measured slightly slower everywhere: default.
constexpr int POL_NT = 2;
typedef unsigned u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pair_swap(unsigned v) {              // value of lane ^ 1 (quad_perm [1, 0, 3, 2])
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);
}
__device__ __forceinline__ unsigned planes_col_off(int c0, bool paired) {   // byte offset of the thread's (first) store in a plane row
    const unsigned tile = (unsigned)(c0 >> 5) * 128u;
    return paired ? tile + ((c0 & 4) ? 64u : 0u) + (unsigned)((c0 & 24) >> 3) * 16u : tile + (unsigned)(c0 & 31) * 2u;
}
__device__ __forceinline__ void store_planes4(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, int c0, const f32x4 o, float ps,
                                              bool paired) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 h01, h23, l01, l23;
    const float t0 = o[0] * ps, t1 = o[1] * ps, t2 = o[2] * ps, t3 = o[3] * ps;
    h01[0] = (_Float16)t0; h01[1] = (_Float16)t1; h23[0] = (_Float16)t2; h23[1] = (_Float16)t3;
    l01[0] = (_Float16)((t0 - (float)h01[0]) * 2048.f); l01[1] = (_Float16)((t1 - (float)h01[1]) * 2048.f);
    l23[0] = (_Float16)((t2 - (float)h23[0]) * 2048.f); l23[1] = (_Float16)((t3 - (float)h23[1]) * 2048.f);
    const unsigned hv0 = __builtin_bit_cast(unsigned, h01), hv1 = __builtin_bit_cast(unsigned, h23);
    const unsigned lv0 = __builtin_bit_cast(unsigned, l01), lv1 = __builtin_bit_cast(unsigned, l23);
    if (paired) {                                                        // uniform: C % 8 == 0 and both lanes of a pair active
        const bool odd = (c0 & 4) != 0;
        const unsigned r0 = pair_swap(odd ? hv0 : lv0), r1 = pair_swap(odd ? hv1 : lv1);
        u32x4 v;
        if (odd) { v[0] = r0; v[1] = r1; v[2] = lv0; v[3] = lv1; }      // lo of channels 8j .. 8j+7
        else { v[0] = hv0; v[1] = hv1; v[2] = r0; v[3] = r1; }           // hi of channels 8j .. 8j+7
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)st_off(voff, soff), 0, 0);
    } else {
        u2v hv, lv;
        hv[0] = hv0; hv[1] = hv1; lv[0] = lv0; lv[1] = lv1;
        __builtin_amdgcn_raw_buffer_store_b64(hv, rs, (int)st_off(voff, soff), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(lv, rs, (int)(st_off(voff, soff) + 64u), 0, 0);
    }
}

// conv mode 2: four consecutive channels as bf16 (RN) into the first half of the pixel's fp32-sized row (conv_planes.hip, BF);
// voff: row in the trip + 2 c0
__device__ __forceinline__ void store_bf16x4(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, const f32x4 o) {
    u2v v;
    v[0] = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)o[0]) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)o[1]) << 16);
    v[1] = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)o[2]) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)o[3]) << 16);
    __builtin_amdgcn_raw_buffer_store_b64(v, rs, (int)st_off(voff, soff), 0, 0);
}

// ---------------------------------------------------------------- BN apply + act (+ skip)
// How the three BatchNorm sweeps address memory.  They run close to BOTH of their limits -- ~5.5 TB/s of HBM traffic with
// ~16 KB per CU in flight, and 30-45 VALU operations per element (Mish and its derivative, the f16 split) against a budget of
// ~45 at that rate -- so nothing per element or per row may be spent on addresses or masks:
//  * a block's trip covers PW_UNROLL * rpb CONSECUTIVE rows; the trip's first row gives a UNIFORM base (a buffer resource in
//    scalar registers, rebuilt per trip by the scalar unit; its size ends at the tensor's last row), each thread adds ONE
//    loop-invariant 32-bit byte offset per tensor (row in the trip * pitch + channel) and the row group of the trip is the
//    instruction's scalar offset: no vector arithmetic per load or store (a 64-bit multiply per tensor and row before);
//  * a trip that lies inside the tensor -- all but the last -- runs without a mask (a uniform test), the last one row by row;
//  * the activation, the skip operand and the width of y are template arguments (the run-time `act` switch and `ybf` test sat
//    inside the element loop; the bf16 branch of the latter even waited for each load before issuing the next).
template <bool YBF> struct YRaw { u32x4 r; };                   // four channels of y as loaded: fp32 ...
template <> struct YRaw<true> { u2v r; };                       // ... or four bf16 (first half of the row), unpacked at use
template <bool YBF, int POL>
__device__ __forceinline__ void y_load(YRaw<YBF>& d, __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    if constexpr (YBF) d.r = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, (int)soff, POL);
    else d.r = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)soff, POL);
}
template <bool YBF> __device__ __forceinline__ f32x4 y_get(const YRaw<YBF>& d) {
    if constexpr (YBF) {
        f32x4 v;
        v[0] = __uint_as_float(d.r[0] << 16); v[1] = __uint_as_float(d.r[0] & 0xffff0000u);
        v[2] = __uint_as_float(d.r[1] << 16); v[3] = __uint_as_float(d.r[1] & 0xffff0000u);
        return v;
    } else {
        return __builtin_bit_cast(f32x4, d.r);
    }
}
template <int POL>
__device__ __forceinline__ f32x4 bn_load4(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)soff, POL));
}
__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (int)st_off(voff, soff), 0, 0);
}
// resource over rows [tb, M) of a tensor with pitch ld (fp32 elements): rows past the end read zeros / are not written
__device__ __forceinline__ __amdgpu_buffer_rsrc_t trip_rsrc(const float* t, long long ld, long long tb, long long M) {
    const unsigned long long left = (unsigned long long)(M - tb) * (unsigned long long)ld * 4ull;
    return y4_make_rsrc(t ? t + tb * ld : t, t ? (unsigned)(left < 0xfffffff0ull ? left : 0xfffffff0ull) : 0u);
}

template <int ACT, bool RES, bool YBF, bool NT>
__global__ __launch_bounds__(PW_THREADS) void bn_act_fwd_kernel(
    const float* __restrict__ y, long long ldy, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ res, long long ldr, float* __restrict__ z, long long ldz,
    long long M, int C, int tpr, int rpb, unsigned* __restrict__ out_amax, int planes, float* __restrict__ twin) {
    // planes = 0: z fp32, max|z| folded into *out_amax.  z == nullptr: measure only (max|z| into *out_amax, nothing stored).
    // planes = 1: z receives the two fp16 pieces of the f16x2 split, per pixel and 32-channel K tile [64 B hi | 64 B lo]
    //             (conv_planes.hip), scaled by the power of two that *out_amax -- a bound or a measured maximum -- implies.
    // twin != nullptr (with planes = 1): z stays fp32 and `twin` receives the pre-split copy (a tensor with both a
    //             plane-consuming conv and fp32 consumers).
    const int tid = threadIdx.x;
    const int cv = tid % tpr, rg = tid / tpr;
    unsigned amax = 0u;
    float ps = 1.f;
    const bool paired = (C & 7) == 0 && tpr >= 2;         // lanes 2k, 2k + 1: same pixel, adjacent channel quads, both in range
    // planes = 2: z (or twin) receives plain bf16 values, dense in the first half of each fp32-sized row (conv mode 2)
    if (planes == 1) {
        const unsigned e8 = (*out_amax >> 23) & 0xffu;
        int se = 268 - (int)e8;                            // as f16x2_scale_exp (conv_f16x2.hip)
        if (e8 == 0u || e8 == 255u) se = 127;
        se = se < 2 ? 2 : (se > 252 ? 252 : se);
        ps = __uint_as_float((unsigned)se << 23);
    }
    // PW_UNROLL row groups per trip, every load issued before the first use, and the block's rows of a trip CONTIGUOUS:
    // all blocks together advance one front through the tensor (rows of a trip spread gridDim apart: 10 % slower, measured)
    const long long trip_rows = (long long)PW_UNROLL * rpb;
    const long long step = (long long)gridDim.x * trip_rows;
    float* const pdst = twin ? twin : z;                   // where the pre-split / bf16 form goes (twin: dense rows)
    const long long ldp = twin ? (long long)C : ldz;
    const unsigned sy = (unsigned)(rpb * ldy * 4), sr = (unsigned)(rpb * ldr * 4), sz = (unsigned)(rpb * ldz * 4),
                   sp = (unsigned)(rpb * ldp * 4);         // byte strides between the row groups of a trip
    for (int c0 = cv * 4; c0 < C; c0 += tpr * 4) {
        f32x4 a, b;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[e] = invstd[c0 + e] * gamma[c0 + e];
            b[e] = beta[c0 + e] - mean[c0 + e] * a[e];
        }
        const unsigned yo = (unsigned)(rg * ldy * 4) + (unsigned)c0 * (YBF ? 2u : 4u), ro = (unsigned)(rg * ldr * 4) + (unsigned)c0 * 4u,
                       zo = (unsigned)(rg * ldz * 4) + (unsigned)c0 * 4u,
                       po = (unsigned)(rg * ldp * 4) + (planes == 2 ? (unsigned)c0 * 2u : planes_col_off(c0, paired));
        for (long long tb = (long long)blockIdx.x * trip_rows; tb < M; tb += step) {
            const __amdgpu_buffer_rsrc_t yrs = trip_rsrc(y, ldy, tb, M), rrs = trip_rsrc(RES ? res : nullptr, ldr, tb, M),
                                         zrs = trip_rsrc(z, ldz, tb, M), prs = trip_rsrc(pdst, ldp, tb, M);
            YRaw<YBF> v[PW_UNROLL];
            f32x4 r[RES ? PW_UNROLL : 1];
            auto emit = [&](int u) {
                const f32x4 vv = y_get<YBF>(v[u]);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = y4_act_t<ACT>(vv[e] * a[e] + b[e]);
                if constexpr (RES) o += r[u];
                if (planes) {
                    if (planes == 2) store_bf16x4(prs, po, (unsigned)u * sp, o);
                    else store_planes4(prs, po, (unsigned)u * sp, c0, o, ps, paired);
                    if (twin) buf_store4(zrs, zo, (unsigned)u * sz, o);
                } else {
                    if (z) buf_store4(zrs, zo, (unsigned)u * sz, o);
                    amax_track(amax, o);
                }
            };
            if (tb + trip_rows <= M) {                     // (uniform) every row of the trip exists
#pragma unroll
                for (int u = 0; u < PW_UNROLL; ++u) {
                    y_load<YBF, NT ? POL_NT : 0>(v[u], yrs, yo, (unsigned)u * sy);
                    if constexpr (RES) r[u] = bn_load4<NT ? POL_NT : 0>(rrs, ro, (unsigned)u * sr);
                }
#pragma unroll
                for (int u = 0; u < PW_UNROLL; ++u) emit(u);
            } else {                                       // the tensor's last trip
#pragma unroll
                for (int u = 0; u < PW_UNROLL; ++u) {
                    if (tb + rg + (long long)u * rpb >= M) continue;
                    y_load<YBF, NT ? POL_NT : 0>(v[u], yrs, yo, (unsigned)u * sy);
                    if constexpr (RES) r[u] = bn_load4<NT ? POL_NT : 0>(rrs, ro, (unsigned)u * sr);
                    emit(u);
                }
            }
        }
    }
    if (out_amax && !planes) amax_commit(amax, out_amax);
}

// backward pass 1: sum_g[c] = sum_m g, sum_gx[c] = sum_m g * xhat,  g = dz * act'(u)
template <int ACT, bool BOUNDS, bool YBF, bool NT>
__global__ __launch_bounds__(PW_THREADS) void bn_act_bwd_reduce_kernel(
    const float* __restrict__ dz, long long lddz, const float* __restrict__ y, long long ldy,
    const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ beta,
    long long M, int C, int tpr, int rpb, int nrows, float* __restrict__ part, unsigned* __restrict__ bounds) {
    __shared__ float red[2][PW_THREADS][4];
    const int tid = threadIdx.x;
    const int cv = tid % tpr, rg = tid / tpr;
    // block b owns the contiguous row groups [b nrows, (b + 1) nrows); four in flight per trip.  (Dealing the groups out so
    // that all blocks advance one front through the tensor, which gained 10 % in the forward / apply sweeps, measured +-0 here.)
    const long long r0 = (long long)blockIdx.x * nrows * rpb;              // the block's first row
    const unsigned sy = (unsigned)(rpb * ldy * 4), sd = (unsigned)(rpb * lddz * 4);
    float gmax = 0.f, xmax = 0.f;                          // max |g|, max |xhat| seen by this thread (plane output only)
    for (int cb = 0; cb < C; cb += tpr * 4) {           // uniform trip count: barriers inside
        const int c0 = cb + cv * 4;
        const bool cok = c0 < C;
        f32x4 s = {0, 0, 0, 0}, sx = {0, 0, 0, 0};
        if (cok) {
            f32x4 mu, is, ga, be;
#pragma unroll
            for (int e = 0; e < 4; ++e) { mu[e] = mean[c0 + e]; is[e] = invstd[c0 + e]; ga[e] = gamma[c0 + e]; be[e] = beta[c0 + e]; }
            const unsigned yo = (unsigned)(rg * ldy * 4) + (unsigned)c0 * (YBF ? 2u : 4u), d_o = (unsigned)(rg * lddz * 4) + (unsigned)c0 * 4u;
            auto add = [&](const f32x4 v, const f32x4 d) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xh = (v[e] - mu[e]) * is[e];
                    const float g = d[e] * y4_act_grad_t<ACT>(ga[e] * xh + be[e]);
                    s[e] += g;
                    sx[e] += g * xh;
                    if constexpr (BOUNDS) {
                        gmax = fmaxf(gmax, fabsf(g));
                        xmax = fmaxf(xmax, fabsf(xh));
                    }
                }
            };
            int i = 0;                                     // 8 independent loads in flight per trip
            for (; i + 3 < nrows; i += 4) {
                const long long tb = r0 + (long long)i * rpb;
                if (tb + 4ll * rpb > M) break;             // (uniform) the tensor ends inside this trip: row by row below
                const __amdgpu_buffer_rsrc_t yrs = trip_rsrc(y, ldy, tb, M), drs = trip_rsrc(dz, lddz, tb, M);
                YRaw<YBF> v[4];
                f32x4 d[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    y_load<YBF, NT ? POL_NT : 0>(v[u], yrs, yo, (unsigned)u * sy);
                    d[u] = bn_load4<NT ? POL_NT : 0>(drs, d_o, (unsigned)u * sd);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) add(y_get<YBF>(v[u]), d[u]);
            }
            for (; i < nrows; ++i) {
                const long long tb = r0 + (long long)i * rpb;
                if (tb + rg < M) {
                    YRaw<YBF> v;
                    y_load<YBF, NT ? POL_NT : 0>(v, trip_rsrc(y, ldy, tb, M), yo, 0u);
                    add(y_get<YBF>(v), bn_load4<NT ? POL_NT : 0>(trip_rsrc(dz, lddz, tb, M), d_o, 0u));
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[0][tid][e] = s[e]; red[1][tid][e] = sx[e]; }
        __syncthreads();
        if (rg == 0 && cok) {
            f32x4 ds = {0, 0, 0, 0}, dsx = {0, 0, 0, 0};
            for (int g = 0; g < rpb; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ds[e] += red[0][g * tpr + cv][e];
                    dsx[e] += red[1][g * tpr + cv][e];
                }
            float* pp = part + (long long)blockIdx.x * 2 * C;
            st4(pp + c0, ds);
            st4(pp + C + c0, dsx);
        }
        __syncthreads();
    }
    if constexpr (BOUNDS) {                                // fmaxf drops NaNs; an Inf stays and poisons the bound (-> scale 1)
        amax_commit(__float_as_uint(gmax), bounds + 0);
        __syncthreads();
        amax_commit(__float_as_uint(xmax), bounds + 1);
    }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* __restrict__ bacc, int nb, int C,
                                                              double* __restrict__ acc, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, const float* __restrict__ gamma,
                                                              const float* __restrict__ invstd, double invM,
                                                              unsigned* __restrict__ bounds) {
    __shared__ double red[2][8][32];
    double s, ss;
    fold_rows32(bacc, nb, C, blockIdx.x * 32, red, s, ss);
    const int c = blockIdx.x * 32 + (threadIdx.x & 31);
    if ((threadIdx.x >> 5) != 0) return;
    const bool cok = c < C;
    if (cok) {
        acc[c] = s; acc[C + c] = ss;                       // read by the apply pass
        dbeta[c] = (float)s;
        dgamma[c] = (float)ss;
    }
    if (bounds) {                                          // wave 0 only: maxima over this block's 32 channels
        float w = cok ? fabsf(gamma[c] * invstd[c]) : 0.f, k1 = cok ? fabsf((float)(s * invM)) : 0.f,
              k2 = cok ? fabsf((float)(ss * invM)) : 0.f;
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) {
            w = fmaxf(w, __shfl_xor(w, off, 64)); k1 = fmaxf(k1, __shfl_xor(k1, off, 64)); k2 = fmaxf(k2, __shfl_xor(k2, off, 64));
        }
        if ((threadIdx.x & 31) == 0) {
            // rounded UP a little: the bound must dominate the fp32 arithmetic of the apply pass
            atomicMax(bounds + 2, __float_as_uint(w * 1.0001f));
            atomicMax(bounds + 3, __float_as_uint(k1 * 1.0001f));
            atomicMax(bounds + 4, __float_as_uint(k2 * 1.0001f));
        }
    }
}

// backward pass 2: dy = gamma*invstd * (g - sum_g/M - xhat * sum_gx/M)
template <int ACT, bool YBF, bool NT>
__global__ __launch_bounds__(PW_THREADS) void bn_act_bwd_apply_kernel(
    const float* __restrict__ dz, long long lddz, const float* __restrict__ y, long long ldy,
    const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ beta,
    const double* __restrict__ acc, float* __restrict__ dy, long long lddy,
    long long M, int C, int tpr, int rpb, unsigned* __restrict__ out_amax, unsigned* __restrict__ bounds, int frozen, int bf,
    float* __restrict__ twin) {
    const int tid = threadIdx.x;
    const int cv = tid % tpr, rg = tid / tpr;
    const double invM = frozen ? 0.0 : 1.0 / (double)M;    // frozen statistics: no batch-statistic terms in dy
    unsigned amax = 0u;
    const bool paired = (C & 7) == 0 && tpr >= 2;
    // plane output (conv mode 3): dy leaves as the two fp16 pieces the conv kernels would otherwise split it into,
    // per pixel and 32-channel K tile [64 B hi | 64 B scaled lo] in the 4C bytes of the fp32 row.  The scale needs max|dy| BEFORE the
    // sweep: |dy| <= max|gamma invstd| (max|g| + max|k1| + max|xhat| max|k2|), all five maxima left by the reduce /
    // finalize kernels in bounds[0..4]; the bound goes to bounds[5] for the consumers (same word -> same scale).
    float ps = 1.f;
    if (bounds) {
        const float bnd = __uint_as_float(bounds[2]) *
                          (__uint_as_float(bounds[0]) + __uint_as_float(bounds[3]) + __uint_as_float(bounds[1]) * __uint_as_float(bounds[4])) * 1.0001f;
        const unsigned bb = __float_as_uint(bnd);
        if (blockIdx.x == 0 && tid == 0) bounds[5] = bb;
        const unsigned e8 = (bb >> 23) & 0xffu;
        int se = 268 - (int)e8;                            // as f16x2_scale_exp (conv_f16x2.hip)
        if (e8 == 0u || e8 == 255u) se = 127;
        se = se < 2 ? 2 : (se > 252 ? 252 : se);
        ps = __uint_as_float((unsigned)se << 23);
    }
    // addressing: as bn_act_fwd_kernel (uniform trip base + loop-invariant 32-bit offsets, the last trip row by row)
    const long long trip_rows = (long long)PW_UNROLL * rpb;
    const long long step = (long long)gridDim.x * trip_rows;
    const bool split = bf || bounds;                       // dy leaves as bf16 / as f16x2 planes (twin: beside the fp32 form)
    float* const pdst = twin ? twin : dy;
    const long long ldp = twin ? (long long)C : lddy;
    const unsigned sy = (unsigned)(rpb * ldy * 4), sd = (unsigned)(rpb * lddz * 4), so = (unsigned)(rpb * lddy * 4),
                   sp = (unsigned)(rpb * ldp * 4);
    for (int c0 = cv * 4; c0 < C; c0 += tpr * 4) {
        f32x4 mu, is, ga, be, k1, k2, gi;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mu[e] = mean[c0 + e]; is[e] = invstd[c0 + e]; ga[e] = gamma[c0 + e]; be[e] = beta[c0 + e];
            k1[e] = (float)(acc[c0 + e] * invM);
            k2[e] = (float)(acc[C + c0 + e] * invM);
            gi[e] = ga[e] * is[e];
        }
        const unsigned yo = (unsigned)(rg * ldy * 4) + (unsigned)c0 * (YBF ? 2u : 4u), d_o = (unsigned)(rg * lddz * 4) + (unsigned)c0 * 4u,
                       oo = (unsigned)(rg * lddy * 4) + (unsigned)c0 * 4u,
                       po = (unsigned)(rg * ldp * 4) + (bf ? (unsigned)c0 * 2u : planes_col_off(c0, paired));
        for (long long tb = (long long)blockIdx.x * trip_rows; tb < M; tb += step) {
            const __amdgpu_buffer_rsrc_t yrs = trip_rsrc(y, ldy, tb, M), drs = trip_rsrc(dz, lddz, tb, M),
                                         ors = trip_rsrc(dy, lddy, tb, M), prs = trip_rsrc(pdst, ldp, tb, M);
            YRaw<YBF> v[PW_UNROLL];
            f32x4 d[PW_UNROLL];
            auto emit = [&](int u) {
                const f32x4 vv = y_get<YBF>(v[u]);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xh = (vv[e] - mu[e]) * is[e];
                    const float g = d[u][e] * y4_act_grad_t<ACT>(ga[e] * xh + be[e]);
                    o[e] = gi[e] * (g - k1[e] - xh * k2[e]);
                }
                if (split) {
                    // twin: dy stays fp32 (+ its maximum in the bf16 mode) for a register-staged dgrad, the pre-split / bf16 copy
                    // (dense rows) goes to the plane wgrad
                    if (bf) store_bf16x4(prs, po, (unsigned)u * sp, o);
                    else store_planes4(prs, po, (unsigned)u * sp, c0, o, ps, paired);
                    if (twin) {
                        buf_store4(ors, oo, (unsigned)u * so, o);
                        if (bf) amax_track(amax, o);
                    }
                } else {
                    buf_store4(ors, oo, (unsigned)u * so, o);
                    amax_track(amax, o);
                }
            };
            if (tb + trip_rows <= M) {
#pragma unroll
                for (int u = 0; u < PW_UNROLL; ++u) {
                    y_load<YBF, NT ? POL_NT : 0>(v[u], yrs, yo, (unsigned)u * sy);
                    d[u] = bn_load4<NT ? POL_NT : 0>(drs, d_o, (unsigned)u * sd);
                }
#pragma unroll
                for (int u = 0; u < PW_UNROLL; ++u) emit(u);
            } else {
#pragma unroll
                for (int u = 0; u < PW_UNROLL; ++u) {
                    if (tb + rg + (long long)u * rpb >= M) continue;
                    y_load<YBF, NT ? POL_NT : 0>(v[u], yrs, yo, (unsigned)u * sy);
                    d[u] = bn_load4<NT ? POL_NT : 0>(drs, d_o, (unsigned)u * sd);
                    emit(u);
                }
            }
        }
    }
    if (out_amax && !bounds && (!bf || twin)) amax_commit(amax, out_amax);
}

// generic column sums (C arbitrary, scalar loads): dbias of the 255-channel head convs.  Two fixed-order stages, no atomics:
// block r adds rows r, r + R, r + 2R, ... of its column group into part[r][c]; the finalize kernel adds the R partials
// of a column in index order in fp64.
__global__ __launch_bounds__(PW_THREADS) void colsum_kernel(const float* __restrict__ x, long long ldx, long long M,
                                                            int C, float* __restrict__ part) {
    const int c = blockIdx.y * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (long long m = blockIdx.x; m < M; m += gridDim.x) s += x[m * ldx + c];
    part[(long long)blockIdx.x * C + c] = s;
}
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ part, int R, int C,
                                                              float* __restrict__ out) {
    // 32 lanes per column: lane l adds partials l, l + 32, ... in index order, then a fixed xor tree (deterministic)
    const int c = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
    double s = 0.0;
    if (c < C)
        for (int r = l; r < R; r += 32) s += (double)part[(long long)r * C + c];
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off, 32);
    if (c < C && l == 0) out[c] = (float)s;
}

__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                               float* scale, float* shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float s = gamma[c] / sqrtf(rv[c] + eps);
    scale[c] = s;
    shift[c] = beta[c] - rm[c] * s;
}

// ---------------------------------------------------------------- copies / adds
__global__ __launch_bounds__(PW_THREADS) void copy_rows_kernel(const float* __restrict__ src, long long lds_,
                                                               float* __restrict__ dst, long long ldd,
                                                               long long M, int C4) {
    const long long total = M * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / C4;
        const int c = (int)(i - m * C4) * 4;
        st4(dst + m * ldd + c, ld4(src + m * lds_ + c));
    }
}
__global__ __launch_bounds__(PW_THREADS) void add_rows_kernel(const float* a, long long lda, const float* b,
                                                              long long ldb, float* out, long long ldo,
                                                              long long M, int C4) {
    const long long total = M * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / C4;
        const int c = (int)(i - m * C4) * 4;
        st4(out + m * ldo + c, ld4(a + m * lda + c) + ld4(b + m * ldb + c));
    }
}

// ---------------------------------------------------------------- max pool (stride 1, same pad)
__global__ __launch_bounds__(PW_THREADS) void maxpool_fwd_kernel(const float* __restrict__ x, long long ldx,
                                                                 float* __restrict__ y, long long ldy,
                                                                 signed char* __restrict__ idx,
                                                                 int B, int H, int W, int C4, int ks) {
    const long long total = (long long)B * H * W * C4;
    const int p = ks / 2;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        const long long pix = i / C4;
        const int w = (int)(pix % W);
        const int h = (int)((pix / W) % H);
        const long long b = pix / ((long long)W * H);
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {-1, -1, -1, -1};
        for (int r = 0; r < ks; ++r) {
            const int hi = h + r - p;
            if ((unsigned)hi >= (unsigned)H) continue;
            for (int q = 0; q < ks; ++q) {
                const int wi = w + q - p;
                if ((unsigned)wi >= (unsigned)W) continue;
                const f32x4 v = ld4(x + ((b * H + hi) * W + wi) * ldx + c);
                const int code = r * ks + q;
#pragma unroll
                for (int e = 0; e < 4; ++e) {            // aten max_pool2d: (val > max) || isnan(val); first max wins
                    if (bi[e] < 0) bi[e] = code;
                    if (v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; bi[e] = code; }
                }
            }
        }
        st4(y + pix * ldy + c, best);
        if (idx) {
            char4 o;
            o.x = (signed char)bi[0]; o.y = (signed char)bi[1]; o.z = (signed char)bi[2]; o.w = (signed char)bi[3];
            *reinterpret_cast<char4*>(idx + pix * (long long)(C4 * 4) + c) = o;
        }
    }
}

template <bool ACC>
__global__ __launch_bounds__(PW_THREADS) void maxpool_bwd_kernel(const float* __restrict__ dy, long long lddy,
                                                                 const signed char* __restrict__ idx,
                                                                 float* __restrict__ dx, long long lddx,
                                                                 int B, int H, int W, int C4, int ks) {
    const long long total = (long long)B * H * W * C4;
    const int p = ks / 2;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        const long long pix = i / C4;
        const int w = (int)(pix % W);
        const int h = (int)((pix / W) % H);
        const long long b = pix / ((long long)W * H);
        f32x4 s = {0, 0, 0, 0};
        for (int r = 0; r < ks; ++r) {
            const int ho = h - r + p;                      // output whose window offset r lands on h
            if ((unsigned)ho >= (unsigned)H) continue;
            for (int q = 0; q < ks; ++q) {
                const int wo = w - q + p;
                if ((unsigned)wo >= (unsigned)W) continue;
                const long long op = (b * H + ho) * W + wo;
                const char4 id = *reinterpret_cast<const char4*>(idx + op * (long long)(C4 * 4) + c);
                const int code = r * ks + q;
                if (id.x == code || id.y == code || id.z == code || id.w == code) {
                    const f32x4 g = ld4(dy + op * lddy + c);
                    if (id.x == code) s[0] += g[0];
                    if (id.y == code) s[1] += g[1];
                    if (id.z == code) s[2] += g[2];
                    if (id.w == code) s[3] += g[3];
                }
            }
        }
        float* o = dx + pix * lddx + c;
        if (ACC) s += ld4(o);
        st4(o, s);
    }
}

// ---------------------------------------------------------------- nearest x2 upsample
__global__ __launch_bounds__(PW_THREADS) void upsample2x_fwd_kernel(const float* __restrict__ x, long long ldx,
                                                                    float* __restrict__ y, long long ldy,
                                                                    int B, int H, int W, int C4) {
    const int H2 = 2 * H, W2 = 2 * W;
    const long long total = (long long)B * H2 * W2 * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        const long long pix = i / C4;
        const int w = (int)(pix % W2);
        const int h = (int)((pix / W2) % H2);
        const long long b = pix / ((long long)W2 * H2);
        st4(y + pix * ldy + c, ld4(x + ((b * H + (h >> 1)) * W + (w >> 1)) * ldx + c));
    }
}
__global__ __launch_bounds__(PW_THREADS) void upsample2x_bwd_kernel(const float* __restrict__ dy, long long lddy,
                                                                    float* __restrict__ dx, long long lddx,
                                                                    int B, int H, int W, int C4) {
    const int W2 = 2 * W, H2 = 2 * H;
    const long long total = (long long)B * H * W * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        const long long pix = i / C4;
        const int w = (int)(pix % W);
        const int h = (int)((pix / W) % H);
        const long long b = pix / ((long long)W * H);
        const long long o = (b * H2 + 2 * h) * W2 + 2 * w;
        // same order as the reference's autograd would sum is irrelevant: 4 terms, fixed order here
        f32x4 s = ld4(dy + o * lddy + c) + ld4(dy + (o + 1) * lddy + c);
        s += ld4(dy + (o + W2) * lddy + c) + ld4(dy + (o + W2 + 1) * lddy + c);
        st4(dx + pix * lddx + c, s);
    }
}

// ---------------------------------------------------------------- general nearest upsample (Upsample, yolov4.py:82-90)
// Source index of destination index d: integer factor f > 0 -> d / f (the eval branch's view/expand/contiguous);
// otherwise min(floor(d * scale), in - 1) with scale = (float)in / out, the arithmetic of F.interpolate(size=...,
// mode='nearest') in the train branch (ATen nearest_neighbor_compute_source_index).
__device__ __forceinline__ int nearest_src(int d, int f, float scale, int n_in) {
    if (f > 0) return d / f;
    const int s = (int)floorf((float)d * scale);
    return s < n_in - 1 ? s : n_in - 1;
}
__global__ __launch_bounds__(PW_THREADS) void upsample_nearest_fwd_kernel(const float* __restrict__ x, long long ldx,
                                                                          float* __restrict__ y, long long ldy,
                                                                          int B, int H, int W, int Ho, int Wo, int C4,
                                                                          int fh, int fw, float sh, float sw) {
    const long long total = (long long)B * Ho * Wo * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        const long long pix = i / C4;
        const int w = (int)(pix % Wo);
        const int h = (int)((pix / Wo) % Ho);
        const long long b = pix / ((long long)Wo * Ho);
        const int hs = nearest_src(h, fh, sh, H), ws = nearest_src(w, fw, sw, W);
        st4(y + pix * ldy + c, ld4(x + ((b * H + hs) * W + ws) * ldx + c));
    }
}
// backward: a source pixel gathers the contiguous block of destination pixels that map to it (the map is monotone),
// summed row-major in a fixed order -> deterministic, no atomics
__device__ __forceinline__ void nearest_range(int s, int f, float scale, int n_in, int n_out, int& lo, int& hi) {
    if (f > 0) { lo = s * f; hi = lo + f; if (hi > n_out) hi = n_out; return; }
    int d = (int)((float)s / scale) - 2;
    if (d < 0) d = 0;
    while (d < n_out && nearest_src(d, 0, scale, n_in) < s) ++d;
    lo = d;
    while (d < n_out && nearest_src(d, 0, scale, n_in) == s) ++d;
    hi = d;
}
__global__ __launch_bounds__(PW_THREADS) void upsample_nearest_bwd_kernel(const float* __restrict__ dy, long long lddy,
                                                                          float* __restrict__ dx, long long lddx,
                                                                          int B, int H, int W, int Ho, int Wo, int C4,
                                                                          int fh, int fw, float sh, float sw) {
    const long long total = (long long)B * H * W * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        const long long pix = i / C4;
        const int w = (int)(pix % W);
        const int h = (int)((pix / W) % H);
        const long long b = pix / ((long long)W * H);
        int h0, h1, w0, w1;
        nearest_range(h, fh, sh, H, Ho, h0, h1);
        nearest_range(w, fw, sw, W, Wo, w0, w1);
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int hh = h0; hh < h1; ++hh)
            for (int ww = w0; ww < w1; ++ww) s += ld4(dy + ((b * Ho + hh) * Wo + ww) * lddy + c);
        st4(dx + pix * lddx + c, s);
    }
}

// ---------------------------------------------------------------- standalone activation (Mish.forward, darknet.py:14-20)
// Flat elementwise sweep over a dense tensor of n floats; the same device functions as the fused epilogues.
__global__ __launch_bounds__(PW_THREADS) void act_fwd_flat_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                  long long n, int act) {
    const long long n4 = n >> 2;
    const bool vec = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
    const long long stride = (long long)gridDim.x * blockDim.x, t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    long long done = 0;
    if (vec) {
        for (long long i = t; i < n4; i += stride) {
            f32x4 v = ld4(x + 4 * i);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = y4_act(v[e], act);
            st4(y + 4 * i, v);
        }
        done = n4 << 2;
    }
    for (long long i = done + t; i < n; i += stride) y[i] = y4_act(x[i], act);
}
__global__ __launch_bounds__(PW_THREADS) void act_bwd_flat_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                  float* __restrict__ dx, long long n, int act) {
    const long long n4 = n >> 2;
    const bool vec = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0;
    const long long stride = (long long)gridDim.x * blockDim.x, t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    long long done = 0;
    if (vec) {
        for (long long i = t; i < n4; i += stride) {
            const f32x4 v = ld4(x + 4 * i), g = ld4(dy + 4 * i);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = g[e] * y4_act_grad(v[e], act);
            st4(dx + 4 * i, o);
        }
        done = n4 << 2;
    }
    for (long long i = done + t; i < n; i += stride) dx[i] = dy[i] * y4_act_grad(x[i], act);
}

inline int grid_for(long long total, int cap = 256 * 16) {
    long long b = (total + PW_THREADS - 1) / PW_THREADS;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}
inline bool vec_ok(const void* p, long long ld, int C) {
    return p && (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (ld & 3) == 0 && (C & 3) == 0 && ld >= C && C > 0;
}

}  // namespace

extern "C" {

static long long bn_blocks(long long M, int C) {
    const RowMap rm = row_map(C);
    const long long rows_per_block = (long long)rm.rpb * stat_rows(M, rm.rpb);
    return (M + rows_per_block - 1) / rows_per_block;
}
// [2C doubles][blocks * 2C floats]
size_t y4_bn_workspace(long long M, int C) {
    if (M <= 0 || C <= 0) return 0;
    return y4_bn_finalize_workspace(C) + (size_t)bn_blocks(M, C) * 2 * C * sizeof(float);
}
// workspace head: acc[2C] | bacc[FOLD_BLOCKS][2C] doubles
size_t y4_bn_finalize_workspace(int C) {
    return C > 0 ? (size_t)(1 + FOLD_BLOCKS) * 2 * C * sizeof(double) : 0;
}
// nt loads in the BatchNorm sweeps (POL_NT above): rows of >= 256 B.  Y4_BN_NT=0 / 1: never / always (A/B on one box, three
// rounds, img/s of the bs = 64 step: never 434.5 / 436.0 / 435.8, always 438.7 / 440.0 / 440.3, always but only on tensors of
// >= 150 MB 438.7 / 439.0 / 439.6)
static inline bool bn_loads_nt(long long M, int C) {
    static const char* e = getenv("Y4_BN_NT");
    if (e && *e) return atoi(e) != 0;
    (void)M;
    return C >= 64;
}
static int fold_partials(const float* part, long long nparts, int C, double* bacc, int* nb, hipStream_t st) {
    // >= 16 rows per block; every block gets at least one row (per = ceil(nparts / blocks) may leave trailing
    // blocks empty -> shrink)
    long long want = (nparts + 15) / 16;
    int blocks = (int)(want < FOLD_BLOCKS ? want : FOLD_BLOCKS);
    if (blocks < 1) blocks = 1;
    const long long per = (nparts + blocks - 1) / blocks;
    blocks = (int)((nparts + per - 1) / per);
    if (blocks < 1) blocks = 1;
    *nb = blocks;
    int cw = 32;
    while (cw < 2 * C && cw < PW_THREADS) cw *= 2;
    hipLaunchKernelGGL(fold_partials_kernel, dim3(blocks), dim3(PW_THREADS), 0, st, part, nparts, 2 * C, cw, bacc);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

int y4_bn_stats_f32(const float* y, int ldy, long long M, int C, float* mean, float* invstd,
                    float* running_mean, float* running_var, long long* num_batches_tracked,
                    float momentum, float eps, void* workspace, size_t workspace_bytes, void* stream) {
    if (!y || !mean || !invstd || !workspace) return Y4_ERR_NULL;
    if (!vec_ok(y, ldy, C) || M <= 0) return Y4_ERR_SHAPE;
    if (workspace_bytes < y4_bn_workspace(M, C)) return Y4_ERR_WORKSPACE;
    hipStream_t st = y4_stream(stream);
    double* acc = static_cast<double*>(workspace);
    double* bacc = acc + 2 * C;
    float* part = reinterpret_cast<float*>(acc + (size_t)(1 + FOLD_BLOCKS) * 2 * C);
    const RowMap rm = row_map(C);
    const int nrows = stat_rows(M, rm.rpb);
    const long long blocks = bn_blocks(M, C);
    hipLaunchKernelGGL(bn_stats_kernel, dim3((unsigned)blocks), dim3(PW_THREADS), 0, st, y, (long long)ldy, M, C,
                       rm.tpr, rm.rpb, nrows, part);
    Y4_CHECK_LAUNCH();
    int nb = 0;
    { const int rc = fold_partials(part, blocks, C, bacc, &nb, st); if (rc != Y4_OK) return rc; }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 31) / 32), dim3(256), 0, st, bacc, nb, M, C, eps, momentum, mean,
                       invstd, running_mean, running_var, num_batches_tracked);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

int y4_bn_finalize_partials_f32(const float* partials, long long nparts, long long M, int C,
                                float* mean, float* invstd, float* running_mean, float* running_var,
                                long long* num_batches_tracked, float momentum, float eps,
                                void* workspace, size_t workspace_bytes, void* stream) {
    if (!partials || !mean || !invstd || !workspace) return Y4_ERR_NULL;
    if (nparts <= 0 || M <= 0 || C <= 0) return Y4_ERR_SHAPE;
    if (workspace_bytes < y4_bn_finalize_workspace(C)) return Y4_ERR_WORKSPACE;
    hipStream_t st = y4_stream(stream);
    double* bacc = static_cast<double*>(workspace) + 2 * C;
    int nb = 0;
    { const int rc = fold_partials(partials, nparts, C, bacc, &nb, st); if (rc != Y4_OK) return rc; }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 31) / 32), dim3(256), 0, st, bacc, nb, M, C, eps, momentum, mean,
                       invstd, running_mean, running_var, num_batches_tracked);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

int y4_bn_planes_bound_f32(const float* gamma, const float* beta, int C, long long M, const unsigned* floor_word, unsigned* out,
                           void* stream) {
    if (!gamma || !beta || !out) return Y4_ERR_NULL;
    if (C <= 0 || M <= 0) return Y4_ERR_SHAPE;
    hipLaunchKernelGGL(bn_planes_bound_kernel, dim3(1), dim3(256), 0, y4_stream(stream), gamma, beta, C,
                       sqrtf((float)(M > 1 ? M - 1 : 1)), nullptr, out, floor_word);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

int y4_bn_act_fwd_f32(const float* y, int ldy, const float* mean, const float* invstd,
                      const float* gamma, const float* beta, int act,
                      const float* residual, int ldr, float* z, int ldz,
                      long long M, int C, unsigned* out_amax, int z_planes, const unsigned* res_amax, float* planes_twin,
                      void* stream) {
    if (!y || !mean || !invstd || !gamma || !beta) return Y4_ERR_NULL;
    const int ybf = (z_planes >> 4) & 1;                                         // + 16: y holds bf16 values (first half of each row)
    z_planes &= 15;
    if (!z && (!out_amax || z_planes)) return Y4_ERR_NULL;                       // measure-only needs the word to fill
    if (z_planes < 0 || z_planes > 3) return Y4_ERR_SHAPE;
    if (planes_twin && (!z_planes || (reinterpret_cast<uintptr_t>(planes_twin) & 15))) return Y4_ERR_SHAPE;
    // planes: whole K tiles; rows dense, or a channel slice of a wider pre-split tensor (a concat buffer: pitch and slice offset
    // in whole 32-channel tiles, i.e. z 128-B aligned)
    if (z_planes && ((!out_amax && z_planes != 3) || (C & 31))) return Y4_ERR_SHAPE;
    if (z_planes && !planes_twin && ldz != C && (ldz < C || (ldz & 31) || (reinterpret_cast<uintptr_t>(z) & 127))) return Y4_ERR_SHAPE;
    if (z_planes == 2) {
        // the scale comes from an analytic bound of max|z| (no measuring pass); a residual must bring its own maximum
        if (residual && !res_amax) return Y4_ERR_NULL;
        hipLaunchKernelGGL(bn_planes_bound_kernel, dim3(1), dim3(256), 0, y4_stream(stream), gamma, beta, C,
                           sqrtf((float)(M > 1 ? M - 1 : 1)), residual ? res_amax : nullptr, out_amax);
        Y4_CHECK_LAUNCH();
    }
    if (!vec_ok(y, ldy, C) || (z && !vec_ok(z, ldz, C)) || (residual && !vec_ok(residual, ldr, C)) || M <= 0)
        return Y4_ERR_SHAPE;
    const RowMap rm = row_map(C);
    long long blocks = (M + rm.rpb * PW_UNROLL - 1) / (rm.rpb * PW_UNROLL);
    if (blocks > 256 * 16) blocks = 256 * 16;
    const bool nt = bn_loads_nt(M, C);
#define Y4_BN_FWD(RES_, YBF_)                                                                                                    \
    if (nt) Y4_BN_FWD_(RES_, YBF_, true); else Y4_BN_FWD_(RES_, YBF_, false)
#define Y4_BN_FWD_(RES_, YBF_, NT_)                                                                                              \
    hipLaunchKernelGGL((bn_act_fwd_kernel<A_, RES_, YBF_, NT_>), dim3((unsigned)blocks), dim3(PW_THREADS), 0, y4_stream(stream), y, \
                       (long long)ldy, mean, invstd, gamma, beta, residual, (long long)ldr, z, (long long)ldz, M, C, rm.tpr,      \
                       rm.rpb, z_planes == 3 ? nullptr : out_amax, z_planes == 3 ? 2 : (z_planes ? 1 : 0), planes_twin)
    if (residual) {
        if (ybf) { Y4_ACT_SWITCH(act, Y4_BN_FWD(true, true)); }
        else { Y4_ACT_SWITCH(act, Y4_BN_FWD(true, false)); }
    } else {
        if (ybf) { Y4_ACT_SWITCH(act, Y4_BN_FWD(false, true)); }
        else { Y4_ACT_SWITCH(act, Y4_BN_FWD(false, false)); }
    }
#undef Y4_BN_FWD
#undef Y4_BN_FWD_
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

static int bn_act_bwd_impl(const float* dz, int lddz, const float* y, int ldy,
                           const float* mean, const float* invstd, const float* gamma, const float* beta,
                           int act, float* dy, int lddy, float* dgamma, float* dbeta,
                           long long M, int C, void* workspace, size_t workspace_bytes, unsigned* out_amax,
                           unsigned* f16_planes, float* planes_twin, int flags, void* stream) {
    if (!dz || !y || !mean || !invstd || !gamma || !beta || !dy || !dgamma || !dbeta || !workspace) return Y4_ERR_NULL;
    const int frozen = flags & 1, bf = (flags >> 1) & 1, ybf = (flags >> 2) & 1;
    if (frozen && f16_planes) return Y4_ERR_SHAPE;         // (the plane bound is derived for batch statistics)
    if (bf && (f16_planes || (!planes_twin && lddy != C) || (C & 31))) return Y4_ERR_SHAPE;
    if (planes_twin && ((!f16_planes && !bf) || (reinterpret_cast<uintptr_t>(planes_twin) & 15))) return Y4_ERR_SHAPE;
    if (!vec_ok(dz, lddz, C) || !vec_ok(y, ldy, C) || !vec_ok(dy, lddy, C) || M <= 0) return Y4_ERR_SHAPE;
    if (workspace_bytes < y4_bn_workspace(M, C)) return Y4_ERR_WORKSPACE;
    hipStream_t st = y4_stream(stream);
    double* acc = static_cast<double*>(workspace);
    double* bacc = acc + 2 * C;
    float* part = reinterpret_cast<float*>(acc + (size_t)(1 + FOLD_BLOCKS) * 2 * C);
    const RowMap rm = row_map(C);
    const int nrows = stat_rows(M, rm.rpb);
    const long long rblocks = bn_blocks(M, C);
    int nb = 0;
    const bool nt = bn_loads_nt(M, C);
    {
#define Y4_BN_RED(BND_, YBF_)                                                                                                    \
        if (nt) Y4_BN_RED_(BND_, YBF_, true); else Y4_BN_RED_(BND_, YBF_, false)
#define Y4_BN_RED_(BND_, YBF_, NT_)                                                                                              \
        hipLaunchKernelGGL((bn_act_bwd_reduce_kernel<A_, BND_, YBF_, NT_>), dim3((unsigned)rblocks), dim3(PW_THREADS), 0, st, dz,  \
                           (long long)lddz, y, (long long)ldy, mean, invstd, gamma, beta, M, C, rm.tpr, rm.rpb, nrows, part,       \
                           f16_planes)
        if (f16_planes && ybf) return Y4_ERR_SHAPE;        // (bf16 conv results exist only in the bf16 plane mode: no bounds there)
        if (f16_planes) { Y4_ACT_SWITCH(act, Y4_BN_RED(true, false)); }
        else if (ybf) { Y4_ACT_SWITCH(act, Y4_BN_RED(false, true)); }
        else { Y4_ACT_SWITCH(act, Y4_BN_RED(false, false)); }
#undef Y4_BN_RED
#undef Y4_BN_RED_
        Y4_CHECK_LAUNCH();
        const int rc = fold_partials(part, rblocks, C, bacc, &nb, st);
        if (rc != Y4_OK) return rc;
    }
    if (f16_planes && ((!planes_twin && lddy != C) || (C & 31))) return Y4_ERR_SHAPE;       // planes: dense rows, whole 32-channel K-tiles
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 31) / 32), dim3(256), 0, st, bacc, nb, C, acc, dgamma, dbeta, gamma,
                       invstd, 1.0 / (double)M, f16_planes);
    Y4_CHECK_LAUNCH();
    long long blocks = (M + rm.rpb * PW_UNROLL - 1) / (rm.rpb * PW_UNROLL);
    if (blocks > 256 * 16) blocks = 256 * 16;
#define Y4_BN_APP(YBF_)                                                                                                          \
    if (nt) Y4_BN_APP_(YBF_, true); else Y4_BN_APP_(YBF_, false)
#define Y4_BN_APP_(YBF_, NT_)                                                                                                    \
    hipLaunchKernelGGL((bn_act_bwd_apply_kernel<A_, YBF_, NT_>), dim3((unsigned)blocks), dim3(PW_THREADS), 0, st, dz,              \
                       (long long)lddz,                                                                                          \
                       y, (long long)ldy, mean, invstd, gamma, beta, acc, dy, (long long)lddy, M, C, rm.tpr, rm.rpb, out_amax,    \
                       f16_planes, frozen ? 1 : 0, bf, planes_twin)
    if (ybf) { Y4_ACT_SWITCH(act, Y4_BN_APP(true)); }
    else { Y4_ACT_SWITCH(act, Y4_BN_APP(false)); }
#undef Y4_BN_APP
#undef Y4_BN_APP_
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

int y4_bn_act_bwd_f32(const float* dz, int lddz, const float* y, int ldy,
                      const float* mean, const float* invstd, const float* gamma, const float* beta,
                      int act, float* dy, int lddy, float* dgamma, float* dbeta,
                      long long M, int C, void* workspace, size_t workspace_bytes, unsigned* out_amax,
                      unsigned* f16_planes, float* planes_twin, int frozen_stats, void* stream) {
    return bn_act_bwd_impl(dz, lddz, y, ldy, mean, invstd, gamma, beta, act, dy, lddy, dgamma, dbeta, M, C, workspace,
                           workspace_bytes, out_amax, f16_planes, planes_twin, frozen_stats, stream);
}

static int colsum_rows(long long M) { return (int)(M < 1024 ? M : 1024); }

size_t y4_bias_grad_workspace(long long M, int C) {
    if (M <= 0 || C <= 0) return 0;
    return (size_t)colsum_rows(M) * (size_t)C * sizeof(float);
}

int y4_bias_grad_f32(const float* dy, int lddy, long long M, int C, float* dbias,
                     void* workspace, size_t workspace_bytes, void* stream) {
    if (!dy || !dbias || !workspace) return Y4_ERR_NULL;
    if (M <= 0 || C <= 0 || lddy < C) return Y4_ERR_SHAPE;
    if (workspace_bytes < y4_bias_grad_workspace(M, C)) return Y4_ERR_WORKSPACE;
    hipStream_t st = y4_stream(stream);
    float* part = static_cast<float*>(workspace);
    const int R = colsum_rows(M);
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)R, (C + 255) / 256), dim3(256), 0, st, dy, (long long)lddy, M, C, part);
    Y4_CHECK_LAUNCH();
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((C + 7) / 8), dim3(256), 0, st, part, R, C, dbias);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

int y4_bn_fold_f32(const float* gamma, const float* beta, const float* running_mean,
                   const float* running_var, float eps, float* scale, float* shift, int C, void* stream) {
    if (!gamma || !beta || !running_mean || !running_var || !scale || !shift) return Y4_ERR_NULL;
    if (C <= 0) return Y4_ERR_SHAPE;
    hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, y4_stream(stream), gamma, beta,
                       running_mean, running_var, eps, scale, shift, C);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

int y4_copy_channels_f32(const float* src, int lds, float* dst, int ldd, long long M, int C, void* stream) {
    if (!vec_ok(src, lds, C) || !vec_ok(dst, ldd, C) || M <= 0) return Y4_ERR_SHAPE;
    hipLaunchKernelGGL(copy_rows_kernel, dim3(grid_for(M * (C / 4))), dim3(PW_THREADS), 0, y4_stream(stream),
                       src, (long long)lds, dst, (long long)ldd, M, C / 4);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

int y4_add_f32(const float* a, int lda, const float* b, int ldb, float* out, int ldo,
               long long M, int C, void* stream) {
    if (!vec_ok(a, lda, C) || !vec_ok(b, ldb, C) || !vec_ok(out, ldo, C) || M <= 0) return Y4_ERR_SHAPE;
    hipLaunchKernelGGL(add_rows_kernel, dim3(grid_for(M * (C / 4))), dim3(PW_THREADS), 0, y4_stream(stream),
                       a, (long long)lda, b, (long long)ldb, out, (long long)ldo, M, C / 4);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

int y4_maxpool_s1_fwd_f32(const float* x, int ldx, float* y, int ldy, signed char* idx,
                          int B, int H, int W, int C, int ksize, void* stream) {
    if (!vec_ok(x, ldx, C) || !vec_ok(y, ldy, C) || B <= 0 || H <= 0 || W <= 0) return Y4_ERR_SHAPE;
    if (ksize < 1 || (ksize & 1) == 0 || ksize > 11) return Y4_ERR_SHAPE;     // idx code r*k+q must fit int8
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for((long long)B * H * W * (C / 4))), dim3(PW_THREADS), 0,
                       y4_stream(stream), x, (long long)ldx, y, (long long)ldy, idx, B, H, W, C / 4, ksize);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

int y4_maxpool_s1_bwd_f32(const float* dy, int lddy, const signed char* idx, float* dx, int lddx,
                          int accumulate, int B, int H, int W, int C, int ksize, void* stream) {
    if (!idx) return Y4_ERR_NULL;
    if (!vec_ok(dy, lddy, C) || !vec_ok(dx, lddx, C) || B <= 0 || H <= 0 || W <= 0) return Y4_ERR_SHAPE;
    if (ksize < 1 || (ksize & 1) == 0 || ksize > 11) return Y4_ERR_SHAPE;
    const int grid = grid_for((long long)B * H * W * (C / 4));
    if (accumulate)
        hipLaunchKernelGGL(maxpool_bwd_kernel<true>, dim3(grid), dim3(PW_THREADS), 0, y4_stream(stream), dy,
                           (long long)lddy, idx, dx, (long long)lddx, B, H, W, C / 4, ksize);
    else
        hipLaunchKernelGGL(maxpool_bwd_kernel<false>, dim3(grid), dim3(PW_THREADS), 0, y4_stream(stream), dy,
                           (long long)lddy, idx, dx, (long long)lddx, B, H, W, C / 4, ksize);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

int y4_upsample2x_fwd_f32(const float* x, int ldx, float* y, int ldy, int B, int H, int W, int C, void* stream) {
    if (!vec_ok(x, ldx, C) || !vec_ok(y, ldy, C) || B <= 0 || H <= 0 || W <= 0) return Y4_ERR_SHAPE;
    hipLaunchKernelGGL(upsample2x_fwd_kernel, dim3(grid_for((long long)B * 4 * H * W * (C / 4))), dim3(PW_THREADS), 0,
                       y4_stream(stream), x, (long long)ldx, y, (long long)ldy, B, H, W, C / 4);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

int y4_upsample2x_bwd_f32(const float* dy, int lddy, float* dx, int lddx, int B, int H, int W, int C, void* stream) {
    if (!vec_ok(dy, lddy, C) || !vec_ok(dx, lddx, C) || B <= 0 || H <= 0 || W <= 0) return Y4_ERR_SHAPE;
    hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(grid_for((long long)B * H * W * (C / 4))), dim3(PW_THREADS), 0,
                       y4_stream(stream), dy, (long long)lddy, dx, (long long)lddx, B, H, W, C / 4);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

int y4_upsample_nearest_fwd_f32(const float* x, int ldx, float* y, int ldy, int B, int H, int W, int Ho, int Wo, int C,
                                int integer_factor, void* stream) {
    if (!vec_ok(x, ldx, C) || !vec_ok(y, ldy, C) || B <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return Y4_ERR_SHAPE;
    int fh = 0, fw = 0;
    if (integer_factor) {
        if (Ho % H || Wo % W) return Y4_ERR_SHAPE;          // the reference's final .view() fails the same way
        fh = Ho / H; fw = Wo / W;
    }
    hipLaunchKernelGGL(upsample_nearest_fwd_kernel, dim3(grid_for((long long)B * Ho * Wo * (C / 4))), dim3(PW_THREADS), 0,
                       y4_stream(stream), x, (long long)ldx, y, (long long)ldy, B, H, W, Ho, Wo, C / 4, fh, fw,
                       (float)H / (float)Ho, (float)W / (float)Wo);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

int y4_upsample_nearest_bwd_f32(const float* dy, int lddy, float* dx, int lddx, int B, int H, int W, int Ho, int Wo, int C,
                                int integer_factor, void* stream) {
    if (!vec_ok(dy, lddy, C) || !vec_ok(dx, lddx, C) || B <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return Y4_ERR_SHAPE;
    int fh = 0, fw = 0;
    if (integer_factor) {
        if (Ho % H || Wo % W) return Y4_ERR_SHAPE;
        fh = Ho / H; fw = Wo / W;
    }
    hipLaunchKernelGGL(upsample_nearest_bwd_kernel, dim3(grid_for((long long)B * H * W * (C / 4))), dim3(PW_THREADS), 0,
                       y4_stream(stream), dy, (long long)lddy, dx, (long long)lddx, B, H, W, Ho, Wo, C / 4, fh, fw,
                       (float)H / (float)Ho, (float)W / (float)Wo);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

int y4_act_fwd_f32(const float* x, float* y, long long n, int act, void* stream) {
    if (!x || !y) return Y4_ERR_NULL;
    if (n <= 0 || act < 0 || act > 3) return Y4_ERR_SHAPE;
    hipLaunchKernelGGL(act_fwd_flat_kernel, dim3(grid_for((n + 3) / 4)), dim3(PW_THREADS), 0, y4_stream(stream), x, y, n, act);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

int y4_act_bwd_f32(const float* x, const float* dy, float* dx, long long n, int act, void* stream) {
    if (!x || !dy || !dx) return Y4_ERR_NULL;
    if (n <= 0 || act < 0 || act > 3) return Y4_ERR_SHAPE;
    hipLaunchKernelGGL(act_bwd_flat_kernel, dim3(grid_for((n + 3) / 4)), dim3(PW_THREADS), 0, y4_stream(stream), x, dy, dx, n, act);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

}  // extern "C"
