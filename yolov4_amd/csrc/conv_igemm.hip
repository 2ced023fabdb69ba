// Implicit-GEMM convolution for gfx950 on v_mfma_f32_32x32x2_f32 (exact fp32, fp32 accumulate).
//
//   forward / dgrad ("gather" kernel):  D[m][n] = sum_k A[m][k] * Bt[n][k]
//       m = output pixel (b,ho,wo), n = output channel, k = (r,q,c)
//       A is gathered on the fly from the NHWC activation (never materialised),
//       Bt is the KRSC filter read as a row-major [N][K] matrix.
//   wgrad:                              D[n][j] = sum_p dy[p][n] * xg[p][j]
//       p = output pixel, j = (r,q,c); both operands are K-major (pixel-major) as stored.
//
// Tile: 256 threads = 4 waves; each wave owns TM x TN tiles of 32x32 (16 accumulator VGPRs each).
// LDS: [rows][BK + 4] floats (row pitch 144 B -> ds_read_b128 conflict-free across the 16-lane
// groups), double buffered, register-staged global loads issued one K-tile ahead.
#include <stdlib.h>
#include "common.h"
#include <cstdarg>
#include <cstdio>
#include "conv_geom.h"

namespace {

constexpr int BK = 32;            // K-tile depth (floats)


using y4::ConvGeom;
using y4::WgradGeom;

// TRANSPOSED = false: source pixel = (hd*stride - pad + r, wd*stride - pad + q)      [forward]
// TRANSPOSED = true : source pixel = ((hd + pad - r)/stride, (wd + pad - q)/stride)   [dgrad]
//                     (stride 2: exact division is guaranteed by the parity-class tiling)
template <int BM, int BN, int WM, int WN, bool TRANSPOSED, int BKT = 32>
__global__ __launch_bounds__(256, (BKT == 16 ? 3 : 2)) void conv_gather_mfma_f32(const ConvGeom g) {
    constexpr int LPR = BKT / 4;                      // lanes per tile row (16 B each)
    constexpr int RPP = 256 / LPR;                    // rows per load pass
    constexpr int PA = BM / RPP, PB = BN / RPP;       // load passes
    constexpr int LDS_PITCH = BKT + 4;                // 144-B / 80-B row pitch: conflict-free ds_read_b128
    constexpr int BK = BKT;
    constexpr int WTM = BM / WM, WTN = BN / WN;       // wave tile
    constexpr int TM = WTM / 32, TN = WTN / 32;
    static_assert(WM * WN == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                                  // [2][BM][LDS_PITCH]
    float* Bs = smem + 2 * BM * LDS_PITCH;             // [2][BN][LDS_PITCH]
    int* row_m = reinterpret_cast<int*>(smem + 2 * (BM + BN) * LDS_PITCH);   // [BM] dest pixel of each tile row (-1: none)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // logical tile: N-tiles innermost so the blocks of one XCD chunk reuse the same A panel
    int r0 = 0, q0 = 0, tstep = 1;
    int ph = 0, pw = 0, mt_local, nt;
    const bool classed = TRANSPOSED && g.stride == 2;
    if (!classed) {
        const int lt = y4_xcd_remap(blockIdx.x, g.mtiles * g.ntiles);
        mt_local = lt / g.ntiles;
        nt = lt - mt_local * g.ntiles;
    } else {
        // Tile cost differs 4:2:2:1 between parity classes, so a contiguous chunk per XCD would
        // leave XCD 0 with 4x the work of XCD 7.  Every XCD instead walks its own eighth of each
        // class, heaviest class (4 taps) first: balanced across XCDs, L2-local within a class.
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        int c = 0;
        while (c < 3 && slot >= g.cls_slot0[c + 1]) ++c;
        const int per = g.cls_slot0[c + 1] - g.cls_slot0[c];
        const int t = xcd * per + (slot - g.cls_slot0[c]);
        const int tiles_c = (g.cls_tile0[c + 1] - g.cls_tile0[c]) * g.ntiles;
        if (t >= tiles_c) return;                     // padding slot (whole block, before any barrier)
        mt_local = t / g.ntiles;
        nt = t - mt_local * g.ntiles;
        ph = (3 - c) >> 1; pw = (3 - c) & 1;          // slot range 0 = class (1,1): 4 taps
        r0 = (ph + g.pad) & 1; q0 = (pw + g.pad) & 1; tstep = 2;
    }
    const int n0 = nt * BN;
    const int nr = (g.k - r0 + tstep - 1) / tstep, nq = (g.k - q0 + tstep - 1) / tstep;

    const int lrow = tid / LPR, kc = tid % LPR;

    // ---- per-thread A-row bookkeeping (rows are fixed for the whole K loop); byte offsets, 32 bit
    // 32-bit buffer offsets are relative to the first image this tile touches (tensors may exceed 4 GiB)
    const int pix_per_img = classed ? g.cls_h[ph] * g.cls_w[pw] : g.Hd * g.Wd;
    const int b_first = (int)(((long long)mt_local * BM) / pix_per_img);
    const unsigned long long img_bytes = (unsigned long long)g.Hs * g.Ws * (unsigned long long)g.lds_ * 4ull;
    const unsigned long long src_skip = (unsigned long long)b_first * img_bytes;
    const unsigned long long src_left = g.src_total_bytes > src_skip ? g.src_total_bytes - src_skip : 0ull;
    const __amdgpu_buffer_rsrc_t src_rsrc =
        y4_make_rsrc(reinterpret_cast<const char*>(g.src) + src_skip, (unsigned)(src_left < 0xfffffff0ull ? src_left : 0xfffffff0ull));
    const __amdgpu_buffer_rsrc_t wt_rsrc = y4_make_rsrc(g.wt, g.wt_bytes);
    const unsigned OOB = 0xffffffffu;                 // > any num_bytes: the load returns zeros
    unsigned a_base[PA];
    int a_h[PA], a_w[PA];
    bool a_ok[PA];
    const unsigned pix_bytes = (unsigned)g.lds_ * 4u;
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const int row = p * RPP + lrow;
        const int i = mt_local * BM + row;
        int b, hd, wd;
        if (!classed) {
            a_ok[p] = i < g.M;
            const int ii = a_ok[p] ? i : 0;
            b = ii / (g.Hd * g.Wd);
            const int rem = ii - b * (g.Hd * g.Wd);
            hd = rem / g.Wd; wd = rem - hd * g.Wd;
        } else {
            const int hc = g.cls_h[ph], wc = g.cls_w[pw];
            a_ok[p] = i < g.B * hc * wc;
            const int ii = a_ok[p] ? i : 0;
            b = ii / (hc * wc);
            const int rem = ii - b * (hc * wc);
            const int hh = rem / wc;
            hd = 2 * hh + ph; wd = 2 * (rem - hh * wc) + pw;
        }
        if (kc == 0) row_m[row] = a_ok[p] ? (b * g.Hd + hd) * g.Wd + wd : -1;
        a_base[p] = (unsigned)(b - b_first) * (unsigned)(g.Hs * g.Ws) * pix_bytes + kc * 16u;
        if (!TRANSPOSED) {
            a_h[p] = hd * g.stride - g.pad;
            a_w[p] = wd * g.stride - g.pad;
        } else {
            a_h[p] = hd + g.pad;
            a_w[p] = wd + g.pad;
        }
    }
    // ---- per-thread B rows (filter rows n): constant byte offset, the K position is a scalar offset
    unsigned b_off[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        const int n = n0 + p * RPP + lrow;
        b_off[p] = n < g.N ? (unsigned)n * (unsigned)g.K * 4u + kc * 16u : OOB;
    }

    f32x4 ra[PA], rb[PB];
    const int CC = g.Cs / BK;
    int r = r0, q = q0, cc = 0;   // position of the K-tile being LOADED

    // Loads are buffer loads with hardware zero-fill for invalid rows: no branches, no selects, a
    // handful of VALU ops per row, so the K loop is one basic block the scheduler can interleave.
    auto load_tile = [&]() {
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            int hi, wi;
            bool ok = a_ok[p];
            if (!TRANSPOSED) {
                hi = a_h[p] + r; wi = a_w[p] + q;
                ok = ok && (unsigned)hi < (unsigned)g.Hs && (unsigned)wi < (unsigned)g.Ws;
            } else {
                const int th = a_h[p] - r, tw = a_w[p] - q;
                hi = th; wi = tw;
                if (g.stride == 2) { hi = th >> 1; wi = tw >> 1; }     // th, tw even by construction
                ok = ok && th >= 0 && tw >= 0 && hi < g.Hs && wi < g.Ws;
            }
            const unsigned off = a_base[p] + (unsigned)(hi * g.Ws + wi) * pix_bytes;
            ra[p] = y4_buf_load4(src_rsrc, ok ? off : OOB, (unsigned)(cc * BK) * 4u);
        }
        const unsigned koff = (unsigned)((r * g.k + q) * CC + cc) * (BK * 4u);
#pragma unroll
        for (int p = 0; p < PB; ++p) rb[p] = y4_buf_load4(wt_rsrc, b_off[p], koff);
        if (++cc == CC) { cc = 0; q += tstep; if (q >= g.k) { q = q0; r += tstep; } }
    };
    // pad channels of dy (Cout = 255 -> 256) may hold anything: zero them on the way to LDS
    int st_cc = 0;
    auto store_tile = [&](int buf) {
        float* as = As + buf * BM * LDS_PITCH;
        float* bs = Bs + buf * BN * LDS_PITCH;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            f32x4 v = ra[p];
            if (TRANSPOSED) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (st_cc * BK + kc * 4 + e >= g.Cs_valid) v[e] = 0.f;
            }
            *reinterpret_cast<f32x4*>(as + (p * RPP + lrow) * LDS_PITCH + kc * 4) = v;
        }
#pragma unroll
        for (int p = 0; p < PB; ++p)
            *reinterpret_cast<f32x4*>(bs + (p * RPP + lrow) * LDS_PITCH + kc * 4) = rb[p];
        if (++st_cc == CC) st_cc = 0;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int KT = nr * nq * CC;
    load_tile();
    store_tile(0);
    __syncthreads();

    const int fr = lane & 31, fh = lane >> 5;
    const int a_frag = (wm * WTM + fr) * LDS_PITCH + fh * 4;
    const int b_frag = (wn * WTN + fr) * LDS_PITCH + fh * 4;

    auto mfma_steps = [&](const float* as, const float* bs, int u0, int u1) {
#pragma unroll
        for (int u = u0; u < u1; ++u) {
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(as + i * 32 * LDS_PITCH + u * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(bs + j * 32 * LDS_PITCH + u * 8);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][t], fb[j][t], acc[i][j], 0, 0, 0);
        }
    };
    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        const float* as = As + cur * BM * LDS_PITCH + a_frag;
        const float* bs = Bs + cur * BN * LDS_PITCH + b_frag;
        // first quarter of the MFMAs is issued before the next tile's address arithmetic + buffer loads,
        // so that block runs in the shadow of MFMAs already queued on the matrix pipe
        mfma_steps(as, bs, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < KT) load_tile();                  // global loads in flight under the remaining MFMAs
        __builtin_amdgcn_sched_barrier(0);
        mfma_steps(as, bs, 1, BK / 8);
        if (kt + 1 < KT) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WTN + j * 32 + fr;
        const bool nok = n < g.N;
        const float sc = (g.scale && nok) ? g.scale[n] : 1.0f;
        const float sh = (g.shift && nok) ? g.shift[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rbase = wm * WTM + i * 32 + 4 * fh;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = row_m[rbase + (e & 3) + 8 * (e >> 2)];
                if (nok && m >= 0) {
                    float v = acc[i][j][e] * sc + sh;
                    v = y4_act(v, g.act);
                    if (g.res) v += g.res[(long long)m * g.ldr + n];
                    g.dst[(long long)m * g.ldd + n] = v;
                }
            }
        }
    }

    // ---- BatchNorm batch statistics fused into the epilogue: column sums / sums of squares of this
    // M-tile (rows beyond M were zero-filled, so they add nothing), one partial row per tile; a tiny
    // second-stage kernel folds the rows in fp64.  Saves one full read of the conv output.
    if (!TRANSPOSED && g.stats) {
        float* red = smem;                                 // [WM][BN][2]; the K loop ended with a barrier
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float cs = 0.f, css = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) { const float v = acc[i][j][e]; cs += v; css += v * v; }
            cs += __shfl_xor(cs, 32, 64);
            css += __shfl_xor(css, 32, 64);
            if (fh == 0) {
                const int c = wn * WTN + j * 32 + fr;
                red[(wm * BN + c) * 2 + 0] = cs;
                red[(wm * BN + c) * 2 + 1] = css;
            }
        }
        __syncthreads();
        for (int c = tid; c < BN; c += 256) {
            float cs = 0.f, css = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) { cs += red[(w * BN + c) * 2]; css += red[(w * BN + c) * 2 + 1]; }
            const int n = n0 + c;
            if (n < g.N) {
                g.stats[((long long)mt_local * 2 + 0) * g.N + n] = cs;
                g.stats[((long long)mt_local * 2 + 1) * g.N + n] = css;
            }
        }
    }
}

// =====================================================================================================
// Split-bf16 variant of the gather kernel ("bf16x3"): every fp32 operand is split EXACTLY into three bf16
// pieces (8+8+8 mantissa bits, truncation split: x = x1 + x2 + x3) when its tile is staged into LDS, and
// the product is formed on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, products exact, fp32
// accumulate) as the six leading terms  a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1.  The dropped terms are
// <= 2^-23 |a b|, i.e. the result carries fp32-grade error (comparable to an fp32 FMA chain's rounding)
// at 6/16 of the fp32-MFMA cost.  Same tiling / gather / epilogues as conv_gather_mfma_f32.
// LDS: 3 planes x [rows][32 + 8] bf16 per operand (80-B pitch: conflict-free ds_read_b128), single stage,
// register prefetch of the next K-tile under the MFMAs.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// pack the upper halves (= truncated bf16) of two fp32 words: result = {hi16(x1), hi16(x0)}
__device__ __forceinline__ unsigned pack_hi16(unsigned x1, unsigned x0) { return __builtin_amdgcn_perm(x1, x0, 0x07060302u); }

// round-to-nearest-even bf16 of 4 floats, packed (plain-bf16 mode: one plane, one MFMA per product)
__device__ __forceinline__ unsigned rn_bf16_bits(float x) {
    const unsigned u = __float_as_uint(x);
    return u + 0x7fffu + ((u >> 16) & 1u);            // upper half = RN bf16 (NaN payloads aside)
}
__device__ __forceinline__ u32x2 round1x4(const f32x4 v) {
    u32x2 p;
    p[0] = pack_hi16(rn_bf16_bits(v[1]), rn_bf16_bits(v[0]));
    p[1] = pack_hi16(rn_bf16_bits(v[3]), rn_bf16_bits(v[2]));
    return p;
}

__device__ __forceinline__ void split3x4(const f32x4 v, u32x2& p1, u32x2& p2, u32x2& p3) {
    unsigned h1[4], h2[4], h3[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h1[e] = __float_as_uint(v[e]);
        const float r1 = v[e] - __uint_as_float(h1[e] & 0xffff0000u);     // exact
        h2[e] = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(h2[e] & 0xffff0000u);       // exact, <= 8 significant bits left
        h3[e] = __float_as_uint(r2);
    }
    p1[0] = pack_hi16(h1[1], h1[0]); p1[1] = pack_hi16(h1[3], h1[2]);
    p2[0] = pack_hi16(h2[1], h2[0]); p2[1] = pack_hi16(h2[3], h2[2]);
    p3[0] = pack_hi16(h3[1], h3[0]); p3[1] = pack_hi16(h3[3], h3[2]);
}

template <int BM, int BN, int WM, int WN, bool TRANSPOSED, int NP = 3>
__global__ __launch_bounds__(256, 2) void conv_gather_bf16x3(const ConvGeom g) {
    constexpr int BK = 32;
    constexpr int PITCH_B = 80;                        // bytes per LDS row (32 bf16 + 16 B pad)
    constexpr int PA = BM / 32;
    constexpr int NB = (BN * 4 + 255) / 256;           // 16-B chunks of the B tile per thread and plane
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    static_assert(WM * WN == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    unsigned char* As = smem_b;                        // [NP][BM][80 B]
    unsigned char* Bs = smem_b + NP * BM * PITCH_B;    // [NP][BN][80 B]
    int* row_m = reinterpret_cast<int*>(smem_b + NP * (BM + BN) * PITCH_B);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    int r0 = 0, q0 = 0, tstep = 1;
    int ph = 0, pw = 0, mt_local, nt;
    const bool classed = TRANSPOSED && g.stride == 2;
    if (!classed) {
        const int lt = y4_xcd_remap(blockIdx.x, g.mtiles * g.ntiles);
        mt_local = lt / g.ntiles;
        nt = lt - mt_local * g.ntiles;
    } else {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        int c = 0;
        while (c < 3 && slot >= g.cls_slot0[c + 1]) ++c;
        const int per = g.cls_slot0[c + 1] - g.cls_slot0[c];
        const int t = xcd * per + (slot - g.cls_slot0[c]);
        const int tiles_c = (g.cls_tile0[c + 1] - g.cls_tile0[c]) * g.ntiles;
        if (t >= tiles_c) return;
        mt_local = t / g.ntiles;
        nt = t - mt_local * g.ntiles;
        ph = (3 - c) >> 1; pw = (3 - c) & 1;
        r0 = (ph + g.pad) & 1; q0 = (pw + g.pad) & 1; tstep = 2;
    }
    const int n0 = nt * BN;
    const int nr = (g.k - r0 + tstep - 1) / tstep, nq = (g.k - q0 + tstep - 1) / tstep;
    // rows are dealt so that the two rows one 16-lane ds_write_b64 group covers lie 4 apart (320 B = bank +16):
    // with adjacent rows (80 B) the groups overlapped on 4 banks -- SQ_LDS_BANK_CONFLICT was 8 % of the wave cycles
    const int t3 = tid >> 3, kc = tid & 7;
    const int lrow = (t3 & 0x18) | ((t3 & 1) << 2) | ((t3 >> 1) & 3);

    // 32-bit buffer offsets are relative to the first image this tile touches (tensors may exceed 4 GiB)
    const int pix_per_img = classed ? g.cls_h[ph] * g.cls_w[pw] : g.Hd * g.Wd;
    const int b_first = (int)(((long long)mt_local * BM) / pix_per_img);
    const unsigned long long img_bytes = (unsigned long long)g.Hs * g.Ws * (unsigned long long)g.lds_ * 4ull;
    const unsigned long long src_skip = (unsigned long long)b_first * img_bytes;
    const unsigned long long src_left = g.src_total_bytes > src_skip ? g.src_total_bytes - src_skip : 0ull;
    const __amdgpu_buffer_rsrc_t src_rsrc =
        y4_make_rsrc(reinterpret_cast<const char*>(g.src) + src_skip, (unsigned)(src_left < 0xfffffff0ull ? src_left : 0xfffffff0ull));
    const __amdgpu_buffer_rsrc_t wt_rsrc = y4_make_rsrc(g.wt_planes, g.wt_bytes);     // 3 planes of N*K bf16
    const unsigned OOB = 0xffffffffu;
    unsigned a_base[PA];
    int a_h[PA], a_w[PA];
    bool a_ok[PA];
    const unsigned pix_bytes = (unsigned)g.lds_ * 4u;
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const int row = p * 32 + lrow;
        const int i = mt_local * BM + row;
        int b, hd, wd;
        if (!classed) {
            a_ok[p] = i < g.M;
            const int ii = a_ok[p] ? i : 0;
            b = ii / (g.Hd * g.Wd);
            const int rem = ii - b * (g.Hd * g.Wd);
            hd = rem / g.Wd; wd = rem - hd * g.Wd;
        } else {
            const int hc = g.cls_h[ph], wc = g.cls_w[pw];
            a_ok[p] = i < g.B * hc * wc;
            const int ii = a_ok[p] ? i : 0;
            b = ii / (hc * wc);
            const int rem = ii - b * (hc * wc);
            const int hh = rem / wc;
            hd = 2 * hh + ph; wd = 2 * (rem - hh * wc) + pw;
        }
        if (kc == 0) row_m[row] = a_ok[p] ? (b * g.Hd + hd) * g.Wd + wd : -1;
        a_base[p] = (unsigned)(b - b_first) * (unsigned)(g.Hs * g.Ws) * pix_bytes + kc * 16u;
        if (!TRANSPOSED) { a_h[p] = hd * g.stride - g.pad; a_w[p] = wd * g.stride - g.pad; }
        else { a_h[p] = hd + g.pad; a_w[p] = wd + g.pad; }
    }
    // B chunks: slot = tid + 256*i -> (row = slot/4, 16-B chunk = slot%4) of each plane
    unsigned b_off[NB];
    int b_lds[NB];
    const unsigned plane_bytes = (unsigned)g.N * (unsigned)g.K * 2u;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int slot = tid + 256 * i;
        const int s2 = slot >> 2, ch = slot & 3;
        const int row = (s2 & ~7) | ((s2 & 1) << 2) | ((s2 >> 1) & 3);     // same dealing for the 8-lane ds_write_b128 groups
        const bool ok = row < BN && (n0 + row) < g.N;
        b_off[i] = ok ? (unsigned)(n0 + row) * (unsigned)g.K * 2u + ch * 16u : OOB;
        b_lds[i] = row < BN ? row * PITCH_B + ch * 16 : -1;
    }

    f32x4 ra[PA];
    u32x4 rb[NB][NP];
    const int CC = g.Cs / BK;
    int r = r0, q = q0, cc = 0;
    // per-tap A offsets: the halo test and the pixel address are recomputed only when the filter tap
    // changes (every CC K-tiles); inside a tap the K position is a scalar offset of the buffer load
    unsigned a_off[PA];
    auto tap_setup = [&]() {
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            int hi, wi;
            bool ok = a_ok[p];
            if (!TRANSPOSED) {
                hi = a_h[p] + r; wi = a_w[p] + q;
                ok = ok && (unsigned)hi < (unsigned)g.Hs && (unsigned)wi < (unsigned)g.Ws;
            } else {
                const int th = a_h[p] - r, tw = a_w[p] - q;
                hi = th; wi = tw;
                if (g.stride == 2) { hi = th >> 1; wi = tw >> 1; }
                ok = ok && th >= 0 && tw >= 0 && hi < g.Hs && wi < g.Ws;
            }
            a_off[p] = ok ? a_base[p] + (unsigned)(hi * g.Ws + wi) * pix_bytes : OOB;
        }
    };
    tap_setup();
    auto load_tile = [&]() {
#pragma unroll
        for (int p = 0; p < PA; ++p) ra[p] = y4_buf_load4(src_rsrc, a_off[p], (unsigned)(cc * BK) * 4u);
        const unsigned koff = (unsigned)((r * g.k + q) * CC + cc) * (BK * 2u);
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                rb[i][pl] = __builtin_amdgcn_raw_buffer_load_b128(wt_rsrc, (int)b_off[i], (int)(koff + pl * plane_bytes), 0);
        if (++cc == CC) { cc = 0; q += tstep; if (q >= g.k) { q = q0; r += tstep; } tap_setup(); }
    };
    int st_cc = 0;
    auto store_tile = [&]() {
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            f32x4 v = ra[p];
            if (TRANSPOSED) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (st_cc * BK + kc * 4 + e >= g.Cs_valid) v[e] = 0.f;
            }
            unsigned char* d = As + (p * 32 + lrow) * PITCH_B + kc * 8;
            if constexpr (NP == 3) {
                u32x2 p1, p2, p3;
                split3x4(v, p1, p2, p3);
                *reinterpret_cast<u32x2*>(d) = p1;
                *reinterpret_cast<u32x2*>(d + BM * PITCH_B) = p2;
                *reinterpret_cast<u32x2*>(d + 2 * BM * PITCH_B) = p3;
            } else {
                *reinterpret_cast<u32x2*>(d) = round1x4(v);
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i)
            if (b_lds[i] >= 0) {
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    *reinterpret_cast<u32x4*>(Bs + pl * BN * PITCH_B + b_lds[i]) = rb[i][pl];
            }
        if (++st_cc == CC) st_cc = 0;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int KT = nr * nq * CC;
    load_tile();
    store_tile();
    __syncthreads();

    const int fr = lane & 31, fh = lane >> 5;
    const unsigned char* a_frag = As + (wm * WTM + fr) * PITCH_B + fh * 16;
    const unsigned char* b_frag = Bs + (wn * WTN + fr) * PITCH_B + fh * 16;

    for (int kt = 0; kt < KT; ++kt) {
        if (kt + 1 < KT) load_tile();                  // next tile's loads fly under this tile's MFMAs
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 fa[TM][NP], fb[TN][NP];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    fa[i][pl] = *reinterpret_cast<const bf16x8*>(a_frag + pl * BM * PITCH_B + i * 32 * PITCH_B + ks * 32);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    fb[j][pl] = *reinterpret_cast<const bf16x8*>(b_frag + pl * BN * PITCH_B + j * 32 * PITCH_B + ks * 32);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    f32x16 c = acc[i][j];
                    if constexpr (NP == 3) {                  // small terms first
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], c, 0, 0, 0);
                    }
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], c, 0, 0, 0);
                    acc[i][j] = c;
                }
        }
        __syncthreads();                               // every wave is done reading this stage
        if (kt + 1 < KT) store_tile();                 // split + write the prefetched tile
        __syncthreads();
    }

#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WTN + j * 32 + fr;
        const bool nok = n < g.N;
        const float sc = (g.scale && nok) ? g.scale[n] : 1.0f;
        const float sh = (g.shift && nok) ? g.shift[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rbase = wm * WTM + i * 32 + 4 * fh;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = row_m[rbase + (e & 3) + 8 * (e >> 2)];
                if (nok && m >= 0) {
                    float v = acc[i][j][e] * sc + sh;
                    v = y4_act(v, g.act);
                    if (g.res) v += g.res[(long long)m * g.ldr + n];
                    g.dst[(long long)m * g.ldd + n] = v;
                }
            }
        }
    }
    if (!TRANSPOSED && g.stats) {
        float* red = reinterpret_cast<float*>(smem_b);     // [WM][BN][2]; the K loop ended with a barrier
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float cs = 0.f, css = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) { const float v = acc[i][j][e]; cs += v; css += v * v; }
            cs += __shfl_xor(cs, 32, 64);
            css += __shfl_xor(css, 32, 64);
            if (fh == 0) {
                const int c = wn * WTN + j * 32 + fr;
                red[(wm * BN + c) * 2 + 0] = cs;
                red[(wm * BN + c) * 2 + 1] = css;
            }
        }
        __syncthreads();
        for (int c = tid; c < BN; c += 256) {
            float cs = 0.f, css = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) { cs += red[(w * BN + c) * 2]; css += red[(w * BN + c) * 2 + 1]; }
            const int n = n0 + c;
            if (n < g.N) {
                g.stats[((long long)mt_local * 2 + 0) * g.N + n] = cs;
                g.stats[((long long)mt_local * 2 + 1) * g.N + n] = css;
            }
        }
    }
}

// fp32 filter [N][K] -> three bf16 planes [3][N][K] (exact truncation split, as split3x4)
__global__ void split_filter_kernel(const float* __restrict__ w, unsigned short* __restrict__ planes, long long n, int np) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = w[i];
        if (np == 1) { planes[i] = (unsigned short)(rn_bf16_bits(v) >> 16); continue; }
        const unsigned h1 = __float_as_uint(v);
        const float r1 = v - __uint_as_float(h1 & 0xffff0000u);
        const unsigned h2 = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(h2 & 0xffff0000u);
        planes[i] = (unsigned short)(h1 >> 16);
        planes[n + i] = (unsigned short)(h2 >> 16);
        planes[2 * n + i] = (unsigned short)(__float_as_uint(r2) >> 16);
    }
}

// [Cout][k][k][Cin] -> [Cin][k][k][Cout4] (zero padded to a multiple of 32 output channels)
__global__ void transpose_filter_kernel(const float* __restrict__ w, float* __restrict__ wt,
                                        int Cout, int Cin, int kk, int Cout_pad) {
    const long long total = (long long)Cin * kk * Cout_pad;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i % Cout_pad);
        const long long t = i / Cout_pad;
        const int tap = (int)(t % kk);
        const int c = (int)(t / kk);
        wt[i] = n < Cout ? w[((long long)n * kk + tap) * Cin + c] : 0.0f;
    }
}

// [Cout][k][k][Cin] fp32 -> 3 bf16 planes of [Cin][k][k][Cout_pad] (dgrad filter for the bf16x3 kernels)
__global__ void transpose_split_filter_kernel(const float* __restrict__ w, unsigned short* __restrict__ planes,
                                              int Cout, int Cin, int kk, int Cout_pad, int np) {
    const long long total = (long long)Cin * kk * Cout_pad;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i % Cout_pad);
        const long long t = i / Cout_pad;
        const int tap = (int)(t % kk);
        const int c = (int)(t / kk);
        const float v = n < Cout ? w[((long long)n * kk + tap) * Cin + c] : 0.0f;
        if (np == 1) { planes[i] = (unsigned short)(rn_bf16_bits(v) >> 16); continue; }
        const unsigned h1 = __float_as_uint(v);
        const float r1 = v - __uint_as_float(h1 & 0xffff0000u);
        const unsigned h2 = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(h2 & 0xffff0000u);
        planes[i] = (unsigned short)(h1 >> 16);
        planes[total + i] = (unsigned short)(h2 >> 16);
        planes[2 * total + i] = (unsigned short)(__float_as_uint(r2) >> 16);
    }
}

thread_local char g_last_kernel[160] = "";          // see y4::note_kernel
}
namespace y4 {
void note_kernel(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_kernel, sizeof(g_last_kernel), fmt, ap);
    va_end(ap);
}
}
namespace {
template <int BM, int BN, int WM, int WN, bool TR, int BKT = 32, int NP = 0>
int launch_gather(const ConvGeom& g0, hipStream_t st) {
    constexpr bool SPLIT = NP > 0;
    ConvGeom g = g0;
    if (TR && g.stride == 2) {
        int t = 0;
        for (int c = 0; c < 4; ++c) {
            const int ph = (3 - c) >> 1, pw = (3 - c) & 1;
            g.cls_h[ph] = (g.Hd + 1 - ph) / 2;
            g.cls_w[pw] = (g.Wd + 1 - pw) / 2;
            g.cls_tile0[c] = t;
            const long long n = (long long)g.B * g.cls_h[ph] * g.cls_w[pw];
            t += (int)((n + BM - 1) / BM);
        }
        g.cls_tile0[4] = t;
        g.mtiles = t;
    } else {
        g.cls_slot0[0] = 0;
        g.mtiles = (g.M + BM - 1) / BM;
    }
    g.ntiles = (g.N + BN - 1) / BN;
    {
        const unsigned long long img = (unsigned long long)g.Hs * g.Ws * (unsigned long long)g.lds_ * 4ull;
        const unsigned long long wb = (unsigned long long)g.N * g.K * 4ull;
        // a tile's window spans the images of BM consecutive output pixels (+1): must fit 32-bit offsets
        const unsigned long long imgs_per_tile = (unsigned long long)BM / (unsigned long long)((g.Hd * g.Wd + 3) / 4 > 0 ? (g.Hd * g.Wd + 3) / 4 : 1) + 2;
        if (img * imgs_per_tile >= 0xfffffff0ull || wb >= 0xfffffff0ull) return Y4_ERR_SHAPE;
        g.src_total_bytes = (unsigned long long)g.B * img; g.wt_bytes = (unsigned)wb;
        if (SPLIT) {
            if (!g.wt_planes) return Y4_ERR_WORKSPACE;
            g.wt_bytes = (unsigned)((unsigned long long)g.N * g.K * 2ull * NP);
        }
    }
    const size_t smem = SPLIT ? (size_t)NP * (BM + BN) * 80 + BM * sizeof(int)
                              : 2ull * (BM + BN) * (BKT + 4) * sizeof(float) + BM * sizeof(int);
    void (*kern)(const ConvGeom);
    if constexpr (SPLIT) kern = conv_gather_bf16x3<BM, BN, WM, WN, TR, (NP > 0 ? NP : 3)>;
    else kern = conv_gather_mfma_f32<BM, BN, WM, WN, TR, BKT>;
    static Y4DynLds lds_attr;                              // per device, see common.h
    if (!lds_attr.ensure(reinterpret_cast<const void*>(kern), smem)) return Y4_ERR_LAUNCH;
    int grid = g.mtiles * g.ntiles;
    if (TR && g.stride == 2) {
        int sl = 0;
        for (int c = 0; c < 4; ++c) {
            g.cls_slot0[c] = sl;
            sl += ((g.cls_tile0[c + 1] - g.cls_tile0[c]) * g.ntiles + 7) / 8;
        }
        g.cls_slot0[4] = sl;
        grid = sl * 8;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, st, g);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

int g_conv_mode = 3;          // 0: fp32 MFMA (exact fma chain), 1: split-bf16 x3 (fp32-grade, 6 bf16 MFMAs),
                              // 2: plain bf16 operands (RN), fp32 accumulate -- mixed precision, BASELINE config 5
                              // 3: split-fp16 x2 (fp32-grade, 3 fp16 MFMAs, per-tensor power-of-two scale; conv_f16x2.hip) -- default

// ---------------------------------------------------------------- streaming 1x1 kernel (small K, small N)
// The 1x1 layers on the 304^2 / 152^2 maps are HBM-bound (K, N <= 128: < 64 flop per byte); the tile machinery
// above spends their time in prologues, barriers and LDS round trips.  Here the whole filter (3 bf16 planes)
// stays in LDS for the life of a persistent block, and every wave streams its own 32 pixel rows from global
// memory straight into the MFMA A-operand layout (lane = pixel, 8 consecutive channels = 32 contiguous bytes),
// splits them in registers and never meets a barrier; the next tile's loads fly under this tile's MFMAs/stores.
// Used for forward (filter planes [N][K]) and for 1x1 dgrad (planes of the transposed filter).
// NW waves per block: 4, or 8 where the filter planes allow one block per CU only (K = N = 128: 104 KB) so that every
// SIMD still holds two waves
template <int KS, int NT, int NW = 4>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void conv1x1_stream_bf16x3(const ConvGeom g) {
    constexpr int NTHR = NW * 64, TROWS = NW * 32;
    constexpr int K = KS * 16;
    constexpr int PITCH = K * 2 + 16;                    // LDS row pitch: conflict-free ds_read_b128 for K = 32/64/128
    constexpr int N32 = NT * 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    {
        constexpr int CPR = K / 8;                       // 16-B chunks per filter row
        const unsigned char* wp = reinterpret_cast<const unsigned char*>(g.wt_planes);
        for (int i = tid; i < 3 * N32 * CPR; i += NTHR) {
            const int pl = i / (N32 * CPR);
            const int rem = i - pl * (N32 * CPR);
            const int row = rem / CPR, ch = rem - row * CPR;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (row < g.N) v = *reinterpret_cast<const u32x4*>(wp + ((size_t)pl * g.N + row) * (K * 2) + ch * 16);
            *reinterpret_cast<u32x4*>(smem_b + (pl * N32 + row) * PITCH + ch * 16) = v;
        }
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t src_rsrc = y4_make_rsrc(g.src, (unsigned)g.src_total_bytes);   // < 4 GiB (host)
    const unsigned pix_bytes = (unsigned)g.lds_ * 4u;
    const int mtiles = g.mtiles;
    // KH k-steps per register set: K <= 64 keeps a whole tile per set (two tiles in flight); K = 128 keeps half a tile
    // per set, the two halves of one tile alternating (the other half's loads fly under this half's MFMAs)
    constexpr int KH = KS == 8 ? 4 : KS;
    f32x4 ra0[KH][2], ra1[KH][2];
    auto load = [&](f32x4 (&ra)[KH][2], int tile, int ks0) {
        const int m = tile * TROWS + wave * 32 + fr;
        const unsigned off = m < g.M ? (unsigned)m * pix_bytes + (unsigned)fh * 32u : 0xffffffffu;
#pragma unroll
        for (int ks = 0; ks < KH; ++ks) {
            ra[ks][0] = y4_buf_load4(src_rsrc, off, (unsigned)(ks0 + ks) * 64u);
            ra[ks][1] = y4_buf_load4(src_rsrc, off, (unsigned)(ks0 + ks) * 64u + 16u);
        }
    };
    float cs[NT], css[NT];
    float sc[NT], sh[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        cs[j] = 0.f; css[j] = 0.f;
        const int n = j * 32 + fr;
        sc[j] = (g.scale && n < g.N) ? g.scale[n] : 1.0f;
        sh[j] = (g.shift && n < g.N) ? g.shift[n] : 0.0f;
    }
    const unsigned char* b_frag = smem_b + fr * PITCH + fh * 16;
    f32x16 acc[NT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    };
    auto mma = [&](f32x4 (&ra)[KH][2], int ks0) {
        // the filter fragments are loop-invariant; at K = 128 keeping them all in registers (up to 384) spills, so the
        // compiler is told the LDS may have changed and re-reads them per tile (ds_read_b128 is cheap beside 6 MFMAs)
        if constexpr (KS == 8) asm volatile("" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < KH; ++ks) {
            u32x2 a1, a2, a3, b1, b2, b3;
            split3x4(ra[ks][0], a1, a2, a3);
            split3x4(ra[ks][1], b1, b2, b3);
            const u32x4 q1 = {a1[0], a1[1], b1[0], b1[1]}, q2 = {a2[0], a2[1], b2[0], b2[1]}, q3 = {a3[0], a3[1], b3[0], b3[1]};
            bf16x8 fa[3];
            fa[0] = __builtin_bit_cast(bf16x8, q1); fa[1] = __builtin_bit_cast(bf16x8, q2); fa[2] = __builtin_bit_cast(bf16x8, q3);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                bf16x8 fb[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    fb[pl] = *reinterpret_cast<const bf16x8*>(b_frag + (pl * N32 + j * 32) * PITCH + (ks0 + ks) * 32);
                f32x16 c = acc[j];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2], fb[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[0], c, 0, 0, 0);
                acc[j] = c;
            }
        }
    };
    auto epilogue = [&](int tile) {
        const int mbase = tile * TROWS + wave * 32 + 4 * fh;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = j * 32 + fr;
            const bool nok = n < g.N;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float raw = acc[j][e];
                cs[j] += raw; css[j] += raw * raw;           // rows past M are exact zeros
                const int m = mbase + (e & 3) + 8 * (e >> 2);
                if (nok && m < g.M) {
                    float v = raw * sc[j] + sh[j];
                    v = y4_act(v, g.act);
                    if (g.res) v += g.res[(long long)m * g.ldr + n];
                    g.dst[(long long)m * g.ldd + n] = v;
                }
            }
        }
    };
    int tile = blockIdx.x;
    if (tile < mtiles) load(ra0, tile, 0);
    if constexpr (KS == 8) {
        while (tile < mtiles) {
            load(ra1, tile, KH);
            zero_acc();
            mma(ra0, 0);
            const int tn = tile + gridDim.x;
            if (tn < mtiles) load(ra0, tn, 0);
            mma(ra1, KH);
            epilogue(tile);
            tile = tn;
        }
    } else {
        while (tile < mtiles) {
            const int t1 = tile + gridDim.x;
            if (t1 < mtiles) load(ra1, t1, 0);
            zero_acc(); mma(ra0, 0); epilogue(tile);
            if (t1 >= mtiles) break;
            const int t2 = t1 + gridDim.x;
            if (t2 < mtiles) load(ra0, t2, 0);
            zero_acc(); mma(ra1, 0); epilogue(t1);
            tile = t2;
        }
    }
    if (g.stats) {                                        // one partial row per block: [gridDim][2][N]
        __syncthreads();                                  // every wave is done with the filter planes
        float* red = reinterpret_cast<float*>(smem_b);    // [NW][N32][2]
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float a = cs[j], b = css[j];
            a += __shfl_xor(a, 32, 64);
            b += __shfl_xor(b, 32, 64);
            if (fh == 0) { red[(wave * N32 + j * 32 + fr) * 2] = a; red[(wave * N32 + j * 32 + fr) * 2 + 1] = b; }
        }
        __syncthreads();
        for (int c = tid; c < N32; c += NTHR) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) { a += red[(w * N32 + c) * 2]; b += red[(w * N32 + c) * 2 + 1]; }
            if (c < g.N) {
                g.stats[((long long)blockIdx.x * 2 + 0) * g.N + c] = a;
                g.stats[((long long)blockIdx.x * 2 + 1) * g.N + c] = b;
            }
        }
    }
}

template <int KS, int NT, int NW = 4>
int launch_stream1x1(const ConvGeom& g0, hipStream_t st, int* nparts) {
    ConvGeom g = g0;
    g.mtiles = (g.M + NW * 32 - 1) / (NW * 32);
    g.ntiles = 1;
    g.src_total_bytes = (unsigned long long)g.M * (unsigned long long)g.lds_ * 4ull;
    const size_t smem = (size_t)3 * NT * 32 * (KS * 32 + 16);
    auto kern = conv1x1_stream_bf16x3<KS, NT, NW>;
    static Y4DynLds lds_attr;                              // per device, see common.h
    if (!lds_attr.ensure(reinterpret_cast<const void*>(kern), smem)) return Y4_ERR_LAUNCH;
    const int resident = smem > 80 * 1024 ? 256 : 512;    // blocks per CU by LDS: 1 (K = N = 128) or 2
    const int grid = g.mtiles < resident ? g.mtiles : resident;
    if (nparts) *nparts = grid;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), smem, st, g);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

// eligibility: bf16x3 arithmetic, 1x1 stride 1, K in {32,64,128}, N <= 128, filter planes <= 52 KB of LDS,
// 32-bit addressable source, every source channel valid, and enough rows to be worth a persistent launch
static bool stream1x1_ok(const ConvGeom& g) {
    if (g_conv_mode != 1 || g.k != 1 || g.stride != 1 || !g.wt_planes) return false;
    if (g.Cs != 32 && g.Cs != 64 && g.Cs != 128) return false;
    if (g.Cs_valid != g.Cs || g.N > 128) return false;
    const int n32 = (g.N + 31) / 32 * 32;
    if (g.Cs * n32 > (g.Cs == 128 ? 16384 : 8192)) return false;     // filter planes: <= 52 KB (2 blocks/CU) or 104 KB at K = 128
    if ((unsigned long long)g.M * (unsigned long long)g.lds_ * 4ull >= 0xfffffff0ull) return false;
    return g.M >= 128 * 1024;
}

static int dispatch_stream1x1(const ConvGeom& g, hipStream_t st, int* nparts) {
    const int nt = (g.N + 31) / 32;
    switch (g.Cs / 16 * 10 + nt) {
        case 21: return launch_stream1x1<2, 1>(g, st, nparts);
        case 22: return launch_stream1x1<2, 2>(g, st, nparts);
        case 23: case 24: return launch_stream1x1<2, 4>(g, st, nparts);
        case 41: return launch_stream1x1<4, 1>(g, st, nparts);
        case 42: return launch_stream1x1<4, 2>(g, st, nparts);
        case 43: case 44: return launch_stream1x1<4, 4>(g, st, nparts);
        case 81: return launch_stream1x1<8, 1>(g, st, nparts);
        case 82: return launch_stream1x1<8, 2>(g, st, nparts);
        case 83: case 84: return launch_stream1x1<8, 4, 8>(g, st, nparts);
        default: return Y4_ERR_SHAPE;
    }
}


template <bool TR>
int dispatch_gather(const ConvGeom& g, hipStream_t st, int* nparts = nullptr) {
    // *nparts: number of BN-statistics partial rows the launch writes (M tiles, or blocks of the streaming kernel)
    if (nparts) *nparts = (g.M + 127) / 128;
    if (stream1x1_ok(g)) return dispatch_stream1x1(g, st, nparts);
    if (g_conv_mode == 1 || g_conv_mode == 2) {
        const bool one = g_conv_mode == 2;
        if (g.N > 64) {
            const long long nt = (g.N + 127) / 128;
            const long long b128 = ((long long)g.M + 127) / 128 * nt, b64 = ((long long)g.M + 63) / 64 * nt;
            const double c128 = (double)((b128 + 511) / 512) * 128.0;
            const double c64 = (double)((b64 + 511) / 512) * 64.0 * 1.10;
            if (c64 < c128 && !(TR && g.stride == 2)) {
                if (nparts) *nparts = (g.M + 63) / 64;
                return one ? launch_gather<64, 128, 2, 2, TR, 32, 1>(g, st) : launch_gather<64, 128, 2, 2, TR, 32, 3>(g, st);
            }
            return one ? launch_gather<128, 128, 2, 2, TR, 32, 1>(g, st) : launch_gather<128, 128, 2, 2, TR, 32, 3>(g, st);
        }
        if (g.N > 32) return one ? launch_gather<128, 64, 2, 2, TR, 32, 1>(g, st) : launch_gather<128, 64, 2, 2, TR, 32, 3>(g, st);
        return one ? launch_gather<128, 32, 4, 1, TR, 32, 1>(g, st) : launch_gather<128, 32, 4, 1, TR, 32, 3>(g, st);
    }
    if (g.N > 64) {
        // LDS allows 2 resident blocks per CU at BK = 32 (74 KB) and 3 at BK = 16 (41 KB).  1x1 convs have
        // short K loops (4..32 tiles), so prologue/epilogue time matters: the third block covers it
        // (+5..10 % measured); on the long 3x3 loops the two depths tie and BK = 32 halves the barriers.
        const bool short_k = g.k == 1;
        const long long slots = short_k ? 768 : 512;
        // When 128-row tiles fill the last round of resident blocks badly (e.g. 724 blocks = 1.41 rounds
        // at 19x19 maps), 64-row tiles (+8 % per-flop cost) win.
        const long long nt = (g.N + 127) / 128;
        const long long b128 = ((long long)g.M + 127) / 128 * nt, b64 = ((long long)g.M + 63) / 64 * nt;
        const double c128 = (double)((b128 + slots - 1) / slots) * 128.0;
        const double c64 = (double)((b64 + slots - 1) / slots) * 64.0 * 1.08;
        if (c64 < c128 && !(TR && g.stride == 2)) {
            if (nparts) *nparts = (g.M + 63) / 64;
            return short_k ? launch_gather<64, 128, 2, 2, TR, 16>(g, st) : launch_gather<64, 128, 2, 2, TR>(g, st);
        }
        return short_k ? launch_gather<128, 128, 2, 2, TR, 16>(g, st) : launch_gather<128, 128, 2, 2, TR>(g, st);
    }
    if (g.k == 1) {      // HBM-bound 1x1 layers at 304^2 / 152^2: more resident blocks = more loads in flight
        if (g.N > 32) return launch_gather<128, 64, 2, 2, TR, 16>(g, st);
    }
    if (g.N > 32) return launch_gather<128, 64, 2, 2, TR>(g, st);
    return launch_gather<128, 32, 4, 1, TR>(g, st);
}

// ------------------------------------------------------------------------------------ wgrad

// D[n][j] = sum_p dy[p][n] * xg[p][j].  Block tile TN_ x TJ_ (64 or 128 each), 4 waves as 2x2,
// K-chunks of 32 pixels, both operands kept pixel-major in LDS exactly as they lie in memory
// (rows of TN_/TJ_ floats): a half-wave's ds_read_b32 covers 32 consecutive floats, conflict-free.
template <int TN_, int TJ_>
__global__ __launch_bounds__(256, 2) void conv_wgrad_mfma_f32(const WgradGeom g) {
    constexpr int MI = TN_ / 64, MJ = TJ_ / 64;           // 32x32 MFMA tiles per wave
    constexpr int LPR_A = TN_ / 4, RPP_A = 256 / LPR_A, PA = 32 / RPP_A;
    constexpr int LPR_B = TJ_ / 4, RPP_B = 256 / LPR_B, PB = 32 / RPP_B;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                           // [2][32][TN_]  dy tile   (pixel-major)
    float* Bs = smem + 2 * 32 * TN_;            // [2][32][TJ_]  x  tile   (pixel-major)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const int tiles = g.ntn * g.ntj;
    int bid = y4_xcd_remap(blockIdx.x, tiles * g.splits);   // tiles of one pixel range share an XCD's L2
    const int split = bid / tiles;
    bid -= split * tiles;
    const int tn = bid / g.ntj, tj = bid - tn * g.ntj;
    const int n0 = tn * TN_, j0 = tj * TJ_;

    const int arow = tid / LPR_A, ac4 = (tid % LPR_A) * 4;
    const int brow = tid / LPR_B, bc4 = (tid % LPR_B) * 4;
    const int Cout4 = (g.Cout + 3) & ~3;
    const bool an_ok = (n0 + ac4) < Cout4;
    const int j = j0 + bc4;
    const bool bj_ok = j < g.J;
    int jr = 0, jq = 0, jc = 0;
    if (bj_ok) { const int tap = j / g.Cin; jc = j - tap * g.Cin; jr = tap / g.k; jq = tap - jr * g.k; }

    const int chunk0 = split * g.chunks_per_split;
    int nchunks = (g.M + 31) / 32 - chunk0;
    if (nchunks > g.chunks_per_split) nchunks = g.chunks_per_split;

    // running (b, ho, wo) of this thread's B rows: advanced by 32 pixels per chunk, no divisions
    // 32-bit buffer windows start at this block's first pixel (dy) / first image (x)
    const long long p_first = (long long)chunk0 * 32;
    const int b_first = (int)(p_first / ((long long)g.Ho * g.Wo));
    const unsigned long long x_skip = (unsigned long long)b_first * g.H * g.W * (unsigned long long)g.ldx * 4ull;
    const unsigned long long dy_skip = (unsigned long long)p_first * (unsigned long long)g.lddy * 4ull;
    const unsigned long long x_left = g.x_total_bytes > x_skip ? g.x_total_bytes - x_skip : 0ull;
    const unsigned long long dy_left = g.dy_total_bytes > dy_skip ? g.dy_total_bytes - dy_skip : 0ull;
    const __amdgpu_buffer_rsrc_t x_rsrc =
        y4_make_rsrc(reinterpret_cast<const char*>(g.x) + x_skip, (unsigned)(x_left < 0xfffffff0ull ? x_left : 0xfffffff0ull));
    const __amdgpu_buffer_rsrc_t dy_rsrc =
        y4_make_rsrc(reinterpret_cast<const char*>(g.dy) + dy_skip, (unsigned)(dy_left < 0xfffffff0ull ? dy_left : 0xfffffff0ull));
    const unsigned OOB = 0xffffffffu;
    int pb_b[PB], pb_h[PB], pb_w[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        const int pix = chunk0 * 32 + p * RPP_B + brow;
        const int pp = pix < g.M ? pix : (int)p_first;
        const int bb = pp / (g.Ho * g.Wo);
        pb_b[p] = bb - b_first;                            // image index relative to the block's window
        const int rem = pp - bb * (g.Ho * g.Wo);
        pb_h[p] = rem / g.Wo;
        pb_w[p] = rem - pb_h[p] * g.Wo;
    }
    unsigned a_off[PA];
#pragma unroll
    for (int p = 0; p < PA; ++p)
        a_off[p] = an_ok ? (unsigned)(p * RPP_A + arow) * (unsigned)g.lddy * 4u + (unsigned)(n0 + ac4) * 4u : OOB;
    const unsigned chunk_bytes = 32u * (unsigned)g.lddy * 4u;
    const unsigned x_pix_bytes = (unsigned)g.ldx * 4u;

    f32x4 ra[PA], rb[PB];
    int ld_chunk = 0;
    auto load_chunk = [&]() {
        const int pbase = (chunk0 + ld_chunk) * 32;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const bool ok = pbase + p * RPP_A + arow < g.M;
            ra[p] = y4_buf_load4(dy_rsrc, ok ? a_off[p] : OOB, (unsigned)ld_chunk * chunk_bytes);
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const int pix = pbase + p * RPP_B + brow;
            const int hi = pb_h[p] * g.stride - g.pad + jr, wi = pb_w[p] * g.stride - g.pad + jq;
            const bool ok = pix < g.M && bj_ok && (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W;
            const unsigned off = (unsigned)((pb_b[p] * g.H + hi) * g.W + wi) * x_pix_bytes + (unsigned)jc * 4u;
            rb[p] = y4_buf_load4(x_rsrc, ok ? off : OOB, 0u);
            pb_w[p] += 32;
            while (pb_w[p] >= g.Wo) { pb_w[p] -= g.Wo; if (++pb_h[p] == g.Ho) { pb_h[p] = 0; ++pb_b[p]; } }
        }
        ++ld_chunk;
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int p = 0; p < PA; ++p)
            *reinterpret_cast<f32x4*>(As + (buf * 32 + p * RPP_A + arow) * TN_ + ac4) = ra[p];
#pragma unroll
        for (int p = 0; p < PB; ++p)
            *reinterpret_cast<f32x4*>(Bs + (buf * 32 + p * RPP_B + brow) * TJ_ + bc4) = rb[p];
    };

    f32x16 acc[MI][MJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jj = 0; jj < MJ; ++jj)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jj][e] = 0.f;

    const int fr = lane & 31, fh = lane >> 5;
    if (nchunks > 0) {
        load_chunk();
        store_chunk(0);
        __syncthreads();
        for (int ch = 0; ch < nchunks; ++ch) {
            const int cur = ch & 1;
            if (ch + 1 < nchunks) load_chunk();
            const float* as = As + (cur * 32 + fh) * TN_ + wm * (TN_ / 2) + fr;
            const float* bs = Bs + (cur * 32 + fh) * TJ_ + wn * (TJ_ / 2) + fr;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                float a[MI], b[MJ];
#pragma unroll
                for (int i = 0; i < MI; ++i) a[i] = as[(2 * t) * TN_ + i * 32];
#pragma unroll
                for (int jj = 0; jj < MJ; ++jj) b[jj] = bs[(2 * t) * TJ_ + jj * 32];
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int jj = 0; jj < MJ; ++jj)
                        acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[jj], acc[i][jj], 0, 0, 0);
            }
            if (ch + 1 < nchunks) store_chunk(cur ^ 1);
            __syncthreads();
        }
    }
    // D[n][j]: row index (n) on the registers, column (j) on the lane
    float* out = g.out + (long long)split * g.Cout * g.J;
#pragma unroll
    for (int jj = 0; jj < MJ; ++jj) {
        const int jcol = j0 + wn * (TJ_ / 2) + jj * 32 + fr;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int nb = n0 + wm * (TN_ / 2) + i * 32 + 4 * fh;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = nb + (e & 3) + 8 * (e >> 2);
                if (n < g.Cout && jcol < g.J) out[(long long)n * g.J + jcol] = acc[i][jj][e];
            }
        }
    }
}

// Split-bf16 ("bf16x3") wgrad: same math as conv_wgrad_mfma_f32, operands split exactly into three bf16
// planes while being transposed into LDS as [n or j][32 pixels] rows (80-B pitch), so the fragment reads
// are the plain ds_read_b128 of the gather kernel.  Each thread owns one 4-pixel x 4-channel block of
// each operand per 32-pixel chunk (4 coalesced 16-B loads, a 4x4 register transpose folded into the
// bf16 packing, 12 ds_write_b64).
template <int TN_, int TJ_, int NP = 3>
__global__ __launch_bounds__(256, 2) void conv_wgrad_bf16x3(const WgradGeom g) {
    constexpr int MI = TN_ / 64, MJ = TJ_ / 64;
    constexpr int PITCH_B = 80;
    constexpr int NBLK_A = 8 * (TN_ / 4), NBLK_B = 8 * (TJ_ / 4);     // 4x4 blocks per chunk (<= 256)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    unsigned char* As = smem_b;                          // [NP][TN_][80 B]
    unsigned char* Bs = smem_b + NP * TN_ * PITCH_B;     // [NP][TJ_][80 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const int tiles = g.ntn * g.ntj;
    int bid = y4_xcd_remap(blockIdx.x, tiles * g.splits);
    const int split = bid / tiles;
    bid -= split * tiles;
    const int tn = bid / g.ntj, tj = bid - tn * g.ntj;
    const int n0 = tn * TN_, j0 = tj * TJ_;

    const int pg = tid & 7, cg = tid >> 3;               // pixel group (4 px), channel group (4 ch)
    const bool a_act = tid < NBLK_A, b_act = tid < NBLK_B;
    const int Cout4 = (g.Cout + 3) & ~3;
    const bool an_ok = a_act && (n0 + cg * 4) < Cout4;
    const int j = j0 + cg * 4;
    const bool bj_ok = b_act && j < g.J;
    int jr = 0, jq = 0, jc = 0;
    if (bj_ok) { const int tap = j / g.Cin; jc = j - tap * g.Cin; jr = tap / g.k; jq = tap - jr * g.k; }

    const int chunk0 = split * g.chunks_per_split;
    int nchunks = (g.M + 31) / 32 - chunk0;
    if (nchunks > g.chunks_per_split) nchunks = g.chunks_per_split;

    // 32-bit buffer windows start at this block's first pixel (dy) / first image (x)
    const long long p_first = (long long)chunk0 * 32;
    const int b_first = (int)(p_first / ((long long)g.Ho * g.Wo));
    const unsigned long long x_skip = (unsigned long long)b_first * g.H * g.W * (unsigned long long)g.ldx * 4ull;
    const unsigned long long dy_skip = (unsigned long long)p_first * (unsigned long long)g.lddy * 4ull;
    const unsigned long long x_left = g.x_total_bytes > x_skip ? g.x_total_bytes - x_skip : 0ull;
    const unsigned long long dy_left = g.dy_total_bytes > dy_skip ? g.dy_total_bytes - dy_skip : 0ull;
    const __amdgpu_buffer_rsrc_t x_rsrc =
        y4_make_rsrc(reinterpret_cast<const char*>(g.x) + x_skip, (unsigned)(x_left < 0xfffffff0ull ? x_left : 0xfffffff0ull));
    const __amdgpu_buffer_rsrc_t dy_rsrc =
        y4_make_rsrc(reinterpret_cast<const char*>(g.dy) + dy_skip, (unsigned)(dy_left < 0xfffffff0ull ? dy_left : 0xfffffff0ull));
    const unsigned OOB = 0xffffffffu;
    int pb_b[4], pb_h[4], pb_w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pix = chunk0 * 32 + pg * 4 + i;
        const int pp = pix < g.M ? pix : (int)p_first;
        const int bb = pp / (g.Ho * g.Wo);
        pb_b[i] = bb - b_first;
        const int rem = pp - bb * (g.Ho * g.Wo);
        pb_h[i] = rem / g.Wo;
        pb_w[i] = rem - pb_h[i] * g.Wo;
    }
    const unsigned a_off0 = an_ok ? (unsigned)(pg * 4) * (unsigned)g.lddy * 4u + (unsigned)(n0 + cg * 4) * 4u : OOB;
    const unsigned dy_pix_bytes = (unsigned)g.lddy * 4u;
    const unsigned chunk_bytes = 32u * dy_pix_bytes;
    const unsigned x_pix_bytes = (unsigned)g.ldx * 4u;

    f32x4 ra[4], rb[4];
    int ld_chunk = 0;
    auto load_chunk = [&]() {
        const int pbase = (chunk0 + ld_chunk) * 32 + pg * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = an_ok && pbase + i < g.M;
            ra[i] = y4_buf_load4(dy_rsrc, ok ? a_off0 + (unsigned)i * dy_pix_bytes : OOB, (unsigned)ld_chunk * chunk_bytes);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int hi = pb_h[i] * g.stride - g.pad + jr, wi = pb_w[i] * g.stride - g.pad + jq;
            const bool ok = bj_ok && pbase + i < g.M && (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W;
            const unsigned off = (unsigned)((pb_b[i] * g.H + hi) * g.W + wi) * x_pix_bytes + (unsigned)jc * 4u;
            rb[i] = y4_buf_load4(x_rsrc, ok ? off : OOB, 0u);
            pb_w[i] += 32;
            while (pb_w[i] >= g.Wo) { pb_w[i] -= g.Wo; if (++pb_h[i] == g.Ho) { pb_h[i] = 0; ++pb_b[i]; } }
        }
        ++ld_chunk;
    };
    // split 4 pixels x 4 channels and write the 4 channel rows (3 planes each) transposed
    auto split_store = [&](const f32x4 (&v)[4], unsigned char* base, int rows) {
        if constexpr (NP == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                u32x2 p;
                p[0] = pack_hi16(rn_bf16_bits(v[1][e]), rn_bf16_bits(v[0][e]));
                p[1] = pack_hi16(rn_bf16_bits(v[3][e]), rn_bf16_bits(v[2][e]));
                *reinterpret_cast<u32x2*>(base + (cg * 4 + e) * PITCH_B + pg * 8) = p;
            }
            return;
        }
        unsigned h1[4][4], h2[4][4], h3[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                h1[i][e] = __float_as_uint(v[i][e]);
                const float r1 = v[i][e] - __uint_as_float(h1[i][e] & 0xffff0000u);
                h2[i][e] = __float_as_uint(r1);
                const float r2 = r1 - __uint_as_float(h2[i][e] & 0xffff0000u);
                h3[i][e] = __float_as_uint(r2);
            }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsigned char* d = base + (cg * 4 + e) * PITCH_B + pg * 8;
            u32x2 p;
            p[0] = pack_hi16(h1[1][e], h1[0][e]); p[1] = pack_hi16(h1[3][e], h1[2][e]);
            *reinterpret_cast<u32x2*>(d) = p;
            p[0] = pack_hi16(h2[1][e], h2[0][e]); p[1] = pack_hi16(h2[3][e], h2[2][e]);
            *reinterpret_cast<u32x2*>(d + rows * PITCH_B) = p;
            p[0] = pack_hi16(h3[1][e], h3[0][e]); p[1] = pack_hi16(h3[3][e], h3[2][e]);
            *reinterpret_cast<u32x2*>(d + 2 * rows * PITCH_B) = p;
        }
    };
    auto store_chunk = [&]() {
        if (a_act) split_store(ra, As, TN_);
        if (b_act) split_store(rb, Bs, TJ_);
    };

    f32x16 acc[MI][MJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jj = 0; jj < MJ; ++jj)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jj][e] = 0.f;

    const int fr = lane & 31, fh = lane >> 5;
    const unsigned char* a_frag = As + (wm * (TN_ / 2) + fr) * PITCH_B + fh * 16;
    const unsigned char* b_frag = Bs + (wn * (TJ_ / 2) + fr) * PITCH_B + fh * 16;
    if (nchunks > 0) {
        load_chunk();
        store_chunk();
        __syncthreads();
        for (int ch = 0; ch < nchunks; ++ch) {
            if (ch + 1 < nchunks) load_chunk();
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 fa[MI][NP], fb[MJ][NP];
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl)
                        fa[i][pl] = *reinterpret_cast<const bf16x8*>(a_frag + pl * TN_ * PITCH_B + i * 32 * PITCH_B + ks * 32);
#pragma unroll
                for (int jj = 0; jj < MJ; ++jj)
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl)
                        fb[jj][pl] = *reinterpret_cast<const bf16x8*>(b_frag + pl * TJ_ * PITCH_B + jj * 32 * PITCH_B + ks * 32);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int jj = 0; jj < MJ; ++jj) {
                        f32x16 c = acc[i][jj];
                        if constexpr (NP == 3) {
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[jj][0], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[jj][1], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[jj][2], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[jj][0], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[jj][1], c, 0, 0, 0);
                        }
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[jj][0], c, 0, 0, 0);
                        acc[i][jj] = c;
                    }
            }
            __syncthreads();
            if (ch + 1 < nchunks) store_chunk();
            __syncthreads();
        }
    }
    float* out = g.out + (long long)split * g.Cout * g.J;
#pragma unroll
    for (int jj = 0; jj < MJ; ++jj) {
        const int jcol = j0 + wn * (TJ_ / 2) + jj * 32 + fr;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int nb = n0 + wm * (TN_ / 2) + i * 32 + 4 * fh;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = nb + (e & 3) + 8 * (e >> 2);
                if (n < g.Cout && jcol < g.J) out[(long long)n * g.J + jcol] = acc[i][jj][e];
            }
        }
    }
}

// vector variant: a block owns VEC float4 columns; its 256/VEC thread groups each add every (256/VEC)-th slab
// (4 independent loads in flight), then the groups are combined through LDS in group order -- a fixed
// summation tree for a given split count, hence deterministic, and hundreds of slabs no longer serialise
// behind one thread's load latency.
template <int VEC>
__global__ __launch_bounds__(256) void slab_reduce_vec_kernel(const f32x4* __restrict__ slabs, f32x4* __restrict__ out,
                                                              long long nvec, int splits) {
    constexpr int SG = 256 / VEC;
    __shared__ f32x4 red[SG][VEC];
    const int v = threadIdx.x % VEC, sg = threadIdx.x / VEC;
    const long long i = (long long)blockIdx.x * VEC + v;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    if (i < nvec) {
        int k = sg;
        for (; k + 3 * SG < splits; k += 4 * SG) {
            const f32x4 a = slabs[(long long)k * nvec + i], b = slabs[(long long)(k + SG) * nvec + i];
            const f32x4 c = slabs[(long long)(k + 2 * SG) * nvec + i], d = slabs[(long long)(k + 3 * SG) * nvec + i];
            s0 += a; s1 += b; s2 += c; s3 += d;
        }
        for (; k < splits; k += SG) s0 += slabs[(long long)k * nvec + i];
    }
    red[sg][v] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sg == 0 && i < nvec) {
        f32x4 t = red[0][v];
#pragma unroll
        for (int g = 1; g < SG; ++g) t += red[g][v];
        out[i] = t;
    }
}

__global__ void slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out,
                                   long long n, int splits) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < splits; ++k) s += slabs[(long long)k * n + i];   // fixed order
        out[i] = s;
    }
}

void wgrad_plan(int B, int H, int W, int Cin, int Cout, int k, int stride, WgradGeom& g) {
    const int pad = (k - 1) / 2;
    g.B = B; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.k = k; g.stride = stride; g.pad = pad;
    g.Ho = (H + 2 * pad - k) / stride + 1;
    g.Wo = (W + 2 * pad - k) / stride + 1;
    g.M = B * g.Ho * g.Wo;
    g.J = k * k * Cin;
    g.tn = Cout <= 64 ? 64 : 128;
    g.tj = g.J <= 64 ? 64 : 128;
    g.ntn = (Cout + g.tn - 1) / g.tn;
    g.ntj = (g.J + g.tj - 1) / g.tj;
    const int tiles = g.ntn * g.ntj;
    const int chunks = (g.M + 31) / 32;
    // resident blocks: LDS = 2*32*(tn+tj)*4 B per block, 160 KiB per CU, <= 8 (wave slots at 4 waves/block)
    int per_cu = (160 * 1024) / (2 * 32 * (g.tn + g.tj) * 4);
    if (per_cu > 4) per_cu = 4;
    if (g.tn + g.tj == 256 && per_cu > 2) per_cu = 2;      // 128x128: 124 VGPRs + 64 KiB
    const int slots = 256 * per_cu;
    // choose the split count in [1, chunks/16] that fills whole rounds of `slots` blocks best
    int max_s = chunks / 16;
    if (max_s < 1) max_s = 1;
    if (max_s > 4096) max_s = 4096;
    int best_s = 1;
    double best_eff = -1.0;
    for (int sp = 1; sp <= max_s; ++sp) {
        const long long blocks = (long long)tiles * sp;
        const long long rounds = (blocks + slots - 1) / slots;
        double eff = (double)blocks / (double)(rounds * slots);
        if (rounds > 6) eff = 1.0;                           // enough rounds: tail is amortised
        if (eff > best_eff + 0.02) { best_eff = eff; best_s = sp; }
        if (blocks >= 2ll * slots && eff >= 0.93) break;     // good enough: fewer slabs to write and fold
        if (blocks >= 6ll * slots) break;
    }
    { static const char* fs = getenv("Y4_WGRAD_SPLITS"); if (fs) { best_s = atoi(fs); if (best_s > max_s) best_s = max_s; if (best_s < 1) best_s = 1; } }
    g.chunks_per_split = (chunks + best_s - 1) / best_s;
    // a block's pixel range (+ 2 images of slack) must fit a 32-bit buffer window; planned for pitches up to 2x the
    // channel count (channel slices of concat buffers), verified against the real pitches at call time
    const unsigned long long per_px = 8ull * (unsigned long long)((Cout + 3) / 4 * 4 > Cin * stride * stride ? (Cout + 3) / 4 * 4 : Cin * stride * stride);
    const unsigned long long slack = 2ull * g.Ho * g.Wo;
    const unsigned long long max_px = 0xf0000000ull / per_px;
    if (max_px > slack + 32 && (unsigned long long)g.chunks_per_split * 32ull + slack > max_px)
        g.chunks_per_split = (int)((max_px - slack) / 32ull);
    g.splits = (chunks + g.chunks_per_split - 1) / g.chunks_per_split;
}

template <int TN_, int TJ_, int NP = 0>
int launch_wgrad(const WgradGeom& g, hipStream_t st) {
    constexpr bool SPLIT = NP > 0;
    const size_t smem = SPLIT ? (size_t)NP * (TN_ + TJ_) * 80 : 2ull * 32 * (TN_ + TJ_) * sizeof(float);
    void (*kern)(const WgradGeom);
    if constexpr (SPLIT) kern = conv_wgrad_bf16x3<TN_, TJ_, (NP > 0 ? NP : 3)>;
    else kern = conv_wgrad_mfma_f32<TN_, TJ_>;
    static Y4DynLds lds_attr;                              // per device, see common.h
    if (!lds_attr.ensure(reinterpret_cast<const void*>(kern), smem)) return Y4_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(g.ntn * g.ntj * g.splits), dim3(256), smem, st, g);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

// ------------------------------------------------------------------------------------ stem
// Cin = 3, k = 3, stride 1, pad 1 (yolo/model/yolov4.py:30).  One thread per output pixel, all
// Cout (<= 32) channels in registers, filter taps come in through the scalar path (uniform
// addresses); the 128-B pixel rows are transposed through LDS so stores are lane-contiguous.
struct StemGeom {
    const float* x; const float* w; float* y; const float* scale; const float* shift;
    float* stats;                 // optional [blocks][2][Cout] column sums of the (raw) output
    long long sxb, sxc, sxh, sxw, ldy;
    int B, H, W, Cout, act;
    long long M;
};

__global__ __launch_bounds__(256) void conv_stem_fwd_kernel(const StemGeom g) {
    __shared__ float tile[4][64][33];
    __shared__ __attribute__((aligned(16))) float wl[27 * 32];          // filter, [tap*3+c][n] (n contiguous)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 27 * 32; i += 256) {
        const int n = i & 31, kk = i >> 5;
        wl[i] = n < g.Cout ? g.w[n * 27 + kk] : 0.f;
    }
    __syncthreads();
    const long long pix0 = (long long)blockIdx.x * 256 + wave * 64;
    const long long pix = pix0 + lane;
    float acc[32];
#pragma unroll
    for (int n = 0; n < 32; ++n) acc[n] = 0.f;
    if (pix < g.M) {
        const int b = (int)(pix / ((long long)g.H * g.W));
        const int rem = (int)(pix - (long long)b * g.H * g.W);
        const int h = rem / g.W, w = rem - h * g.W;
        const float* xb = g.x + b * g.sxb;
        float xv[27];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int hi = h + r - 1, wi = w + q - 1;
                const bool ok = (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W;
#pragma unroll
                for (int c = 0; c < 3; ++c) xv[(r * 3 + q) * 3 + c] = ok ? xb[c * g.sxc + hi * g.sxh + wi * g.sxw] : 0.f;
            }
        // taps one at a time (sched_barrier keeps the 216 LDS broadcasts from being hoisted into 864 registers)
#pragma unroll
        for (int kk = 0; kk < 27; ++kk) {
            const f32x4* wp = reinterpret_cast<const f32x4*>(wl + kk * 32);
            const float v = xv[kk];
#pragma unroll
            for (int n4 = 0; n4 < 8; ++n4) {
                const f32x4 wv = wp[n4];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[n4 * 4 + e] = fmaf(v, wv[e], acc[n4 * 4 + e]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int n = 0; n < 32; ++n) {
        float v = acc[n];
        if (n < g.Cout) {
            v = v * (g.scale ? g.scale[n] : 1.f) + (g.shift ? g.shift[n] : 0.f);
            v = y4_act(v, g.act);
        }
        tile[wave][lane][n] = v;
    }
    __syncthreads();
    // 64 pixels x 32 ch: lane -> (pixel = it*2 + lane/32, ch = lane%32): 128-B contiguous per half wave
    const int ch = lane & 31;
    if (ch < g.Cout) {
#pragma unroll 4
        for (int it = 0; it < 32; ++it) {
            const int pl = it * 2 + (lane >> 5);
            const long long p = pix0 + pl;
            if (p < g.M) g.y[p * g.ldy + ch] = tile[wave][pl][ch];
        }
    }
    if (g.stats) {                                   // uniform branch: column sums of this block's 256 pixels
        __shared__ float sred[8][32][2];
        const int grp = tid >> 5;                    // 8 groups of 32 pixels
        float cs = 0.f, css = 0.f;
        for (int i = 0; i < 32; ++i) { const float v = tile[grp >> 1][(grp & 1) * 32 + i][ch]; cs += v; css += v * v; }
        sred[grp][ch][0] = cs; sred[grp][ch][1] = css;
        __syncthreads();
        if (tid < 64) {
            const int c = tid & 31, which = tid >> 5;
            float t = 0.f;
            for (int k = 0; k < 8; ++k) t += sred[k][c][which];
            if (c < g.Cout) g.stats[((long long)blockIdx.x * 2 + which) * g.Cout + c] = t;
        }
    }
}

// Stem forward on the matrix cores (bf16x3 mode): the 27-value patch of a pixel is the A row (K = 27 -> 32,
// ordered k = r*9 + c*3 + q so that the three q of one (r, c) are neighbours in memory), gathered straight from
// the strided input into the MFMA A layout (lane = pixel, 2 x 8 k values), split in registers; the filter is
// 24 VGPRs of B fragments per wave.  12 MFMAs per 32 pixels x 32 channels; the C layout (lane = channel) gives
// 128-byte contiguous stores.  Persistent blocks walk groups of 256 pixels and leave one BN-statistics row
// per group (same contract as the VALU kernel above).
__global__ __launch_bounds__(256) void conv_stem_fwd_bf16x3_kernel(const StemGeom g, const int ngroups, const float inv_hw,
                                                                   const float inv_w) {
    __shared__ float sred[4][32][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    const int HW = g.H * g.W;
    bf16x8 fb[2][3];
    int koff[16];          // element offset of k relative to the pixel, per lane
    int ktap[16];          // r*3+q (bit index into the 9-bit validity mask), 9 = padding k
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k = (i >> 3) * 16 + fh * 8 + (i & 7);
        const int r = k / 9, c = (k - r * 9) / 3, q = k - r * 9 - c * 3;
        const bool kv = k < 27;
        koff[i] = kv ? (int)(c * g.sxc + (r - 1) * g.sxh + (q - 1) * g.sxw) * 4 : 0;
        ktap[i] = kv ? r * 3 + q : 9;
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        f32x4 w0, w1;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = ks * 16 + fh * 8 + i;
            const int r = k / 9, c = (k - r * 9) / 3, q = k - r * 9 - c * 3;
            const float v = (k < 27 && fr < g.Cout) ? g.w[fr * 27 + (r * 3 + q) * 3 + c] : 0.f;
            if (i < 4) w0[i] = v; else w1[i - 4] = v;
        }
        u32x2 a1, a2, a3, b1, b2, b3;
        split3x4(w0, a1, a2, a3);
        split3x4(w1, b1, b2, b3);
        const u32x4 q1 = {a1[0], a1[1], b1[0], b1[1]}, q2 = {a2[0], a2[1], b2[0], b2[1]}, q3 = {a3[0], a3[1], b3[0], b3[1]};
        fb[ks][0] = __builtin_bit_cast(bf16x8, q1); fb[ks][1] = __builtin_bit_cast(bf16x8, q2); fb[ks][2] = __builtin_bit_cast(bf16x8, q3);
    }
    const float sc = (g.scale && fr < g.Cout) ? g.scale[fr] : 1.f;
    const float sh = (g.shift && fr < g.Cout) ? g.shift[fr] : 0.f;
    const __amdgpu_buffer_rsrc_t rsrc = y4_make_rsrc(g.x, 0xfffffff0u);      // extent checked on the host
    float xv0[16], xv1[16];
    auto load = [&](float (&xv)[16], long long tile) {
        const long long p = tile * 32 + fr;
        unsigned base = 0xffffffffu;
        unsigned m9 = 0;
        if (p < g.M) {
            int b = (int)((float)p * inv_hw);
            int rem = (int)(p - (long long)b * HW);
            if (rem < 0) { --b; rem += HW; } else if (rem >= HW) { ++b; rem -= HW; }
            int h = (int)((float)rem * inv_w);
            int w = rem - h * g.W;
            if (w < 0) { --h; w += g.W; } else if (w >= g.W) { ++h; w -= g.W; }
            base = (unsigned)(b * g.sxb + h * g.sxh + w * g.sxw) * 4u;
            const unsigned hm = (h > 0 ? 1u : 0u) | 2u | (h + 1 < g.H ? 4u : 0u);
            const unsigned wm = (w > 0 ? 1u : 0u) | 2u | (w + 1 < g.W ? 4u : 0u);
            m9 = ((hm & 1u) ? wm : 0u) | ((hm & 2u) ? wm << 3 : 0u) | ((hm & 4u) ? wm << 6 : 0u);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const bool ok = (m9 >> ktap[i]) & 1u;
            xv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, ok ? (int)(base + (unsigned)koff[i]) : -1, 0, 0));
        }
    };
    float cs = 0.f, css = 0.f;
    auto compute = [&](float (&xv)[16], long long tile) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const f32x4 v0 = {xv[ks * 8 + 0], xv[ks * 8 + 1], xv[ks * 8 + 2], xv[ks * 8 + 3]};
            const f32x4 v1 = {xv[ks * 8 + 4], xv[ks * 8 + 5], xv[ks * 8 + 6], xv[ks * 8 + 7]};
            u32x2 a1, a2, a3, b1, b2, b3;
            split3x4(v0, a1, a2, a3);
            split3x4(v1, b1, b2, b3);
            const u32x4 q1 = {a1[0], a1[1], b1[0], b1[1]}, q2 = {a2[0], a2[1], b2[0], b2[1]}, q3 = {a3[0], a3[1], b3[0], b3[1]};
            const bf16x8 f1 = __builtin_bit_cast(bf16x8, q1), f2 = __builtin_bit_cast(bf16x8, q2), f3 = __builtin_bit_cast(bf16x8, q3);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f3, fb[ks][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f2, fb[ks][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, fb[ks][2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f2, fb[ks][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, fb[ks][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, fb[ks][0], acc, 0, 0, 0);
        }
        const long long mbase = tile * 32 + 4 * fh;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float raw = acc[e];
            cs += raw; css += raw * raw;                  // pixels past M contribute exact zeros
            const long long m = mbase + (e & 3) + 8 * (e >> 2);
            if (fr < g.Cout && m < g.M) g.y[m * g.ldy + fr] = y4_act(raw * sc + sh, g.act);
        }
    };
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const long long t0 = (long long)grp * 8 + wave * 2;
        load(xv0, t0);
        load(xv1, t0 + 1);
        cs = 0.f; css = 0.f;
        compute(xv0, t0);
        compute(xv1, t0 + 1);
        if (g.stats) {
            cs += __shfl_xor(cs, 32, 64);
            css += __shfl_xor(css, 32, 64);
            if (fh == 0) { sred[wave][fr][0] = cs; sred[wave][fr][1] = css; }
            __syncthreads();
            if (tid < 64) {
                const int c = tid & 31, which = tid >> 5;
                const float t = (sred[0][c][which] + sred[1][c][which]) + (sred[2][c][which] + sred[3][c][which]);
                if (c < g.Cout) g.stats[((long long)grp * 2 + which) * g.Cout + c] = t;
            }
            __syncthreads();
        }
    }
}

// wgrad of the stem: D[n][j] (32 x 27->32) = sum_p dy[p][n] * x[p + tap(j)][c(j)], operands
// straight from global memory into the MFMA (a dy pixel row IS the 32-float A fragment).
struct StemWgradGeom {
    const float* x; const float* dy; float* slabs;
    long long sxb, sxc, sxh, sxw, lddy;
    int B, H, W, Cout;
    long long M, pix_per_wave;
};

__global__ __launch_bounds__(256) void conv_stem_wgrad_kernel(const StemWgradGeom g) {
    const int tid = threadIdx.x, lane = tid & 63;
    const long long wave_id = (long long)blockIdx.x * 4 + (tid >> 6);
    const long long p_begin = wave_id * g.pix_per_wave;
    long long p_end = p_begin + g.pix_per_wave;
    if (p_end > g.M) p_end = g.M;
    const int fr = lane & 31, fh = lane >> 5;
    const bool jok = fr < 27;
    const int tap = jok ? fr / 3 : 0, c = jok ? fr - tap * 3 : 0;
    const int r = tap / 3 - 1, q = tap - (tap / 3) * 3 - 1;
    const bool nok = fr < g.Cout;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    // running (b, h, w) of this lane's pixel p = p_begin + fh + 2*iter: no divisions in the loop
    long long p = p_begin + fh;
    int b = 0, h = 0, w = 0;
    if (p < g.M) {
        b = (int)(p / ((long long)g.H * g.W));
        const int rem = (int)(p - (long long)b * g.H * g.W);
        h = rem / g.W; w = rem - h * g.W;
    }
    // 8 pixel pairs per trip: 16 independent buffer loads (out-of-range / halo lanes read zeros through the
    // descriptor's bounds check) are in flight before the 8 MFMAs consume them -- the loop is latency-bound otherwise
    const long long dy_first = p_begin < g.M ? p_begin : 0;
    const __amdgpu_buffer_rsrc_t dy_rsrc = y4_make_rsrc(g.dy + dy_first * g.lddy, (unsigned)(g.pix_per_wave * g.lddy * 4));
    const __amdgpu_buffer_rsrc_t x_rsrc = y4_make_rsrc(g.x, 0xfffffff0u);                     // extent checked on the host
    const unsigned dy_step = (unsigned)g.lddy * 8u;                                            // 2 pixels
    unsigned dy_off = ((unsigned)fh * (unsigned)g.lddy + (unsigned)fr) * 4u;
    const long long coff = c * g.sxc;
    for (long long pc = p_begin; pc < p_end; pc += 16) {
        float a[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool pok = p < p_end;
            const int hi = h + r, wi = w + q;
            const bool xok = pok && jok && (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W;
            const unsigned xo = (unsigned)(b * g.sxb + coff + hi * g.sxh + wi * g.sxw) * 4u;
            a[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dy_rsrc, (pok && nok) ? (int)dy_off : -1, 0, 0));
            bv[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, xok ? (int)xo : -1, 0, 0));
            dy_off += dy_step;
            p += 2;
            w += 2;
            if (w >= g.W) { w -= g.W; if (++h == g.H) { h = 0; ++b; } }  // W >= 2
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], bv[u], acc, 0, 0, 0);
    }
    float* out = g.slabs + wave_id * 1024;     // [32 n][32 j]
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int n = (e & 3) + 8 * (e >> 2) + 4 * fh;
        out[n * 32 + fr] = acc[e];
    }
}

// Two coalesced stages over the per-wave slabs [nslabs][32 n][32 j]: 64 blocks each fold nslabs/64 slabs for all
// 1024 entries (thread <-> 4 consecutive entries, fp64), then one block folds the 64 partial rows in a fixed order.
__global__ __launch_bounds__(256) void stem_wgrad_reduce1_kernel(const float* __restrict__ slabs, double* __restrict__ part,
                                                                 int nslabs) {
    const int per = (nslabs + gridDim.x - 1) / gridDim.x;
    const int k0 = blockIdx.x * per;
    const int k1 = min(nslabs, k0 + per);
    double s[4] = {0, 0, 0, 0};
    for (int k = k0; k < k1; ++k) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(slabs + (long long)k * 1024 + threadIdx.x * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += (double)v[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) part[(long long)blockIdx.x * 1024 + threadIdx.x * 4 + e] = s[e];
}
__global__ __launch_bounds__(256) void stem_wgrad_reduce2_kernel(const double* __restrict__ part, float* __restrict__ dw,
                                                                 int nparts, int Cout) {
    for (int i = threadIdx.x; i < Cout * 27; i += 256) {
        const int n = i / 27, j = i - n * 27;
        double s = 0.0;
        for (int k = 0; k < nparts; ++k) s += part[(long long)k * 1024 + n * 32 + j];
        dw[i] = (float)s;
    }
}

constexpr int STEM_WAVES = 4096;

}  // namespace

namespace y4 {
int slab_reduce(const float* slabs, float* dw, long long n, int splits, hipStream_t st) {
    if ((n & 3) == 0 && !(reinterpret_cast<uintptr_t>(dw) & 15) && !(reinterpret_cast<uintptr_t>(slabs) & 15)) {
        const long long nvec = n / 4;
        if (nvec >= 64 * 1024)
            hipLaunchKernelGGL(slab_reduce_vec_kernel<64>, dim3((unsigned)((nvec + 63) / 64)), dim3(256), 0, st,
                               reinterpret_cast<const f32x4*>(slabs), reinterpret_cast<f32x4*>(dw), nvec, splits);
        else
            hipLaunchKernelGGL(slab_reduce_vec_kernel<16>, dim3((unsigned)((nvec + 15) / 16)), dim3(256), 0, st,
                               reinterpret_cast<const f32x4*>(slabs), reinterpret_cast<f32x4*>(dw), nvec, splits);
    } else {
        const int blocks = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, st, slabs, dw, n, splits);
    }
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}
}  // namespace y4

// ======================================================================================== C ABI
extern "C" {

int y4_set_conv_mode(int mode) {
    if (mode < 0 || mode > 3) return Y4_ERR_SHAPE;
    g_conv_mode = mode;
    return Y4_OK;
}
int y4_get_conv_mode(void) { return g_conv_mode; }

// Per-call workspace of the forward conv (modes 1-3): [64 B of amax words: [0] filter, [1] gathered tensor when the caller
// gave none][4 KiB of per-block filter maxima][filter planes].  Nothing is shared between calls, so convs may be issued
// from any number of streams / devices / threads at once.
constexpr size_t FWD_WS_HDR = 64 + 4096;

size_t y4_conv2d_fwd_workspace(int Cin, int Cout, int k) {
    if (Cin <= 0 || Cout <= 0 || k <= 0) return 0;
    return FWD_WS_HDR + (size_t)Cout * k * k * Cin * 6;
}

static int conv_fwd_impl(const float* x, int ldx, const float* w, float* y, int ldy,
                         int B, int H, int W, int Cin, int Cout, int k, int stride,
                         const float* scale, const float* shift, int act,
                         const float* residual, int ldr, float* stats, int* nparts, const unsigned* x_amax,
                         unsigned* y_amax, void* workspace, size_t workspace_bytes, void* stream,
                         void* w_prepared = nullptr, void* dgrad_filter = nullptr, size_t dgrad_filter_bytes = 0) {
    if (!x || (!w && !w_prepared) || !y) return Y4_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || (k != 1 && k != 3) || (stride != 1 && stride != 2))
        return Y4_ERR_SHAPE;
    if (Cin <= 0 || Cin % BK != 0 || ldx < Cin || ldy < Cout || (ldx & 3) || (residual && ldr < Cout))
        return Y4_ERR_SHAPE;
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(w) & 15)) return Y4_ERR_SHAPE;
    const int pad = (k - 1) / 2;
    ConvGeom g{};
    g.src = x; g.wt = w; g.dst = y; g.scale = scale; g.shift = shift; g.res = residual; g.stats = stats;
    g.lds_ = ldx; g.ldd = ldy; g.ldr = ldr;
    g.B = B; g.Hs = H; g.Ws = W; g.Cs = Cin; g.Cs_valid = Cin;
    g.Hd = (H + 2 * pad - k) / stride + 1;
    g.Wd = (W + 2 * pad - k) / stride + 1;
    g.N = Cout; g.k = k; g.stride = stride; g.pad = pad;
    const long long M = (long long)B * g.Hd * g.Wd;
    if (M >= (1ll << 31)) return Y4_ERR_SHAPE;
    g.M = (int)M; g.K = k * k * Cin; g.act = act;
    if (w_prepared) {
        // filter already split (y4_conv2d_prepare_filter_f32): [word 0: max|w| bits][word 1: scratch][.. 64 B][4 KiB][planes]
        if (g_conv_mode != 3) return Y4_ERR_SHAPE;
        if (reinterpret_cast<uintptr_t>(w_prepared) & 15) return Y4_ERR_SHAPE;
        unsigned* hdr = static_cast<unsigned*>(w_prepared);
        g.wt_planes = reinterpret_cast<unsigned short*>(static_cast<char*>(w_prepared) + 64 + 4096);
        if (!x_amax) {
            const int rc = y4::amax_launch(x, ldx, (long long)B * H * W, Cin, hdr + 1, y4_stream(stream));
            if (rc != Y4_OK) return rc;
            x_amax = hdr + 1;
        }
        g.src_amax = x_amax; g.wt_amax = hdr; g.dst_amax = y_amax;
        return y4::f16x2_gather(g, false, y4_stream(stream), nparts);
    }
    if (g_conv_mode != 0) {
        const long long nw = (long long)Cout * g.K;
        if (!workspace) return Y4_ERR_NULL;
        if (workspace_bytes < y4_conv2d_fwd_workspace(Cin, Cout, k)) return Y4_ERR_WORKSPACE;
        if (reinterpret_cast<uintptr_t>(workspace) & 15) return Y4_ERR_SHAPE;
        unsigned* hdr = static_cast<unsigned*>(workspace);
        unsigned short* planes = reinterpret_cast<unsigned short*>(static_cast<char*>(workspace) + FWD_WS_HDR);
        g.wt_planes = planes;
        if (g_conv_mode == 3) {
            int rc;
            if (dgrad_filter) {
                // the backward pass of this layer will want the transposed planes of the same filter: both in one launch,
                // into a buffer laid out as y4_conv2d_dgrad_f32's workspace (which that call then takes with w == NULL)
                if (dgrad_filter_bytes < y4_conv2d_dgrad_workspace(Cin, Cout, k)) return Y4_ERR_WORKSPACE;
                if (reinterpret_cast<uintptr_t>(dgrad_filter) & 15) return Y4_ERR_SHAPE;
                const int cp = (Cout + 31) / 32 * 32;
                unsigned* hdr_t = reinterpret_cast<unsigned*>(static_cast<char*>(dgrad_filter) + (size_t)Cin * k * k * cp * 6);
                // (the 2-D tile kernel runs dgrad in forward form: it wants the taps mirrored; same test as in conv_dgrad_impl)
                rc = y4::f16x2_filter_planes_dual(w, planes, hdr, hdr + 16, static_cast<unsigned short*>(dgrad_filter), hdr_t, Cout, Cin,
                                                  k * k, cp, y4::tile_conv_ok(cp, Cout, Cin, k, stride, H, W), y4_stream(stream));
            } else {
                rc = y4::f16x2_filter_planes(w, planes, Cout, g.K, hdr, hdr + 16, y4_stream(stream));
            }
            if (rc != Y4_OK) return rc;
            if (!x_amax) {                                  // no producer-side maximum: one extra pass over the input
                rc = y4::amax_launch(x, ldx, (long long)B * H * W, Cin, hdr + 1, y4_stream(stream));
                if (rc != Y4_OK) return rc;
                x_amax = hdr + 1;
            }
            g.src_amax = x_amax; g.wt_amax = hdr; g.dst_amax = y_amax;
            return y4::f16x2_gather(g, false, y4_stream(stream), nparts);
        }
        const int blocks = (int)((nw + 255) / 256 > 4096 ? 4096 : (nw + 255) / 256);
        hipLaunchKernelGGL(split_filter_kernel, dim3(blocks), dim3(256), 0, y4_stream(stream), w, planes, nw,
                           g_conv_mode == 2 ? 1 : 3);
        Y4_CHECK_LAUNCH();
    }
    return dispatch_gather<false>(g, y4_stream(stream), nparts);
}

int y4_conv2d_fwd_f32(const float* x, int ldx, const float* w, float* y, int ldy,
                      int B, int H, int W, int Cin, int Cout, int k, int stride,
                      const float* scale, const float* shift, int act,
                      const float* residual, int ldr, const unsigned* x_amax, unsigned* y_amax,
                      void* workspace, size_t workspace_bytes, void* stream) {
    return conv_fwd_impl(x, ldx, w, y, ldy, B, H, W, Cin, Cout, k, stride, scale, shift, act, residual, ldr, nullptr,
                         nullptr, x_amax, y_amax, workspace, workspace_bytes, stream);
}

int y4_amax_f32(const float* x, int ldx, long long M, int C, unsigned* amax_bits, void* stream) {
    if (!x || !amax_bits) return Y4_ERR_NULL;
    if (M < 0 || C <= 0 || ldx < C) return Y4_ERR_SHAPE;
    return y4::amax_launch(x, ldx, M, C, amax_bits, y4_stream(stream));
}

int y4_amax_merge_u32(unsigned* dst, const unsigned* src, void* stream) {
    if (!dst || !src) return Y4_ERR_NULL;
    return y4::amax_merge(dst, src, y4_stream(stream));
}

size_t y4_conv2d_bnstats_workspace(int B, int H, int W, int Cin, int Cout, int k, int stride) {
    if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || (k != 1 && k != 3) || (stride != 1 && stride != 2)) return 0;
    const int pad = (k - 1) / 2;
    const long long M = (long long)B * ((H + 2 * pad - k) / stride + 1) * ((W + 2 * pad - k) / stride + 1);
    long long rows = Cin == 3 ? (M + 255) / 256 : (M + 63) / 64;            // smallest M-tile of any variant
    const long long per_img = (long long)B * ((M / B + 127) / 128);         // the 3x3 halo kernel tiles image by image
    if (per_img > rows) rows = per_img;
    return (size_t)rows * 2 * Cout * sizeof(float);
}

int y4_conv2d_fwd_bnstats_f32(const float* x, int ldx, const float* w, float* y, int ldy,
                              int B, int H, int W, int Cin, int Cout, int k, int stride,
                              float* partials, size_t partial_bytes, long long* nparts_host, const unsigned* x_amax,
                              void* workspace, size_t workspace_bytes, void* dgrad_filter, size_t dgrad_filter_bytes,
                              void* stream) {
    if (!partials || !nparts_host) return Y4_ERR_NULL;
    if (dgrad_filter && g_conv_mode != 3) return Y4_ERR_SHAPE;
    if (partial_bytes < y4_conv2d_bnstats_workspace(B, H, W, Cin, Cout, k, stride)) return Y4_ERR_WORKSPACE;
    int np = 0;
    const int rc = conv_fwd_impl(x, ldx, w, y, ldy, B, H, W, Cin, Cout, k, stride, nullptr, nullptr, Y4_ACT_LINEAR,
                                 nullptr, 0, partials, &np, x_amax, nullptr, workspace, workspace_bytes, stream, nullptr,
                                 dgrad_filter, dgrad_filter_bytes);
    if (rc != Y4_OK) return rc;
    *nparts_host = np;
    return Y4_OK;
}

size_t y4_conv2d_dgrad_workspace(int Cin, int Cout, int k) {
    const size_t cp = (size_t)((Cout + 31) / 32) * 32;
    // fp32 transposed filter (4 B), 3 bf16 planes (6 B) or 2 fp16 planes (4 B) per element, + 64 B of amax words
    // + 4 KiB of per-block filter maxima
    return (size_t)Cin * k * k * cp * 6 + 64 + 4096;
}

static int conv_dgrad_impl(const float* dy, int lddy, const float* w, float* dx, int lddx,
                           int B, int H, int W, int Cin, int Cout, int k, int stride,
                           void* workspace, size_t workspace_bytes, const unsigned* dy_amax,
                           const float* residual, int ldr, void* stream) {
    if (!dy || !dx || !workspace) return Y4_ERR_NULL;
    if (!w && g_conv_mode != 3) return Y4_ERR_NULL;        // w == NULL: the workspace already holds the transposed planes
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (k != 1 && k != 3) || (stride != 1 && stride != 2))
        return Y4_ERR_SHAPE;
    const int Cout_pad = (Cout + 31) / 32 * 32;
    // the padded channels of dy are read (multiplied by zero filter rows): they must exist
    if (lddy < Cout_pad || (lddy & 3) || lddx < Cin) return Y4_ERR_SHAPE;
    if (workspace_bytes < y4_conv2d_dgrad_workspace(Cin, Cout, k)) return Y4_ERR_WORKSPACE;
    if ((reinterpret_cast<uintptr_t>(dy) & 15) || (reinterpret_cast<uintptr_t>(workspace) & 15)) return Y4_ERR_SHAPE;
    hipStream_t st = y4_stream(stream);
    float* wt = static_cast<float*>(workspace);
    const long long total = (long long)Cin * k * k * Cout_pad;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    unsigned* hdr = reinterpret_cast<unsigned*>(static_cast<char*>(workspace) + (size_t)total * 6);
    unsigned* wamax = hdr;                                 // the call's own words: nothing shared between calls
    if (g_conv_mode == 3) {
        if (w) {
            const int rc = y4::f16x2_filter_planes_transposed(w, static_cast<unsigned short*>(workspace), Cout, Cin, k * k, Cout_pad,
                                                              wamax, hdr + 16, st, y4::tile_conv_ok(Cout_pad, Cout, Cin, k, stride, H, W));
            if (rc != Y4_OK) return rc;
        }                                                  // else: left there by y4_conv2d_fwd_bnstats_f32(..., dgrad_filter)
    } else if (g_conv_mode != 0)
        hipLaunchKernelGGL(transpose_split_filter_kernel, dim3(blocks), dim3(256), 0, st, w,
                           static_cast<unsigned short*>(workspace), Cout, Cin, k * k, Cout_pad, g_conv_mode == 2 ? 1 : 3);
    else
        hipLaunchKernelGGL(transpose_filter_kernel, dim3(blocks), dim3(256), 0, st, w, wt, Cout, Cin, k * k, Cout_pad);
    Y4_CHECK_LAUNCH();
    const int pad = (k - 1) / 2;
    ConvGeom g{};
    if (residual && ldr < Cin) return Y4_ERR_SHAPE;
    g.src = dy; g.wt = wt; g.dst = dx; g.scale = nullptr; g.shift = nullptr; g.res = residual;
    g.wt_planes = static_cast<const unsigned short*>(workspace);
    g.lds_ = lddy; g.ldd = lddx; g.ldr = ldr;
    g.B = B;
    g.Hs = (H + 2 * pad - k) / stride + 1;
    g.Ws = (W + 2 * pad - k) / stride + 1;
    g.Cs = Cout_pad; g.Cs_valid = Cout;
    g.Hd = H; g.Wd = W; g.N = Cin; g.k = k; g.stride = stride; g.pad = pad;
    const long long M = (long long)B * H * W;
    if (M >= (1ll << 31)) return Y4_ERR_SHAPE;
    g.M = (int)M; g.K = k * k * Cout_pad; g.act = Y4_ACT_LINEAR;
    if (g_conv_mode == 3) {
        if (!dy_amax) {
            // pad channels of dy may hold anything: the maximum is taken over the valid channels only
            const int rc = y4::amax_launch(dy, lddy, (long long)B * g.Hs * g.Ws, Cout, hdr + 1, st);
            if (rc != Y4_OK) return rc;
            dy_amax = hdr + 1;
        }
        g.src_amax = dy_amax; g.wt_amax = wamax;
        // few channels on a large map: the 2-D tile kernel, in forward form on the mirrored transposed planes (above)
        if (y4::tile_conv_ok(Cout_pad, Cout, Cin, k, stride, H, W)) return y4::f16x2_tile(g, st, nullptr);
        if (y4::tile_dgrad_s2_ok(Cout_pad, Cout, Cin, k, stride, g.Hs, g.Ws)) return y4::f16x2_tile_dgrad_s2(g, st);
        return y4::f16x2_gather(g, true, st, nullptr);
    }
    return dispatch_gather<true>(g, st);
}

int y4_conv2d_dgrad_f32(const float* dy, int lddy, const float* w, float* dx, int lddx,
                        int B, int H, int W, int Cin, int Cout, int k, int stride,
                        void* workspace, size_t workspace_bytes, const unsigned* dy_amax,
                        const float* residual, int ldr, void* stream) {
    return conv_dgrad_impl(dy, lddy, w, dx, lddx, B, H, W, Cin, Cout, k, stride, workspace, workspace_bytes, dy_amax,
                           residual, ldr, stream);
}

int y4_last_conv_kernel(char* buf, int cap) {
    if (!buf || cap <= 0) return Y4_ERR_NULL;
    snprintf(buf, (size_t)cap, "%s", g_last_kernel);
    g_last_kernel[0] = 0;
    return Y4_OK;
}

size_t y4_conv2d_prepared_bytes(int Cout, int K) {
    if (Cout <= 0 || K <= 0) return 0;
    return 64 + 4096 + (size_t)Cout * (size_t)K * 4;
}

int y4_conv2d_prepare_filter_f32(const float* w, int Cout, int K, void* prepared, size_t prepared_bytes, void* stream) {
    if (!w || !prepared) return Y4_ERR_NULL;
    if (Cout <= 0 || K <= 0 || (K & 31)) return Y4_ERR_SHAPE;
    if (prepared_bytes < y4_conv2d_prepared_bytes(Cout, K)) return Y4_ERR_WORKSPACE;
    if ((reinterpret_cast<uintptr_t>(prepared) & 15) || (reinterpret_cast<uintptr_t>(w) & 15)) return Y4_ERR_SHAPE;
    return y4::f16x2_refresh_prepared(w, prepared, Cout, K, y4_stream(stream));
}

int y4_conv2d_fwd_prepared_f32(const float* x, int ldx, void* w_prepared, float* y, int ldy,
                               int B, int H, int W, int Cin, int Cout, int k, int stride,
                               const float* scale, const float* shift, int act,
                               const float* residual, int ldr, const unsigned* x_amax, unsigned* y_amax, void* stream) {
    if (!w_prepared) return Y4_ERR_NULL;
    return conv_fwd_impl(x, ldx, nullptr, y, ldy, B, H, W, Cin, Cout, k, stride, scale, shift, act, residual, ldr, nullptr,
                         nullptr, x_amax, y_amax, nullptr, 0, stream, w_prepared);
}

size_t y4_conv2d_wgrad_workspace(int B, int H, int W, int Cin, int Cout, int k, int stride) {
    WgradGeom g{};
    wgrad_plan(B, H, W, Cin, Cout, k, stride, g);
    // slabs of the split-K partial sums + 64 B of amax words (f16x2 mode); the tile kernel (wgrad_tile.hip) writes one slab per block
    int splits = g.splits;
    if (y4::tile_wgrad_ok(Cin, Cout, k, stride, H, W, 0, 0) && y4::tile_wgrad_slabs(Cin, Cout) > splits) splits = y4::tile_wgrad_slabs(Cin, Cout);
    return (splits > 1 ? (size_t)splits * Cout * g.J * sizeof(float) : 0) + 64;
}

int y4_conv2d_wgrad_f32(const float* x, int ldx, const float* dy, int lddy, float* dw,
                        int B, int H, int W, int Cin, int Cout, int k, int stride,
                        void* workspace, size_t workspace_bytes, const unsigned* x_amax, const unsigned* dy_amax,
                        void* stream) {
    if (!x || !dy || !dw) return Y4_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || (k != 1 && k != 3) || (stride != 1 && stride != 2))
        return Y4_ERR_SHAPE;
    if (Cin <= 0 || (Cin & 3) || ldx < Cin || (ldx & 3) || (lddy & 3) || lddy < ((Cout + 3) & ~3)) return Y4_ERR_SHAPE;
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(dy) & 15)) return Y4_ERR_SHAPE;
    WgradGeom g{};
    wgrad_plan(B, H, W, Cin, Cout, k, stride, g);
    if ((long long)B * g.Ho * g.Wo >= (1ll << 31)) return Y4_ERR_SHAPE;
    g.x = x; g.dy = dy; g.ldx = ldx; g.lddy = lddy;
    {
        g.x_total_bytes = (unsigned long long)B * H * W * (unsigned long long)ldx * 4ull;
        g.dy_total_bytes = (unsigned long long)B * g.Ho * g.Wo * (unsigned long long)lddy * 4ull;
        const unsigned long long range_px = (unsigned long long)g.chunks_per_split * 32ull + 2ull * g.Ho * g.Wo;
        const unsigned long long per_px = 4ull * (unsigned long long)(lddy > ldx * stride * stride ? lddy : ldx * stride * stride);
        if (range_px * per_px >= 0xfffffff0ull) return Y4_ERR_SHAPE;      // pitch more than 2x the channel count at > 4 GiB
    }
    hipStream_t st = y4_stream(stream);
    // few channels on a large map (3x3): the tile kernel -- every x pixel staged once for nine taps, one slab per block
    const bool tile = g_conv_mode == 3 && y4::tile_wgrad_ok(Cin, Cout, k, stride, H, W, ldx, lddy);
    if (tile) g.splits = y4::tile_wgrad_slabs(Cin, Cout);
    const size_t slab_bytes = g.splits > 1 ? (size_t)g.splits * Cout * g.J * sizeof(float) : 0;
    if (g.splits > 1) {
        if (!workspace) return Y4_ERR_NULL;
        if (workspace_bytes < slab_bytes) return Y4_ERR_WORKSPACE;
        g.out = static_cast<float*>(workspace);
    } else {
        g.out = dw;
    }
    int rc;
    if (g_conv_mode == 3) {
        if (!x_amax || !dy_amax) {
            if (!workspace || workspace_bytes < slab_bytes + 64) return Y4_ERR_WORKSPACE;
            unsigned* hdr = reinterpret_cast<unsigned*>(static_cast<char*>(workspace) + slab_bytes);
            if (!x_amax) {
                rc = y4::amax_launch(x, ldx, (long long)B * H * W, Cin, hdr, st);
                if (rc != Y4_OK) return rc;
                x_amax = hdr;
            }
            if (!dy_amax) {
                rc = y4::amax_launch(dy, lddy, (long long)B * g.Ho * g.Wo, Cout, hdr + 1, st);
                if (rc != Y4_OK) return rc;
                dy_amax = hdr + 1;
            }
        }
        g.x_amax = x_amax; g.dy_amax = dy_amax;
        rc = tile ? y4::f16x2_wgrad_tile(g, st) : y4::f16x2_wgrad(g, st);
    } else if (g_conv_mode == 1) {
        if (g.tn == 128 && g.tj == 128) rc = launch_wgrad<128, 128, 3>(g, st);
        else if (g.tn == 128) rc = launch_wgrad<128, 64, 3>(g, st);
        else if (g.tj == 128) rc = launch_wgrad<64, 128, 3>(g, st);
        else rc = launch_wgrad<64, 64, 3>(g, st);
    } else if (g_conv_mode == 2) {
        if (g.tn == 128 && g.tj == 128) rc = launch_wgrad<128, 128, 1>(g, st);
        else if (g.tn == 128) rc = launch_wgrad<128, 64, 1>(g, st);
        else if (g.tj == 128) rc = launch_wgrad<64, 128, 1>(g, st);
        else rc = launch_wgrad<64, 64, 1>(g, st);
    } else if (g.tn == 128 && g.tj == 128) rc = launch_wgrad<128, 128>(g, st);
    else if (g.tn == 128) rc = launch_wgrad<128, 64>(g, st);
    else if (g.tj == 128) rc = launch_wgrad<64, 128>(g, st);
    else rc = launch_wgrad<64, 64>(g, st);
    if (rc != Y4_OK) return rc;
    if (g.splits > 1) {
        const int rc2 = y4::slab_reduce(static_cast<const float*>(workspace), dw, (long long)Cout * g.J, g.splits, st);
        if (rc2 != Y4_OK) return rc2;
    }
    return Y4_OK;
}

int y4_conv2d_stem_fwd_f32(const float* x, long long sxb, long long sxc, long long sxh, long long sxw,
                           const float* w, float* y, int ldy, int B, int H, int W, int Cout,
                           const float* scale, const float* shift, int act,
                           float* bnstats_partials, void* stream) {
    if (!x || !w || !y) return Y4_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cout > 32 || ldy < Cout) return Y4_ERR_SHAPE;
    StemGeom g{};
    g.x = x; g.w = w; g.y = y; g.scale = scale; g.shift = shift; g.stats = bnstats_partials;
    g.sxb = sxb; g.sxc = sxc; g.sxh = sxh; g.sxw = sxw; g.ldy = ldy;
    g.B = B; g.H = H; g.W = W; g.Cout = Cout; g.act = act;
    g.M = (long long)B * H * W;
    const long long blocks = (g.M + 255) / 256;
    if (blocks >= (1ll << 31)) return Y4_ERR_SHAPE;
    // matrix-core variant: bf16x3 arithmetic, 32-bit byte offsets into x (non-negative strides)
    const bool mfma_ok = (g_conv_mode >= 1 && g_conv_mode <= 3) && sxb >= 0 && sxc >= 0 && sxh >= 0 && sxw >= 0 &&
                         ((long long)(B - 1) * sxb + 2 * sxc + (long long)(H - 1) * sxh + (long long)(W - 1) * sxw + 1) * 4 < 0xfffffff0ll &&
                         (long long)H * W < (1 << 24);
    if (mfma_ok) {
        const int grid = (int)(blocks < 2048 ? blocks : 2048);
        hipLaunchKernelGGL(conv_stem_fwd_bf16x3_kernel, dim3(grid), dim3(256), 0, y4_stream(stream), g, (int)blocks,
                           1.0f / (float)((long long)H * W), 1.0f / (float)W);
        Y4_CHECK_LAUNCH();
        return Y4_OK;
    }
    hipLaunchKernelGGL(conv_stem_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, y4_stream(stream), g);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

size_t y4_conv2d_stem_wgrad_workspace(int, int, int, int) {
    return (size_t)STEM_WAVES * 1024 * sizeof(float) + 64 * 1024 * sizeof(double);
}

int y4_conv2d_stem_wgrad_f32(const float* x, long long sxb, long long sxc, long long sxh, long long sxw,
                             const float* dy, int lddy, float* dw, int B, int H, int W, int Cout,
                             void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !dy || !dw || !workspace) return Y4_ERR_NULL;
    if (B <= 0 || H <= 0 || W < 2 || Cout <= 0 || Cout > 32 || lddy < Cout) return Y4_ERR_SHAPE;
    if (workspace_bytes < y4_conv2d_stem_wgrad_workspace(B, H, W, Cout)) return Y4_ERR_WORKSPACE;
    StemWgradGeom g{};
    g.x = x; g.dy = dy; g.slabs = static_cast<float*>(workspace);
    g.sxb = sxb; g.sxc = sxc; g.sxh = sxh; g.sxw = sxw; g.lddy = lddy;
    g.B = B; g.H = H; g.W = W; g.Cout = Cout;
    g.M = (long long)B * H * W;
    long long ppw = (g.M + STEM_WAVES - 1) / STEM_WAVES;
    ppw = (ppw + 15) & ~15ll;                     // 16-pixel trips (8 MFMA pixel pairs) never straddle waves
    if ((long long)ppw * lddy * 4 >= 0xfffffff0ll || sxb < 0 || sxc < 0 || sxh < 0 || sxw < 0 ||
        ((long long)(B - 1) * sxb + 2 * sxc + (long long)(H - 1) * sxh + (long long)(W - 1) * sxw + 1) * 4 >= 0xfffffff0ll)
        return Y4_ERR_SHAPE;
    g.pix_per_wave = ppw;
    hipStream_t st = y4_stream(stream);
    hipLaunchKernelGGL(conv_stem_wgrad_kernel, dim3(STEM_WAVES / 4), dim3(256), 0, st, g);
    Y4_CHECK_LAUNCH();
    double* part = reinterpret_cast<double*>(static_cast<char*>(workspace) + (size_t)STEM_WAVES * 1024 * sizeof(float));
    hipLaunchKernelGGL(stem_wgrad_reduce1_kernel, dim3(64), dim3(256), 0, st, static_cast<const float*>(workspace), part,
                       STEM_WAVES);
    Y4_CHECK_LAUNCH();
    hipLaunchKernelGGL(stem_wgrad_reduce2_kernel, dim3(1), dim3(256), 0, st, part, dw, 64, Cout);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

}  // extern "C"
