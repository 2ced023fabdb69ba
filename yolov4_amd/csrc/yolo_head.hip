// YOLO head: anchor decode, target assignment + detection loss (+ its gradient), and the
// per-class greedy NMS of the post-processing.  No MFMA here: HBM-bound sweeps, wavefront /
// block reductions, and one block-wide bitonic sort + bitmask NMS per (image, class) segment.
//
// This file is compiled with -ffp-contract=off: IoU comparisons against thresholds must follow
// the reference's separate fp32 multiply / add / divide sequence bit for bit
// (yolo/model/yololoss.py:56-91, yolo/util/utils.py:64-77).
#include <stdlib.h>
#include "common.h"

namespace {

struct Anchors { float w[16]; float h[16]; };

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ------------------------------------------------------------------------------------ decode
// one block sweeps pixels; thread t <-> logit channel t = a*n_ch + ch of that pixel
template <bool EVAL>
__global__ __launch_bounds__(256) void yolo_decode_kernel(const float* __restrict__ logits, long long ldl,
                                                          float* __restrict__ output, float* __restrict__ pred,
                                                          long long n_total, long long box_off, float stride,
                                                          int B, int F, int A, int n_ch, Anchors anc) {
    const long long npix = (long long)B * F * F;
    const int nt = A * n_ch;
    for (int t = threadIdx.x; t < nt; t += blockDim.x) {
        const int a = t / n_ch, ch = t - a * n_ch;
        for (long long p = blockIdx.x; p < npix; p += gridDim.x) {
            const int i = (int)(p % F);
            const int j = (int)((p / F) % F);
            const long long b = p / ((long long)F * F);
            const float v = logits[p * ldl + t];
            float o = v, pv = v;
            if (ch != 2 && ch != 3) { o = sigmoidf_(v); pv = o; }
            if (ch == 0) pv = o + (float)i;
            else if (ch == 1) pv = o + (float)j;
            else if (ch == 2) pv = expf(v) * anc.w[a];
            else if (ch == 3) pv = expf(v) * anc.h[a];
            if (EVAL) {
                if (ch < 4) pv = pv * stride;
                output[(b * n_total + box_off + ((long long)a * F + j) * F + i) * n_ch + ch] = pv;
            } else {
                const long long cell = ((b * A + a) * F + j) * F + i;
                output[cell * n_ch + ch] = o;
                if (ch < 4) pred[cell * 4 + ch] = pv;
            }
        }
    }
}

// Tiled form of the decode: a block takes DEC_TP consecutive pixels of ONE image with all their A * n_ch logit channels.
//   phase 1: coalesced 16-B loads of whole pixel rows (a thread's four columns -- hence its anchor / channel roles -- are
//            fixed for the whole tile: no division per element), activation applied on the way into an LDS image
//            [pixel][column];
//   phase 2: for an anchor a, the tile's part of the result is ONE contiguous run of cnt * n_ch floats (rows of n_ch
//            floats of consecutive pixels follow each other in [B, N, n_ch] / [B, A, F, F, n_ch]): written with aligned
//            16-B stores gathered from the LDS image (scalar stores for the <= 3 floats at either end of the run).
// The per-channel kernel above read 4 bytes per lane with a 1-KiB stride between lanes' pixels and wrote three 340-B
// segments per pixel (0.22 of the HBM peak) and spent 30 VALU per element on IEEE expf + division; this one moves whole
// lines both ways with one v_exp_f32 + one v_rcp_f32 per element (~1e-7 relative, far inside the 1e-4 parity bar):
// 4.6 - 4.9 TB/s algorithmic on the 76 x 76 layer at bs = 32 (rows of 32 pixels; 16 on small maps for occupancy).
template <bool EVAL, int DEC_TP>                           // DEC_TP pixels per tile: DEC_TP x 256 floats of LDS
__global__ __launch_bounds__(256) void yolo_decode_tiled_kernel(const float* __restrict__ logits, long long ldl,
                                                                float* __restrict__ output, float* __restrict__ pred,
                                                                long long n_total, long long box_off, float stride,
                                                                int F, int A, int n_ch, Anchors anc, int tiles_per_img) {
    __shared__ __attribute__((aligned(16))) float tile[DEC_TP][256];
    __shared__ __attribute__((aligned(16))) float predt[EVAL ? 1 : 4][DEC_TP][4];      // train: decoded boxes per anchor
    const int tid = threadIdx.x;
    const int b = blockIdx.x / tiles_per_img, t = blockIdx.x - b * tiles_per_img;
    const int FF = F * F;
    const int p0 = t * DEC_TP;
    const int cnt = FF - p0 < DEC_TP ? FF - p0 : DEC_TP;
    const int nt = A * n_ch;
    // ---- phase 1.  Per column (fixed for the thread): value = rcp(exp(-v) + one) * mul + (ci * i + cj * j)
    //   sigmoid channels: one = 1 (1 / (1 + e^-v)); exp channels (w, h): one = 0 (1 / e^-v = e^v), mul = anchor (* stride)
    // -> one v_exp, one v_rcp and three plain VALU per element, no per-element role tests
    const int col4 = tid & 63, r0 = tid >> 6;
    f32x4 one, mul, ci, cj;
    int ca[4], cch[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = 4 * col4 + e;
        const int a = c < nt ? c / n_ch : -1;
        const int ch = c < nt ? c - a * n_ch : 4;
        ca[e] = a; cch[e] = ch;
        const float sc = (EVAL && ch < 4) ? stride : 1.0f;
        one[e] = (ch == 2 || ch == 3) ? 0.f : 1.f;
        mul[e] = ch == 2 ? anc.w[a < 0 ? 0 : a] * sc : ch == 3 ? anc.h[a < 0 ? 0 : a] * sc : sc;
        ci[e] = ch == 0 ? sc : 0.f;
        cj[e] = ch == 1 ? sc : 0.f;
    }
    const bool anyc = 4 * col4 < nt;
    // all of a thread's DEC_TP / 4 row loads go out before the first is used (the loop was latency-bound otherwise)
    f32x4 vv[DEC_TP / 4];
#pragma unroll
    for (int u = 0; u < DEC_TP / 4; ++u) {
        const int r = r0 + 4 * u;
        vv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (anyc && r < cnt) vv[u] = *reinterpret_cast<const f32x4*>(logits + ((long long)b * FF + p0 + r) * ldl + 4 * col4);
    }
#pragma unroll
    for (int u = 0; u < DEC_TP / 4; ++u) {
        const int r = r0 + 4 * u;
        if (r >= cnt) break;
        const int pix = p0 + r;
        const int j = pix / F, i = pix - j * F;
        const float fi = (float)i, fj = (float)j;
        const f32x4 v = vv[u];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float t = __expf(-v[e]);
            const float rr = __frcp_rn(t + one[e]);
            const float pv = rr * mul[e] + (ci[e] * fi + cj[e] * fj);
            if (EVAL) {
                o[e] = pv;
            } else {
                const int ch = cch[e];
                o[e] = (ch == 2 || ch == 3) ? v[e] : rr;          // `output` keeps the raw w / h logits (yololayer.py:136-139)
                if (ca[e] >= 0 && ch < 4) predt[ca[e]][r][ch] = pv;
            }
        }
        *reinterpret_cast<f32x4*>(&tile[r][4 * col4]) = o;
    }
    __syncthreads();
    // ---- phase 2
    for (int a = 0; a < A; ++a) {
        const long long row0 = EVAL ? (long long)b * n_total + box_off + (long long)a * FF + p0
                                    : ((long long)b * A + a) * FF + p0;
        float* dst = output + row0 * n_ch;                 // the run [dst, dst + cnt * n_ch)
        const int len = cnt * n_ch;
        const int mis = (int)((reinterpret_cast<uintptr_t>(dst) >> 2) & 3);      // floats past a 16-B boundary
        const int head = mis ? 4 - mis : 0;                // scalar floats before the first aligned float4
        const int nv = len > head ? (len - head) >> 2 : 0;
        const int colbase = a * n_ch;
        // (row, channel) of a thread's float4 advance by 1024 floats per trip: one division up front, none in the loop
        const int adv_pl = 1024 / n_ch, adv_ch = 1024 - adv_pl * n_ch;
        int pl0 = (head + 4 * tid) / n_ch, ch0 = head + 4 * tid - pl0 * n_ch;
        for (int q = tid; q < nv; q += 256) {
            const int o0 = head + 4 * q;
            int pl = pl0, ch = ch0;
            f32x4 w;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                w[e] = tile[pl][colbase + ch];
                if (++ch == n_ch) { ch = 0; ++pl; }
            }
            *reinterpret_cast<f32x4*>(dst + o0) = w;
            pl0 += adv_pl; ch0 += adv_ch;
            if (ch0 >= n_ch) { ch0 -= n_ch; ++pl0; }
        }
        const int tail0 = head + 4 * nv;                   // scalar ends: [0, head) and [tail0, len)
        if (tid < head && tid < len) { const int pl = tid / n_ch; dst[tid] = tile[pl][colbase + tid - pl * n_ch]; }
        if (tid >= 4 && tid < 4 + (len - tail0) && len > head) {
            const int o = tail0 + tid - 4;
            const int pl = o / n_ch;
            dst[o] = tile[pl][colbase + o - pl * n_ch];
        }
        if (!EVAL) {
            // pred [B, A, F, F, 4]: one float4 per cell, always 16-B aligned
            if (tid < cnt)
                *reinterpret_cast<f32x4*>(pred + (((long long)b * A + a) * FF + p0 + tid) * 4) = *reinterpret_cast<const f32x4*>(predt[a][tid]);
        }
    }
}

static int decode_tp(long long npix) { return npix >= 100000 ? 32 : 16; }      // tile height: measured best per map size
static bool decode_tiled_ok(const float* logits, int ldl, int A, int n_ch, const float* output, const float* pred) {
    return A * n_ch <= 256 && A <= 4 && (ldl & 3) == 0 && ldl >= ((A * n_ch + 3) & ~3) &&
           (reinterpret_cast<uintptr_t>(logits) & 15) == 0 && (reinterpret_cast<uintptr_t>(output) & 3) == 0 &&
           (!pred || (reinterpret_cast<uintptr_t>(pred) & 15) == 0);
}

__global__ __launch_bounds__(256) void yolo_decode_bwd_kernel(const float* __restrict__ logits, long long ldl,
                                                              const float* __restrict__ g_out,
                                                              const float* __restrict__ g_pred,
                                                              float* __restrict__ g_logits,
                                                              int B, int F, int A, int n_ch, Anchors anc) {
    const long long npix = (long long)B * F * F;
    const int nt = A * n_ch;
    for (int t = threadIdx.x; t < (int)ldl; t += blockDim.x) {
        const int a = t / n_ch, ch = t - a * n_ch;
        for (long long p = blockIdx.x; p < npix; p += gridDim.x) {
            float g = 0.f;
            if (t < nt) {
                const int i = (int)(p % F);
                const int j = (int)((p / F) % F);
                const long long b = p / ((long long)F * F);
                const long long cell = ((b * A + a) * F + j) * F + i;
                const float v = logits[p * ldl + t];
                if (g_out) g = g_out[cell * n_ch + ch];
                if (ch < 4 && g_pred) {
                    const float gp = g_pred[cell * 4 + ch];
                    if (ch < 2) g += gp;
                    else g += gp * (expf(v) * (ch == 2 ? anc.w[a] : anc.h[a]));
                }
                if (ch != 2 && ch != 3) {
                    const float o = sigmoidf_(v);
                    g = g * ((1.0f - o) * o);
                }
            }
            g_logits[p * ldl + t] = g;
        }
    }
}

// ------------------------------------------------------------------------------------ pairwise IoU
// bboxes_iou, yolo/model/yololoss.py:16-91: [Na,4] x [Nb,4] -> [Na,Nb]; xyxy corners or centre/size; the intersection
// counts only where tl < br on both axes; area_i / (area_a + area_b - area_i), no epsilon.  torch.max/min
// propagate NaN (fmaxf/fminf do not), kept here.  One thread per (a, b) pair, the fp32 operation sequence of the
// reference (this file is built with -ffp-contract=off).
__device__ __forceinline__ float tmax_(float a, float b) { return (a != a || b != b) ? NAN : fmaxf(a, b); }
__device__ __forceinline__ float tmin_(float a, float b) { return (a != a || b != b) ? NAN : fminf(a, b); }
__global__ __launch_bounds__(256) void bboxes_iou_kernel(const float* __restrict__ A_, long long Na, const float* __restrict__ B_,
                                                         long long Nb, int xyxy, float* __restrict__ out) {
    const long long total = Na * Nb;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long ia = i / Nb, ib = i - ia * Nb;
        const float a0 = A_[ia * 4], a1 = A_[ia * 4 + 1], a2 = A_[ia * 4 + 2], a3 = A_[ia * 4 + 3];
        const float b0 = B_[ib * 4], b1 = B_[ib * 4 + 1], b2 = B_[ib * 4 + 2], b3 = B_[ib * 4 + 3];
        float tlx, tly, brx, bry, area_a, area_b;
        if (xyxy) {
            tlx = tmax_(a0, b0); tly = tmax_(a1, b1);
            brx = tmin_(a2, b2); bry = tmin_(a3, b3);
            area_a = (a2 - a0) * (a3 - a1);
            area_b = (b2 - b0) * (b3 - b1);
        } else {
            tlx = tmax_(a0 - a2 / 2.f, b0 - b2 / 2.f); tly = tmax_(a1 - a3 / 2.f, b1 - b3 / 2.f);
            brx = tmin_(a0 + a2 / 2.f, b0 + b2 / 2.f); bry = tmin_(a1 + a3 / 2.f, b1 + b3 / 2.f);
            area_a = a2 * a3;
            area_b = b2 * b3;
        }
        const float en = ((tlx < brx) ? 1.f : 0.f) * ((tly < bry) ? 1.f : 0.f);
        const float area_i = ((brx - tlx) * (bry - tly)) * en;
        out[i] = area_i / ((area_a + area_b) - area_i);
    }
}

// ------------------------------------------------------------------------------------ loss
struct LossWs {
    size_t nlabel, npos, truth, pos_cell, pos_val, pos_cls, pos_index, partials, total;
    int cw, nblocks;
};
static inline size_t al16(size_t x) { return (x + 15) & ~(size_t)15; }
constexpr int LOSS_BLOCK = 256;
static LossWs loss_ws(int B, int F, int A, int K, int C) {
    LossWs w;
    w.cw = (C + 31) / 32;
    const long long cells = (long long)B * A * F * F;
    w.nblocks = (int)((cells + LOSS_BLOCK - 1) / LOSS_BLOCK);
    size_t o = 0;
    w.nlabel = o; o = al16(o + (size_t)B * 4);
    w.npos = o; o = al16(o + (size_t)B * 4);
    w.truth = o; o = al16(o + (size_t)B * K * 4 * 4);
    w.pos_cell = o; o = al16(o + (size_t)B * K * 4);
    w.pos_val = o; o = al16(o + (size_t)B * K * 5 * 4);
    w.pos_cls = o; o = al16(o + (size_t)B * K * w.cw * 4);
    w.pos_index = o; o = al16(o + (size_t)cells * 4);
    w.partials = o; o = al16(o + (size_t)w.nblocks * 4 * 8);
    w.total = o;
    return w;
}

struct LossPtrs {
    int* nlabel; int* npos; float* truth; int* pos_cell; float* pos_val; unsigned* pos_cls; int* pos_index;
    double* partials;
};
static LossPtrs loss_ptrs(const void* ws, const LossWs& l) {
    char* b = static_cast<char*>(const_cast<void*>(ws));
    LossPtrs p;
    p.nlabel = reinterpret_cast<int*>(b + l.nlabel);
    p.npos = reinterpret_cast<int*>(b + l.npos);
    p.truth = reinterpret_cast<float*>(b + l.truth);
    p.pos_cell = reinterpret_cast<int*>(b + l.pos_cell);
    p.pos_val = reinterpret_cast<float*>(b + l.pos_val);
    p.pos_cls = reinterpret_cast<unsigned*>(b + l.pos_cls);
    p.pos_index = reinterpret_cast<int*>(b + l.pos_index);
    p.partials = reinterpret_cast<double*>(b + l.partials);
    return p;
}

// yololoss.py:196-265,304-369: one thread per image walks its <= K labels in order (the
// sequential semantics -- last writer wins for xy/wh/scale, class bits OR -- are kept by
// construction); cost is negligible (B*K*9 IoUs).
__global__ void yolo_assign_kernel(const float* __restrict__ labels, int K, int B, int F, int A, int C, int cw,
                                   float stride, Anchors all_anc, int n_all, int m0, int m1, int m2,
                                   Anchors masked, LossPtrs w) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* lab = labels + (long long)b * K * 5;
    int n = 0;
    for (int t = 0; t < K; ++t) {
        const float s = (((lab[t * 5] + lab[t * 5 + 1]) + lab[t * 5 + 2]) + lab[t * 5 + 3]) + lab[t * 5 + 4];
        if (s > 0.f) ++n;
    }
    w.nlabel[b] = n;
    int np = 0;
    const float Ff = (float)F;
    for (int t = 0; t < n; ++t) {
        const float tx = lab[t * 5] / stride, ty = lab[t * 5 + 1] / stride;
        const float tw = lab[t * 5 + 2] / stride, th = lab[t * 5 + 3] / stride;
        float* tr = w.truth + ((long long)b * K + t) * 4;
        tr[0] = tx; tr[1] = ty; tr[2] = tw; tr[3] = th;
        // IoU of (0,0,tw,th) against the n_all anchors (0,0,aw,ah), xyxy form, first max wins
        int best = 0;
        float best_iou = -INFINITY;
        bool first = true;
        for (int k = 0; k < n_all; ++k) {
            const float aw = all_anc.w[k], ah = all_anc.h[k];
            const float brx = fminf(tw, aw), bry = fminf(th, ah);
            const float en = (0.f < brx && 0.f < bry) ? 1.f : 0.f;
            const float ai = (brx * bry) * en;
            const float iou = ai / ((tw * th + aw * ah) - ai);
            // torch.argmax: NaN counts as maximal; first occurrence wins
            if (first || iou > best_iou || (iou != iou && !(best_iou != best_iou))) { best = k; best_iou = iou; first = false; }
        }
        if (!(best == m0 || best == m1 || best == m2)) continue;
        const int a = best % 3;
        const int i = (int)(short)(int)tx, j = (int)(short)(int)ty;                    // .to(torch.int16): truncation
        if (i < 0 || i >= F || j < 0 || j >= F || a >= A) continue;          // reference would raise / wrap: skipped
        const int cell = (a * F + j) * F + i;
        int rec = -1;
        for (int u = 0; u < np; ++u)
            if (w.pos_cell[(long long)b * K + u] == cell) { rec = u; break; }
        if (rec < 0) {
            rec = np++;
            w.pos_cell[(long long)b * K + rec] = cell;
            for (int q = 0; q < cw; ++q) w.pos_cls[((long long)b * K + rec) * cw + q] = 0u;
            w.pos_index[(long long)b * A * F * F + cell] = rec;
        }
        float* pv = w.pos_val + ((long long)b * K + rec) * 5;
        pv[0] = tx - (float)(short)(int)tx;
        pv[1] = ty - (float)(short)(int)ty;
        pv[2] = logf(tw / masked.w[a] + 1e-16f);
        pv[3] = logf(th / masked.h[a] + 1e-16f);
        pv[4] = sqrtf(2.0f - tw * th / Ff / Ff);
        const int cls = (int)(short)(int)lab[t * 5 + 4];
        if (cls >= 0 && cls < C) w.pos_cls[((long long)b * K + rec) * cw + (cls >> 5)] |= 1u << (cls & 31);
    }
    w.npos[b] = np;
}

// The same assignment with one wave per image (K <= 64 labels, one lane each): the per-label work (anchor argmax, cell,
// regression targets) is independent; the sequential rules of the loop above become lane-order rules -- a cell's record
// belongs to the first label that hits it (records numbered in that order), its values come from the LAST such label,
// its class bits are the OR over all of them.
__global__ __launch_bounds__(64) void yolo_assign_wave_kernel(const float* __restrict__ labels, int K, int B, int F, int A,
                                                              int C, int cw, float stride, Anchors all_anc, int n_all,
                                                              int m0, int m1, int m2, Anchors masked, LossPtrs w) {
    const int b = blockIdx.x, t = threadIdx.x;
    const float* lab = labels + (long long)b * K * 5;
    float l0 = 0.f, l1 = 0.f, l2 = 0.f, l3 = 0.f, l4 = 0.f;
    if (t < K) { l0 = lab[t * 5]; l1 = lab[t * 5 + 1]; l2 = lab[t * 5 + 2]; l3 = lab[t * 5 + 3]; l4 = lab[t * 5 + 4]; }
    const float s = (((l0 + l1) + l2) + l3) + l4;
    const int n = __popcll(__ballot(t < K && s > 0.f));
    if (t == 0) w.nlabel[b] = n;
    const bool act = t < n;
    const float tx = l0 / stride, ty = l1 / stride, tw = l2 / stride, th = l3 / stride;
    if (act) {
        float* tr = w.truth + ((long long)b * K + t) * 4;
        tr[0] = tx; tr[1] = ty; tr[2] = tw; tr[3] = th;
    }
    int best = 0;
    float best_iou = -INFINITY;
    bool first = true;
    for (int k = 0; k < n_all; ++k) {
        const float aw = all_anc.w[k], ah = all_anc.h[k];
        const float brx = fminf(tw, aw), bry = fminf(th, ah);
        const float en = (0.f < brx && 0.f < bry) ? 1.f : 0.f;
        const float ai = (brx * bry) * en;
        const float iou = ai / ((tw * th + aw * ah) - ai);
        if (first || iou > best_iou || (iou != iou && !(best_iou != best_iou))) { best = k; best_iou = iou; first = false; }
    }
    const int a = best % 3;
    const int i = (int)(short)(int)tx, j = (int)(short)(int)ty;
    const bool pos = act && (best == m0 || best == m1 || best == m2) && i >= 0 && i < F && j >= 0 && j < F && a < A;
    const int cell = pos ? (a * F + j) * F + i : -1 - t;                      // non-positive lanes never match anyone
    const int cls = (int)(short)(int)l4;
    const unsigned long long pm = __ballot(pos);
    int owner = t;
    bool later_same = false;
    for (int u = 0; u < 64; ++u) {
        const int cu = __shfl(cell, u, 64);
        if (((pm >> u) & 1ull) && cu == cell) {
            if (u < owner) owner = u;
            if (u > t) later_same = true;
        }
    }
    const bool is_owner = pos && owner == t;
    const unsigned long long om = __ballot(is_owner);
    const int my_rank = __popcll(om & ((1ull << t) - 1ull));
    const int rec = __shfl(my_rank, owner, 64);
    for (int q = 0; q < cw; ++q) {
        unsigned word = 0u;
        for (int u = 0; u < 64; ++u) {
            const int cu = __shfl(cell, u, 64), ku = __shfl(cls, u, 64);
            if (((pm >> u) & 1ull) && cu == cell && ku >= 0 && ku < C && (ku >> 5) == q) word |= 1u << (ku & 31);
        }
        if (is_owner) w.pos_cls[((long long)b * K + rec) * cw + q] = word;
    }
    if (is_owner) {
        w.pos_cell[(long long)b * K + rec] = cell;
        w.pos_index[(long long)b * A * F * F + cell] = rec;
    }
    if (pos && !later_same) {
        const float Ff = (float)F;
        float* pv = w.pos_val + ((long long)b * K + rec) * 5;
        pv[0] = tx - (float)(short)(int)tx;
        pv[1] = ty - (float)(short)(int)ty;
        pv[2] = logf(tw / masked.w[a] + 1e-16f);
        pv[3] = logf(th / masked.h[a] + 1e-16f);
        pv[4] = sqrtf(2.0f - tw * th / Ff / Ff);
    }
    if (t == 0) w.npos[b] = __popcll(om);
}

__device__ __forceinline__ float bce_term(float o, float t) {
    const float lo = fmaxf(logf(o), -100.f), l1 = fmaxf(logf(1.0f - o), -100.f);
    return -(t * lo + (1.0f - t) * l1);
}
__device__ __forceinline__ float bce_grad(float o, float t) { return (o - t) / fmaxf((1.0f - o) * o, 1e-12f); }

// yololoss.py:276-301 (ignore mask) + :402-427 (loss terms), one thread per cell.
__global__ __launch_bounds__(LOSS_BLOCK) void yolo_loss_dense_kernel(
    const float* __restrict__ output, const float* __restrict__ pred, int B, int F, int A, int K, int C, int cw,
    float thresh, float* __restrict__ obj_mask, LossPtrs w) {
    __shared__ double red[4][LOSS_BLOCK / 64];
    const long long cells = (long long)B * A * F * F;
    const long long cell = (long long)blockIdx.x * LOSS_BLOCK + threadIdx.x;
    const int n_ch = 5 + C;
    double part[4] = {0, 0, 0, 0};
    if (cell < cells) {
        const int per_img = A * F * F;
        const int b = (int)(cell / per_img);
        const int n = w.nlabel[b];
        const float px = pred[cell * 4], py = pred[cell * 4 + 1], pw = pred[cell * 4 + 2], ph = pred[cell * 4 + 3];
        const float pl = px - pw / 2.f, pr = px + pw / 2.f, pt = py - ph / 2.f, pb = py + ph / 2.f;
        const float area_a = pw * ph;
        float best = -INFINITY;
        bool nan_seen = false;
        const float* tr = w.truth + (long long)b * K * 4;
        for (int t = 0; t < n; ++t) {
            const float tx = tr[t * 4], ty = tr[t * 4 + 1], tw = tr[t * 4 + 2], th = tr[t * 4 + 3];
            const float tlx = fmaxf(pl, tx - tw / 2.f), tly = fmaxf(pt, ty - th / 2.f);
            const float brx = fminf(pr, tx + tw / 2.f), bry = fminf(pb, ty + th / 2.f);
            // torch.max/min propagate NaN, fmaxf does not: track NaN inputs explicitly
            const bool in_nan = (pl != pl) || (pr != pr) || (pt != pt) || (pb != pb);
            const float en = (tlx < brx && tly < bry) ? 1.f : 0.f;
            const float ai = ((brx - tlx) * (bry - tly)) * en;
            const float iou = ai / ((area_a + tw * th) - ai);
            if (in_nan || iou != iou) nan_seen = true;
            else best = fmaxf(best, iou);
        }
        const bool ignore = !nan_seen && best > thresh;
        const int rec = w.pos_index[cell];
        const float m = (rec >= 0 || !ignore) ? 1.f : 0.f;
        obj_mask[cell] = m;
        const float* o = output + cell * n_ch;
        if (m != 0.f) part[2] = (double)bce_term(o[4], rec >= 0 ? 1.f : 0.f);
        if (rec >= 0) {
            const float* pv = w.pos_val + ((long long)b * K + rec) * 5;
            const unsigned* pc = w.pos_cls + ((long long)b * K + rec) * cw;
            const float s = pv[4];
            const float ww = s * s;
            part[0] = (double)(ww * bce_term(o[0], pv[0])) + (double)(ww * bce_term(o[1], pv[1]));
            const float d2 = o[2] * s - pv[2] * s, d3 = o[3] * s - pv[3] * s;
            part[1] = ((double)(d2 * d2) + (double)(d3 * d3)) * 0.5;
            double c = 0;
            for (int k = 0; k < C; ++k) c += (double)bce_term(o[5 + k], ((pc[k >> 5] >> (k & 31)) & 1u) ? 1.f : 0.f);
            part[3] = c;
        }
    }
    // block reduction (wave shuffles, then 4 waves through LDS), fixed order -> deterministic
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double v = part[q];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0) red[q][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        double v = 0;
        for (int k = 0; k < LOSS_BLOCK / 64; ++k) v += red[threadIdx.x][k];
        w.partials[(long long)blockIdx.x * 4 + threadIdx.x] = v;
    }
}

__global__ __launch_bounds__(256) void yolo_loss_final_kernel(const double* __restrict__ partials, int nblocks,
                                                              double* __restrict__ out) {
    __shared__ double red[4][256];
    double v[4] = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < nblocks; i += 256)
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += partials[(long long)i * 4 + q];
#pragma unroll
    for (int q = 0; q < 4; ++q) red[q][threadIdx.x] = v[q];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s)
#pragma unroll
            for (int q = 0; q < 4; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x < 4) out[threadIdx.x] = red[threadIdx.x][0];
}

// grad wrt `output`, dense: thread per element
__global__ __launch_bounds__(256) void yolo_loss_bwd_kernel(const float* __restrict__ output, int masked,
                                                            const float* __restrict__ obj_mask,
                                                            const float* __restrict__ gscale_p,
                                                            float* __restrict__ g_out, int B, int F, int A, int K,
                                                            int C, int cw, LossPtrs w) {
    const int n_ch = 5 + C;
    const long long total = (long long)B * A * F * F * n_ch;
    const int per_img = A * F * F;
    const float gscale = *gscale_p;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const long long cell = e / n_ch;
        const int ch = (int)(e - cell * n_ch);
        const int rec = w.pos_index[cell];
        float g = 0.f;
        if (ch == 4) {
            if (obj_mask[cell] != 0.f) g = bce_grad(output[e], rec >= 0 ? 1.f : 0.f);
        } else if (rec >= 0) {
            const int b = (int)(cell / per_img);
            const float* pv = w.pos_val + ((long long)b * K + rec) * 5;
            const float s = pv[4];
            const float o = output[e];
            if (ch < 2) g = (s * s) * bce_grad(o, pv[ch]);
            else if (ch < 4) g = ((masked ? o : o * s) - pv[ch] * s) * s;
            else {
                const int k = ch - 5;
                const unsigned bit = (w.pos_cls[((long long)b * K + rec) * cw + (k >> 5)] >> (k & 31)) & 1u;
                g = bce_grad(o, bit ? 1.f : 0.f);
            }
        }
        g_out[e] = g * gscale;
    }
}

// yololoss.py:402-407: output[...,4]*=obj_mask; output[...,{0-3,5..}]*=tgt_mask; output[...,2:4]*=tgt_scale
__global__ __launch_bounds__(256) void yolo_loss_mask_output_kernel(float* __restrict__ output,
                                                                    const float* __restrict__ obj_mask,
                                                                    int B, int F, int A, int K, int C, LossPtrs w) {
    const int n_ch = 5 + C;
    const long long total = (long long)B * A * F * F * n_ch;
    const int per_img = A * F * F;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const long long cell = e / n_ch;
        const int ch = (int)(e - cell * n_ch);
        const int rec = w.pos_index[cell];
        float v = output[e];
        if (ch == 4) v = v * obj_mask[cell];
        else {
            v = v * (rec >= 0 ? 1.f : 0.f);
            if (ch == 2 || ch == 3) {
                const int b = (int)(cell / per_img);
                v = v * (rec >= 0 ? w.pos_val[((long long)b * K + rec) * 5 + 4] : 0.f);
            }
        }
        output[e] = v;
    }
}

__global__ void yolo_dense_targets_kernel(float* __restrict__ target, float* __restrict__ tgt_mask,
                                          float* __restrict__ tgt_scale, int B, int F, int A, int K, int C, int cw,
                                          LossPtrs w) {
    // one thread per (image, record)
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * K) return;
    const int b = idx / K, rec = idx - b * K;
    if (rec >= w.npos[b]) return;
    const long long cell = (long long)b * A * F * F + w.pos_cell[idx];
    const int n_ch = 5 + C;
    const float* pv = w.pos_val + (long long)idx * 5;
    float* t = target + cell * n_ch;
    t[0] = pv[0]; t[1] = pv[1]; t[2] = pv[2]; t[3] = pv[3]; t[4] = 1.f;
    for (int k = 0; k < C; ++k)
        if ((w.pos_cls[(long long)idx * cw + (k >> 5)] >> (k & 31)) & 1u) t[5 + k] = 1.f;
    float* tm = tgt_mask + cell * (4 + C);
    for (int k = 0; k < 4 + C; ++k) tm[k] = 1.f;
    tgt_scale[cell * 2] = pv[4];
    tgt_scale[cell * 2 + 1] = pv[4];
}

// ------------------------------------------------------------------------------------ postprocess
// The prediction rows are [5 + C] floats (340 B at C = 80): one thread per row reads 4 B from 64 different lines per
// load (measured: 25x the tensor's bytes on the fabric, 276 GB/s).  The tiled forms below copy POST_ROWS consecutive rows
// -- a contiguous span -- into LDS with coalesced loads (row pitch forced odd: conflict-free column walks) and let
// thread (row, class) pairs read them from there; candidate counts go through an LDS histogram first.
constexpr int POST_ROWS = 64;
constexpr int POST_MAX_NCH = 255;                          // 64 rows x 255 floats = 63.8 KB of LDS

__device__ __forceinline__ void post_load_tile(const float* __restrict__ pred, long long row0, int rows, int n_ch,
                                               int pitch, float* __restrict__ tile) {
    const float* src = pred + row0 * n_ch;
    const int total = rows * n_ch;
    int e = threadIdx.x;
    int r = e / n_ch, c = e - r * n_ch;
    const int dr = 256 / n_ch, dc = 256 - dr * n_ch;
    for (; e < total; e += 256) {
        tile[r * pitch + c] = src[e];
        r += dr; c += dc;
        if (c >= n_ch) { c -= n_ch; ++r; }
    }
}

__global__ __launch_bounds__(256) void post_count_tiled_kernel(float* __restrict__ pred, int B, long long N, int C,
                                                               float conf, int convert, int* __restrict__ counts) {
    extern __shared__ float post_smem[];
    const int n_ch = 5 + C, pitch = n_ch | 1;
    float* tile = post_smem;
    int* hist = reinterpret_cast<int*>(post_smem + POST_ROWS * pitch);       // [2][C]: the tile's first / second image
    const long long total = (long long)B * N;
    const long long ntiles = (total + POST_ROWS - 1) / POST_ROWS;
    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const long long row0 = t * POST_ROWS;
        const int rows = (int)((total - row0) < POST_ROWS ? (total - row0) : POST_ROWS);
        post_load_tile(pred, row0, rows, n_ch, pitch, tile);
        for (int i = threadIdx.x; i < 2 * C; i += 256) hist[i] = 0;
        __syncthreads();
        const int b0 = (int)(row0 / N);
        const long long b0_end = (long long)(b0 + 1) * N;                     // rows >= b0_end belong to image b0 + 1 ...
        if (convert && threadIdx.x < rows) {                                  // utils.py:117-126, written back in place
            float* q = tile + threadIdx.x * pitch;
            const float x = q[0], y = q[1], w = q[2], h = q[3];
            float* p = pred + (row0 + threadIdx.x) * n_ch;
            p[0] = x - w / 2.f; p[1] = y - h / 2.f; p[2] = x + w / 2.f; p[3] = y + h / 2.f;
        }
        for (int e = threadIdx.x; e < POST_ROWS * C; e += 256) {
            const int r = e & (POST_ROWS - 1), c = e / POST_ROWS;
            if (r < rows) {
                const float* q = tile + r * pitch;
                if (q[5 + c] * q[4] >= conf) {
                    // ... or a later one when N < POST_ROWS: those go straight to the global counter
                    const long long i = row0 + r;
                    if (i < b0_end) atomicAdd(&hist[c], 1);
                    else if (i < b0_end + N) atomicAdd(&hist[C + c], 1);
                    else atomicAdd(&counts[(int)(i / N) * C + c], 1);
                }
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * C; i += 256) {
            const int v = hist[i];
            const int b = b0 + (i >= C ? 1 : 0);
            if (v && b < B) atomicAdd(&counts[b * C + (i >= C ? i - C : i)], v);
        }
        __syncthreads();
    }
}

__device__ __forceinline__ unsigned long long make_key(float score, unsigned idx);

__global__ __launch_bounds__(256) void post_fill_tiled_kernel(const float* __restrict__ pred, int B, long long N, int C,
                                                              float conf, const int* __restrict__ seg_off,
                                                              int* __restrict__ cursor,
                                                              unsigned long long* __restrict__ keys) {
    extern __shared__ float post_smem[];
    const int n_ch = 5 + C, pitch = n_ch | 1;
    float* tile = post_smem;
    const long long total = (long long)B * N;
    const long long ntiles = (total + POST_ROWS - 1) / POST_ROWS;
    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const long long row0 = t * POST_ROWS;
        const int rows = (int)((total - row0) < POST_ROWS ? (total - row0) : POST_ROWS);
        post_load_tile(pred, row0, rows, n_ch, pitch, tile);
        __syncthreads();
        for (int e = threadIdx.x; e < POST_ROWS * C; e += 256) {
            const int r = e & (POST_ROWS - 1), c = e / POST_ROWS;
            if (r < rows) {
                const float* q = tile + r * pitch;
                const float obj = q[4], cc = q[5 + c];
                if (cc * obj >= conf) {
                    const long long i = row0 + r;
                    const int b = (int)(i / N);
                    const int seg = b * C + c;
                    const int slot = atomicAdd(&cursor[seg], 1);
                    // (a segment truncated by a too-small candidate buffer takes what fits; the host re-runs with room)
                    if (slot < seg_off[seg + 1] - seg_off[seg])
                        keys[seg_off[seg] + slot] = make_key(obj * cc, (unsigned)(i - (long long)b * N));   // utils.py:209
                }
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void post_count_kernel(float* __restrict__ pred, int B, long long N, int C,
                                                         float conf, int convert, int* __restrict__ counts) {
    const long long total = (long long)B * N;
    const int n_ch = 5 + C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        float* p = pred + i * n_ch;
        if (convert) {                                           // utils.py:117-126
            const float x = p[0], y = p[1], w = p[2], h = p[3];
            p[0] = x - w / 2.f; p[1] = y - h / 2.f; p[2] = x + w / 2.f; p[3] = y + h / 2.f;
        }
        const int b = (int)(i / N);
        const float obj = p[4];
        for (int c = 0; c < C; ++c)
            if (p[5 + c] * obj >= conf) atomicAdd(&counts[b * C + c], 1);
    }
}

// key: high 32 = ~ordered(score) (so ascending key = descending score), low 32 = box index
__device__ __forceinline__ unsigned long long make_key(float score, unsigned idx) {
    unsigned u = __float_as_uint(score);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);            // total order on floats
    return ((unsigned long long)(~u) << 32) | idx;
}

__global__ __launch_bounds__(256) void post_fill_kernel(const float* __restrict__ pred, int B, long long N, int C,
                                                        float conf, const int* __restrict__ seg_off,
                                                        int* __restrict__ cursor, unsigned long long* __restrict__ keys) {
    const long long total = (long long)B * N;
    const int n_ch = 5 + C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const float* p = pred + i * n_ch;
        const int b = (int)(i / N);
        const unsigned box = (unsigned)(i - (long long)b * N);
        const float obj = p[4];
        for (int c = 0; c < C; ++c) {
            const float cc = p[5 + c];
            if (cc * obj >= conf) {
                const int seg = b * C + c;
                const int slot = atomicAdd(&cursor[seg], 1);
                if (slot < seg_off[seg + 1] - seg_off[seg])
                    keys[seg_off[seg] + slot] = make_key(obj * cc, box);  // utils.py:209 score = obj*cls_conf
            }
        }
    }
}

// Block-wide bitonic sort of n keys in global memory, n arbitrary.  All-ascending network (the
// first step of every merge mirrors the upper half), so virtual +inf padding at indices >= n
// never moves and a compare-exchange whose upper index is >= n is simply skipped.
__device__ void block_bitonic_sort(unsigned long long* k, int n) {
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    for (int size = 2; size <= np2; size <<= 1) {
        const int half = size >> 1;
        for (int t = threadIdx.x; t < np2 / 2; t += blockDim.x) {
            const int blk = t / half, off = t - blk * half;
            const int lo = blk * size + off, hi = blk * size + size - 1 - off;
            if (hi < n) {
                const unsigned long long a = k[lo], b = k[hi];
                if (a > b) { k[lo] = b; k[hi] = a; }
            }
        }
        __syncthreads();
        for (int stride = half >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < np2 / 2; t += blockDim.x) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                if (hi < n) {
                    const unsigned long long a = k[lo], b = k[hi];
                    if (a > b) { k[lo] = b; k[hi] = a; }
                }
            }
            __syncthreads();
        }
    }
}

// utils.py:64-81 greedy suppression on a sorted list.  box(i): xyxy of sorted candidate i.
// Chunks of 256 candidates: (A) test against everything kept so far, (B) 256x256 bitmask of
// intra-chunk suppression + a serial scan by one thread.  keep_flag[i] in sorted order.
struct NmsShared {
    float4 kb[256];
    float ka[256];
    unsigned long long mask[256][4];
    unsigned long long alive[4];
    int kept_in_chunk[256];
    int nk;
};

template <typename BoxFn>
__device__ int block_greedy_nms(int n, float thresh, int limit, BoxFn box_of, float4* kept_box, float* kept_area,
                                int* kept_sorted_pos, NmsShared& sh) {
    int kept = 0;                                               // uniform
    for (int s = 0; s < n && (limit <= 0 || kept < limit); s += 256) {
        const int i = s + threadIdx.x;
        const bool valid = i < n;
        float4 bx = make_float4(0, 0, 0, 0);
        if (valid) bx = box_of(i);
        const float area = (bx.z - bx.x) * (bx.w - bx.y);
        bool alive = valid;
        // (A) against earlier kept boxes, staged through LDS 256 at a time
        for (int k0 = 0; k0 < kept; k0 += 256) {
            __syncthreads();
            if (k0 + (int)threadIdx.x < kept) { sh.kb[threadIdx.x] = kept_box[k0 + threadIdx.x]; sh.ka[threadIdx.x] = kept_area[k0 + threadIdx.x]; }
            __syncthreads();
            const int kn = min(256, kept - k0);
            if (alive)
                for (int k = 0; k < kn; ++k) {
                    const float4 o = sh.kb[k];
                    const float tlx = fmaxf(bx.x, o.x), tly = fmaxf(bx.y, o.y);
                    const float brx = fminf(bx.z, o.z), bry = fminf(bx.w, o.w);
                    const float en = (tlx < brx && tly < bry) ? 1.f : 0.f;
                    const float inter = ((brx - tlx) * (bry - tly)) * en;
                    const float iou = inter / ((area + sh.ka[k]) - inter);
                    if (iou >= thresh) { alive = false; break; }
                }
        }
        // (B) intra-chunk
        __syncthreads();
        sh.kb[threadIdx.x] = bx;
        sh.ka[threadIdx.x] = area;
        if (threadIdx.x < 4) sh.alive[threadIdx.x] = 0ull;
        __syncthreads();
        {
            const unsigned long long bal = __ballot(alive);
            if ((threadIdx.x & 63) == 0) sh.alive[threadIdx.x >> 6] = bal;
        }
        unsigned long long mk[4] = {0, 0, 0, 0};
        if (alive)
            for (int u = threadIdx.x + 1; u < 256 && s + u < n; ++u) {
                const float4 o = sh.kb[u];
                // candidate u is tested against already-selected t: area_u + area_t (utils.py:76)
                const float tlx = fmaxf(o.x, bx.x), tly = fmaxf(o.y, bx.y);
                const float brx = fminf(o.z, bx.z), bry = fminf(o.w, bx.w);
                const float en = (tlx < brx && tly < bry) ? 1.f : 0.f;
                const float inter = ((brx - tlx) * (bry - tly)) * en;
                const float iou = inter / ((sh.ka[u] + area) - inter);
                if (iou >= thresh) mk[u >> 6] |= 1ull << (u & 63);
            }
#pragma unroll
        for (int q = 0; q < 4; ++q) sh.mask[threadIdx.x][q] = mk[q];
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long al[4] = {sh.alive[0], sh.alive[1], sh.alive[2], sh.alive[3]};
            int nk = 0;
            int room = limit > 0 ? limit - kept : 0x7fffffff;
            for (int t = 0; t < 256 && room > 0; ++t) {
                if ((al[t >> 6] >> (t & 63)) & 1ull) {
                    sh.kept_in_chunk[nk++] = t;
                    --room;
#pragma unroll
                    for (int q = 0; q < 4; ++q) al[q] &= ~sh.mask[t][q];
                }
            }
            sh.nk = nk;
        }
        __syncthreads();
        const int nk = sh.nk;
        if ((int)threadIdx.x < nk) {
            const int t = sh.kept_in_chunk[threadIdx.x];
            kept_box[kept + threadIdx.x] = sh.kb[t];
            kept_area[kept + threadIdx.x] = sh.ka[t];
            kept_sorted_pos[kept + threadIdx.x] = s + t;
        }
        kept += nk;
        __syncthreads();
    }
    return kept;
}

constexpr int NMS_LDS_KEYS = 2048;

__global__ __launch_bounds__(256) void post_nms_kernel(const float* __restrict__ pred, long long N, int C,
                                                       float thresh, const int* __restrict__ seg_off,
                                                       unsigned long long* __restrict__ keys,
                                                       float4* __restrict__ kbox, float* __restrict__ karea,
                                                       int* __restrict__ kpos, float* __restrict__ det_rows,
                                                       int* __restrict__ kept_out) {
    __shared__ NmsShared sh;
    const int seg = blockIdx.x;
    const int off = seg_off[seg];
    const int n = seg_off[seg + 1] - off;
    if (n == 0) { if (threadIdx.x == 0) kept_out[seg] = 0; return; }
    const int b = seg / C, c = seg - b * C;
    const int n_ch = 5 + C;
    unsigned long long* k = keys + off;
    if (n <= NMS_LDS_KEYS) {
        // the usual segment (a few hundred candidates): the bitonic network runs on an LDS copy of the keys
        __shared__ unsigned long long skeys[NMS_LDS_KEYS];
        for (int i = threadIdx.x; i < n; i += blockDim.x) skeys[i] = k[i];
        __syncthreads();
        block_bitonic_sort(skeys, n);
        for (int i = threadIdx.x; i < n; i += blockDim.x) k[i] = skeys[i];
        __syncthreads();
    } else {
        block_bitonic_sort(k, n);
    }
    const float* img = pred + (long long)b * N * n_ch;
    auto box_of = [&](int i) {
        const unsigned idx = (unsigned)(k[i] & 0xffffffffull);
        const float* p = img + (long long)idx * n_ch;
        return make_float4(p[0], p[1], p[2], p[3]);
    };
    const int kept = block_greedy_nms(n, thresh, 0, box_of, kbox + off, karea + off, kpos + off, sh);
    for (int r = threadIdx.x; r < kept; r += blockDim.x) {
        const unsigned idx = (unsigned)(k[kpos[off + r]] & 0xffffffffull);
        const float* p = img + (long long)idx * n_ch;
        float* o = det_rows + (long long)(off + r) * 7;
        o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; o[3] = p[3]; o[4] = p[4]; o[5] = p[5 + c]; o[6] = (float)c;
    }
    if (threadIdx.x == 0) kept_out[seg] = kept;
}

__global__ __launch_bounds__(256) void nms_keys_kernel(const float* __restrict__ scores, long long R,
                                                       unsigned long long* __restrict__ keys) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < R; i += (long long)gridDim.x * blockDim.x)
        keys[i] = scores ? make_key(scores[i], (unsigned)i) : (unsigned long long)i;
}

__global__ __launch_bounds__(256) void nms_single_kernel(const float* __restrict__ boxes, int R, float thresh, int limit,
                                                         unsigned long long* __restrict__ keys, float4* kbox,
                                                         float* karea, int* kpos, int* __restrict__ keep_idx,
                                                         int* __restrict__ n_keep) {
    __shared__ NmsShared sh;
    block_bitonic_sort(keys, R);
    auto box_of = [&](int i) {
        const unsigned idx = (unsigned)(keys[i] & 0xffffffffull);
        return make_float4(boxes[idx * 4], boxes[idx * 4 + 1], boxes[idx * 4 + 2], boxes[idx * 4 + 3]);
    };
    const int kept = block_greedy_nms(R, thresh, limit, box_of, kbox, karea, kpos, sh);
    for (int r = threadIdx.x; r < kept; r += blockDim.x) keep_idx[r] = (int)(keys[kpos[r]] & 0xffffffffull);
    if (threadIdx.x == 0) *n_keep = kept;
}

// ---- device-side prefix sums of the postprocess: segment offsets from the candidate counts, output offsets from the
// survivor counts.  One block; thread t owns a run of consecutive segments, runs are combined by an LDS scan.
__device__ __forceinline__ long long block_exclusive_scan_1024(long long v, long long* sh) {
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const long long add = t >= d ? sh[t - d] : 0;
        __syncthreads();
        sh[t] += add;
        __syncthreads();
    }
    return sh[t] - v;
}
__global__ __launch_bounds__(1024) void post_scan_kernel(const int* __restrict__ counts, int nseg, long long cap,
                                                         int* __restrict__ seg_off, int* __restrict__ info) {
    __shared__ long long sh[1024];
    const int per = (nseg + 1023) / 1024;
    const int s0 = threadIdx.x * per, s1 = min(nseg, s0 + per);
    long long mine = 0;
    for (int s = s0; s < s1; ++s) mine += counts[s];
    long long run = block_exclusive_scan_1024(mine, sh);
    for (int s = s0; s < s1; ++s) {
        seg_off[s] = (int)(run < cap ? run : cap);         // clamped: a too-small buffer truncates, never overflows
        run += counts[s];
    }
    if (threadIdx.x == 1023) {
        const long long total = sh[1023];
        seg_off[nseg] = (int)(total < cap ? total : cap);
        info[0] = (int)(total < 0x7fffffffll ? total : 0x7fffffffll);
        info[1] = total > cap ? 1 : 0;
    }
}
__global__ __launch_bounds__(1024) void post_out_scan_kernel(const int* __restrict__ kept, int nseg, int C, int B,
                                                             int* __restrict__ out_off, int* __restrict__ img_off) {
    __shared__ long long sh[1024];
    const int per = (nseg + 1023) / 1024;
    const int s0 = threadIdx.x * per, s1 = min(nseg, s0 + per);
    long long mine = 0;
    for (int s = s0; s < s1; ++s) mine += kept[s];
    long long run = block_exclusive_scan_1024(mine, sh);
    for (int s = s0; s < s1; ++s) {
        out_off[s] = (int)run;
        if (s % C == 0) img_off[s / C] = (int)run;
        run += kept[s];
    }
    if (threadIdx.x == 1023) { out_off[nseg] = (int)sh[1023]; img_off[B] = (int)sh[1023]; }
}
// rows of segment s: det_rows[seg_off[s] .. + kept[s]) -> out_rows[out_off[s] ..): class asc, score desc per image
__global__ __launch_bounds__(64) void post_compact_kernel(const float* __restrict__ det_rows, const int* __restrict__ seg_off,
                                                          const int* __restrict__ kept, const int* __restrict__ out_off,
                                                          float* __restrict__ out_rows) {
    const int s = blockIdx.x;
    const int n = kept[s] * 7;
    const float* src = det_rows + (long long)seg_off[s] * 7;
    float* dst = out_rows + (long long)out_off[s] * 7;
    for (int i = threadIdx.x; i < n; i += 64) dst[i] = src[i];
}

struct PostWs { size_t keys, cursor, kbox, karea, kpos, total; };
static PostWs post_ws(long long total, int nseg) {
    PostWs w;
    size_t o = 0;
    const size_t t = (size_t)(total > 0 ? total : 1);
    w.keys = o; o = al16(o + t * 8);
    w.cursor = o; o = al16(o + (size_t)(nseg > 0 ? nseg : 1) * 4);
    w.kbox = o; o = al16(o + t * 16);
    w.karea = o; o = al16(o + t * 4);
    w.kpos = o; o = al16(o + t * 4);
    w.total = o;
    return w;
}

inline int grid_for(long long total) {
    long long b = (total + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}
static bool fill_anchors(Anchors& a, const float* host_wh, int n) {
    if (!host_wh || n < 1 || n > 16) return false;
    for (int i = 0; i < 16; ++i) { a.w[i] = 1.f; a.h[i] = 1.f; }
    for (int i = 0; i < n; ++i) { a.w[i] = host_wh[2 * i]; a.h[i] = host_wh[2 * i + 1]; }
    return true;
}

}  // namespace

extern "C" {

int y4_bboxes_iou_f32(const float* boxes_a, long long Na, const float* boxes_b, long long Nb, int xyxy, float* iou,
                      void* stream) {
    if (Na < 0 || Nb < 0) return Y4_ERR_SHAPE;
    if (Na == 0 || Nb == 0) return Y4_OK;
    if (!boxes_a || !boxes_b || !iou) return Y4_ERR_NULL;
    long long blocks = (Na * Nb + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(bboxes_iou_kernel, dim3((unsigned)blocks), dim3(256), 0, y4_stream(stream), boxes_a, Na, boxes_b, Nb,
                       xyxy, iou);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

int y4_yolo_decode_train_f32(const float* logits, int ldl, float* output, float* pred,
                             int B, int F, int A, int n_classes, const float* anchors_wh_host, void* stream) {
    if (!logits || !output || !pred) return Y4_ERR_NULL;
    const int n_ch = 5 + n_classes;
    Anchors anc;
    if (B <= 0 || F <= 0 || A <= 0 || n_classes <= 0 || ldl < A * n_ch || !fill_anchors(anc, anchors_wh_host, A))
        return Y4_ERR_SHAPE;
    const long long npix = (long long)B * F * F;
    const int tp = decode_tp(npix);
    const long long tpi = ((long long)F * F + tp - 1) / tp;
    if (decode_tiled_ok(logits, ldl, A, n_ch, output, pred) && (long long)B * tpi < (1ll << 31)) {
        auto kern = tp == 16 ? yolo_decode_tiled_kernel<false, 16> : yolo_decode_tiled_kernel<false, 32>;
        hipLaunchKernelGGL(kern, dim3((unsigned)(B * tpi)), dim3(256), 0, y4_stream(stream), logits,
                           (long long)ldl, output, pred, 0ll, 0ll, 1.0f, F, A, n_ch, anc, (int)tpi);
        Y4_CHECK_LAUNCH();
        return Y4_OK;
    }
    hipLaunchKernelGGL(yolo_decode_kernel<false>, dim3((unsigned)(npix < 8192 ? npix : 8192)), dim3(256), 0,
                       y4_stream(stream), logits, (long long)ldl, output, pred, 0ll, 0ll, 1.0f, B, F, A, n_ch, anc);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

int y4_yolo_decode_eval_f32(const float* logits, int ldl, float* out, long long n_total, long long box_off,
                            int B, int F, int A, int n_classes, const float* anchors_wh_host,
                            float stride, void* stream) {
    if (!logits || !out) return Y4_ERR_NULL;
    const int n_ch = 5 + n_classes;
    Anchors anc;
    if (B <= 0 || F <= 0 || A <= 0 || n_classes <= 0 || ldl < A * n_ch || !fill_anchors(anc, anchors_wh_host, A))
        return Y4_ERR_SHAPE;
    if (box_off < 0 || box_off + (long long)A * F * F > n_total) return Y4_ERR_SHAPE;
    const long long npix = (long long)B * F * F;
    const int tp = decode_tp(npix);
    const long long tpi = ((long long)F * F + tp - 1) / tp;
    if (decode_tiled_ok(logits, ldl, A, n_ch, out, nullptr) && (long long)B * tpi < (1ll << 31)) {
        auto kern = tp == 16 ? yolo_decode_tiled_kernel<true, 16> : yolo_decode_tiled_kernel<true, 32>;
        hipLaunchKernelGGL(kern, dim3((unsigned)(B * tpi)), dim3(256), 0, y4_stream(stream), logits,
                           (long long)ldl, out, (float*)nullptr, n_total, box_off, stride, F, A, n_ch, anc, (int)tpi);
        Y4_CHECK_LAUNCH();
        return Y4_OK;
    }
    hipLaunchKernelGGL(yolo_decode_kernel<true>, dim3((unsigned)(npix < 8192 ? npix : 8192)), dim3(256), 0,
                       y4_stream(stream), logits, (long long)ldl, out, (float*)nullptr, n_total, box_off, stride, B, F,
                       A, n_ch, anc);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

int y4_yolo_decode_bwd_f32(const float* logits, int ldl, const float* g_output, const float* g_pred,
                           float* g_logits, int B, int F, int A, int n_classes,
                           const float* anchors_wh_host, void* stream) {
    if (!logits || !g_logits) return Y4_ERR_NULL;
    const int n_ch = 5 + n_classes;
    Anchors anc;
    if (B <= 0 || F <= 0 || A <= 0 || n_classes <= 0 || ldl < A * n_ch || !fill_anchors(anc, anchors_wh_host, A))
        return Y4_ERR_SHAPE;
    const long long npix = (long long)B * F * F;
    hipLaunchKernelGGL(yolo_decode_bwd_kernel, dim3((unsigned)(npix < 8192 ? npix : 8192)), dim3(256), 0,
                       y4_stream(stream), logits, (long long)ldl, g_output, g_pred, g_logits, B, F, A, n_ch, anc);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

size_t y4_yolo_loss_workspace(int B, int F, int A, int K, int n_classes) {
    if (B <= 0 || F <= 0 || A <= 0 || K <= 0 || n_classes <= 0) return 0;
    return loss_ws(B, F, A, K, n_classes).total;
}

int y4_yolo_loss_fwd_f32(const float* output, const float* pred, const float* labels, int K,
                         int B, int F, int A, int n_classes, float stride, float ignore_thresh,
                         const float* all_anchors_host, int n_all_anchors, const int* anch_mask_host,
                         float* obj_mask, double* loss_parts,
                         void* workspace, size_t workspace_bytes, void* stream) {
    if (!output || !pred || !labels || !obj_mask || !loss_parts || !workspace || !anch_mask_host) return Y4_ERR_NULL;
    if (B <= 0 || F <= 0 || A != 3 || K <= 0 || n_classes <= 0 || (long long)A * F * F >= (1ll << 31)) return Y4_ERR_SHAPE;
    Anchors all, masked;
    if (!fill_anchors(all, all_anchors_host, n_all_anchors)) return Y4_ERR_SHAPE;
    float mwh[6];
    for (int a = 0; a < 3; ++a) {
        const int id = anch_mask_host[a];
        if (id < 0 || id >= n_all_anchors) return Y4_ERR_SHAPE;
        mwh[2 * a] = all_anchors_host[2 * id]; mwh[2 * a + 1] = all_anchors_host[2 * id + 1];
    }
    fill_anchors(masked, mwh, 3);
    const LossWs l = loss_ws(B, F, A, K, n_classes);
    if (workspace_bytes < l.total) return Y4_ERR_WORKSPACE;
    const LossPtrs w = loss_ptrs(workspace, l);
    hipStream_t st = y4_stream(stream);
    if (hipMemsetAsync(w.pos_index, 0xff, (size_t)B * A * F * F * 4, st) != hipSuccess) return Y4_ERR_LAUNCH;
    if (K <= 64) {
        hipLaunchKernelGGL(yolo_assign_wave_kernel, dim3(B), dim3(64), 0, st, labels, K, B, F, A, n_classes, l.cw,
                           stride, all, n_all_anchors, anch_mask_host[0], anch_mask_host[1], anch_mask_host[2], masked, w);
    } else {
        hipLaunchKernelGGL(yolo_assign_kernel, dim3((B + 63) / 64), dim3(64), 0, st, labels, K, B, F, A, n_classes, l.cw,
                           stride, all, n_all_anchors, anch_mask_host[0], anch_mask_host[1], anch_mask_host[2], masked, w);
    }
    Y4_CHECK_LAUNCH();
    hipLaunchKernelGGL(yolo_loss_dense_kernel, dim3(l.nblocks), dim3(LOSS_BLOCK), 0, st, output, pred, B, F, A, K,
                       n_classes, l.cw, ignore_thresh, obj_mask, w);
    Y4_CHECK_LAUNCH();
    hipLaunchKernelGGL(yolo_loss_final_kernel, dim3(1), dim3(256), 0, st, w.partials, l.nblocks, loss_parts);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

int y4_yolo_loss_bwd_f32(const float* output, int output_is_masked, const float* obj_mask,
                         const float* gscale, float* g_output,
                         int B, int F, int A, int K, int n_classes,
                         const void* workspace, size_t workspace_bytes, void* stream) {
    if (!output || !obj_mask || !g_output || !workspace || !gscale) return Y4_ERR_NULL;
    if (B <= 0 || F <= 0 || A <= 0 || K <= 0 || n_classes <= 0) return Y4_ERR_SHAPE;
    const LossWs l = loss_ws(B, F, A, K, n_classes);
    if (workspace_bytes < l.total) return Y4_ERR_WORKSPACE;
    const LossPtrs w = loss_ptrs(workspace, l);
    const long long total = (long long)B * A * F * F * (5 + n_classes);
    hipLaunchKernelGGL(yolo_loss_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, y4_stream(stream), output,
                       output_is_masked, obj_mask,
                       gscale, g_output, B, F, A, K, n_classes, l.cw, w);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

int y4_yolo_loss_mask_output_f32(float* output, const float* obj_mask, int B, int F, int A, int K,
                                 int n_classes, const void* workspace, size_t workspace_bytes, void* stream) {
    if (!output || !obj_mask || !workspace) return Y4_ERR_NULL;
    if (B <= 0 || F <= 0 || A <= 0 || K <= 0 || n_classes <= 0) return Y4_ERR_SHAPE;
    const LossWs l = loss_ws(B, F, A, K, n_classes);
    if (workspace_bytes < l.total) return Y4_ERR_WORKSPACE;
    const LossPtrs w = loss_ptrs(workspace, l);
    const long long total = (long long)B * A * F * F * (5 + n_classes);
    hipLaunchKernelGGL(yolo_loss_mask_output_kernel, dim3(grid_for(total)), dim3(256), 0, y4_stream(stream), output,
                       obj_mask, B, F, A, K, n_classes, w);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

int y4_yolo_loss_dense_targets_f32(float* target, float* tgt_mask, float* tgt_scale,
                                   int B, int F, int A, int K, int n_classes,
                                   const void* workspace, size_t workspace_bytes, void* stream) {
    if (!target || !tgt_mask || !tgt_scale || !workspace) return Y4_ERR_NULL;
    if (B <= 0 || F <= 0 || A <= 0 || K <= 0 || n_classes <= 0) return Y4_ERR_SHAPE;
    const LossWs l = loss_ws(B, F, A, K, n_classes);
    if (workspace_bytes < l.total) return Y4_ERR_WORKSPACE;
    const LossPtrs w = loss_ptrs(workspace, l);
    hipStream_t st = y4_stream(stream);
    const size_t cells = (size_t)B * A * F * F;
    if (hipMemsetAsync(target, 0, cells * (5 + n_classes) * 4, st) != hipSuccess) return Y4_ERR_LAUNCH;
    if (hipMemsetAsync(tgt_mask, 0, cells * (4 + n_classes) * 4, st) != hipSuccess) return Y4_ERR_LAUNCH;
    if (hipMemsetAsync(tgt_scale, 0, cells * 2 * 4, st) != hipSuccess) return Y4_ERR_LAUNCH;
    hipLaunchKernelGGL(yolo_dense_targets_kernel, dim3((B * K + 255) / 256), dim3(256), 0, st, target, tgt_mask,
                       tgt_scale, B, F, A, K, n_classes, l.cw, w);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

int y4_post_count_f32(float* prediction, int B, long long N, int n_classes, float conf_thre,
                      int convert_xyxy, int* counts, void* stream) {
    if (!prediction || !counts) return Y4_ERR_NULL;
    if (B <= 0 || N <= 0 || n_classes <= 0) return Y4_ERR_SHAPE;
    hipStream_t st = y4_stream(stream);
    if (hipMemsetAsync(counts, 0, (size_t)B * n_classes * 4, st) != hipSuccess) return Y4_ERR_LAUNCH;
    const int pitch = (5 + n_classes) | 1;
    const size_t smem = (size_t)POST_ROWS * pitch * 4 + (size_t)2 * n_classes * 4;     // tile + the two histograms
    if (5 + n_classes <= POST_MAX_NCH && smem <= 65536) {                             // (64 KiB: the default dynamic-LDS limit)
        const long long ntiles = ((long long)B * N + POST_ROWS - 1) / POST_ROWS;
        hipLaunchKernelGGL(post_count_tiled_kernel, dim3((unsigned)(ntiles < 2048 ? ntiles : 2048)), dim3(256), smem, st,
                           prediction, B, N, n_classes, conf_thre, convert_xyxy, counts);
    } else {
        hipLaunchKernelGGL(post_count_kernel, dim3(grid_for((long long)B * N)), dim3(256), 0, st, prediction, B, N,
                           n_classes, conf_thre, convert_xyxy, counts);
    }
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

int y4_post_scan_i32(const int* counts, int n_segments, long long cap, int* seg_offsets, int* info, void* stream) {
    if (!counts || !seg_offsets || !info) return Y4_ERR_NULL;
    if (n_segments <= 0 || cap <= 0 || cap >= (1ll << 31)) return Y4_ERR_SHAPE;
    hipLaunchKernelGGL(post_scan_kernel, dim3(1), dim3(1024), 0, y4_stream(stream), counts, n_segments, cap, seg_offsets, info);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

int y4_post_compact_f32(const float* det_rows, const int* seg_offsets, const int* kept, int B, int n_classes,
                        float* out_rows, int* out_offsets, int* img_offsets, void* stream) {
    if (!det_rows || !seg_offsets || !kept || !out_rows || !out_offsets || !img_offsets) return Y4_ERR_NULL;
    if (B <= 0 || n_classes <= 0) return Y4_ERR_SHAPE;
    const int nseg = B * n_classes;
    hipStream_t st = y4_stream(stream);
    hipLaunchKernelGGL(post_out_scan_kernel, dim3(1), dim3(1024), 0, st, kept, nseg, n_classes, B, out_offsets, img_offsets);
    Y4_CHECK_LAUNCH();
    hipLaunchKernelGGL(post_compact_kernel, dim3(nseg), dim3(64), 0, st, det_rows, seg_offsets, kept, out_offsets, out_rows);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

size_t y4_post_nms_workspace(long long total_candidates, int n_segments) {
    return post_ws(total_candidates, n_segments).total;
}

int y4_post_nms_f32(const float* prediction, int B, long long N, int n_classes, float conf_thre,
                    float nms_thre, const int* seg_offsets, long long total_candidates,
                    float* det_rows, int* kept, void* workspace, size_t workspace_bytes, void* stream) {
    if (!prediction || !seg_offsets || !det_rows || !kept || !workspace) return Y4_ERR_NULL;
    if (B <= 0 || N <= 0 || N >= (1ll << 31) || n_classes <= 0 || total_candidates < 0 ||
        total_candidates >= (1ll << 31)) return Y4_ERR_SHAPE;
    const int nseg = B * n_classes;
    const PostWs l = post_ws(total_candidates, nseg);
    if (workspace_bytes < l.total) return Y4_ERR_WORKSPACE;
    char* base = static_cast<char*>(workspace);
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(base + l.keys);
    int* cursor = reinterpret_cast<int*>(base + l.cursor);
    hipStream_t st = y4_stream(stream);
    if (hipMemsetAsync(cursor, 0, (size_t)nseg * 4, st) != hipSuccess) return Y4_ERR_LAUNCH;
    const int pitch = (5 + n_classes) | 1;
    const size_t smem = (size_t)POST_ROWS * pitch * 4;
    if (5 + n_classes <= POST_MAX_NCH && smem <= 65536) {
        const long long ntiles = ((long long)B * N + POST_ROWS - 1) / POST_ROWS;
        hipLaunchKernelGGL(post_fill_tiled_kernel, dim3((unsigned)(ntiles < 2048 ? ntiles : 2048)), dim3(256), smem, st,
                           prediction, B, N, n_classes, conf_thre, seg_offsets, cursor, keys);
    } else {
        hipLaunchKernelGGL(post_fill_kernel, dim3(grid_for((long long)B * N)), dim3(256), 0, st, prediction, B, N,
                           n_classes, conf_thre, seg_offsets, cursor, keys);
    }
    Y4_CHECK_LAUNCH();
    hipLaunchKernelGGL(post_nms_kernel, dim3(nseg), dim3(256), 0, st, prediction, N, n_classes, nms_thre, seg_offsets,
                       keys, reinterpret_cast<float4*>(base + l.kbox), reinterpret_cast<float*>(base + l.karea),
                       reinterpret_cast<int*>(base + l.kpos), det_rows, kept);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

size_t y4_nms_workspace(long long R) { return post_ws(R, 1).total; }

int y4_nms_f32(const float* boxes, const float* scores, long long R, float thresh, int limit,
               int* keep_idx, int* n_keep, void* workspace, size_t workspace_bytes, void* stream) {
    if (!boxes || !keep_idx || !n_keep || !workspace) return Y4_ERR_NULL;
    if (R <= 0 || R >= (1ll << 30)) return Y4_ERR_SHAPE;
    const PostWs l = post_ws(R, 1);
    if (workspace_bytes < l.total) return Y4_ERR_WORKSPACE;
    char* base = static_cast<char*>(workspace);
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(base + l.keys);
    hipStream_t st = y4_stream(stream);
    hipLaunchKernelGGL(nms_keys_kernel, dim3(grid_for(R)), dim3(256), 0, st, scores, R, keys);
    Y4_CHECK_LAUNCH();
    hipLaunchKernelGGL(nms_single_kernel, dim3(1), dim3(256), 0, st, boxes, (int)R, thresh, limit, keys,
                       reinterpret_cast<float4*>(base + l.kbox), reinterpret_cast<float*>(base + l.karea),
                       reinterpret_cast<int*>(base + l.kpos), keep_idx, n_keep);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

}  // extern "C"
