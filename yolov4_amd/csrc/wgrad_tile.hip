// Filter gradient of the 3x3 layers with FEW channels on LARGE maps (CSP stage 1 / 2: 32 -> 64 at 304^2, stride 1 and
// stride 2 from 608^2; 64 -> 64 at 152^2), f16x2 arithmetic, fp32 operands in HBM (these tensors have fp32 consumers), gfx950.
//
// dW[n][tap][c] = sum_p dy[p][n] x[p s + tap][c] is a GEMM whose K dimension is the pixel.  The split-K kernel
// (conv_wgrad_f16x2) fetches and splits every x pixel once per filter tap (9x) for 32 - 64 MFMA columns and ran these
// layers at 118 - 138 TFLOP/s, 2 - 5x their HBM time.  Here:
//   * a persistent 8-wave block owns 64 output channels x ONE 32-channel chunk of the input x all nine taps: the whole
//     64 x 288 slice of dW stays in its accumulators (72 registers per lane) for the life of the block -- one slab per block,
//     folded by the fixed-order slab reduce (deterministic);
//   * it walks over tiles of TH x 16 output pixels; per tile the dy tile and the x PATCH the nine taps touch are read ONCE,
//     scaled, split into the two fp16 pieces and stored pixel-major in LDS (rows of 128 B = [32 hi | 32 lo], the 32-B segments
//     XOR-swizzled by g(row) = (row >> 1 & 1) | (row >> 3 & 1) << 1 as in conv_planes.hip);
//   * the MFMA fragments (k = 8 consecutive pixels of one channel) come out of that image through `ds_read_b64_tr_b16`, the
//     hardware transpose read; a filter tap is a constant row offset into the patch (stride 2: the patch is stored
//     de-interleaved by column parity, so that the pixels a tap reads for 8 consecutive outputs are 8 consecutive rows);
//   * two LDS images: the next tile's loads are in flight in registers under the MFMAs of the current one, one barrier per tile.
// Wave w: output-channel tiles 2 (w & 1), + 1 (32 of the 64), input-channel half (w >> 1) & 1 (16 of the 32), taps 0-4 or 5-8
// (w >> 2): 30 / 24 MFMAs, 8 + 20 / 16 transposed reads per 32 pixels.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "conv_geom.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float accv __attribute__((ext_vector_type(4)));
typedef __fp16 trh4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

__device__ __host__ __forceinline__ unsigned wt_scale_exp(unsigned amax_bits) {      // as f16x2_scale_exp (conv_f16x2.hip)
    const unsigned e = (amax_bits >> 23) & 0xffu;
    if (e == 0u || e == 255u) return 127u;
    int se = 268 - (int)e;
    if (se < 2) se = 2;
    if (se > 252) se = 252;
    return (unsigned)se;
}

struct TileWgradGeom {
    const float* x; const float* dy; float* out;      // out: [slabs][Cout][9 Cin]
    long long ldx, lddy;
    int B, H, W, Cin, Ho, Wo, Cout;                   // H, W: input map; Ho, Wo: dy map
    int tiles_y, tiles_x, tiles;                      // tiles of TH x 16 output pixels per image, and in all (B tiles_y tiles_x)
    int groups_c, groups_n, bpg;                      // Cin / 32, Cout / 64, blocks per group
    const unsigned* x_amax; const unsigned* dy_amax;
};

constexpr int WROW = 128;

// row -> byte offset of logical 32-B segment s of that row
__device__ __forceinline__ int seg_addr(int row, int s) {
    const int g = ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
    return row * WROW + ((s ^ g) << 5);
}

template <int S>
struct TileShape {
    static constexpr int TH = S == 1 ? 8 : 4;                       // output rows per tile (16 columns)
    static constexpr int KS = TH / 2;                                // K-steps of 32 pixels (2 rows x 16) per tile
    static constexpr int PH = S == 1 ? TH + 2 : 2 * TH + 1;          // patch rows
    static constexpr int PW = S == 1 ? 18 : 33;                      // patch columns
    // LDS row stride of a patch row (stride 2: 17 even + 17 odd slots, padded).  Chosen so that a K-step (2 output rows = 2 or
    // 4 patch rows) advances by a multiple of 16 LDS rows: the segment swizzle g(row) is then the same in every K-step and a
    // lane's fragment addresses are its K-step-0 addresses plus a constant
    static constexpr int PWS = S == 1 ? 24 : 36;
    static constexpr int ODD0 = S == 1 ? 0 : 18;                     // stride 2: first slot of the odd columns inside a patch row
    static constexpr int XROWS = (PH * PWS + 7) / 8 * 8;
    static constexpr int DROWS = 2 * TH * 16;                        // two 32-channel chunks of the dy tile
    static constexpr int IMG = (XROWS + DROWS) * WROW;               // one LDS image
    static constexpr int NX4 = PH * PW * 8;                          // float4 slots of the patch (32 channels = 8 float4 per pixel)
    static constexpr int ND4 = TH * 16 * 16;                         // float4 slots of the dy tile (64 channels)
    static constexpr int NSLOT = (NX4 + ND4 + 511) / 512;            // float4 per thread and tile
};

template <int S>
__global__ __launch_bounds__(512, 1) void wgrad_tile_f16x2(const TileWgradGeom g) {
    using T = TileShape<S>;
    constexpr int TH = T::TH, KS = T::KS, PW = T::PW, PWS = T::PWS, XROWS = T::XROWS, IMG = T::IMG;
    constexpr int NX4 = T::NX4, ND4 = T::ND4, NSLOT = T::NSLOT;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ngroups = g.groups_c * g.groups_n;
    const int grp_id = blockIdx.x % ngroups, bi = blockIdx.x / ngroups;       // the groups of a tile range run side by side (shared reads)
    const int gc = grp_id % g.groups_c, gn = grp_id / g.groups_c;
    const int c0 = gc * 32, n0 = gn * 64;

    const float sx = __uint_as_float(wt_scale_exp(g.x_amax ? *g.x_amax : 0u) << 23);
    const float sdy = __uint_as_float(wt_scale_exp(g.dy_amax ? *g.dy_amax : 0u) << 23);
    const unsigned pitch_x = (unsigned)g.ldx * 4u, pitch_dy = (unsigned)g.lddy * 4u;
    const unsigned long long ximg = (unsigned long long)g.H * g.W * pitch_x, dimg = (unsigned long long)g.Ho * g.Wo * pitch_dy;

    // ---- staging: thread t owns float4 slots t, t + 512, ...; slot < NX4: patch (pixel slot / 8, channels 4 (slot % 8)),
    // else dy tile (pixel / 16 of the tile, channels 4 (slot % 16))
    f32x4 stg[NSLOT];
    auto load_tile = [&](int tile) {
        const int b = tile / (g.tiles_y * g.tiles_x);
        const int rem = tile - b * (g.tiles_y * g.tiles_x);
        const int ty0 = (rem / g.tiles_x) * TH, tx0 = (rem - (rem / g.tiles_x) * g.tiles_x) * 16;
        const __amdgpu_buffer_rsrc_t xr = y4_make_rsrc(reinterpret_cast<const char*>(g.x) + (unsigned long long)b * ximg, (unsigned)ximg);
        const __amdgpu_buffer_rsrc_t dr = y4_make_rsrc(reinterpret_cast<const char*>(g.dy) + (unsigned long long)b * dimg, (unsigned)dimg);
        const int hx0 = ty0 * S - 1, wx0 = tx0 * S - 1;
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            const int slot = tid + i * 512;
            if (slot < NX4) {
                const int pix = slot >> 3, c4 = slot & 7;
                const int py = pix / PW, px = pix - py * PW;
                const int h = hx0 + py, w = wx0 + px;
                const bool ok = (unsigned)h < (unsigned)g.H && (unsigned)w < (unsigned)g.W;
                const unsigned off = ok ? (unsigned)(h * g.W + w) * pitch_x + (unsigned)(c0 + c4 * 4) * 4u : 0xffffffffu;
                stg[i] = y4_buf_load4(xr, off, 0u);
            } else if (slot < NX4 + ND4) {
                const int d = slot - NX4;
                const int pix = d >> 4, c4 = d & 15;
                const int h = ty0 + (pix >> 4), w = tx0 + (pix & 15);
                const bool ok = h < g.Ho && w < g.Wo;
                const unsigned off = ok ? (unsigned)(h * g.Wo + w) * pitch_dy + (unsigned)(n0 + c4 * 4) * 4u : 0xffffffffu;
                stg[i] = y4_buf_load4(dr, off, 0u);
            }
        }
    };
    auto store_tile = [&](unsigned char* img) {
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            const int slot = tid + i * 512;
            int row, c;
            float s;
            if (slot < NX4) {
                const int pix = slot >> 3;
                c = (slot & 7) * 4;
                const int py = pix / PW, px = pix - py * PW;
                row = S == 1 ? py * PWS + px : py * PWS + (px & 1) * T::ODD0 + (px >> 1);
                s = sx;
            } else if (slot < NX4 + ND4) {
                const int d = slot - NX4;
                const int pix = d >> 4;
                c = (d & 15) * 4;
                row = XROWS + (c >> 5) * (TH * 16) + pix;
                c &= 31;
                s = sdy;
            } else {
                continue;
            }
            h4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = stg[i][e] * s;
                hi[e] = (_Float16)t;
                lo[e] = (_Float16)((t - (float)hi[e]) * 2048.f);
            }
            // channels c .. c + 3 of the chunk: segment c >> 4 (hi) / 2 + (c >> 4) (lo), byte (c & 15) 2 inside it
            *reinterpret_cast<h4*>(img + seg_addr(row, c >> 4) + (c & 15) * 2) = hi;
            *reinterpret_cast<h4*>(img + seg_addr(row, 2 + (c >> 4)) + (c & 15) * 2) = lo;
        }
    };

    // ---- fragments.  lane -> (k group grp = 8 pixels, block row qrow, 8-B piece pp); a transposed read takes rows r .. r + 3
    // of one 32-B segment, a fragment is two of them (rows r and r + 4).  All byte offsets of a lane are formed ONCE, for
    // K-step 0 of an image; K-step ks adds a constant (KDELTA rows, a multiple of 16: same swizzle), the image its base.
    const int grp = lane >> 4, qrow = (lane >> 2) & 3, pp = lane & 3;
    typedef __attribute__((address_space(3))) trh4* trp;
    static_assert((S == 1 ? 2 * PWS : 4 * PWS) % 16 == 0, "K-step stride must keep the swizzle");
    constexpr int KDELTA_B = (S == 1 ? 2 * PWS : 4 * PWS) * WROW, KDELTA_A = 32 * WROW;
    // wave -> (pair of output-channel tiles np: 32 of the 64, input-channel half ch: 16 of the 32, tap group tg: taps 0-4 / 5-8).
    // Two row tiles share every x fragment: 8 + 20 transposed reads per 30 MFMAs and K-step (one row tile x nine taps per wave
    // was 4 + 36 per 27, and the kernel is bound by its LDS reads)
    const int np = wave & 1, ch = (wave >> 1) & 1, tg = wave >> 2;
    const int ntap = tg ? 4 : 5;
    // dy rows of this lane in K-step 0: chunk np, pixel 8 grp + qrow; row tile i: segment i (hi), 2 + i (lo)
    const int a_row0 = XROWS + np * (TH * 16) + 8 * grp + qrow;
    // x rows of tap (0, 0) in K-step 0: output row grp >> 1, columns 8 (grp & 1) .. + 7
    const int b_row0 = (S == 1 ? (grp >> 1) * PWS : (grp >> 1) * 2 * PWS) + (grp & 1) * 8 + qrow;
    int aoff[2][2][2], boff[5][2][2];                      // [row tile][plane][row / row + 4], [tap of the group][plane][row / row + 4]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int h = 0; h < 2; ++h) aoff[i][pl][h] = seg_addr(a_row0 + 4 * h, 2 * pl + i) + pp * 8;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int tap = tg * 5 + t < 9 ? tg * 5 + t : 8;
        const int r = tap / 3, q = tap - 3 * r;
        const int row = b_row0 + r * PWS + (S == 1 ? q : (q & 1) * T::ODD0 + (q >> 1));
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int h = 0; h < 2; ++h) boff[t][pl][h] = seg_addr(row + 4 * h, 2 * pl + ch) + pp * 8;
    }
    auto frag = [&](const unsigned char* base, int o0, int o1) -> f16x8 {
        const trh4 a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((trp)(base + o0));
        const trh4 b = __builtin_amdgcn_ds_read_tr16_b64_v4f16((trp)(base + o1));
        f16x8 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) { r[e] = (_Float16)a[e]; r[4 + e] = (_Float16)b[e]; }
        return r;
    };

    accv acc0[2][5], acc1[2][5];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 5; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc0[i][t][e] = 0.f; acc1[i][t][e] = 0.f; }

    auto compute = [&](const unsigned char* img) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const unsigned char* ab = img + ks * KDELTA_A;     // (wave-uniform bases: one add per read, nothing to hoist)
            const unsigned char* bb = img + ks * KDELTA_B;
            f16x8 fa[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) fa[i][pl] = frag(ab, aoff[i][pl][0], aoff[i][pl][1]);
            f16x8 fb0, fb1, nb0, nb1;
            fb0 = frag(bb, boff[0][0][0], boff[0][0][1]); fb1 = frag(bb, boff[0][1][0], boff[0][1][1]);
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                if (t + 1 < ntap) {                             // (wave-uniform)
                    nb0 = frag(bb, boff[(t + 1) % 5][0][0], boff[(t + 1) % 5][0][1]);
                    nb1 = frag(bb, boff[(t + 1) % 5][1][0], boff[(t + 1) % 5][1][1]);
                }
                if (t < ntap) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        acc1[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][1], fb0, acc1[i][t], 0, 0, 0);
                        acc0[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][0], fb0, acc0[i][t], 0, 0, 0);
                        acc1[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][0], fb1, acc1[i][t], 0, 0, 0);
                    }
                }
                fb0 = nb0; fb1 = nb1;
            }
        }
    };

    // ---- tile loop: block bi of its group takes tiles bi, bi + bpg, ...
    int tile = bi;
    int cur = 0;
    if (tile < g.tiles) {
        load_tile(tile);
        store_tile(smem);
    }
    __syncthreads();
    for (; tile < g.tiles; tile += g.bpg) {
        const int nxt = tile + g.bpg;
        if (nxt < g.tiles) load_tile(nxt);                 // in flight under the MFMAs below
        compute(smem + cur * IMG);
        // (dealing this split + store work out between the K-steps of the tile being computed measured +-0: 0.84 vs 0.83 ms)
        if (nxt < g.tiles) store_tile(smem + (cur ^ 1) * IMG);
        __syncthreads();                                   // image cur ^ 1 complete, image cur free for the tile after next
        cur ^= 1;
    }

    // ---- epilogue: slab[bi][n][tap Cin + c]; 16x16 tile: column = lane & 15 (input channel), row = 4 (lane >> 4) + e (output channel)
    const float un = __uint_as_float((254u - wt_scale_exp(g.x_amax ? *g.x_amax : 0u)) << 23) *
                     __uint_as_float((254u - wt_scale_exp(g.dy_amax ? *g.dy_amax : 0u)) << 23);
    const float un1 = un * (1.0f / 2048.0f);
    const int J = 9 * g.Cin;
    float* out = g.out + (long long)bi * g.Cout * J;
    const int fr = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        if (t >= ntap) continue;
        const int tap = tg * 5 + t;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = n0 + (2 * np + i) * 16 + 4 * kq + e;
                out[(long long)n * J + tap * g.Cin + c0 + ch * 16 + fr] = acc0[i][t][e] * un + acc1[i][t][e] * un1;
            }
    }
}

template <int S>
int launch_tile_wgrad(const TileWgradGeom& g, hipStream_t st) {
    using T = TileShape<S>;
    constexpr size_t smem = 2ull * T::IMG;
    static_assert(smem <= 160 * 1024, "LDS");
    auto kern = wgrad_tile_f16x2<S>;
    static Y4DynLds lds_attr;
    if (!lds_attr.ensure(reinterpret_cast<const void*>(kern), smem)) return Y4_ERR_LAUNCH;
    y4::note_kernel("wgrad_tile_f16x2<%d>", S);
    hipLaunchKernelGGL(kern, dim3(g.bpg * g.groups_c * g.groups_n), dim3(512), smem, st, g);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

}  // namespace

namespace y4 {

// the layers this kernel serves: 3x3, 64 output channels per group, 32-channel input chunks, large maps; stride 2 on even maps
bool tile_wgrad_ok(int Cin, int Cout, int k, int stride, int H, int W, long long ldx, long long lddy) {
    static const bool off = getenv("Y4_NO_TILE_WGRAD") != nullptr;
    if (off || k != 3 || Cout != 64 || (Cin != 32 && Cin != 64)) return false;
    if (stride == 2 && (Cin != 32 || (H & 1) || (W & 1))) return false;
    if (stride != 1 && stride != 2) return false;
    if ((long long)H * W < 100ll * 100ll) return false;
    // one image of either tensor inside a 32-bit buffer window
    if (ldx > 0 && (unsigned long long)H * W * (unsigned long long)ldx * 4ull >= 0xfffffff0ull) return false;
    if (lddy > 0 && (unsigned long long)H * W * (unsigned long long)lddy * 4ull >= 0xfffffff0ull) return false;
    return true;
}

// slabs the kernel writes (= blocks per group): one block per CU over all groups
int tile_wgrad_slabs(int Cin, int Cout) {
    const int groups = (Cin / 32) * (Cout / 64);
    int bpg = 256 / (groups > 0 ? groups : 1);
    return bpg < 1 ? 1 : bpg;
}

int f16x2_wgrad_tile(const WgradGeom& w, hipStream_t st) {
    TileWgradGeom g{};
    g.x = w.x; g.dy = w.dy; g.out = w.out; g.ldx = w.ldx; g.lddy = w.lddy;
    g.B = w.B; g.H = w.H; g.W = w.W; g.Cin = w.Cin; g.Ho = w.Ho; g.Wo = w.Wo; g.Cout = w.Cout;
    const int TH = w.stride == 1 ? 8 : 4;
    g.tiles_y = (w.Ho + TH - 1) / TH; g.tiles_x = (w.Wo + 15) / 16;
    const long long tiles = (long long)w.B * g.tiles_y * g.tiles_x;
    if (tiles >= (1ll << 31)) return Y4_ERR_SHAPE;
    g.tiles = (int)tiles;
    g.groups_c = w.Cin / 32; g.groups_n = w.Cout / 64;
    g.bpg = w.splits;                                      // (= tile_wgrad_slabs: the planner sized the slabs for it)
    g.x_amax = w.x_amax; g.dy_amax = w.dy_amax;
    if ((reinterpret_cast<uintptr_t>(w.x) & 15) || (reinterpret_cast<uintptr_t>(w.dy) & 15) || (w.ldx & 3) || (w.lddy & 3)) return Y4_ERR_SHAPE;
    return w.stride == 1 ? launch_tile_wgrad<1>(g, st) : launch_tile_wgrad<2>(g, st);
}

}  // namespace y4
