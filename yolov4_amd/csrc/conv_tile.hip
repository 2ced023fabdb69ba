// 3x3 stride-1 convolution for FEW channels on LARGE maps (the stage-1 / stage-2 layers: 32 -> 64 @304^2, 64 -> 64 @152^2,
// forward and dgrad), gfx950 only, f16x2 arithmetic (conv_f16x2.hip, header).
//
// On these layers the gather kernel is bound by its staging, not by the matrix pipe: it fetches and splits every input
// pixel once per filter tap (9x) for only 32-64 output columns of MFMA work.  Here a persistent block owns 2-D output tiles
// of TH x 16 (16 x 16) pixels of one image and stages, per 32-channel chunk, the PATCH of (TH + 2) x 18 input pixels those outputs
// touch -- each input element is read from HBM, scaled and split ONCE -- and runs the nine taps out of it at shifted row
// offsets (an MFMA row tile is 16 consecutive pixels of one output row, i.e. 16 consecutive patch rows for any tap).
//   * filter: Cs * BN <= 2048, so the whole [BN][9 Cs] filter of the block's N tile (two fp16 planes, 72 KB) stays in LDS
//     for the life of the block, brought in once by LDS-DMA from the per-call planes (conv_f16x2.hip: f16x2_filter_planes)
//   * patch: two chunk buffers of (TH + 2) * 18 rows x 128 B ([64 B hi | 64 B lo] per pixel, 16-B chunks XOR-swizzled by
//     (row >> 1) & 7 as in conv_planes.hip); the loads of chunk c + 1 (the next tile's first chunk after the last one) are in
//     flight in registers under the nine taps of chunk c; ONE barrier per chunk
//   * wave w of 8 owns output rows 2w, 2w + 1 of the tile (two 16-pixel row tiles) and all BN columns
//   * epilogue: raw fp32 result (+ residual for dgrad); BatchNorm column sums accumulate in registers over ALL tiles of
//     the block -> one partial row per block
// dgrad runs the same kernel on the mirrored, transposed filter (as conv_planes.hip does).
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "conv_geom.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float accv __attribute__((ext_vector_type(4)));

__device__ __host__ __forceinline__ unsigned tl_scale_exp(unsigned amax_bits) {      // as f16x2_scale_exp (conv_f16x2.hip)
    const unsigned e = (amax_bits >> 23) & 0xffu;
    if (e == 0u || e == 255u) return 127u;
    int se = 268 - (int)e;
    if (se < 2) se = 2;
    if (se > 252) se = 252;
    return (unsigned)se;
}
__device__ __forceinline__ float tl_scale(const unsigned* amax) { return __uint_as_float(tl_scale_exp(amax ? *amax : 0u) << 23); }
__device__ __forceinline__ float tl_unscale(const unsigned* amax) { return __uint_as_float((254u - tl_scale_exp(amax ? *amax : 0u)) << 23); }

__device__ __forceinline__ void tl_split4(const f32x4 v, const float s, u32x2& hi, u32x2& lo) {
    f16x2v h0, h1, l0, l1;
    const float t0 = v[0] * s, t1 = v[1] * s, t2 = v[2] * s, t3 = v[3] * s;
    h0[0] = (_Float16)t0; h0[1] = (_Float16)t1; h1[0] = (_Float16)t2; h1[1] = (_Float16)t3;
    l0[0] = (_Float16)((t0 - (float)h0[0]) * 2048.f); l0[1] = (_Float16)((t1 - (float)h0[1]) * 2048.f);
    l1[0] = (_Float16)((t2 - (float)h1[0]) * 2048.f); l1[1] = (_Float16)((t3 - (float)h1[1]) * 2048.f);
    hi[0] = __builtin_bit_cast(unsigned, h0); hi[1] = __builtin_bit_cast(unsigned, h1);
    lo[0] = __builtin_bit_cast(unsigned, l0); lo[1] = __builtin_bit_cast(unsigned, l1);
}

constexpr int TW = 16;

struct TileGeom {
    y4::ConvGeom g;
    int tiles_h, tiles_w, ntiles, sp_tiles;               // spatial tiles per image (h, w), N tiles, spatial tiles in total
    int step_b, step_th, step_tw;                         // a block's stride through the tile sequence, as (image, tile row, tile column)
};

// S2 = true: the dgrad of a STRIDE-2 conv (32 -> 64 @608 -> 304: the most expensive single layer of the step).  A tile is 16 x 16
// positions of the dy grid; every filter tap (r, q) feeds exactly one parity class (h & 1, w & 1) of the 32 x 32 output pixels
// the tile owns -- r = 1 -> even rows from dy row i, r = 0 / 2 -> odd rows from dy rows i + 1 / i -- so the nine taps run out
// of ONE staged patch of 17 x 17 dy pixels into four accumulator sets, no tap is multiplied by a zero and dy is read once.
template <int CS, int BN, int TH, bool S2 = false>
__global__ __launch_bounds__(TH * 32, 1) void conv3x3_tile_f16x2(const TileGeom tg) {
    const y4::ConvGeom& g = tg.g;
    constexpr int NW = TH / 2, NT = NW * 64;
    constexpr int CC = CS / 32, KT = 9 * CC;
    constexpr int PW = S2 ? TW + 1 : TW + 2, PH = S2 ? TH + 1 : TH + 2, ORG = S2 ? 0 : 1, NCLS = S2 ? 4 : 1;
    constexpr int PROWS = PH * PW, PBUF = PROWS * 128;
    constexpr int FROW = KT * 128 + (KT % 2 == 0 ? 128 : 0);      // an ODD number of 128-B lines: rows n, n + 1 fall on different bank halves
    constexpr int FBYTES = BN * FROW;
    constexpr int TN = BN / 16;
    constexpr int SLOTS = PROWS * 8, PP = (SLOTS + NT - 1) / NT;
    static_assert(CS * BN <= 2048 && FBYTES + 2 * PBUF <= 160 * 1024, "filter + two patch buffers must fit the LDS");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* const fsm = smem;
    unsigned char* const psm = smem + FBYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = blockIdx.x % tg.ntiles;
    const int n0 = nt * BN;
    const int sp_first = blockIdx.x / tg.ntiles, sp_step = gridDim.x / tg.ntiles;
    const int H = g.Hs, W = g.Ws;
    const unsigned pix_bytes = (unsigned)g.lds_ * 4u;
    const unsigned long long img_bytes = (unsigned long long)H * W * pix_bytes;
    const unsigned OOB = 0xffffffffu;
    const float sa = tl_scale(g.src_amax);

    // ---- filter of this N tile -> LDS, once (rows >= N come back as zeros)
    {
        const __amdgpu_buffer_rsrc_t wt_rsrc = y4_make_rsrc(g.wt_planes, g.wt_bytes);
        typedef __attribute__((address_space(3))) void* lds_ptr;
        for (int base = wave * 1024; base < FBYTES; base += NW * 1024) {
            const int L = base + lane * 16;
            const int row = L / FROW, rem = L - row * FROW;
            const int t = rem >> 7, p = (rem >> 4) & 7;                  // (t == KT: the pad line of the row, never read)
            const int c = p ^ ((row >> 1) & 7);
            const unsigned off = ((n0 + row) < g.N && t < KT) ? (unsigned)(n0 + row) * (unsigned)(KT * 128) + (unsigned)t * 128u + (unsigned)c * 16u : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wt_rsrc, (lds_ptr)(fsm + base), 16, (int)off, 0, 0, 0);
        }
    }

    // ---- patch slots of this thread: (patch pixel, 16-B group of 4 channels), fixed for the kernel
    int s_ph[PP], s_pw[PP], s_lds[PP];
#pragma unroll
    for (int i = 0; i < PP; ++i) {
        const int slot = tid + i * NT;
        const int prow = slot >> 3, kc = slot & 7;
        s_ph[i] = slot < SLOTS ? prow / PW : -100000;      // (a row no image has: never valid)
        s_pw[i] = prow % PW;
        s_lds[i] = prow * 128 + ((((kc >> 1) ^ ((prow >> 1) & 7)) << 4) | ((kc & 1) << 3));
    }

    // ---- fragment addressing: lane -> (row fr of a 16-row tile, K quarter kq)
    const int fr = lane & 15, kq = lane >> 4;
    int b_hi[TN], b_lo[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = j * 16 + fr, sw = (n >> 1) & 7;
        b_hi[j] = n * FROW + ((kq ^ sw) << 4);
        b_lo[j] = n * FROW + (((4 + kq) ^ sw) << 4);
    }
    // A fragments: patch row of (output row 2 wave + i, column fr) at tap (r, q) and its swizzled hi chunk, for all 12 (i + r, q)
    // -- fixed for the kernel (the lo chunk is the same offset ^ 64)
    int a_off[4][3];
#pragma unroll
    for (int ir = 0; ir < 4; ++ir)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int prow = (2 * wave + ir) * PW + q + fr;
            a_off[ir][q] = prow * 128 + ((kq ^ ((prow >> 1) & 7)) << 4);
        }

    accv acc0[NCLS][2][TN], acc1[NCLS][2][TN];
    // BatchNorm column sums over ALL tiles of the block (up to ~25 000 pixels per lane and column, from many images): a tile's
    // eight values per lane are summed first and added to the running sum with a compensated (Kahan) add, so the result does
    // not depend on which images the block happens to visit (a plain running fp32 sum made the batch-permutation test's
    // gradients move 4x further than a one-ulp input change)
    float cs[TN], css[TN], cc[TN], ccs[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) { cs[j] = 0.f; css[j] = 0.f; cc[j] = 0.f; ccs[j] = 0.f; }

    const float un = tl_unscale(g.src_amax) * tl_unscale(g.wt_amax);
    const float un1 = un * (1.0f / 2048.0f);

    // ---- chunk stream: (spatial tile, channel chunk) in order; `ld_*` = the chunk whose loads go out next
    f32x4 ra[PP];
    // tile coordinates advance by carries (a division per tile and call site costs ~40 VALU: measured as the longest part
    // of a tile's fixed cost)
    struct Tile { int sp, b, th, tw; };
    auto advance = [&](Tile t) {
        t.sp += sp_step; t.b += tg.step_b; t.th += tg.step_th; t.tw += tg.step_tw;
        if (t.tw >= tg.tiles_w) { t.tw -= tg.tiles_w; ++t.th; }
        if (t.th >= tg.tiles_h) { t.th -= tg.tiles_h; ++t.b; }
        return t;
    };
    auto load_chunk = [&](const Tile& tl, int cc) {
        const int b = tl.b, h0 = tl.th * TH, w0 = tl.tw * TW;
        const __amdgpu_buffer_rsrc_t rs = y4_make_rsrc(reinterpret_cast<const char*>(g.src) + (unsigned long long)b * img_bytes, (unsigned)img_bytes);
#pragma unroll
        for (int i = 0; i < PP; ++i) {
            const int hh = h0 - ORG + s_ph[i], ww = w0 - ORG + s_pw[i];
            const bool ok = (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W;
            const unsigned off = ok ? (unsigned)(hh * W + ww) * pix_bytes + (unsigned)((tid + i * NT) & 7) * 16u : OOB;
            ra[i] = y4_buf_load4(rs, off, (unsigned)cc * 128u);
        }
    };
    auto store_chunk = [&](int buf) {
        unsigned char* pb = psm + buf * PBUF;
#pragma unroll
        for (int i = 0; i < PP; ++i) {
            if (tid + i * NT < SLOTS) {
                u32x2 hi, lo;
                tl_split4(ra[i], sa, hi, lo);
                *reinterpret_cast<u32x2*>(pb + s_lds[i]) = hi;
                *reinterpret_cast<u32x2*>(pb + (s_lds[i] ^ 64)) = lo;       // logical chunk 4 + c sits at position (4 + c) ^ sw = (c ^ sw) ^ 4
            }
        }
    };
    auto compute = [&](int buf, int cc) {
        const unsigned char* pb = psm + buf * PBUF;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int r = tap / 3, q = tap - 3 * r;
            // stride 1: patch row offset (r, q), one accumulator set; stride-2 dgrad: tap -> (parity class, dy offset)
            const int cls = S2 ? ((r == 1 ? 0 : 2) | (q == 1 ? 0 : 1)) : 0;
            const int dr = S2 ? (r == 0 ? 1 : 0) : r, dq = S2 ? (q == 0 ? 1 : 0) : q;
            const int t = tap * CC + cc;
            f16x8 fb[TN][2];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                fb[j][0] = *reinterpret_cast<const f16x8*>(fsm + b_hi[j] + t * 128);
                fb[j][1] = *reinterpret_cast<const f16x8*>(fsm + b_lo[j] + t * 128);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f16x8 fa0 = *reinterpret_cast<const f16x8*>(pb + a_off[i + dr][dq]);
                const f16x8 fa1 = *reinterpret_cast<const f16x8*>(pb + (a_off[i + dr][dq] ^ 64));
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc1[cls][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa1, fb[j][0], acc1[cls][i][j], 0, 0, 0);
                    acc1[cls][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa0, fb[j][1], acc1[cls][i][j], 0, 0, 0);
                    acc0[cls][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa0, fb[j][0], acc0[cls][i][j], 0, 0, 0);
                }
            }
        }
    };
    // Branch-free epilogue: stores and residual loads are raw buffer accesses whose offset is pushed out of range for rows /
    // columns outside the image or the filter count (the hardware drops them), one 32-bit offset per output PIXEL of the lane
    // with the TN column tiles as immediate offsets.  (With `if (valid) store` per element the compiler built ~160 basic
    // blocks of exec-mask branches: 570 scalar + 610 vector instructions per tile around 216 MFMAs.)
    bool nok[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) nok[j] = n0 + j * 16 + fr < g.N;
    const bool allcols = n0 + BN <= g.N;
    const unsigned dpix_bytes = (unsigned)g.ldd * 4u, rpix_bytes = (unsigned)g.ldr * 4u;
    const int Ho = S2 ? g.Hd : H, Wo = S2 ? g.Wd : W;      // produced map (stride-2 dgrad: twice the dy grid)
    const unsigned long long dimg_bytes = (unsigned long long)Ho * Wo * dpix_bytes, rimg_bytes = (unsigned long long)Ho * Wo * rpix_bytes;
    auto epilogue_impl = [&](const Tile& tl, auto ALLC, auto RES) {
        constexpr bool allc = decltype(ALLC)::value, with_res = decltype(RES)::value;
        const int b = tl.b, h0 = tl.th * TH, w0 = tl.tw * TW;
        float ts[TN], tss[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) { ts[j] = 0.f; tss[j] = 0.f; }
        const __amdgpu_buffer_rsrc_t drs = y4_make_rsrc(reinterpret_cast<char*>(g.dst) + (unsigned long long)b * dimg_bytes, (unsigned)dimg_bytes);
        const __amdgpu_buffer_rsrc_t rrs = y4_make_rsrc(with_res ? reinterpret_cast<const char*>(g.res) + (unsigned long long)b * rimg_bytes : nullptr,
                                                        with_res ? (unsigned)rimg_bytes : 0u);
#pragma unroll
        for (int cls = 0; cls < NCLS; ++cls)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int hg = h0 + 2 * wave + i;                          // row of the tile grid (stride 1: the output row itself)
            const int h = S2 ? 2 * hg + (cls >> 1) : hg;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int wg = w0 + 4 * kq + e;
                const int w = S2 ? 2 * wg + (cls & 1) : wg;
                const bool pok = h < Ho && w < Wo;
                const unsigned pix = (unsigned)(h * Wo + w);
                const unsigned doff = pok ? pix * dpix_bytes + (unsigned)(n0 + fr) * 4u : OOB;
                const unsigned roff = pok ? pix * rpix_bytes + (unsigned)(n0 + fr) * 4u : OOB;
                const float keep = pok ? 1.0f : 0.0f;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    float v = acc0[cls][i][j][e] * un + acc1[cls][i][j][e] * un1;
                    const bool cok = allc || nok[j];
                    if constexpr (!S2) {                               // statistics (forward): raw result, valid rows / columns only
                        const float vs = cok ? v * keep : 0.0f;
                        ts[j] += vs; tss[j] += vs * vs;
                    }
                    if constexpr (with_res)
                        v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, (int)(cok ? roff : OOB), j * 64, 0));
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), drs, (int)(cok ? doff : OOB), j * 64, 0);
                }
            }
        }
        if constexpr (!S2) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float y1 = ts[j] - cc[j], t1 = cs[j] + y1;
                cc[j] = (t1 - cs[j]) - y1; cs[j] = t1;
                const float y2 = tss[j] - ccs[j], t2 = css[j] + y2;
                ccs[j] = (t2 - css[j]) - y2; css[j] = t2;
            }
        }
    };
    auto epilogue = [&](const Tile& tl) {                  // four straight-line versions behind two uniform tests
        using T = std::true_type; using F = std::false_type;
        if (allcols) { if (g.res) epilogue_impl(tl, T{}, T{}); else epilogue_impl(tl, T{}, F{}); }
        else { if (g.res) epilogue_impl(tl, F{}, T{}); else epilogue_impl(tl, F{}, F{}); }
    };

    Tile cur;
    {
        const int per_img = tg.tiles_h * tg.tiles_w;
        cur.sp = sp_first; cur.b = sp_first / per_img;
        const int r = sp_first - cur.b * per_img;
        cur.th = r / tg.tiles_w; cur.tw = r - cur.th * tg.tiles_w;
    }
    int cur_cc = 0, buf = 0;
    if (cur.sp < tg.sp_tiles) {
        load_chunk(cur, 0);
        store_chunk(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // filter DMA (and the first patch) landed
    __syncthreads();
    while (cur.sp < tg.sp_tiles) {
        Tile nx = cur;
        int nx_cc = cur_cc + 1;
        if (nx_cc == CC) { nx_cc = 0; nx = advance(cur); }
        const bool more = nx.sp < tg.sp_tiles;
        if (more) load_chunk(nx, nx_cc);                   // in flight under the nine taps below
        if (cur_cc == 0) {
#pragma unroll
            for (int c = 0; c < NCLS; ++c)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) { acc0[c][i][j][e] = 0.f; acc1[c][i][j][e] = 0.f; }
        }
        compute(buf, cur_cc);
        if (cur_cc == CC - 1) epilogue(cur);
        if (more) store_chunk(buf ^ 1);                    // (every wave left that buffer at the previous barrier)
        __syncthreads();
        buf ^= 1; cur = nx; cur_cc = nx_cc;
    }

    // ---- BatchNorm column sums of everything this block produced -> ONE partial row [2][N] per block and N tile
    if (g.stats) {
        float* red = reinterpret_cast<float*>(psm);        // [NW][BN][2]
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float a = cs[j], b2 = css[j];
            a += __shfl_xor(a, 16, 64); b2 += __shfl_xor(b2, 16, 64);
            a += __shfl_xor(a, 32, 64); b2 += __shfl_xor(b2, 32, 64);
            if (kq == 0) {
                red[(wave * BN + j * 16 + fr) * 2 + 0] = a;
                red[(wave * BN + j * 16 + fr) * 2 + 1] = b2;
            }
        }
        __syncthreads();
        for (int c = tid; c < BN; c += NT) {
            float a = 0.f, b2 = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) { a += red[(w * BN + c) * 2]; b2 += red[(w * BN + c) * 2 + 1]; }
            const int n = n0 + c;
            if (n < g.N) {
                const long long rowi = blockIdx.x / tg.ntiles;
                g.stats[(rowi * 2 + 0) * g.N + n] = a;
                g.stats[(rowi * 2 + 1) * g.N + n] = b2;
            }
        }
    }
}

template <int CS, int BN, int TH, bool S2 = false>
int launch_tile(const y4::ConvGeom& g, hipStream_t st, int* nparts) {
    TileGeom tg{};
    tg.g = g;
    tg.tiles_h = (g.Hs + TH - 1) / TH;
    tg.tiles_w = (g.Ws + TW - 1) / TW;
    tg.ntiles = (g.N + BN - 1) / BN;
    const long long sp = (long long)g.B * tg.tiles_h * tg.tiles_w;
    if (sp >= (1ll << 30)) return Y4_ERR_SHAPE;
    tg.sp_tiles = (int)sp;
    long long per = 256 / tg.ntiles;                       // one persistent block per CU
    if (per > sp) per = sp;
    if (per < 1) per = 1;
    const int grid = (int)per * tg.ntiles;
    {
        const int per_img = tg.tiles_h * tg.tiles_w;
        tg.step_b = (int)per / per_img;
        const int r = (int)per - tg.step_b * per_img;
        tg.step_th = r / tg.tiles_w; tg.step_tw = r - tg.step_th * tg.tiles_w;
    }
    if (nparts) *nparts = (int)per;
    constexpr int CC = CS / 32, KT = 9 * CC;
    constexpr size_t smem = (size_t)BN * (KT * 128 + (KT % 2 == 0 ? 128 : 0)) + 2ull * (S2 ? (TH + 1) * (TW + 1) : (TH + 2) * (TW + 2)) * 128;
    tg.g.wt_bytes = (unsigned)((unsigned long long)g.N * g.K * 4ull);
    auto kern = conv3x3_tile_f16x2<CS, BN, TH, S2>;
    static Y4DynLds lds_attr;                              // per device, see common.h
    if (!lds_attr.ensure(reinterpret_cast<const void*>(kern), smem)) return Y4_ERR_LAUNCH;
    y4::note_kernel(S2 ? "conv3x3_tile_f16x2<%d, %d, %d, true>" : "conv3x3_tile_f16x2<%d, %d, %d>", CS, BN, TH);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(TH * 32), smem, st, tg);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

}  // namespace

namespace y4 {

// 3x3 / stride 1 / pad 1, 32 or 64 gathered channels, <= 64 produced channels, a map large enough that the gather kernel's
// nine-fold staging is what bounds it, and nothing in the epilogue but the raw result (+ residual, + column sums)
// The kernels address one image of each tensor through a 32-bit window (H * W * pitch * 4 B < 4 GiB).  The predicates below
// are asked where the pitches are not known yet (the forward call prepares the dgrad filter in the form THIS dgrad kernel wants),
// so they assume the widest pitch a tile layer meets in a cat buffer, 256 channels: larger maps (> 2047 x 2047) go to the gather
// kernels, whose addresses are 64-bit.  (The launchers keep the exact test for callers with wider pitches.)
static bool tile_window_ok(long long H, long long W) { return (unsigned long long)(H * W) * 4ull * 256ull < 0xfffffff0ull; }

bool tile_conv_ok(int Cs, int Cs_valid, int N, int k, int stride, int H, int W) {
    static const bool off = getenv("Y4_NO_TILE") != nullptr;
    if (off || k != 3 || stride != 1 || Cs != Cs_valid || (Cs != 32 && Cs != 64) || N < 1 || N > 64) return false;
    if ((long long)H * W < 100 * 100) return false;
    return tile_window_ok(H, W);
}

int f16x2_tile(const ConvGeom& g, hipStream_t st, int* nparts) {
    if (!g.wt_planes || g.scale || g.shift || g.dst_amax || g.act != Y4_ACT_LINEAR) return Y4_ERR_SHAPE;
    const unsigned long long px = (unsigned long long)g.Hs * g.Ws * 4ull;      // 32-bit windows per image on all three tensors
    if (px * (unsigned long long)g.lds_ >= 0xfffffff0ull || px * (unsigned long long)g.ldd >= 0xfffffff0ull ||
        (g.res && px * (unsigned long long)g.ldr >= 0xfffffff0ull)) return Y4_ERR_SHAPE;
    // 16 x 16 outputs per tile, 8 waves (two per SIMD): 4 waves on 8 x 16 tiles were 10-25 % slower (one wave per SIMD has
    // nobody to hide its LDS round trips and its epilogue behind)
    return g.Cs == 32 ? launch_tile<32, 64, 16>(g, st, nparts) : launch_tile<64, 32, 16>(g, st, nparts);
}

// dgrad of a 3x3 / stride-2 / pad-1 conv with 64 (padded) output channels and <= 32 input channels on a large map: the
// gathered tensor is dy ([B][Hs][Ws][64]), the produced one dx ([B][Hd][Wd][N]); filter planes transposed, NOT mirrored
bool tile_dgrad_s2_ok(int Cs, int Cs_valid, int N, int k, int stride, int Hs, int Ws) {
    static const bool off = getenv("Y4_NO_TILE") != nullptr;
    return !off && k == 3 && stride == 2 && Cs == 64 && Cs_valid == 64 && N >= 1 && N <= 32 && (long long)Hs * Ws >= 100 * 100 &&
           tile_window_ok(2ll * Hs, 2ll * Ws);            // (dx lives on the 2 Hs x 2 Ws grid)
}

int f16x2_tile_dgrad_s2(const ConvGeom& g, hipStream_t st) {
    if (!g.wt_planes || g.scale || g.shift || g.dst_amax || g.act != Y4_ACT_LINEAR || g.stats) return Y4_ERR_SHAPE;
    const unsigned long long pxs = (unsigned long long)g.Hs * g.Ws * 4ull, pxd = (unsigned long long)g.Hd * g.Wd * 4ull;
    if (pxs * (unsigned long long)g.lds_ >= 0xfffffff0ull || pxd * (unsigned long long)g.ldd >= 0xfffffff0ull ||
        (g.res && pxd * (unsigned long long)g.ldr >= 0xfffffff0ull)) return Y4_ERR_SHAPE;
    return launch_tile<64, 32, 16, true>(g, st, nullptr);
}

}  // namespace y4
