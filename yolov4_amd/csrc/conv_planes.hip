// Implicit-GEMM convolution over PRE-SPLIT operands ("planes"), gfx950 only: LDS-DMA + ds_read_b128 + MFMA, no VALU in
// the K loop.
//
// Arithmetic: the f16x2 split of conv_f16x2.hip (every operand tensor scaled by a power of two from an upper bound of
// its max|x|, each element two fp16 pieces hi = RN16(s x), lo = RN16((s x - hi) 2^11); three fp16 MFMAs per product into
// two fp32 accumulators; c = (acc0 + 2^-11 acc1) / (s_A s_B)).  What changes is WHO splits: here both operands arrive
// in HBM already split -- activations / gradients written that way by the BatchNorm sweeps that produce them
// (pointwise.hip, `planes` outputs), filters by the per-call split kernels -- in the layout the DMA wants:
//
//     tensor[pixel or filter row][K-tile of 32 channels][ 64 B: 32 hi halfs | 64 B: 32 lo halfs ]
//
// i.e. 4 bytes per element like fp32, one K-tile of one row = one whole 128-B line.  A block stages a K-tile of BM
// activation rows and BN filter rows with `buffer_load_dwordx4 ... lds` (16 B per lane, 8 lanes per row: a wave
// instruction moves 8 whole lines; per-lane SOURCE addresses make it a gather over the filter taps, the descriptor's
// range check turns halo / tail rows into zeros), into a lane-linear LDS image whose 16-B chunks are XOR-swizzled on the
// source address (chunk c of row r sits at position c ^ ((r >> 1) & 7): every ds_read_b128 lane group of the 16x16x32
// fragment reads is conflict-free).  Three LDS stages, two K-tiles of DMA in flight across ONE raw s_barrier per K-tile
// with a counted vmcnt, one 8-wave 256 x 128 block per CU (wave tile 64 x 64: 48 MFMAs + 16 ds_read_b128 + 6 DMA issues
// per wave and K-tile).
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "conv_geom.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __host__ __forceinline__ unsigned pl_scale_exp(unsigned amax_bits) {      // as f16x2_scale_exp (conv_f16x2.hip)
    const unsigned e = (amax_bits >> 23) & 0xffu;
    if (e == 0u || e == 255u) return 127u;
    int se = 268 - (int)e;
    if (se < 2) se = 2;
    if (se > 252) se = 252;
    return (unsigned)se;
}
__device__ __forceinline__ float pl_scale(const unsigned* amax) { return __uint_as_float(pl_scale_exp(amax ? *amax : 0u) << 23); }
__device__ __forceinline__ float pl_unscale(const unsigned* amax) { return __uint_as_float((254u - pl_scale_exp(amax ? *amax : 0u)) << 23); }

struct PlaneConvGeom {
    const unsigned char* src;          // activation planes [B][Hs][Ws][Cs / 32][128 B]
    const unsigned char* wt;           // filter planes [N][K / 32][128 B], K order (tap, channel)
    float* dst; long long ldd;         // raw fp32 result [M][ldd]
    const float* res; long long ldr;   // optional: added in the epilogue (dgrad: gradient arriving over a skip connection)
    float* stats;                      // optional: per-M-tile column sums [mtiles][2][N] of the result
    int dst_bf16;                      // the result leaves as bf16 (RN) in the first half of each fp32-sized row (pitch 4 ldd bytes)
    // DG2 (dgrad of a 3x3 stride-2 conv, one launch per parity class (ph, pw) of the dx grid): the rows are the class's pixels
    // (hc, wc) <-> dx pixel (2 hc + ph, 2 wc + pw) of an Hfull x Wfull map (Hd, Wd = Hfull / 2, Wfull / 2); the class's 1 / 2 /
    // 2 / 4 taps read dy pixel (hc + dh, wc + dw), dh, dw in {0, 1}, and filter K-tile group `slot` (= tap r * 3 + q)
    int cls_ntaps, cls_ph, cls_pw, Hfull, Wfull;
    int cls_dh[4], cls_dw[4], cls_slot[4];
    int B, Hs, Ws, Cs, Hd, Wd, N, k, stride, pad, M, K;
    int mtiles, ntiles;
    unsigned long long src_total_bytes;
    unsigned wt_bytes;
    const unsigned* src_amax; const unsigned* wt_amax;
};

constexpr int PROW = 128;                                  // bytes per LDS row: one 32-deep K-tile, [64 B hi | 64 B lo]

// Two shapes of the same kernel:
//   <256, 128, 4, 2>: 8 waves, three 48-KB stages, ONE block per CU -- the MFMA-bound layers (3x3, long K).
//   <128, 128, 2, 2>: 4 waves, two 32-KB stages, TWO blocks per CU -- short-K layers (1x1 with K <= a few hundred), which
//                     are HBM-bound: with one block per CU its load, MFMA and store phases run back to back and the
//                     memory pipes idle for most of them; two co-resident blocks overlap one's epilogue with the other's loads.
//
// BF (conv mode 2, BASELINE configs[4]: bf16 MFMA conv, fp32 everything else): the SAME kernel over plain bf16 operands.
// A 128-B row then holds 64 channels of one pixel (activations: dense bf16 NHWC in the first half of the fp32-sized row,
// filters: the dense bf16 [N][K] matrix), a K-tile is 64 deep, the two 16-B chunks a lane reads per row are k 0-31 and
// k 32-63 of ONE plane: two bf16 MFMAs into ONE accumulator set, no scales.  With half the accumulator registers the wave
// tile grows to 128 x 64 (<256, 256, 2, 4>: two 64-KB stages), which brings the staged bytes per MFMA cycle back to the
// f16x2 kernel's (32 B / clk / CU) -- at 256 x 128 the bf16 form would wait for its DMA most of the time.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <bool BF>
__device__ __forceinline__ f32x4v pl_mma(const f16x8 a, const f16x8 b, const f32x4v c) {
    if constexpr (BF) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// YB (with BF): the result leaves as bf16 (PlaneConvGeom::dst_bf16) -- a variant of its own, so that the three store forms of
// the epilogue do not share one register allocation (as run-time branches they cost the 128 x 64 wave tile 14 spills)
// DG2: the parity-class dgrad of the 3x3 stride-2 layers (PlaneConvGeom::cls_*): a tap list instead of the k x k raster, and a
// scattered destination (every second pixel of every second row).
// Cache policy of the result stores: nt (streaming).  The result of a conv is read once, by the BatchNorm sweep that follows
// (itself with streaming loads), and is far larger than what the next K sweep of this kernel wants to find in the caches:
// 449.2 / 450.1 / 448.2 -> 453.8 / 453.4 / 453.1 img/s (three A/B rounds on one box, forward + stride-1 dgrad epilogues).
#ifndef PL_ST_POL
#define PL_ST_POL 2
#endif
template <int BM, int BN, int WM, int WN, bool BF, bool YB = false, bool DG2 = false>
__global__ __launch_bounds__(WM * WN * 64, WM * WN == 8 ? 1 : 2) void conv_planes_mfma(const PlaneConvGeom g) {
    constexpr int NWAVE = WM * WN, NTHR = NWAVE * 64;
    static_assert(NWAVE == 8 || NWAVE == 4, "8 or 4 waves");
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 16, TN = WTN / 16;
    constexpr int A_BYTES = BM * PROW, B_BYTES = BN * PROW, STAGE = A_BYTES + B_BYTES;
    constexpr int NSTAGE = (NWAVE == 8 && 3 * STAGE <= 160 * 1024) ? 3 : 2;
    constexpr int KSH = BF ? 6 : 5;                        // log2 of the channels in one 128-B row
    constexpr unsigned WEB = BF ? 2u : 4u;                 // filter bytes per element
    constexpr int PA = BM / 8 / NWAVE, PB = BN / 8 / NWAVE, NDMA = PA + PB;      // 1-KiB DMA pieces per wave and K-tile
    static_assert(PA >= 1 && PB >= 1 && ((NSTAGE == 3 && NDMA == 6) || (NSTAGE == 2 && NDMA == 8)),
                  "the counted vmcnt below assumes 6 (three stages) or 8 (two stages) pieces per wave and K-tile");
    typedef float accv __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int lt = y4_xcd_remap(blockIdx.x, g.mtiles * g.ntiles);
    const int mt = lt / g.ntiles, nt = lt - mt * g.ntiles;
    const int n0 = nt * BN;
    const int CC = g.Cs >> KSH;
    const int KT = (DG2 ? g.cls_ntaps : g.k * g.k) * CC;

    // ---- 32-bit window of the source tensor, re-based at the first image this tile touches
    const int pix_per_img = g.Hd * g.Wd;
    const int b_first = (int)(((long long)mt * BM) / pix_per_img);
    const unsigned pitch = (unsigned)g.Cs * 4u;
    const unsigned long long img_bytes = (unsigned long long)g.Hs * g.Ws * pitch;
    const unsigned long long skip = (unsigned long long)b_first * img_bytes;
    const unsigned long long left = g.src_total_bytes > skip ? g.src_total_bytes - skip : 0ull;
    const __amdgpu_buffer_rsrc_t src_rsrc = y4_make_rsrc(g.src + skip, (unsigned)(left < 0xfffffff0ull ? left : 0xfffffff0ull));
    const __amdgpu_buffer_rsrc_t wt_rsrc = y4_make_rsrc(g.wt, g.wt_bytes);
    const unsigned OOB = 0xffffffffu;                      // beyond any window: the range check returns zeros (halo, tail rows)

    // ---- DMA slots of this lane: 8 lanes per row, lane -> (row of the piece, position in the row)
    const int lr = lane >> 3, pos = lane & 7;
    unsigned a_base[PA], a_valid[PA], a_voff[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int row = (wave * PA + i) * 8 + lr;
        const int m = mt * BM + row;
        const bool ok = m < g.M;
        const int mm = ok ? m : 0;
        const int b = mm / pix_per_img;
        const int rem = mm - b * pix_per_img;
        const int hd = rem / g.Wd, wd = rem - hd * g.Wd;
        const int hc = DG2 ? hd : hd * g.stride - g.pad, wc = DG2 ? wd : wd * g.stride - g.pad;
        const int chunk = pos ^ ((row >> 1) & 7);
        a_base[i] = (unsigned)(((b - b_first) * g.Hs + hc) * g.Ws + wc) * pitch + (unsigned)chunk * 16u;   // mod 2^32
        unsigned v = 0u;
        if constexpr (DG2) {
            for (int t = 0; t < g.cls_ntaps; ++t)
                if (ok && hc + g.cls_dh[t] < g.Hs && wc + g.cls_dw[t] < g.Ws) v |= 1u << t;
        } else {
            for (int r = 0; r < g.k; ++r)
                for (int q = 0; q < g.k; ++q)
                    if (ok && (unsigned)(hc + r) < (unsigned)g.Hs && (unsigned)(wc + q) < (unsigned)g.Ws) v |= 1u << (r * g.k + q);
        }
        a_valid[i] = v;
    }
    unsigned b_off[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        const int row = (wave * PB + j) * 8 + lr;
        const int chunk = pos ^ ((row >> 1) & 7);
        b_off[j] = (n0 + row) < g.N ? (unsigned)(n0 + row) * (unsigned)g.K * WEB + (unsigned)chunk * 16u : OOB;
    }
    // ---- DMA issue state: the K-tile (ld_tap, ld_cc) whose pieces are going out next; taps outer, channel chunks inner
    int ld_tap = 0, ld_r = 0, ld_q = 0, ld_cc = 0;
    int ld_slot = DG2 ? g.cls_slot[0] : 0;                 // filter K-tile group of the tap under way (DG2: from the class's list)
    auto tap_setup = [&]() {
        unsigned toff;
        if constexpr (DG2) {
            const int t = ld_tap < g.cls_ntaps ? ld_tap : 0;
            toff = (unsigned)(g.cls_dh[t] * g.Ws + g.cls_dw[t]) * pitch;
            ld_slot = g.cls_slot[t];
        } else {
            toff = (unsigned)(ld_r * g.Ws + ld_q) * pitch;
            ld_slot = ld_tap;
        }
#pragma unroll
        for (int i = 0; i < PA; ++i) a_voff[i] = ((a_valid[i] >> ld_tap) & 1u) ? a_base[i] + toff : OOB;
    };
    tap_setup();
    typedef __attribute__((address_space(3))) void* lds_ptr;
    // pieces [p0, p1) of the current K-tile into stage `stage` (pieces 0..PA-1: activation rows, PA..NDMA-1: filter rows)
    auto issue = [&](int stage, auto P0, auto P1) {
        constexpr int p0 = decltype(P0)::value, p1 = decltype(P1)::value;
        unsigned char* st = smem + stage * STAGE;
        const unsigned soff_a = (unsigned)ld_cc * 128u;
        const unsigned soff_b = (unsigned)(ld_slot * CC + ld_cc) * 128u;
#pragma unroll
        for (int p = p0; p < p1; ++p) {
            if (p < PA)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(src_rsrc, (lds_ptr)(st + (wave * PA + p) * 1024), 16, (int)a_voff[p], (int)soff_a, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wt_rsrc, (lds_ptr)(st + A_BYTES + (wave * PB + (p - PA)) * 1024), 16,
                                                         (int)b_off[p - PA], (int)soff_b, 0, 0);
        }
        if constexpr (p1 == NDMA) {                        // K-tile complete: advance
            if (++ld_cc == CC) {
                ld_cc = 0; ++ld_tap;
                if (++ld_q == g.k) { ld_q = 0; ++ld_r; }
                tap_setup();
            }
        }
    };
    using I0 = std::integral_constant<int, 0>; using I2 = std::integral_constant<int, 2>; using I4 = std::integral_constant<int, 4>;
    using I6 = std::integral_constant<int, NDMA>;          // (= all pieces of a K-tile)

    accv acc0[TM][TN], acc1[BF ? 1 : TM][BF ? 1 : TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc0[i][j][e] = 0.f;
                if constexpr (!BF) acc1[i][j][e] = 0.f;
            }

    // ---- fragment addresses: lane -> (row fr of a 16-row tile, K quarter kq); logical chunks kq (hi) and 4 + kq (lo)
    const int fr = lane & 15, kq = lane >> 4;
    const int sw = (fr >> 1) & 7;                          // tile bases are multiples of 16 rows: the swizzle depends on fr only
    const int a_hi = (wm * WTM + fr) * PROW + ((kq ^ sw) << 4), a_lo = (wm * WTM + fr) * PROW + (((4 + kq) ^ sw) << 4);
    const int b_hi = A_BYTES + (wn * WTN + fr) * PROW + ((kq ^ sw) << 4), b_lo = A_BYTES + (wn * WTN + fr) * PROW + (((4 + kq) ^ sw) << 4);

    // ---- K loop.  Three stages, ring position = K-tile index mod 3 (compile time: the loop is unrolled by 3).  Step t:
    //   fragments of tile t -> MFMAs of the first TM - 1 row tiles, the DMA pieces of a tile under way spread between them;
    //   then [own reads returned | own pieces of tile t + 1 landed: counted vmcnt | barrier]: tile t + 1 is readable and
    //   stage t mod 3 is free for EVERY wave -> the first pieces of tile t + 3 go out; last row tile's MFMAs.
    // The barrier sits in front of the last MFMA group, so the next step's first fragment reads (after it in program
    // order) are hoisted under those MFMAs by the compiler: no wave starts a step with an empty matrix pipe.
    bool pend = false;                                     // a K-tile's pieces are partly issued
    auto step = [&](auto SC, const int kt) {
        constexpr int S = decltype(SC)::value;
        const unsigned char* base = smem + S * STAGE;
        f16x8 fb[TN][2];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            fb[j][0] = *reinterpret_cast<const f16x8*>(base + b_hi + j * 16 * PROW);
            fb[j][1] = *reinterpret_cast<const f16x8*>(base + b_lo + j * 16 * PROW);
        }
        // A fragments one row tile ahead: those of row tile i + 1 are requested BEFORE the MFMAs of row tile i, so the LDS
        // round trip runs under 12 MFMAs (+2-4 % on the 3x3 layers, A/B on one box); the scheduling barrier keeps the MFMAs
        // of row tile TM - 2 in front of the waits (the compiler otherwise sinks them below the barrier)
        f16x8 fap[2][2];
        fap[0][0] = *reinterpret_cast<const f16x8*>(base + a_hi);
        fap[0][1] = *reinterpret_cast<const f16x8*>(base + a_lo);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (i + 1 < TM) {
                fap[(i + 1) & 1][0] = *reinterpret_cast<const f16x8*>(base + a_hi + (i + 1) * 16 * PROW);
                fap[(i + 1) & 1][1] = *reinterpret_cast<const f16x8*>(base + a_lo + (i + 1) * 16 * PROW);
            }
            const f16x8 fa0 = fap[i & 1][0], fa1 = fap[i & 1][1];
            if (i == TM - 1) {
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // every read of stage S has returned
                if (kt + 1 < KT) {
                    if constexpr (NSTAGE == 3) {
                        if (kt + 2 < KT) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        // (all six at once here is 3-5 % SLOWER on the 3x3 layers, measured: this kernel's pieces mostly hit L2)
                        if (kt + 3 < KT) { issue(S, I0{}, I2{}); pend = true; }
                    } else {
                        // two stages: only tile t + 1 is in flight (issued one whole step ago); the other block on this CU
                        // covers the rest of the latency.  All of tile t + 2 goes out at once into the stage just freed.
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        if (kt + 2 < KT) issue(S, I0{}, I6{});
                    }
                }
            }
            if constexpr (BF) {                            // k 0-31, then k 32-63: TN independent MFMAs between the dependent pair
#pragma unroll
                for (int j = 0; j < TN; ++j) acc0[i][j] = pl_mma<true>(fa0, fb[j][0], acc0[i][j]);
#pragma unroll
                for (int j = 0; j < TN; ++j) acc0[i][j] = pl_mma<true>(fa1, fb[j][1], acc0[i][j]);
            } else {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc1[i][j] = pl_mma<false>(fa1, fb[j][0], acc1[i][j]);
                    acc1[i][j] = pl_mma<false>(fa0, fb[j][1], acc1[i][j]);
                    acc0[i][j] = pl_mma<false>(fa0, fb[j][0], acc0[i][j]);
                }
            }
            if constexpr (NSTAGE == 3) {
                if (i == 0 && pend) issue((S + 2) % NSTAGE, I2{}, I4{});
                if (i == 1 && pend) { issue((S + 2) % NSTAGE, I4{}, I6{}); pend = false; }
            }
        }
    };
    static_assert(TM >= 3, "the DMA pieces are spread over the first three row tiles");

    for (int t = 0; t < NSTAGE && t < KT; ++t) issue(t, I0{}, I6{});
    if constexpr (NSTAGE == 3) {
        if (KT >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (KT == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        if (KT >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if constexpr (NSTAGE == 3) {
        int kt = 0;
        for (; kt + 3 <= KT; kt += 3) {
            step(std::integral_constant<int, 0>{}, kt);
            step(std::integral_constant<int, 1>{}, kt + 1);
            step(std::integral_constant<int, 2>{}, kt + 2);
        }
        if (kt < KT) step(std::integral_constant<int, 0>{}, kt);
        if (kt + 1 < KT) step(std::integral_constant<int, 1>{}, kt + 1);
    } else {
        int kt = 0;
        for (; kt + 2 <= KT; kt += 2) {
            step(std::integral_constant<int, 0>{}, kt);
            step(std::integral_constant<int, 1>{}, kt + 1);
        }
        if (kt < KT) step(std::integral_constant<int, 0>{}, kt);
    }

    // ---- epilogue: c = (acc0 + 2^-11 acc1) / (s_A s_B); 16x16 tile: col = lane & 15, row = 4 (lane >> 4) + e.
    // Branch-free raw buffer stores straight from the accumulators (a wave instruction writes 4 rows x 64 B) on a window
    // re-based at the tile's first row and ending at row M (rows past M fall out of range and are dropped), the lane's row /
    // column in the vector offset, the accumulator register e in the scalar offset, the column tile j as the immediate.
    // 1-4 % faster on every layer shape than passing the tile through per-wave LDS patches to store float4 rows (A/B on
    // one box), and it leaves the stage memory alone.
    const float un = BF ? 1.0f : pl_unscale(g.src_amax) * pl_unscale(g.wt_amax);
    const float un1 = un * (1.0f / 2048.0f);
    auto result = [&](int i, int j, int e) -> float {
        if constexpr (BF) return acc0[i][j][e];
        else return acc0[i][j][e] * un + acc1[i][j][e] * un1;
    };
    {
        const unsigned drow = (unsigned)g.ldd * 4u, rrow = (unsigned)g.ldr * 4u;
        const long long row0 = (long long)mt * BM;
        const unsigned long long left = (unsigned long long)(g.M - row0);
        // (ldr == 0: ONE row of N values added to every result row -- the bias of a conv without BatchNorm)
        const unsigned long long dby = left * drow, rby = g.ldr ? left * rrow : (unsigned long long)g.N * 4ull;
        const __amdgpu_buffer_rsrc_t drs = y4_make_rsrc(reinterpret_cast<char*>(g.dst) + row0 * (long long)drow, (unsigned)(dby < 0xfffffff0ull ? dby : 0xfffffff0ull));
        const __amdgpu_buffer_rsrc_t rrs = y4_make_rsrc(g.res ? reinterpret_cast<const char*>(g.res) + row0 * (long long)rrow : nullptr,
                                                        g.res ? (unsigned)(rby < 0xfffffff0ull ? rby : 0xfffffff0ull) : 0u);
        // the epilogue's lane arithmetic starts from an opaque copy of the lane id: otherwise the compiler forms the addresses of
        // all three store variants before the K loop, where the registers are taken (256 x 256 bf16 shape: 14 spilled)
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int fr = lane_e & 15, kq = lane_e >> 4;
        const int colb = n0 + wn * WTN + fr;
        const bool allc = n0 + BN <= g.N;
        if constexpr (DG2) {
            // scattered destination: row m of the class grid -> dx pixel (b, 2 hc + ph, 2 wc + pw); one division pair per 4-row
            // group of a lane, the other three rows by stepping (wc, hc, b).  Window: the whole dx tensor (< 4 GiB, host-checked)
            const __amdgpu_buffer_rsrc_t xrs = y4_make_rsrc(g.dst, (unsigned)((unsigned long long)g.B * g.Hfull * g.Wfull * drow));
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m0 = mt * BM + wm * WTM + 16 * i + 4 * kq;
                int b = m0 / pix_per_img;
                const int rem = m0 - b * pix_per_img;
                int hc = rem / g.Wd, wc = rem - hc * g.Wd;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool rok = m0 + e < g.M;
                    const unsigned po = (unsigned)((b * g.Hfull + 2 * hc + g.cls_ph) * g.Wfull + 2 * wc + g.cls_pw) * drow + (unsigned)colb * 4u;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float v = result(i, j, e);
                        const bool cok = rok && (allc || colb + 16 * j < g.N);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), xrs, (int)(cok ? po + 64u * j : 0xffffffffu), 0, PL_ST_POL);
                    }
                    if (++wc == g.Wd) { wc = 0; if (++hc == g.Hd) { hc = 0; ++b; } }
                }
            }
        } else if constexpr (!YB) {
        if (g.res) {
            // skip operand (dgrad: the gradient arriving over a ResBlock skip connection): the 4 x TN loads of a row tile all go
            // out before the first of them is needed -- issued one by one in front of their stores, each store waited for
            // its own dependent load (128->128 1x1 @76^2 dgrad ran at half its forward rate)
            constexpr int EB = TM > 4 ? 1 : 4;             // accumulator rows per batch of loads (the 128 x 64 wave tile has 16 fewer registers to spare)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const unsigned rl = (unsigned)(wm * WTM + 16 * i + 4 * kq);
                const unsigned dvo = rl * drow + (unsigned)colb * 4u, rvo = rl * rrow + (unsigned)colb * 4u;
#pragma unroll
                for (int e0 = 0; e0 < 4; e0 += EB) {
                    float rv[EB][TN];
#pragma unroll
                    for (int e = 0; e < EB; ++e)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const bool cok = allc || colb + 16 * j < g.N;
                            rv[e][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, (int)(cok ? rvo + 64u * j : 0xffffffffu), (int)((e0 + e) * rrow), 0));
                        }
#pragma unroll
                    for (int e = 0; e < EB; ++e)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const float v = result(i, j, e0 + e);
                            acc0[i][j][e0 + e] = v;        // kept for the column sums
                            const bool cok = allc || colb + 16 * j < g.N;
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v + rv[e][j]), drs, (int)(cok ? dvo + 64u * j : 0xffffffffu), (int)((e0 + e) * drow), PL_ST_POL);
                        }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const unsigned rl = (unsigned)(wm * WTM + 16 * i + 4 * kq);
                const unsigned dvo = rl * drow + (unsigned)colb * 4u;
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float v = result(i, j, e);
                        acc0[i][j][e] = v;                 // kept for the column sums
                        const bool cok = allc || colb + 16 * j < g.N;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), drs, (int)(cok ? dvo + 64u * j : 0xffffffffu), (int)(e * drow), PL_ST_POL);
                    }
            }
        }
        } else {
            // bf16 result (conv mode 'bf16': what the BatchNorm sweeps then read is half the bytes): rounded to nearest even,
            // the column sums are taken over the ROUNDED values (they normalise what is stored); lanes 2k / 2k + 1 hold adjacent
            // columns: the even lane packs both and stores one dword (N is a multiple of 32 on this path: pairs are whole)
            // (the column sums are formed right here, row tile by row tile, and parked in acc0[0][j][0..1] for the block fold below:
            // a second pass over 128 live accumulators cost the 128 x 64 wave tile a spill)
            float csum[TN], cssum[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) { csum[j] = 0.f; cssum[j] = 0.f; }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const unsigned rl = (unsigned)(wm * WTM + 16 * i + 4 * kq);
                const unsigned dvo = rl * drow + (unsigned)colb * 2u;
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float v = result(i, j, e);
                        unsigned u = __builtin_bit_cast(unsigned, v);
                        u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;          // RN-even to bf16 (finite values)
                        const float vr = __builtin_bit_cast(float, u);
                        csum[j] += vr; cssum[j] += vr * vr;
                        const unsigned nb = (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0xB1, 0xf, 0xf, true);   // lane ^ 1
                        const bool cok = (allc || colb + 16 * j < g.N) && !(fr & 1);
                        __builtin_amdgcn_raw_buffer_store_b32((u >> 16) | nb, drs, (int)(cok ? dvo + 32u * j : 0xffffffffu), (int)(e * drow), PL_ST_POL);
                    }
                __builtin_amdgcn_sched_barrier(0);         // (one row tile at a time)
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) { acc0[0][j][0] = csum[j]; acc0[0][j][1] = cssum[j]; }
        }
    }
    if (g.stats) {                                         // rows past M are exact zeros (their operand rows were zero-filled)
        __syncthreads();                                   // every wave has left the last stage
        float* red = reinterpret_cast<float*>(smem);       // [WM][BN][2]
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float cs = 0.f, css = 0.f;
            if constexpr (YB) {
                cs = acc0[0][j][0]; css = acc0[0][j][1];   // (summed in the store loop)
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float v = acc0[i][j][e]; cs += v; css += v * v; }
            }
            cs += __shfl_xor(cs, 16, 64); css += __shfl_xor(css, 16, 64);
            cs += __shfl_xor(cs, 32, 64); css += __shfl_xor(css, 32, 64);
            if (kq == 0) {
                const int c = wn * WTN + j * 16 + fr;
                red[(wm * BN + c) * 2 + 0] = cs;
                red[(wm * BN + c) * 2 + 1] = css;
            }
        }
        __syncthreads();
        for (int c = tid; c < BN; c += NTHR) {
            float cs = 0.f, css = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) { cs += red[(w * BN + c) * 2]; css += red[(w * BN + c) * 2 + 1]; }
            const int n = n0 + c;
            if (n < g.N) {
                g.stats[((long long)mt * 2 + 0) * g.N + n] = cs;
                g.stats[((long long)mt * 2 + 1) * g.N + n] = css;
            }
        }
    }
}

// ==================================================================================== wgrad over planes
// dW[n][j] = sum_p dy[p][n] * x[p @ tap(j)][c(j)], j = (tap, c): a GEMM whose K dimension is the pixel index, i.e. both
// operands are needed with the CONTRACTION index along the MFMA k axis although they sit in HBM pixel-major.  The LDS
// image keeps them pixel-major as the DMA delivers them -- one row per (32-channel chunk, pixel), rows of 128 B
// [hi | lo] -- and the fragments come out of it through `ds_read_b64_tr_b16`, the hardware transpose read: a 16-lane
// group reads a block of 4 pixel rows x 16 channels and each lane receives ONE channel's 4 consecutive pixels, which is
// exactly half of an 8-deep k fragment; two such reads per plane and tile, no VALU, no register transposes.
//   * K-step = 32 consecutive pixels (raster order over batch x H x W); split-K over pixel ranges into fp32 slabs.
//   * block tile 128 (n) x 256 (j = 8 chunks of 32 channels, each chunk with its own filter tap), 8 waves as 2 x 4,
//     wave tile 64 x 64, 48 MFMAs per wave and K-step; the same three-stage DMA ring as the forward kernel.
//   * swizzle: the four 32-B segments of a row ([hi 0-15 | hi 16-31 | lo 0-15 | lo 16-31]) are permuted by
//     g(p) = ((p >> 1) & 1) | (((p >> 3) & 1) << 1) (XOR on the segment index), which makes every transposed read
//     conflict-free (a 32-lane half reads pixels {0-3, 8-11} or {4-7, 12-15} of one segment: 8 distinct 32-B slots).
//   * stride 1 only (the layers this serves): pixel p of dy is pixel p of x, so a lane's x address is linear in p and a
//     tap is a constant offset plus a validity test on (h, w).
struct PlaneWgradGeom {
    const unsigned char* x;            // planes [B][H][W][Cin / 32][128 B]
    const unsigned char* dy;           // planes [B][H][W][Cout / 32][128 B]
    float* out;                        // [splits][Cout][J] slabs, or dW itself when splits == 1
    int B, H, W, Cin, Cout, k, pad, M, J;   // H, W, M: the dy grid (= the x grid at stride 1; at stride 2 x is 2H x 2W)
    int ldy_ch;                        // channels per dy pixel row: Cout rounded up to whole K tiles (pad channels hold zeros)
    int ntn, ntj, splits, steps_per_split;
    unsigned long long x_total_bytes, dy_total_bytes;
    const unsigned* x_amax; const unsigned* dy_amax;
};

typedef __fp16 trh4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__device__ __forceinline__ f16x8 tr_frag(const unsigned char* p) {      // k 0..3 from p, k 4..7 four pixel rows further
    typedef __attribute__((address_space(3))) trh4* lp;
    const trh4 a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp)(p));
    const trh4 b = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp)(p + 4 * PROW));
    f16x8 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) { r[e] = (_Float16)a[e]; r[4 + e] = (_Float16)b[e]; }
    return r;
}

template <int N> __device__ __forceinline__ void pl_wait_vm() {      // counted wait with a compile-time literal
    static_assert(N == 0 || N == 3 || N == 4 || N == 6 || N == 8 || N == 12, "add the literal");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
}

// BF (conv mode 2): plain bf16 operands.  A 128-B LDS row is then one pixel's 64-channel chunk, its four 32-B segments are
// four 16-channel tiles of ONE plane (f16x2: [hi 0-15 | hi 16-31 | lo 0-15 | lo 16-31]), one bf16 MFMA per tile pair and
// K-step into one accumulator set; <128, 256>: 24 KB per K-step, <256, 256> (wave tile 128 x 64): 32 KB per K-step, the f16x2
// kernel's bytes per MFMA cycle.  The result leaves through raw buffer stores (the LDS patches of the f16x2 form do not fit).
// SD = 2: the 3x3 stride-2 layers (H, W of the input even): dy pixel p = (b, h, w) reads x pixel (b, 2h + r - 1, 2w + q - 1),
// whose raster index is 4 p - 2 w + (r - 1) 2W + (q - 1) -- still one multiply-add per lane and K-step from the running (p, w).
template <int TN_, int TJ_, bool BF, int SD = 1>
__global__ __launch_bounds__(512, 1) void wgrad_planes_mfma(const PlaneWgradGeom g) {
    constexpr int NWAVE = 8, WN2 = 2, WJ = 4;
    constexpr int BAR = 2;                                 // row tile in front of whose MFMAs the barrier sits (see `step`)
    constexpr int CH = BF ? 64 : 32, CSH = BF ? 6 : 5;     // channels per 128-B row
    constexpr int WTN = TN_ / WN2, WTJ = TJ_ / WJ, TM = WTN / 16, TN = WTJ / 16;
    static_assert(TN == 4 && (TM == 4 || (BF && TM == 8)), "wave tile 64 x 64 (bf16 also 128 x 64)");
    constexpr int A_BYTES = (TN_ / CH) * 32 * PROW, B_BYTES = (TJ_ / CH) * 32 * PROW, STAGE = A_BYTES + B_BYTES, NSTAGE = 3;
    constexpr int PA = A_BYTES / 1024 / NWAVE, PB = B_BYTES / 1024 / NWAVE, NDMA = PA + PB;
    static_assert(PA >= 1 && PB >= 1 && (NDMA == 6 || NDMA == 3 || NDMA == 4), "DMA pieces per wave and K-step (counted vmcnt literals)");
    typedef float accv __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn2 = wave >> 2, wj = wave & 3;
    const int tiles = g.ntn * g.ntj;
    int bid = y4_xcd_remap(blockIdx.x, tiles * g.splits);
    const int split = bid / tiles;
    bid -= split * tiles;
    const int tn = bid / g.ntj, tj = bid - tn * g.ntj;
    const int n0 = tn * TN_, j0 = tj * TJ_;
    const int CC = g.Cin >> CSH;

    const int steps_total = (g.M + 31) >> 5;
    const int s0 = split * g.steps_per_split;
    int KT = steps_total - s0;
    if (KT > g.steps_per_split) KT = g.steps_per_split;
    if (KT < 0) KT = 0;
    const long long P0 = (long long)s0 * 32;               // first pixel of this split

    // ---- 32-bit windows: dy from pixel P0; x from pixel P0 - (pad W + pad) (the earliest pixel a tap can reach), >= 0
    const unsigned pitch_x = (unsigned)g.Cin * 4u, pitch_dy = (unsigned)g.ldy_ch * 4u;
    const long long back = SD == 1 ? (long long)g.pad * g.W + g.pad : 2ll * g.W + 2ll * g.W + 1;     // S = 2: 2 w <= 2 W, one x row (2 W) + 1
    const long long Pw = SD * SD * P0 > back ? SD * SD * P0 - back : 0;     // first x pixel of the window (x raster)
    const unsigned long long dy_skip = (unsigned long long)P0 * pitch_dy, x_skip = (unsigned long long)Pw * pitch_x;
    const unsigned long long dy_left = g.dy_total_bytes > dy_skip ? g.dy_total_bytes - dy_skip : 0ull;
    const unsigned long long x_left = g.x_total_bytes > x_skip ? g.x_total_bytes - x_skip : 0ull;
    const __amdgpu_buffer_rsrc_t dy_rsrc = y4_make_rsrc(g.dy + dy_skip, (unsigned)(dy_left < 0xfffffff0ull ? dy_left : 0xfffffff0ull));
    const __amdgpu_buffer_rsrc_t x_rsrc = y4_make_rsrc(g.x + x_skip, (unsigned)(x_left < 0xfffffff0ull ? x_left : 0xfffffff0ull));
    const unsigned OOB = 0xffffffffu;

    // ---- DMA slots: 8 lanes per row; lane -> (pixel inside its octet, 16-B position in the row)
    const int lr = lane >> 3, pos = lane & 7;
    // dy pieces a = wave * 2 + i: channel chunk a >> 2, pixel octet a & 3
    unsigned a_voff[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int a = wave * PA + i;
        const int chunk = a >> 2, p = (a & 3) * 8 + lr;
        const int gsw = ((p >> 1) & 1) | (((p >> 3) & 1) << 1);
        const unsigned srcoff = (unsigned)((((pos >> 1) ^ gsw) << 5) + ((pos & 1) << 4));
        a_voff[i] = (n0 + chunk * CH) < g.Cout ? (unsigned)p * pitch_dy + (unsigned)(n0 / CH + chunk) * 128u + srcoff : OOB;
    }
    // x pieces: pixel octet wave & 3, j chunks (wave >> 2) * 4 + i: ONE pixel per lane, four taps / channel chunks
    const int xp = (wave & 3) * 8 + lr;                    // pixel of this lane inside the K-step
    unsigned x_const[PB];                                  // tap offset + channel chunk + swizzled position; OOB: chunk beyond J
    unsigned x_pk = 0u;                                    // per piece 8 bits: (dh + 2) | (dw + 2) << 4; dh + 2 = 15: never valid
    {
        const int gsw = ((xp >> 1) & 1) | (((xp >> 3) & 1) << 1);
        const unsigned srcoff = (unsigned)((((pos >> 1) ^ gsw) << 5) + ((pos & 1) << 4));
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int jg = j0 / CH + (wave >> 2) * PB + i;
            const bool ok = jg * CH < g.J;
            const int tap = ok ? jg / CC : 0;
            const int c32 = jg - tap * CC;
            const int r = tap / g.k, q = tap - r * g.k;
            x_pk |= (unsigned)((ok ? r - g.pad + 2 : 15) | ((q - g.pad + 2) << 4)) << (8 * i);
            x_const[i] = (unsigned)((long long)(SD * SD * P0 - Pw) * pitch_x) + (unsigned)(((r - g.pad) * (SD * g.W) + (q - g.pad)) * (int)pitch_x) +
                         (unsigned)c32 * 128u + srcoff;
        }
    }
    // running raster position of this lane's pixel of the NEXT K-step to be issued
    int lp = (int)P0 + xp;                                 // (M < 2^31, checked on the host; may run past M by < 2^12)
    int lh, lw;
    {
        const long long hw = (long long)g.H * g.W;
        const long long rem = (long long)lp % hw;
        lh = (int)(rem / g.W); lw = (int)(rem - (long long)lh * g.W);
    }
    unsigned lbase = (unsigned)xp * pitch_x;               // (pixel - P0) * pitch, mod 2^32
    unsigned ldy = 0u;                                     // K-step offset of the dy rows
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto issue = [&](int stage, auto P0c, auto P1c) {
        constexpr int p0 = decltype(P0c)::value, p1 = decltype(P1c)::value;
        unsigned char* st = smem + stage * STAGE;
#pragma unroll
        for (int p = p0; p < p1; ++p) {
            if (p < PA) {
                const unsigned vo = a_voff[p] == OOB ? OOB : a_voff[p] + ldy;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(dy_rsrc, (lds_ptr)(st + (wave * PA + p) * 1024), 16, (int)vo, 0, 0, 0);
            } else {
                const int i = p - PA;
                const int dh = (int)((x_pk >> (8 * i)) & 15u) - 2, dw = (int)((x_pk >> (8 * i + 4)) & 15u) - 2;
                const bool ok = lp < g.M && (unsigned)(SD * lh + dh) < (unsigned)(SD * g.H) && (unsigned)(SD * lw + dw) < (unsigned)(SD * g.W);
                const unsigned lb = SD == 1 ? lbase : (unsigned)(4 * (lp - (int)P0) - 2 * lw) * pitch_x;     // (x pixel - S^2 P0) * pitch
                const unsigned vo = ok ? lb + x_const[i] : OOB;
                const int b = ((wave >> 2) * PB + i) * 4 + (wave & 3);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_ptr)(st + A_BYTES + b * 1024), 16, (int)vo, 0, 0, 0);
            }
        }
        if constexpr (p1 == NDMA) {                        // K-step complete: advance by 32 pixels
            lp += 32; lbase += 32u * pitch_x; ldy += 32u * pitch_dy;
            lw += 32;
            while (lw >= g.W) { lw -= g.W; if (++lh == g.H) lh = 0; }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I6 = std::integral_constant<int, NDMA>;          // (= all pieces of a K-tile)

    accv acc0[TM][TN], acc1[BF ? 1 : TM][BF ? 1 : TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc0[i][j][e] = 0.f;
                if constexpr (!BF) acc1[i][j][e] = 0.f;
            }

    // ---- transposed fragment reads: lane -> (k group grp: pixels 8 grp + {0..3} and + 4, block row qrow, 8-B piece pp)
    const int grp = lane >> 4, qrow = (lane >> 2) & 3, pp = lane & 3;
    const int gsw = ((qrow >> 1) & 1) | ((grp & 1) << 1);
    const int rowoff = (8 * grp + qrow) * PROW + pp * 8;
    // logical segment s = plane * 2 + (tile & 1) lives at position s ^ gsw
    int segoff[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) segoff[s4] = rowoff + ((s4 ^ gsw) << 5);
    const int a_tile0 = (wn2 * (WTN / CH)) * 32 * PROW;                // chunk of this wave's first n tile
    const int b_tile0 = A_BYTES + (wj * (WTJ / CH)) * 32 * PROW;
    // 16-channel tile t of a wave's range, plane pl (f16x2 only) -> offset of its fragment inside the operand's stage image
    auto frag_off = [&](int t, int pl) -> int {
        if constexpr (BF) return (t >> 2) * 32 * PROW + segoff[t & 3];
        else return (t >> 1) * 32 * PROW + segoff[2 * pl + (t & 1)];
    };
    constexpr int NPL = BF ? 1 : 2;

    auto step = [&](auto SC, const int kt) {
        constexpr int S = decltype(SC)::value;
        const unsigned char* base = smem + S * STAGE;
        f16x8 fb[TN][NPL];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) fb[j][pl] = tr_frag(base + b_tile0 + frag_off(j, pl));
        // A fragments ahead of their MFMAs: row tile i + 1 is requested before the MFMAs of row tile i; at the barrier
        // position BAR everything still missing is requested, so that the stage goes back to the DMA 24 MFMAs before the step
        // ends (BAR = 3 -> 2: +5...9 %; BAR = 1 or 0 need three / four fragment sets live and spill inside the loop: 1.6x slower)
        f16x8 fap[TM][NPL];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) fap[0][pl] = tr_frag(base + a_tile0 + frag_off(0, pl));
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = i + 1; r < TM; ++r)
                if ((i < BAR && r == i + 1) || (i == BAR && r > i)) {
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) fap[r][pl] = tr_frag(base + a_tile0 + frag_off(r, pl));
                }
            const f16x8 fa0 = fap[i][0], fa1 = fap[i][NPL - 1];
            if (i == BAR) {
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (kt + 1 < KT) {
                    if (kt + 2 < KT) pl_wait_vm<NDMA>();
                    else pl_wait_vm<0>();
                    __builtin_amdgcn_s_barrier();
                    // all six pieces of tile t + 3 at once, the moment its stage is free: this kernel streams fresh pixels every
                    // K-step and waits for them (with the DMA switched off it runs 1.4-1.8x faster), so two whole steps of lead
                    // beat spreading the issues between the MFMA groups (+3...27 %, 12 % on average, A/B on one box)
                    if (kt + 3 < KT) issue(S, I0{}, I6{});
                }
            }
            if constexpr (BF) {
#pragma unroll
                for (int j = 0; j < TN; ++j) acc0[i][j] = pl_mma<true>(fa0, fb[j][0], acc0[i][j]);
            } else {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc1[i][j] = pl_mma<false>(fa1, fb[j][0], acc1[i][j]);
                    acc1[i][j] = pl_mma<false>(fa0, fb[j][NPL - 1], acc1[i][j]);
                    acc0[i][j] = pl_mma<false>(fa0, fb[j][0], acc0[i][j]);
                }
            }
        }
    };

    for (int t = 0; t < 3 && t < KT; ++t) issue(t, I0{}, I6{});
    if (KT >= 3) pl_wait_vm<2 * NDMA>();
    else if (KT == 2) pl_wait_vm<NDMA>();
    else pl_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    {
        int kt = 0;
        for (; kt + 3 <= KT; kt += 3) {
            step(std::integral_constant<int, 0>{}, kt);
            step(std::integral_constant<int, 1>{}, kt + 1);
            step(std::integral_constant<int, 2>{}, kt + 2);
        }
        if (kt < KT) step(std::integral_constant<int, 0>{}, kt);
        if (kt + 1 < KT) step(std::integral_constant<int, 1>{}, kt + 1);
    }

    if constexpr (BF) {
        // ---- epilogue (bf16): raw buffer stores straight from the accumulators; a wave instruction writes 4 rows x 64 B
        float* outb = g.out + (long long)split * g.Cout * g.J;
        const __amdgpu_buffer_rsrc_t ors = y4_make_rsrc(outb, (unsigned)((unsigned long long)g.Cout * g.J * 4ull));
        int lane_b = lane;
        asm volatile("" : "+v"(lane_b));
        const int frb = lane_b & 15, kqb = lane_b >> 4;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = n0 + wn2 * WTN + i * 16 + 4 * kqb + e;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int jj = j0 + wj * WTJ + j * 16 + frb;
                    const unsigned off = (n < g.Cout && jj < g.J) ? ((unsigned)n * (unsigned)g.J + (unsigned)jj) * 4u : 0xffffffffu;
                    const float v = acc0[i][j][e];         // (a bit_cast straight on the vector-element lvalue reads element 0)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ors, (int)off, 0, 0);
                }
            }
        return;
    }
    // ---- epilogue: out[n][j] = (acc0 + 2^-11 acc1) / (s_dy s_x), through per-wave LDS patches as float4 rows of j
    const float un = pl_unscale(g.dy_amax) * pl_unscale(g.x_amax);
    const float un1 = un * (1.0f / 2048.0f);
    __syncthreads();
    constexpr int EP = WTJ + 4;
    static_assert(BF || NWAVE * WTN * EP * 4 <= NSTAGE * STAGE, "epilogue patches must fit the stages");
    float* patch = reinterpret_cast<float*>(smem) + wave * (WTN * EP);
    // the epilogue's lane arithmetic starts from an opaque copy of the lane id: otherwise the compiler forms these addresses
    // before the K loop, where every one of the 256 registers is taken, and parks them in scratch
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int fr = lane_e & 15, kq = lane_e >> 4;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) patch[(i * 16 + 4 * kq + e) * EP + j * 16 + fr] = acc0[i][j][e] * un + acc1[BF ? 0 : i][BF ? 0 : j][e] * un1;
    float* out = g.out + (long long)split * g.Cout * g.J;
    const int c4 = (lane_e & 15) * 4;
    const int jv = j0 + wj * WTJ + c4;
#pragma unroll
    for (int it = 0; it < WTN / 4; ++it) {
        const int row = it * 4 + (lane_e >> 4);
        const int n = n0 + wn2 * WTN + row;
        if (n < g.Cout && jv < g.J)
            *reinterpret_cast<f32x4v*>(out + (long long)n * g.J + jv) = *reinterpret_cast<const f32x4v*>(patch + row * EP + c4);
    }
}

// ---- fp32 NHWC (pitch ld) -> planes; one thread per 4 channels of a pixel
__global__ __launch_bounds__(256) void planes_split_kernel(const float* __restrict__ x, long long ld, long long M, int C,
                                                           const unsigned* __restrict__ amax, unsigned char* __restrict__ planes,
                                                           long long pitch, int cvalid) {
    const float s = pl_scale(amax);
    const int C4 = C >> 2;
    const long long total = M * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / C4;
        const int c = (int)(i - m * C4) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + m * ld + c);
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        h4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float t = c + e < cvalid ? v[e] * s : 0.f;   // (channels >= cvalid: pad of the source rows, written as zeros)
            hi[e] = (_Float16)t;
            lo[e] = (_Float16)((t - (float)hi[e]) * 2048.f);
        }
        unsigned char* row = planes + m * pitch + (c >> 5) * 128 + (c & 31) * 2;
        *reinterpret_cast<h4*>(row) = hi;
        *reinterpret_cast<h4*>(row + 64) = lo;
    }
}

// ---- conv mode 2 (bf16): the "planes" of an activation are its bf16 values, dense per pixel in the FIRST HALF of the fp32-sized
// row (pitch 4 C bytes: same allocation, same addressing as the f16x2 planes; the second half is never touched)
__device__ __forceinline__ unsigned short pl_bf16_bits(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }
__device__ __forceinline__ unsigned pair_swap_u(unsigned v) {             // value of lane ^ 1 (quad_perm [1, 0, 3, 2])
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);
}
__global__ __launch_bounds__(256) void planes_split_bf16_kernel(const float* __restrict__ x, long long ld, long long M, int C,
                                                                unsigned char* __restrict__ planes, long long pitch, int cvalid) {
    const int C4 = C >> 2;
    const long long total = M * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / C4;
        const int c = (int)(i - m * C4) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + m * ld + c);
        typedef unsigned short us4 __attribute__((ext_vector_type(4)));
        us4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = c + e < cvalid ? pl_bf16_bits(v[e]) : (unsigned short)0;
        *reinterpret_cast<us4*>(planes + m * pitch + c * 2) = o;
    }
}
// ==================================================================================== bf16 streaming 1x1 (small K, small N)
// The 1x1 layers with K, N <= 128 on the large maps (stage 1 / 2 in the bf16 mode) are pure byte movers: 128-256 B in and out per
// pixel for 8-32 k MACs.  The tile kernel above spends a block launch, two barriers and a filter fetch per 128 pixels on them and
// reaches 40 % of the HBM rate (64 -> 64 @304^2: 1.39 ms forward against 0.55 of bytes).  Here, as in conv1x1_stream_f16x2
// (conv_f16x2.hip): the whole bf16 filter stays in LDS for the life of a persistent block, every wave streams its own 32 pixel
// rows from global memory straight into the A layout of v_mfma_f32_32x32x16_bf16 (lane = pixel, 8 consecutive channels = 16
// contiguous bytes of the pixel's bf16 row: no conversion, no LDS for activations, no barrier in the loop), the next tile's loads
// fly under this tile's MFMAs and stores, BatchNorm column sums accumulate in registers (one partial row per block).
// YB: the result leaves as bf16 (column pairs packed through DPP, sums over the rounded values) -- else fp32 (+ skip operand).
template <int KS, int NT, int NW, bool YB>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void conv1x1_stream_bf16(const PlaneConvGeom g) {
    constexpr int NTHR = NW * 64, TROWS = NW * 32;
    constexpr int K = KS * 16;
    constexpr int PITCH = K * 2 + 16;                      // LDS row pitch: conflict-free ds_read_b128 for K = 64 / 128
    constexpr int N32 = NT * 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_sb[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    {
        constexpr int CPR = K / 8;                         // 16-B chunks per filter row
        for (int i = tid; i < N32 * CPR; i += NTHR) {
            const int row = i / CPR, ch = i - row * CPR;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (row < g.N) v = *reinterpret_cast<const u32x4*>(g.wt + (size_t)row * (K * 2) + ch * 16);
            *reinterpret_cast<u32x4*>(smem_sb + row * PITCH + ch * 16) = v;
        }
    }
    __syncthreads();
    const unsigned pix_bytes = (unsigned)g.Cs * 4u;
    const int mtiles = g.mtiles;
    u32x4 ra0[KS], ra1[KS];
    const unsigned lane_off = (unsigned)(wave * 32 + fr) * pix_bytes + (unsigned)fh * 16u;
    auto load = [&](u32x4 (&ra)[KS], int tile) {
        // window re-based at the tile's first row, ending at row M: rows past M read zeros (the 128-channel maps of stage 1
        // exceed one 32-bit window at bs = 128)
        const long long row0 = (long long)tile * TROWS;
        const unsigned long long left = (unsigned long long)(g.M - row0) * pix_bytes;
        const __amdgpu_buffer_rsrc_t src_rsrc = y4_make_rsrc(g.src + row0 * (long long)pix_bytes,
                                                            (unsigned)(left < 0xfffffff0ull ? left : 0xfffffff0ull));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) ra[ks] = __builtin_amdgcn_raw_buffer_load_b128(src_rsrc, (int)lane_off, ks * 32, 0);
    };
    float cs[NT], css[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) { cs[j] = 0.f; css[j] = 0.f; }
    const unsigned char* b_frag = smem_sb + fr * PITCH + fh * 16;
    f32x16 acc[NT];
    auto mma = [&](u32x4 (&ra)[KS]) {
        // the filter fragments are loop-invariant: up to 8 of them (32 registers) stay in registers across tiles, more are
        // re-read from LDS per tile (the K = N = 128 form would spill)
        if constexpr (KS * NT > 8) asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 fa = __builtin_bit_cast(bf16x8, ra[ks]);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const bf16x8 fb = *reinterpret_cast<const bf16x8*>(b_frag + (j * 32) * PITCH + ks * 32);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[j], 0, 0, 0);
            }
        }
    };
    // Branch-free epilogue: raw buffer stores on a window re-based at the tile's first row whose extent ends at row M (rows past M
    // fall out of range and are dropped), the lane's row and column in the vector offset, the accumulator register's row
    // ((e & 3) + 8 (e >> 2)) times the pitch in the scalar offset (dword stores: no wide-store hazard, see pointwise.hip)
    const unsigned drow = (unsigned)g.ldd * 4u, rrow = (unsigned)g.ldr * 4u;
    auto epilogue = [&](int tile) {
        const long long row0 = (long long)tile * TROWS;
        const unsigned long long left = (unsigned long long)(g.M - row0);
        const unsigned long long dby = left * drow, rby = left * rrow;
        const __amdgpu_buffer_rsrc_t drs = y4_make_rsrc(reinterpret_cast<char*>(g.dst) + row0 * (long long)drow,
                                                        (unsigned)(dby < 0xfffffff0ull ? dby : 0xfffffff0ull));
        const __amdgpu_buffer_rsrc_t rrs = y4_make_rsrc(g.res ? reinterpret_cast<const char*>(g.res) + row0 * (long long)rrow : nullptr,
                                                        g.res ? (unsigned)(rby < 0xfffffff0ull ? rby : 0xfffffff0ull) : 0u);
        const unsigned lrow = (unsigned)(wave * 32 + 4 * fh);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = j * 32 + fr;
            const bool nok = n < g.N;
            if constexpr (YB) {
                // adjacent lanes hold adjacent columns of one row: the even lane packs both and stores one dword at byte 2 n
                const bool st = nok && !(fr & 1);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned vo = st ? (lrow + 8u * q) * drow + (unsigned)n * 2u : 0xffffffffu;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const unsigned me = (unsigned)pl_bf16_bits(acc[j][q * 4 + r]);
                        const float vr = __uint_as_float(me << 16);
                        cs[j] += vr; css[j] += vr * vr;        // rows past M are exact zeros
                        const unsigned nb = pair_swap_u(me);
                        __builtin_amdgcn_raw_buffer_store_b32(me | (nb << 16), drs, (int)vo, (int)(r * drow), PL_ST_POL);
                    }
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned vo = nok ? (lrow + 8u * q) * drow + (unsigned)n * 4u : 0xffffffffu;
                    const unsigned ro = nok ? (lrow + 8u * q) * rrow + (unsigned)n * 4u : 0xffffffffu;
                    float rr[4] = {0.f, 0.f, 0.f, 0.f};
                    if (g.res) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            rr[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, (int)ro, (int)(r * rrow), 0));
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = acc[j][q * 4 + r];
                        cs[j] += v; css[j] += v * v;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v + rr[r]), drs, (int)vo, (int)(r * drow), PL_ST_POL);
                    }
                }
            }
        }
    };
    int tile = blockIdx.x;
    if (tile < mtiles) load(ra0, tile);
    while (tile < mtiles) {
        const int t1 = tile + gridDim.x;
        if (t1 < mtiles) load(ra1, t1);
        mma(ra0); epilogue(tile);
        if (t1 >= mtiles) break;
        const int t2 = t1 + gridDim.x;
        if (t2 < mtiles) load(ra0, t2);
        mma(ra1); epilogue(t1);
        tile = t2;
    }
    if (g.stats) {                                         // one partial row per block: [gridDim][2][N]
        __syncthreads();                                   // every wave is done with the filter
        float* red = reinterpret_cast<float*>(smem_sb);    // [NW][N32][2]
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

template <int KS, int NT, int NW, bool YB>
static int launch_stream_bf16(const PlaneConvGeom& g0, hipStream_t st, int* nparts) {
    PlaneConvGeom g = g0;
    g.mtiles = (g.M + NW * 32 - 1) / (NW * 32);
    g.ntiles = 1;
    size_t smem = (size_t)NT * 32 * (KS * 32 + 16);
    const size_t red = (size_t)NW * NT * 32 * 2 * sizeof(float);
    if (smem < red) smem = red;
    auto kern = conv1x1_stream_bf16<KS, NT, NW, YB>;
    static Y4DynLds lds_attr;                              // per device, see common.h
    if (!lds_attr.ensure(reinterpret_cast<const void*>(kern), smem)) return Y4_ERR_LAUNCH;
    const int resident = NW == 8 ? 256 : 512;              // blocks per CU: 1 (8 waves) or 2
    const int grid = g.mtiles < resident ? g.mtiles : resident;
    if (nparts) *nparts = grid;
    y4::note_kernel("conv1x1_stream_bf16<%d, %d, %d, %s>", KS, NT, NW, YB ? "true" : "false");
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), smem, st, g);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}
// eligibility: bf16 operands, 1x1 stride 1, K in {64, 128}, N <= 128, rows enough for a persistent launch (Y4_BF_STREAM=0: off)
static bool stream_bf16_ok(const PlaneConvGeom& g, bool dst_bf16) {
    static const bool off = getenv("Y4_BF_STREAM") && atoi(getenv("Y4_BF_STREAM")) == 0;
    if (off || g.k != 1 || g.stride != 1 || (g.Cs != 64 && g.Cs != 128) || g.N > 128 || g.N < 1) return false;
    if (dst_bf16 && (g.res || (g.N & 1))) return false;
    return g.M >= 128 * 1024;
}
template <bool YB>
static int dispatch_stream_bf16(const PlaneConvGeom& g, hipStream_t st, int* nparts) {
    const int nt = (g.N + 31) / 32;
    if (g.Cs == 64) {
        if (nt == 1) return launch_stream_bf16<4, 1, 4, YB>(g, st, nparts);
        if (nt == 2) return launch_stream_bf16<4, 2, 4, YB>(g, st, nparts);
        return launch_stream_bf16<4, 4, 4, YB>(g, st, nparts);
    }
    if (nt == 1) return launch_stream_bf16<8, 1, 4, YB>(g, st, nparts);
    if (nt == 2) return launch_stream_bf16<8, 2, 4, YB>(g, st, nparts);
    return launch_stream_bf16<8, 4, 4, YB>(g, st, nparts);
}

// filter [Cout][kk][Cin] fp32 -> bf16, same order (forward operand), and optionally [Cin][kk (mirrored)][Cout] (dgrad operand)
__global__ __launch_bounds__(256) void bf16_filter_kernel(const float* __restrict__ w, unsigned short* __restrict__ fwd,
                                                          unsigned short* __restrict__ tr, int Cout, int Cin, int kk, int mirror) {
    const long long n = (long long)Cout * kk * Cin;
    const long long i0 = blockIdx.x * (long long)blockDim.x + threadIdx.x, step = (long long)gridDim.x * blockDim.x;
    if (fwd)
        for (long long i = i0; i < n; i += step) fwd[i] = pl_bf16_bits(w[i]);
    if (tr)
        for (long long i = i0; i < n; i += step) {
            const int nn = (int)(i % Cout);
            const long long t = i / Cout;
            const int tp = (int)(t % kk);
            const int tap = mirror ? kk - 1 - tp : tp;
            const int c = (int)(t / kk);
            tr[i] = pl_bf16_bits(w[((long long)nn * kk + tap) * Cin + c]);
        }
}
int bf16_filter(const float* w, unsigned short* fwd, unsigned short* tr, int Cout, int Cin, int kk, bool mirror, hipStream_t st) {
    const long long n = (long long)Cout * kk * Cin;
    const int blocks = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    hipLaunchKernelGGL(bf16_filter_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, st, w, fwd, tr, Cout, Cin, kk, mirror ? 1 : 0);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

template <int BM, int BN, int WM, int WN, bool BF, bool YB = false, bool DG2 = false>
int launch_conv_planes(const PlaneConvGeom& g0, hipStream_t st) {
    PlaneConvGeom g = g0;
    g.mtiles = (g.M + BM - 1) / BM;
    g.ntiles = (g.N + BN - 1) / BN;
    constexpr int NW = WM * WN;
    constexpr size_t stage = (size_t)(BM + BN) * PROW;
    constexpr size_t smem = ((NW == 8 && 3 * stage <= 160 * 1024) ? 3ull : 2ull) * stage;   // the stages; the epilogue needs WM * BN * 8 B of them
    static_assert(NW == 8 || 2 * smem <= 160 * 1024, "two blocks per CU");
    static_assert(smem <= 160 * 1024, "LDS");
    auto kern = conv_planes_mfma<BM, BN, WM, WN, BF, YB, DG2>;
    static Y4DynLds lds_attr;                              // per device, see common.h
    if (!lds_attr.ensure(reinterpret_cast<const void*>(kern), smem)) return Y4_ERR_LAUNCH;
    y4::note_kernel("conv_planes_mfma<%d, %d, %d, %d, %s, %s, %s>", BM, BN, WM, WN, BF ? "true" : "false", YB ? "true" : "false",
                    DG2 ? "true" : "false");
    hipLaunchKernelGGL(kern, dim3(g.mtiles * g.ntiles), dim3(NW * 64), smem, st, g);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

}  // namespace

namespace y4 {

bool planes_conv_ok(int Cin, int Cout, int k, int stride) {
    return Cin > 0 && (Cin & 31) == 0 && Cout > 0 && (Cout & 3) == 0 && (k == 1 || k == 3) && (stride == 1 || stride == 2);
}

// The DMA kernels address their operands through 32-bit buffer windows.  One predicate per kernel family, used by the launchers
// below AND by y4_conv_planes_fit (the host asks BEFORE it lets a producer emit planes: an operand that exists only pre-split
// has no register-staged kernel to fall back to).
// forward form: the window starts at the first image a 256-row tile touches
static bool conv_window_ok(int B, int Hs, int Ws, int Cs, int N, int k, int stride, bool bf) {
    const int pad = (k - 1) / 2;
    const long long Hd = (Hs + 2 * pad - k) / stride + 1, Wd = (Ws + 2 * pad - k) / stride + 1;
    if (Hd <= 0 || Wd <= 0 || (long long)B * Hd * Wd >= (1ll << 31)) return false;
    const unsigned long long img = (unsigned long long)Hs * Ws * (unsigned long long)Cs * 4ull;
    const unsigned long long wb = (unsigned long long)N * k * k * Cs * (bf ? 2ull : 4ull);
    const unsigned long long imgs_per_tile = 256ull / (unsigned long long)(Hd * Wd) + 2;
    return img * imgs_per_tile < 0xfffffff0ull && wb < 0xfffffff0ull;
}
// stride-2 dgrad: the parity-class launches scatter into one window over the whole dx tensor
static bool dgrad_s2_window_ok(int B, int H, int W, int Cin, int Cout, long long lddx) {
    if ((H & 1) || (W & 1)) return false;
    return conv_window_ok(B, H / 2, W / 2, Cout, Cin, 3, 1, false) &&
           (unsigned long long)B * H * W * (unsigned long long)lddx * 4ull < 0xfffffff0ull;
}
// wgrad (H, W: the dy grid): a block's windows cover its K range plus the taps' reach
static bool wgrad_window_ok(int B, int H, int W, int Cin, int Cout, int k, int stride, bool bf) {
    const long long M = (long long)B * H * W;
    if (M >= (1ll << 31) - 65536) return false;            // (32-bit pixel counters in the kernel, with room for a K-step past M)
    if ((unsigned long long)Cout * k * k * Cin * 4ull >= 0xfffffff0ull) return false;
    int ntn, ntj, splits, sps;
    planes_wgrad_plan(B, H, W, Cin, Cout, k, &ntn, &ntj, &splits, &sps, planes_wgrad_tn(Cout, bf), bf);
    const int pad = (k - 1) / 2;
    const unsigned long long range_px = (unsigned long long)sps * 32ull + 2ull * (unsigned long long)(pad * W + pad) + 64ull;
    const unsigned long long range_x = stride == 1 ? range_px : 4ull * range_px + 8ull * (unsigned long long)W + 64ull;
    return range_px * (unsigned long long)Cout * 4ull < 0xfffffff0ull && range_x * (unsigned long long)Cin * 4ull < 0xfffffff0ull;
}
// every plane kernel a ConvBNAct layer with this geometry would launch (x: [B][H][W][Cin]); dgrad_planes: its dgrad too
bool planes_fit(int B, int H, int W, int Cin, int Cout, int k, int stride, bool bf, bool dgrad_planes) {
    if (B <= 0 || H <= 0 || W <= 0 || !planes_conv_ok(Cin, Cout, k, stride) || (Cout & 31)) return false;
    if (stride == 2 && (k != 3 || (H & 1) || (W & 1))) return false;
    if (!conv_window_ok(B, H, W, Cin, Cout, k, stride, bf)) return false;
    const int Ho = H / stride, Wo = W / stride;
    if (dgrad_planes) {
        if (stride == 1 ? !conv_window_ok(B, H, W, Cout, Cin, k, 1, bf) : !dgrad_s2_window_ok(B, H, W, Cin, Cout, Cin)) return false;
    }
    return wgrad_window_ok(B, Ho, Wo, Cin, Cout, k, stride, bf);
}

// forward-form conv over planes: src [B][Hs][Ws][Cs] planes, filter planes [N][k*k*Cs], raw fp32 result (+ res) and
// optional per-M-tile column sums (256-row tiles).  bf: plain bf16 operands (conv mode 2), amax words unused.
int planes_conv(const void* src, const unsigned* src_amax, const void* wt_planes, const unsigned* wt_amax, float* dst, long long ldd,
                const float* res, long long ldr, float* stats, int* nparts, int B, int Hs, int Ws, int Cs, int N, int k, int stride,
                hipStream_t st, bool bf, bool dst_bf16) {
    PlaneConvGeom g{};
    if (dst_bf16 && (res || (N & 31))) return Y4_ERR_SHAPE;
    g.dst_bf16 = dst_bf16 ? 1 : 0;
    const int pad = (k - 1) / 2;
    g.src = static_cast<const unsigned char*>(src); g.wt = static_cast<const unsigned char*>(wt_planes);
    g.dst = dst; g.ldd = ldd; g.res = res; g.ldr = ldr; g.stats = stats;
    g.B = B; g.Hs = Hs; g.Ws = Ws; g.Cs = Cs; g.N = N; g.k = k; g.stride = stride; g.pad = pad;
    g.Hd = (Hs + 2 * pad - k) / stride + 1;
    g.Wd = (Ws + 2 * pad - k) / stride + 1;
    if (!conv_window_ok(B, Hs, Ws, Cs, N, k, stride, bf)) return Y4_ERR_SHAPE;
    const long long M = (long long)B * g.Hd * g.Wd;
    if (bf && (Cs & 63)) return Y4_ERR_SHAPE;              // whole 64-channel K tiles
    g.M = (int)M; g.K = k * k * Cs;
    const unsigned long long img = (unsigned long long)Hs * Ws * (unsigned long long)Cs * 4ull;
    const unsigned long long wb = (unsigned long long)N * g.K * (bf ? 2ull : 4ull);
    g.src_total_bytes = (unsigned long long)B * img;
    g.wt_bytes = (unsigned)wb;
    g.src_amax = src_amax; g.wt_amax = wt_amax;
    // short K (1x1 layers up to 256 input channels): the two-blocks-per-CU shape (3-8 % faster there, measured; from K = 512
    // on the 256-row shape wins again).  Y4_PLANES_SMALL = 0 never, 2 always (experiments)
    static const int small_mode = getenv("Y4_PLANES_SMALL") ? atoi(getenv("Y4_PLANES_SMALL")) : 1;
    // ... and grids that quantise badly on 256 CUs with 256-row tiles (e.g. 364 blocks = 1.42 rounds at 19 x 19): in units of
    // the time of ONE 128-row block alone on a CU, a round of 256-row blocks costs 2; 128-row blocks sit two per CU, a full
    // round of 512 costs 2, a last round of <= 256 blocks (one per CU) costs 1 (the two shapes run long K at the same rate)
    bool quant = false;
    {
        const long long nt = (N + 127) / 128;
        const long long b256 = (long long)((g.M + 255) / 256) * nt, b128 = (long long)((g.M + 127) / 128) * nt;
        const long long c256 = ((b256 + 255) / 256) * 2;
        const long long tail = b128 % 512;
        const long long c128 = (b128 / 512) * 2 + (tail == 0 ? 0 : (tail <= 256 ? 1 : 2));
        quant = (double)c128 * 1.08 < (double)c256;
    }
    const bool small = small_mode == 2 || (small_mode == 1 && (g.K <= 256 || quant)) || (small_mode == 4 && g.K <= 256);   // 4: short K only
    if (bf && stream_bf16_ok(g, dst_bf16)) {
        g.K = Cs;
        return dst_bf16 ? dispatch_stream_bf16<true>(g, st, nparts) : dispatch_stream_bf16<false>(g, st, nparts);
    }
    if (bf) {
        // 256 x 256 (wave tile 128 x 64) unless its grid quantises badly on 256 CUs or N fits one 128-column tile
        const char* bt = getenv("Y4_BF_TILE");             // experiments / tests: 1: 256x128 always, 2: 256x256 always
        const int bf_tile = bt ? atoi(bt) : 0;
        const long long mt = (g.M + 255) / 256;
        const long long r256 = (mt * ((N + 255) / 256) + 255) / 256, r128 = (mt * ((N + 127) / 128) + 255) / 256;
        const bool big = bf_tile == 2 || (bf_tile == 0 && N > 128 && (double)r256 * 2.0 <= (double)r128 * 1.25);
        if (small && !big) {
            if (nparts) *nparts = (g.M + 127) / 128;
            return dst_bf16 ? launch_conv_planes<128, 128, 2, 2, true, true>(g, st) : launch_conv_planes<128, 128, 2, 2, true>(g, st);
        }
        if (nparts) *nparts = (g.M + 255) / 256;
        if (dst_bf16) return big ? launch_conv_planes<256, 256, 2, 4, true, true>(g, st) : launch_conv_planes<256, 128, 4, 2, true, true>(g, st);
        return big ? launch_conv_planes<256, 256, 2, 4, true>(g, st) : launch_conv_planes<256, 128, 4, 2, true>(g, st);
    }
    if (dst_bf16) return Y4_ERR_SHAPE;                     // (bf16 results: the bf16 operand kernels only)
    if (nparts) *nparts = small ? (g.M + 127) / 128 : (g.M + 255) / 256;
    return small ? launch_conv_planes<128, 128, 2, 2, false>(g, st) : launch_conv_planes<256, 128, 4, 2, false>(g, st);
}

// dgrad of a 3x3 stride-2 conv over dy planes [B][Ho][Wo][Cout] (H = 2 Ho, W = 2 Wo): four forward-form launches, one per parity
// class of the dx grid, each over its own 1 / 2 / 2 / 4 taps of the TRANSPOSED (un-mirrored) filter planes [Cin][9][Cout]
int planes_dgrad_s2(const void* dy, const unsigned* dy_amax, const void* wt_planes, const unsigned* wt_amax, float* dx, long long lddx,
                    int B, int H, int W, int Cin, int Cout, hipStream_t st, bool bf) {
    if ((H & 1) || (W & 1) || (Cout & 31) || (Cin & 3) || (bf && (Cout & 63))) return Y4_ERR_SHAPE;
    const int Ho = H / 2, Wo = W / 2;
    PlaneConvGeom g{};
    g.src = static_cast<const unsigned char*>(dy); g.wt = static_cast<const unsigned char*>(wt_planes);
    g.dst = dx; g.ldd = lddx;
    g.B = B; g.Hs = Ho; g.Ws = Wo; g.Cs = Cout; g.N = Cin; g.k = 3; g.stride = 1; g.pad = 0;
    g.Hd = Ho; g.Wd = Wo; g.Hfull = H; g.Wfull = W;
    if (B <= 0 || Ho <= 0 || Wo <= 0 || !dgrad_s2_window_ok(B, H, W, Cin, Cout, lddx)) return Y4_ERR_SHAPE;
    const long long M = (long long)B * Ho * Wo;
    g.M = (int)M; g.K = 9 * Cout;
    const unsigned long long img = (unsigned long long)Ho * Wo * (unsigned long long)Cout * 4ull;
    const unsigned long long wb = (unsigned long long)Cin * g.K * (bf ? 2ull : 4ull);
    g.src_total_bytes = (unsigned long long)B * img;
    g.wt_bytes = (unsigned)wb;
    g.src_amax = dy_amax; g.wt_amax = wt_amax;
    // heaviest class first (4 taps), so that the launch that ends the sequence is the short one
    for (int c = 3; c >= 0; --c) {
        const int ph = c >> 1, pw = c & 1;
        g.cls_ph = ph; g.cls_pw = pw; g.cls_ntaps = 0;
        for (int r = 0; r < 3; ++r)
            for (int q = 0; q < 3; ++q) {
                if (((ph + 1 - r) & 1) || ((pw + 1 - q) & 1)) continue;
                const int t = g.cls_ntaps++;
                g.cls_dh[t] = (ph + 1 - r) / 2; g.cls_dw[t] = (pw + 1 - q) / 2; g.cls_slot[t] = r * 3 + q;
            }
        const int rc = bf ? launch_conv_planes<256, 128, 4, 2, true, false, true>(g, st)
                          : launch_conv_planes<256, 128, 4, 2, false, false, true>(g, st);
        if (rc != Y4_OK) return rc;
    }
    return Y4_OK;
}

// split-K plan of the plane wgrad: tiles x splits blocks on 256 CUs (one block per CU), minimising rounds x K-steps per block
void planes_wgrad_plan(int B, int H, int W, int Cin, int Cout, int k, int* ntn, int* ntj, int* splits, int* sps, int tn, bool bf) {
    // blocks that run at once: one per CU -- two for the bf16 128-row form (72 KB of LDS, 121 registers: two blocks fit a CU), whose
    // small tiles are latency-bound per K step (a barrier and a DMA wait per 32 pixels): 64 -> 64 @304^2 1.45 -> see DESIGN section 5
    static const bool two = !(getenv("Y4_BF_WGRAD_2CU") && atoi(getenv("Y4_BF_WGRAD_2CU")) == 0);
    const int cap = (bf && tn == 128 && two) ? 512 : 256;
    const long long M = (long long)B * H * W;
    const int J = k * k * Cin;
    *ntn = (Cout + tn - 1) / tn;
    *ntj = (J + 255) / 256;
    const int tiles = *ntn * *ntj;
    const long long steps = (M + 31) / 32;
    long long best = -1; int bs = 1;
    for (int s = 1; s <= cap; ++s) {
        const long long per = (steps + s - 1) / s;
        if (s > 1 && per < 12) break;
        const long long rounds = ((long long)tiles * s + cap - 1) / cap;
        const long long cost = rounds * (per + 6);         // + prologue / epilogue of a block, in K-steps
        if (best < 0 || cost < best) { best = cost; bs = s; }
    }
    const long long per = (steps + bs - 1) / bs;
    *sps = (int)per;
    *splits = (int)((steps + per - 1) / per);
}

// bf16 form: 128-row n tiles, two blocks per CU (the 256-row form, wave tile 128 x 64, on request)
int planes_wgrad_tn(int Cout, bool bf) {
    if (!bf) return 128;
    const char* bt = getenv("Y4_BF_WGRAD_TILE");           // experiments / tests: 128 / 256 forced
    const int bf_tile = bt ? atoi(bt) : 0;
    if (bf_tile == 128 || (bf_tile == 256 && Cout >= 256)) return bf_tile;
    // 128-row tiles everywhere: two of those blocks share a CU (planes_wgrad_plan), which beats one block with the 128 x 64 wave
    // tile -- 758.4 / 756.5 -> 766.3 / 765.4 img/s at bs = 128 (the 256-row form stays for experiments: Y4_BF_WGRAD_TILE=256)
    return 128;
}

template <int TN_, bool BF, int S = 1>
static int launch_wgrad_planes(const PlaneWgradGeom& g, hipStream_t st) {
    constexpr int CH = BF ? 64 : 32;
    constexpr size_t smem = 3ull * (TN_ / CH + 256 / CH) * 32 * PROW;
    static_assert(smem <= 160 * 1024, "LDS");
    auto kern = wgrad_planes_mfma<TN_, 256, BF, S>;
    static Y4DynLds lds_attr;                              // per device, see common.h
    if (!lds_attr.ensure(reinterpret_cast<const void*>(kern), smem)) return Y4_ERR_LAUNCH;
    y4::note_kernel(BF ? "wgrad_planes_mfma<%d, 256, true, %d>" : "wgrad_planes_mfma<%d, 256, false, %d>", TN_, S);
    hipLaunchKernelGGL(kern, dim3(g.ntn * g.ntj * g.splits), dim3(512), smem, st, g);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

// H, W: the INPUT (x) grid; stride 2 (k = 3, H and W even): dy lives on H/2 x W/2
int planes_wgrad(const void* x, const unsigned* x_amax, const void* dy, const unsigned* dy_amax, float* dw, void* workspace,
                 size_t workspace_bytes, int B, int H, int W, int Cin, int Cout, int k, hipStream_t st, bool bf, int stride) {
    PlaneWgradGeom g{};
    g.x = static_cast<const unsigned char*>(x); g.dy = static_cast<const unsigned char*>(dy);
    if (stride == 2) {
        if (k != 3 || (H & 1) || (W & 1)) return Y4_ERR_SHAPE;
        H /= 2; W /= 2;                                    // from here on: the dy grid
    } else if (stride != 1) return Y4_ERR_SHAPE;
    g.B = B; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.k = k; g.pad = (k - 1) / 2;
    const long long M = (long long)B * H * W;
    const int q = bf ? 64 : 32;
    g.ldy_ch = (Cout + q - 1) / q * q;                     // dy rows hold whole K tiles; channels >= Cout are zero (caller)
    if (bf && (Cin & 63)) return Y4_ERR_SHAPE;
    if (!wgrad_window_ok(B, H, W, Cin, g.ldy_ch, k, stride, bf)) return Y4_ERR_SHAPE;
    g.M = (int)M; g.J = k * k * Cin;
    const int tn = planes_wgrad_tn(Cout, bf);
    planes_wgrad_plan(B, H, W, Cin, Cout, k, &g.ntn, &g.ntj, &g.splits, &g.steps_per_split, tn, bf);
    g.x_total_bytes = (unsigned long long)M * stride * stride * Cin * 4ull;
    g.dy_total_bytes = (unsigned long long)M * g.ldy_ch * 4ull;
    g.x_amax = x_amax; g.dy_amax = dy_amax;
    const size_t slab = (size_t)Cout * g.J * sizeof(float);
    if (g.splits > 1) {
        if (!workspace || workspace_bytes < slab * g.splits) return Y4_ERR_WORKSPACE;
        g.out = static_cast<float*>(workspace);
    } else {
        g.out = dw;
    }
    int rc;
    if (!bf) rc = stride == 2 ? launch_wgrad_planes<128, false, 2>(g, st) : launch_wgrad_planes<128, false>(g, st);
    else if (stride == 2) rc = tn == 256 ? launch_wgrad_planes<256, true, 2>(g, st) : launch_wgrad_planes<128, true, 2>(g, st);
    else if (tn == 256) rc = launch_wgrad_planes<256, true>(g, st);
    else rc = launch_wgrad_planes<128, true>(g, st);
    if (rc != Y4_OK) return rc;
    if (g.splits > 1) return y4::slab_reduce(static_cast<const float*>(workspace), dw, (long long)Cout * g.J, g.splits, st);
    return Y4_OK;
}

int planes_split(const float* x, long long ld, long long M, int C, const unsigned* amax, void* planes, hipStream_t st, bool bf,
                 long long pitch_ch, int c_valid) {
    // c_valid (0: C): source channels >= c_valid are pad (whatever the rows hold there) and leave as zeros
    // pitch_ch: channels per pixel row of the destination (0: C, dense; > C: a channel slice of a wider pre-split tensor)
    const long long pitch = (pitch_ch > 0 ? pitch_ch : (long long)C) * 4;
    const long long total = M * (C / 4);
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    if (bf) {
        hipLaunchKernelGGL(planes_split_bf16_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, st, x, ld, M, C,
                           static_cast<unsigned char*>(planes), pitch, c_valid > 0 ? c_valid : C);
        Y4_CHECK_LAUNCH();
        return Y4_OK;
    }
    hipLaunchKernelGGL(planes_split_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, st, x, ld, M, C, amax,
                       static_cast<unsigned char*>(planes), pitch, c_valid > 0 ? c_valid : C);
    Y4_CHECK_LAUNCH();
    return Y4_OK;
}

}  // namespace y4

// ======================================================================================== C ABI
// conv mode 3: f16x2 planes (amax words required); conv mode 2, or mode 3 with y4_set_planes_bf16(1) ("hybrid": bf16 MFMA on
// the plane layers, fp32-grade f16x2 kernels on the others): bf16 "planes" (amax words ignored, may be NULL)
static int g_planes_bf16 = 0;
static inline bool pl_mode_ok() { const int m = y4_get_conv_mode(); return m == 3 || m == 2; }
static inline bool pl_bf() { const int m = y4_get_conv_mode(); return m == 2 || (m == 3 && g_planes_bf16); }

extern "C" {

int y4_set_planes_bf16(int on) { g_planes_bf16 = on ? 1 : 0; return Y4_OK; }
int y4_conv_planes_fit(int B, int H, int W, int Cin, int Cout, int k, int stride, int dgrad_planes) {
    if (!pl_mode_ok()) return 0;
    const bool bf = pl_bf();
    if (bf && ((Cin & 63) || (Cout & 63))) return 0;
    return y4::planes_fit(B, H, W, Cin, Cout, k, stride, bf, dgrad_planes != 0) ? 1 : 0;
}
int y4_get_planes_bf16(void) { return g_planes_bf16; }

int y4_planes_split_f32(const float* x, int ldx, long long M, int C, const unsigned* amax, void* planes, void* stream) {
    if (!x || !planes || (!amax && !pl_bf())) return Y4_ERR_NULL;
    if (M <= 0 || C <= 0 || (C & 31) || ldx < C || (ldx & 3)) return Y4_ERR_SHAPE;
    if (pl_bf() && (C & 63)) return Y4_ERR_SHAPE;
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(planes) & 15)) return Y4_ERR_SHAPE;
    return y4::planes_split(x, ldx, M, C, amax, planes, y4_stream(stream), pl_bf());
}

int y4_planes_split_into_f32(const float* x, int ldx, long long M, int C, const unsigned* amax, void* planes, int ld_planes, int c_valid,
                             void* stream) {
    if (!x || !planes || (!amax && !pl_bf())) return Y4_ERR_NULL;
    if (M <= 0 || C <= 0 || (C & 31) || ldx < C || (ldx & 3) || ld_planes < C || (ld_planes & 31) || c_valid < 0 || c_valid > C) return Y4_ERR_SHAPE;
    if (pl_bf() && (C & 63)) return Y4_ERR_SHAPE;
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(planes) & 127)) return Y4_ERR_SHAPE;
    return y4::planes_split(x, ldx, M, C, amax, planes, y4_stream(stream), pl_bf(), ld_planes, c_valid);
}

int y4_conv2d_fwd_planes_f32(const void* x_planes, const float* w, float* y, int ldy,
                             int B, int H, int W, int Cin, int Cout, int k, int stride,
                             float* partials, size_t partial_bytes, long long* nparts_host, const unsigned* x_amax,
                             void* workspace, size_t workspace_bytes, void* dgrad_filter, size_t dgrad_filter_bytes,
                             int y_bf16, const float* bias, void* stream) {
    if (!x_planes || !w || !y || !workspace) return Y4_ERR_NULL;
    if (bias && (y_bf16 || partials || (reinterpret_cast<uintptr_t>(bias) & 3))) return Y4_ERR_SHAPE;   // (a conv without BatchNorm: no statistics)
    if (!pl_mode_ok()) return Y4_ERR_SHAPE;
    const bool bf = pl_bf();
    if (!bf && !x_amax) return Y4_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || !y4::planes_conv_ok(Cin, Cout, k, stride) || ldy < Cout || (ldy & 3)) return Y4_ERR_SHAPE;
    if (bf && (Cin & 63)) return Y4_ERR_SHAPE;
    if (workspace_bytes < y4_conv2d_fwd_workspace(Cin, Cout, k)) return Y4_ERR_WORKSPACE;
    if ((reinterpret_cast<uintptr_t>(x_planes) & 15) || (reinterpret_cast<uintptr_t>(w) & 15) ||
        (reinterpret_cast<uintptr_t>(workspace) & 15) || (reinterpret_cast<uintptr_t>(y) & 15)) return Y4_ERR_SHAPE;
    const int pad = (k - 1) / 2;
    const long long M = (long long)B * ((H + 2 * pad - k) / stride + 1) * ((W + 2 * pad - k) / stride + 1);
    if (partials && partial_bytes < (size_t)((M + 127) / 128) * 2 * Cout * sizeof(float)) return Y4_ERR_WORKSPACE;   // 128-row tiles at most
    hipStream_t st = y4_stream(stream);
    unsigned* hdr = static_cast<unsigned*>(workspace);
    unsigned short* planes = reinterpret_cast<unsigned short*>(static_cast<char*>(workspace) + 64 + 4096);
    int rc;
    if (dgrad_filter) {
        // also the mirrored transposed planes for y4_conv2d_dgrad_planes_f32 (which then takes them with w == NULL): one launch
        if ((Cout & 31) || (bf && (Cout & 63))) return Y4_ERR_SHAPE;
        if (dgrad_filter_bytes < y4_conv2d_dgrad_workspace(Cin, Cout, k)) return Y4_ERR_WORKSPACE;
        if (reinterpret_cast<uintptr_t>(dgrad_filter) & 15) return Y4_ERR_SHAPE;
        unsigned* hdr_t = reinterpret_cast<unsigned*>(static_cast<char*>(dgrad_filter) + (size_t)Cin * k * k * Cout * 6);
        // stride 1: mirrored (the plane dgrad is the forward kernel); stride 2: as the register-staged dgrad wants them
        // (bf16 planes at stride 2: the un-mirrored transposed bf16 planes the bf16 parity-class dgrad takes)
        if (bf) rc = bf16_filter(w, planes, static_cast<unsigned short*>(dgrad_filter), Cout, Cin, k * k, stride == 1, st);
        else rc = y4::f16x2_filter_planes_dual(w, planes, hdr, hdr + 16, static_cast<unsigned short*>(dgrad_filter), hdr_t, Cout, Cin, k * k,
                                               Cout, stride == 1, st);
    } else {
        if (bf) rc = bf16_filter(w, planes, nullptr, Cout, Cin, k * k, false, st);
        else rc = y4::f16x2_filter_planes(w, planes, Cout, k * k * Cin, hdr, hdr + 16, st);
    }
    if (rc != Y4_OK) return rc;
    int np = 0;
    rc = y4::planes_conv(x_planes, x_amax, planes, hdr, y, ldy, bias, 0, partials, &np, B, H, W, Cin, Cout, k, stride, st, bf, y_bf16 != 0);
    if (nparts_host) *nparts_host = np;
    return rc;
}


// dgrad of a stride-1 conv over dy planes: the forward kernel on the mirrored, transposed filter (N = Cin, K = k*k*Cout)
int y4_conv2d_dgrad_planes_f32(const void* dy_planes, const float* w, float* dx, int lddx,
                               int B, int H, int W, int Cin, int Cout, int k, int stride,
                               void* workspace, size_t workspace_bytes, const unsigned* dy_amax,
                               const float* residual, int ldr, void* stream) {
    if (!dy_planes || !dx || !workspace) return Y4_ERR_NULL;     // w == NULL: workspace filled by the forward call
    if (!pl_mode_ok()) return Y4_ERR_SHAPE;
    const bool bf = pl_bf();
    if (!bf && !dy_amax) return Y4_ERR_NULL;
    if (stride == 2) {
        // 3x3 stride 2 on an even map: one launch per parity class on the un-mirrored transposed planes
        if ((bf && (Cout & 63)) || k != 3 || residual || B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || !y4::planes_conv_ok(Cout, Cin, k, 1) ||
            lddx < Cin || (lddx & 3)) return Y4_ERR_SHAPE;
        if (workspace_bytes < y4_conv2d_dgrad_workspace(Cin, Cout, k)) return Y4_ERR_WORKSPACE;
        if ((reinterpret_cast<uintptr_t>(dy_planes) & 15) || (reinterpret_cast<uintptr_t>(w) & 15) ||
            (reinterpret_cast<uintptr_t>(workspace) & 15) || (reinterpret_cast<uintptr_t>(dx) & 15)) return Y4_ERR_SHAPE;
        hipStream_t st2 = y4_stream(stream);
        unsigned* hdr2 = reinterpret_cast<unsigned*>(static_cast<char*>(workspace) + (size_t)Cin * 9 * Cout * 6);
        if (w) {
            const int rc = bf ? bf16_filter(w, nullptr, static_cast<unsigned short*>(workspace), Cout, Cin, 9, false, st2)
                              : y4::f16x2_filter_planes_transposed(w, static_cast<unsigned short*>(workspace), Cout, Cin, 9, Cout, hdr2, hdr2 + 16, st2, false);
            if (rc != Y4_OK) return rc;
        }
        return y4::planes_dgrad_s2(dy_planes, dy_amax, workspace, hdr2, dx, lddx, B, H, W, Cin, Cout, st2, bf);
    }
    if (stride != 1) return Y4_ERR_SHAPE;
    if (B <= 0 || H <= 0 || W <= 0 || !y4::planes_conv_ok(Cout, Cin, k, 1) || lddx < Cin || (lddx & 3)) return Y4_ERR_SHAPE;
    if (bf && (Cout & 63)) return Y4_ERR_SHAPE;
    if (residual && (ldr < Cin || (ldr & 3) || (reinterpret_cast<uintptr_t>(residual) & 15))) return Y4_ERR_SHAPE;
    if (workspace_bytes < y4_conv2d_dgrad_workspace(Cin, Cout, k)) return Y4_ERR_WORKSPACE;
    if ((reinterpret_cast<uintptr_t>(dy_planes) & 15) || (reinterpret_cast<uintptr_t>(w) & 15) ||
        (reinterpret_cast<uintptr_t>(workspace) & 15) || (reinterpret_cast<uintptr_t>(dx) & 15)) return Y4_ERR_SHAPE;
    hipStream_t st = y4_stream(stream);
    const long long total = (long long)Cin * k * k * Cout;
    unsigned* hdr = reinterpret_cast<unsigned*>(static_cast<char*>(workspace) + (size_t)total * 6);
    if (w) {
        const int rc = bf ? bf16_filter(w, nullptr, static_cast<unsigned short*>(workspace), Cout, Cin, k * k, true, st)
                          : y4::f16x2_filter_planes_transposed(w, static_cast<unsigned short*>(workspace), Cout, Cin, k * k, Cout, hdr, hdr + 16, st, true);
        if (rc != Y4_OK) return rc;
    }
    return y4::planes_conv(dy_planes, dy_amax, workspace, hdr, dx, lddx, residual, ldr, nullptr, nullptr, B, H, W, Cout, Cin, k, 1, st, bf);
}

size_t y4_conv2d_wgrad_planes_workspace(int B, int H, int W, int Cin, int Cout, int k, int stride) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || k <= 0 || (stride != 1 && stride != 2)) return 0;
    if (stride == 2) { H = (H + 1) / 2; W = (W + 1) / 2; }  // the K dimension runs over the dy grid
    int ntn, ntj, splits, sps;
    // the larger of the two modes' split counts (the mode may be switched between the size query and the call)
    size_t most = 0;
    for (int bf = 0; bf < 2; ++bf) {
        y4::planes_wgrad_plan(B, H, W, Cin, Cout, k, &ntn, &ntj, &splits, &sps, y4::planes_wgrad_tn(Cout, bf != 0), bf != 0);
        const size_t need = splits > 1 ? (size_t)splits * Cout * k * k * Cin * sizeof(float) : 0;
        if (need > most) most = need;
    }
    return most + 64;
}

int y4_conv2d_wgrad_planes_f32(const void* x_planes, const void* dy_planes, float* dw,
                               int B, int H, int W, int Cin, int Cout, int k, int stride,
                               void* workspace, size_t workspace_bytes, const unsigned* x_amax, const unsigned* dy_amax,
                               void* stream) {
    if (!x_planes || !dy_planes || !dw) return Y4_ERR_NULL;
    if (!pl_mode_ok()) return Y4_ERR_SHAPE;
    const bool bf = pl_bf();
    if (!bf && (!x_amax || !dy_amax)) return Y4_ERR_NULL;
    // (Cout need not be whole K tiles: dy's pixel rows then hold ceil(Cout / 32) tiles -- 64-channel units in the bf16 mode -- whose
    //  pad channels are zero; dW has Cout rows)
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || (Cin & 31) || Cout <= 0 || (k != 1 && k != 3)) return Y4_ERR_SHAPE;
    if ((reinterpret_cast<uintptr_t>(x_planes) & 15) || (reinterpret_cast<uintptr_t>(dy_planes) & 15) ||
        (reinterpret_cast<uintptr_t>(dw) & 15) || (reinterpret_cast<uintptr_t>(workspace) & 15)) return Y4_ERR_SHAPE;
    if (workspace_bytes < y4_conv2d_wgrad_planes_workspace(B, H, W, Cin, Cout, k, stride)) return Y4_ERR_WORKSPACE;
    return y4::planes_wgrad(x_planes, x_amax, dy_planes, dy_amax, dw, workspace, workspace_bytes, B, H, W, Cin, Cout, k, y4_stream(stream), bf, stride);
}

}  // extern "C"
