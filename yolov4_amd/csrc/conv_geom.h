// Problem descriptors shared by the implicit-GEMM convolution kernels (conv_igemm.hip: fp32-MFMA, bf16x3, bf16;
// conv_f16x2.hip: the two-piece fp16 split).
#pragma once
#include "common.h"

namespace y4 {

struct ConvGeom {
    // gathered ("source") tensor and produced ("dest") tensor, both NHWC with pitch
    const float* src; const float* wt; float* dst;
    const float* scale; const float* shift; const float* res;
    float* stats;                 // forward only, optional: per-M-tile column sums [mtiles][2][N] of the raw output
    long long lds_, ldd, ldr;     // pixel pitches (elements)
    int B, Hs, Ws, Cs;            // source dims (Cs = GEMM-K channels, multiple of 32)
    int Cs_valid;                 // channels >= Cs_valid of the source are treated as zero (pad lanes)
    int Hd, Wd, N;                // dest spatial dims, N = dest channels (any)
    int k, stride, pad;
    int M;                        // B*Hd*Wd
    int K;                        // k*k*Cs
    int act;
    int mtiles, ntiles;
    // stride-2 dgrad: dest pixels are tiled per parity class (h&1, w&1) so that every row of a
    // tile sees the SAME set of contributing filter taps (1, 2, 2 or 4 of the 9) and the K loop
    // visits only those -- no zero-filled MFMAs.  cls_tile0[c] = first M-tile of class c.
    int cls_tile0[5];
    int cls_h[2], cls_w[2];       // class extents: (Hd + 1 - ph) / 2, (Wd + 1 - pw) / 2
    int cls_slot0[5];             // classed launch: per-XCD slot ranges (each XCD gets 1/8 of EVERY class)
    unsigned long long src_total_bytes;   // whole source tensor; each block re-bases its 32-bit buffer window at its first image
    unsigned wt_bytes;            // filter extent for the buffer descriptor (< 4 GiB, checked on the host)
    const unsigned short* wt_planes;   // split modes: filter pre-split into 16-bit planes (the call's workspace)
    // f16x2 mode: device words holding the bit pattern of max|finite element| of the gathered tensor / the filter
    // (NULL: operand taken unscaled); every element is multiplied by the power of two that brings that maximum
    // into [2^14, 2^15) before it is split into fp16 pieces, and the epilogue undoes both scales
    const unsigned* src_amax; const unsigned* wt_amax;
    unsigned* dst_amax;           // f16x2 forward, optional: max|finite output| folded in with atomicMax
};

struct WgradGeom {
    const float* x; const float* dy; float* out;    // out: dw (splits==1) or slab base
    long long ldx, lddy;
    int B, H, W, Cin, Ho, Wo, Cout;
    int k, stride, pad;
    int M;          // B*Ho*Wo
    int J;          // k*k*Cin
    int ntn, ntj, splits, chunks_per_split;   // chunks of 32 pixels
    int tn, tj;     // tile edges chosen by the planner (64 or 128)
    const unsigned* x_amax; const unsigned* dy_amax;   // f16x2 mode, as ConvGeom::src_amax
    unsigned long long x_total_bytes, dy_total_bytes;   // whole tensors; blocks re-base their 32-bit windows
};

// conv_f16x2.hip
int f16x2_gather(const ConvGeom& g, bool transposed, hipStream_t st, int* nparts);
int f16x2_wgrad(const WgradGeom& g, hipStream_t st);
// prepared filter (inference): [64-B header][4 KiB of fingerprint partials][planes]; re-split only when the bits changed
int f16x2_refresh_prepared(const float* w, void* prepared, int Cout, int K, hipStream_t st);
// filter maximum + split in two launches without atomics or pre-zeroed words: per-block maxima into part[<= 1024], folded
// by every block of the split kernel (block 0 leaves the result in *amax_out for the conv kernel's epilogue)
int f16x2_filter_planes(const float* w, unsigned short* planes, int Cout, int K, unsigned* amax_out, unsigned* part, hipStream_t st);
int f16x2_filter_planes_dual(const float* w, unsigned short* planes, unsigned* amax_out, unsigned* part, unsigned short* planes_t,
                             unsigned* amax_out_t, int Cout, int Cin, int kk, int Cout_pad, bool mirror, hipStream_t st);
int f16x2_filter_planes_transposed(const float* w, unsigned short* planes, int Cout, int Cin, int kk, int Cout_pad,
                                   unsigned* amax_out, unsigned* part, hipStream_t st, bool mirror = false);
// conv_igemm.hip: dw[n] = sum over `splits` fp32 slabs of n elements each, fixed order (deterministic)
int slab_reduce(const float* slabs, float* dw, long long n, int splits, hipStream_t st);
int amax_launch(const float* x, long long ld, long long M, int C, unsigned* amax_bits, hipStream_t st,
                bool prezeroed = false);     // zeroes the word first unless the caller hands in a zeroed one
int amax_merge(unsigned* dst, const unsigned* src, hipStream_t st);


// conv_tile.hip: 3x3 stride-1 layers with few channels on large maps (2-D tiles, every input element staged once);
// forward form only: dgrad hands in the mirrored transposed filter planes
bool tile_conv_ok(int Cs, int Cs_valid, int N, int k, int stride, int H, int W);
int f16x2_tile(const ConvGeom& g, hipStream_t st, int* nparts);
bool tile_dgrad_s2_ok(int Cs, int Cs_valid, int N, int k, int stride, int Hs, int Ws);
int f16x2_tile_dgrad_s2(const ConvGeom& g, hipStream_t st);      // dgrad of a stride-2 3x3 conv, all four parity classes per tile

// wgrad_tile.hip: filter gradient of the 3x3 layers with 32 / 64 input channels and 64 output channels on large maps
// (x patch and dy tile staged once per tile of outputs, all nine taps out of LDS, the whole dW slice in one block's accumulators)
bool tile_wgrad_ok(int Cin, int Cout, int k, int stride, int H, int W, long long ldx, long long lddy);
int tile_wgrad_slabs(int Cin, int Cout);
int f16x2_wgrad_tile(const WgradGeom& g, hipStream_t st);     // g.out: g.splits (= tile_wgrad_slabs) slabs of [Cout][9 Cin]

// conv_planes.hip: DMA-fed kernels over pre-split operands
bool planes_conv_ok(int Cin, int Cout, int k, int stride);
int planes_conv(const void* src, const unsigned* src_amax, const void* wt_planes, const unsigned* wt_amax, float* dst, long long ldd,
                const float* res, long long ldr, float* stats, int* nparts, int B, int Hs, int Ws, int Cs, int N, int k, int stride,
                hipStream_t st, bool bf = false, bool dst_bf16 = false);
int planes_dgrad_s2(const void* dy, const unsigned* dy_amax, const void* wt_planes, const unsigned* wt_amax, float* dx, long long lddx,
                    int B, int H, int W, int Cin, int Cout, hipStream_t st, bool bf = false);
int planes_split(const float* x, long long ld, long long M, int C, const unsigned* amax, void* planes, hipStream_t st, bool bf = false,
                 long long pitch_ch = 0, int c_valid = 0);
void planes_wgrad_plan(int B, int H, int W, int Cin, int Cout, int k, int* ntn, int* ntj, int* splits, int* sps, int tn = 128, bool bf = false);
int planes_wgrad_tn(int Cout, bool bf);
int planes_wgrad(const void* x, const unsigned* x_amax, const void* dy, const unsigned* dy_amax, float* dw, void* workspace,
                 size_t workspace_bytes, int B, int H, int W, int Cin, int Cout, int k, hipStream_t st, bool bf = false,
                 int stride = 1);

// Name of the conv kernel launched last on this host thread, spelled as rocprofv3 prints the symbol (bench.py names the
// dominant kernel with it instead of restating the dispatch rules).
void note_kernel(const char* fmt, ...);
}  // namespace y4
