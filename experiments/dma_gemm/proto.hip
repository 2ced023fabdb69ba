// PROTOTYPE (not part of libyolov4_amd.so): what does the conv arithmetic reach when both operands arrive as bf16
// planes by LDS-DMA instead of being split in registers?  C[M][N] (fp32) = sum_k A[m][k] * B[n][k], both operands given
// as three exact bf16 planes [3][rows][K] (x = p1 + p2 + p3), six MFMAs per product as in conv_gather_bf16x3.
// Structure: one 8-wave block per CU, tile 256 x 256, BK = 16, three LDS stages of 48 KB, global_load_lds 16 B per lane
// into a lane-linear image whose 16-B chunks are XOR-swizzled on the SOURCE address, two K-tiles of DMA in flight across
// raw s_barriers with a counted vmcnt, 48 MFMAs per wave and K-tile, one barrier per K-tile.
// (First variant, 256 x 128 x 32 with two stages and vmcnt(0) + __syncthreads(): 88-131 TFLOP/s, i.e. no better than the
// register-staged product kernel -- one tile of DMA in flight does not cover the L2 latency.)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int BM = 256, BN = 256, BK = 16, NSTAGE = 3;
constexpr int ROWB = BK * 2;                                  // 32 B per row per plane
constexpr int A_BYTES = 3 * BM * ROWB, B_BYTES = 3 * BN * ROWB, STAGE = A_BYTES + B_BYTES;   // 72 KB
constexpr int NDMA = STAGE / (512 * 16);                      // 6 wave-instructions per wave and K-tile

__global__ void split_planes_kernel(const float* __restrict__ x, unsigned short* __restrict__ planes, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = x[i];
        const unsigned h1 = __float_as_uint(v);
        const float r1 = v - __uint_as_float(h1 & 0xffff0000u);
        const unsigned h2 = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(h2 & 0xffff0000u);
        planes[i] = (unsigned short)(h1 >> 16);
        planes[n + i] = (unsigned short)(h2 >> 16);
        planes[2 * n + i] = (unsigned short)(__float_as_uint(r2) >> 16);
    }
}

// one stage: [A plane 0..2][256 rows][32 B] | [B plane 0..2][256 rows][32 B]; 16-B chunk c of row r sits at slot c ^ ((r >> 3) & 1)
__global__ __launch_bounds__(512, 1) void gemm_planes_dma_kernel(const unsigned short* __restrict__ Ap, const unsigned short* __restrict__ Bp,
                                                                 float* __restrict__ C, int M, int N, int K, int mode) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;                  // 2 x 4 waves, 128 x 64 each
    const int ntn = N / BN;
    // XCD-aware order: blocks b, b+8, b+16, ... share an XCD (and its L2); give each XCD a contiguous run of tiles
    const int nblk = gridDim.x, xcd = blockIdx.x & 7, q = nblk >> 3, r = nblk & 7;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    const int mt = bid / ntn, nt = bid - mt * ntn;
    const long long m0 = (long long)mt * BM, n0 = (long long)nt * BN;
    const long long a_plane = (long long)M * K, b_plane = (long long)N * K;    // elements per plane
    const int KT = K / BK;

    // DMA slots of this lane: instruction i of wave w covers stage bytes [(w * NDMA + i) * 1024, +1024)
    const unsigned short* src[NDMA];
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
        const int byte = (wave * NDMA + i) * 1024 + lane * 16;
        const bool isA = byte < A_BYTES;
        const int b2 = isA ? byte : byte - A_BYTES;
        const int rows = isA ? BM : BN;
        const int pl = b2 / (rows * ROWB);
        const int rem = b2 - pl * rows * ROWB;
        const int row = rem / ROWB, slot = (rem % ROWB) / 16;
        const int chunk = slot ^ ((row >> 3) & 1);
        src[i] = isA ? Ap + pl * a_plane + (m0 + row) * K + chunk * 8
                     : Bp + pl * b_plane + (n0 + row) * K + chunk * 8;
    }
    auto issue = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < NDMA; ++i)
            __builtin_amdgcn_global_load_lds(src[i] + kt * BK,
                                             reinterpret_cast<__attribute__((address_space(3))) void*>(
                                                 reinterpret_cast<uintptr_t>(smem + buf * STAGE + (wave * NDMA + i) * 1024)),
                                             16, 0, 0);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int fr = lane & 31, fh = lane >> 5;
    // three stages, two K-tiles of DMA in flight: wait only for the OLDER one (counted vmcnt), raw barrier (no vmcnt(0)),
    // then refill the stage everybody has just finished reading, then compute
    issue(0, 0);
    if (KT > 1) issue(1, 1);
    for (int kt = 0; kt < KT; ++kt) {
        if (kt + 1 < KT) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < KT && mode != 1) issue(kt + 2, (kt + 2) % NSTAGE);
        const unsigned char* As = smem + ((mode == 2 ? 0 : kt) % NSTAGE) * STAGE;
        const unsigned char* Bs = As + A_BYTES;
        bf16x8 fa[4][3], fb[2][3];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wm * 128 + i * 32 + fr;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                fa[i][pl] = *reinterpret_cast<const bf16x8*>(As + (pl * BM + row) * ROWB + ((fh ^ ((row >> 3) & 1)) << 4));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = wn * 64 + j * 32 + fr;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                fb[j][pl] = *reinterpret_cast<const bf16x8*>(Bs + (pl * BN + row) * ROWB + ((fh ^ ((row >> 3) & 1)) << 4));
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x16 c = acc[i][j];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], c, 0, 0, 0);
                acc[i][j] = c;
            }
        __builtin_amdgcn_s_setprio(0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long long n = n0 + wn * 64 + j * 32 + fr;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const long long m = m0 + wm * 128 + i * 32 + 4 * fh + (e & 3) + 8 * (e >> 2);
                if (m < M) C[m * N + n] = acc[i][j][e];
            }
        }
}
// Variant B: the same arithmetic as two INDEPENDENT 4-wave blocks per CU (tile 128 x 256, BK = 16, two 36-KB stages each,
// one K-tile of DMA in flight, vmcnt(0) + barrier per K-tile): the two blocks' barrier phases interleave by themselves.
namespace vb {
constexpr int BM = 128, BN = 256, BK = 16, ROWB = 32;
constexpr int A_BYTES = 3 * BM * ROWB, B_BYTES = 3 * BN * ROWB, STAGE = A_BYTES + B_BYTES;   // 36 KB
constexpr int NDMA = STAGE / (256 * 16);                                                     // 9
__global__ __launch_bounds__(256, 2) void gemm_planes_dma2_kernel(const unsigned short* __restrict__ Ap, const unsigned short* __restrict__ Bp,
                                                                  float* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;                  // 2 x 2 waves, 64 x 128 each
    const int ntn = N / BN;
    const int nblk = gridDim.x, xcd = blockIdx.x & 7, q = nblk >> 3, r = nblk & 7;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    const int mt = bid / ntn, nt = bid - mt * ntn;
    const long long m0 = (long long)mt * BM, n0 = (long long)nt * BN;
    const long long a_plane = (long long)M * K, b_plane = (long long)N * K;
    const int KT = K / BK;
    const unsigned short* src[NDMA];
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
        const int byte = (wave * NDMA + i) * 1024 + lane * 16;
        const bool isA = byte < A_BYTES;
        const int b2 = isA ? byte : byte - A_BYTES;
        const int rows = isA ? BM : BN;
        const int pl = b2 / (rows * ROWB);
        const int rem = b2 - pl * rows * ROWB;
        const int row = rem / ROWB, slot = (rem % ROWB) / 16;
        const int chunk = slot ^ ((row >> 3) & 1);
        src[i] = isA ? Ap + pl * a_plane + (m0 + row) * K + chunk * 8 : Bp + pl * b_plane + (n0 + row) * K + chunk * 8;
    }
    auto issue = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < NDMA; ++i)
            __builtin_amdgcn_global_load_lds(src[i] + kt * BK,
                                             reinterpret_cast<__attribute__((address_space(3))) void*>(
                                                 reinterpret_cast<uintptr_t>(smem + buf * STAGE + (wave * NDMA + i) * 1024)),
                                             16, 0, 0);
    };
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int fr = lane & 31, fh = lane >> 5;
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < KT) issue(kt + 1, buf ^ 1);
        const unsigned char* As = smem + buf * STAGE;
        const unsigned char* Bs = As + A_BYTES;
        bf16x8 fa[2][3], fb[4][3];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wm * 64 + i * 32 + fr;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                fa[i][pl] = *reinterpret_cast<const bf16x8*>(As + (pl * BM + row) * ROWB + ((fh ^ ((row >> 3) & 1)) << 4));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = wn * 128 + j * 32 + fr;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                fb[j][pl] = *reinterpret_cast<const bf16x8*>(Bs + (pl * BN + row) * ROWB + ((fh ^ ((row >> 3) & 1)) << 4));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x16 c = acc[i][j];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], c, 0, 0, 0);
                acc[i][j] = c;
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long n = n0 + wn * 128 + j * 32 + fr;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const long long m = m0 + wm * 64 + i * 32 + 4 * fh + (e & 3) + 8 * (e >> 2);
                if (m < M) C[m * N + n] = acc[i][j][e];
            }
        }
}
}  // namespace vb

// Variant C: A stays fp32 in HBM and in LDS (4 B/element through the DMA instead of 6), every wave splits the A fragments it
// reads into the three bf16 pieces in registers, between its MFMAs; B (the small operand) arrives as pre-split planes.
namespace vc {
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
constexpr int BM = 256, BN = 256, BK = 16, NSTAGE = 4;
constexpr int A_BYTES = BM * BK * 4, B_BYTES = 3 * BN * BK * 2, STAGE = A_BYTES + B_BYTES;   // 16 + 24 = 40 KB
constexpr int NDMA = STAGE / (512 * 16);                                                     // 5
__device__ __forceinline__ unsigned pack_hi16(unsigned hi, unsigned lo) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); }
__device__ __forceinline__ void split8(const f32x4 v0, const f32x4 v1, bf16x8& p1, bf16x8& p2, bf16x8& p3) {
    unsigned h1[8], h2[8], h3[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = e < 4 ? v0[e] : v1[e - 4];
        h1[e] = __float_as_uint(x);
        const float r1 = x - __uint_as_float(h1[e] & 0xffff0000u);
        h2[e] = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(h2[e] & 0xffff0000u);
        h3[e] = __float_as_uint(r2);
    }
    const u32x4 q1 = {pack_hi16(h1[1], h1[0]), pack_hi16(h1[3], h1[2]), pack_hi16(h1[5], h1[4]), pack_hi16(h1[7], h1[6])};
    const u32x4 q2 = {pack_hi16(h2[1], h2[0]), pack_hi16(h2[3], h2[2]), pack_hi16(h2[5], h2[4]), pack_hi16(h2[7], h2[6])};
    const u32x4 q3 = {pack_hi16(h3[1], h3[0]), pack_hi16(h3[3], h3[2]), pack_hi16(h3[5], h3[4]), pack_hi16(h3[7], h3[6])};
    p1 = __builtin_bit_cast(bf16x8, q1); p2 = __builtin_bit_cast(bf16x8, q2); p3 = __builtin_bit_cast(bf16x8, q3);
}
__global__ __launch_bounds__(512, 1) void gemm_f32a_dma_kernel(const float* __restrict__ A, const unsigned short* __restrict__ Bp,
                                                               float* __restrict__ C, int M, int N, int K, int mode) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;                  // 2 x 4 waves, 128 x 64 each
    const int ntn = N / BN;
    const int nblk = gridDim.x, xcd = blockIdx.x & 7, q = nblk >> 3, r = nblk & 7;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    const int mt = bid / ntn, nt = bid - mt * ntn;
    const long long m0 = (long long)mt * BM, n0 = (long long)nt * BN;
    const long long b_plane = (long long)N * K;
    const int KT = K / BK;
    const unsigned char* src[NDMA];
    int kstep[NDMA];
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
        const int byte = (wave * NDMA + i) * 1024 + lane * 16;
        if (byte < A_BYTES) {                                  // fp32 rows of 64 B, 4 chunks, slot = chunk ^ ((row >> 2) & 3)
            const int row = byte / 64, slot = (byte % 64) / 16;
            const int chunk = slot ^ ((row >> 2) & 3);
            src[i] = reinterpret_cast<const unsigned char*>(A + (m0 + row) * K + chunk * 4);
            kstep[i] = BK * 4;
        } else {                                               // bf16 plane rows of 32 B, 2 chunks, slot = chunk ^ ((row >> 3) & 1)
            const int b2 = byte - A_BYTES;
            const int pl = b2 / (BN * 32);
            const int rem = b2 - pl * BN * 32;
            const int row = rem / 32, slot = (rem % 32) / 16;
            const int chunk = slot ^ ((row >> 3) & 1);
            src[i] = reinterpret_cast<const unsigned char*>(Bp + pl * b_plane + (n0 + row) * K + chunk * 8);
            kstep[i] = BK * 2;
        }
    }
    auto issue = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < NDMA; ++i)
            __builtin_amdgcn_global_load_lds(src[i] + (long long)kt * kstep[i],
                                             reinterpret_cast<__attribute__((address_space(3))) void*>(
                                                 reinterpret_cast<uintptr_t>(smem + buf * STAGE + (wave * NDMA + i) * 1024)),
                                             16, 0, 0);
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int fr = lane & 31, fh = lane >> 5;
    issue(0, 0);
    if (KT > 1) issue(1, 1);
    if (KT > 2) issue(2, 2);
    for (int kt = 0; kt < KT; ++kt) {
        if (kt + 2 < KT) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");      // two younger tiles may still be in flight
        else if (kt + 1 < KT) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 3 < KT && mode != 1) issue(kt + 3, (kt + 3) % NSTAGE);
        const unsigned char* As = smem + (kt % NSTAGE) * STAGE;
        const unsigned char* Bs = As + A_BYTES;
        f32x4 ra[4][2];
        bf16x8 fb[2][3];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wm * 128 + i * 32 + fr;
            const int s = (row >> 2) & 3;
            ra[i][0] = *reinterpret_cast<const f32x4*>(As + row * 64 + (((2 * fh) ^ s) << 4));
            ra[i][1] = *reinterpret_cast<const f32x4*>(As + row * 64 + (((2 * fh + 1) ^ s) << 4));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = wn * 64 + j * 32 + fr;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                fb[j][pl] = *reinterpret_cast<const bf16x8*>(Bs + (pl * BN + row) * 32 + ((fh ^ ((row >> 3) & 1)) << 4));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bf16x8 f1, f2, f3;
            split8(ra[i][0], ra[i][1], f1, f2, f3);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x16 c = acc[i][j];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f3, fb[j][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f2, fb[j][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, fb[j][2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f2, fb[j][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, fb[j][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, fb[j][0], c, 0, 0, 0);
                acc[i][j] = c;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long long n = n0 + wn * 64 + j * 32 + fr;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const long long m = m0 + wm * 128 + i * 32 + 4 * fh + (e & 3) + 8 * (e >> 2);
                if (m < M) C[m * N + n] = acc[i][j][e];
            }
        }
}
}  // namespace vc

// Sustained matrix-pipe rate with nothing else going on: every wave issues 32x32x16 bf16 MFMAs on register operands
// (random data, 4 independent accumulators), 8 waves per CU.  The ceiling the power envelope leaves.
__global__ __launch_bounds__(512, 1) void mfma_peak_kernel(const unsigned* __restrict__ seed, float* __restrict__ out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    u32x4 a = {seed[tid & 1023], seed[(tid + 1) & 1023], seed[(tid + 2) & 1023], seed[(tid + 3) & 1023]};
    u32x4 b = {seed[(tid + 4) & 1023], seed[(tid + 5) & 1023], seed[(tid + 6) & 1023], seed[(tid + 7) & 1023]};
    const bf16x8 fa = __builtin_bit_cast(bf16x8, a), fb = __builtin_bit_cast(bf16x8, b);
    f32x16 c0, c1, c2, c3;
#pragma unroll
    for (int e = 0; e < 16; ++e) { c0[e] = 0.f; c1[e] = 0.f; c2[e] = 0.f; c3[e] = 0.f; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, fa, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fa, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, fb, c3, 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
    out[tid] = s;
}
}  // namespace

extern "C" {
int proto_gemm2(const unsigned short* Ap, const unsigned short* Bp, float* C, int M, int N, int K, void* stream) {
    if (M % vb::BM || N % vb::BN || K % vb::BK) return 2;
    static bool done = false;
    if (!done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(vb::gemm_planes_dma2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                2 * vb::STAGE) != hipSuccess) return 3;
        done = true;
    }
    hipLaunchKernelGGL(vb::gemm_planes_dma2_kernel, dim3((M / vb::BM) * (N / vb::BN)), dim3(256), 2 * vb::STAGE, (hipStream_t)stream, Ap, Bp, C, M, N, K);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
int proto_gemm3(const float* A, const unsigned short* Bp, float* C, int M, int N, int K, void* stream) {
    if (M % vc::BM || N % vc::BN || K % vc::BK) return 2;
    const char* e = getenv("PROTO_MODE");
    const int mode = e ? atoi(e) : 0;
    static bool done = false;
    if (!done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(vc::gemm_f32a_dma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                vc::NSTAGE * vc::STAGE) != hipSuccess) return 3;
        done = true;
    }
    hipLaunchKernelGGL(vc::gemm_f32a_dma_kernel, dim3((M / vc::BM) * (N / vc::BN)), dim3(512), vc::NSTAGE * vc::STAGE, (hipStream_t)stream,
                       A, Bp, C, M, N, K, mode);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
int proto_mfma_peak(const unsigned* seed, float* out, int iters, void* stream) {
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(256), dim3(512), 0, (hipStream_t)stream, seed, out, iters);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
int proto_split(const float* x, unsigned short* planes, long long n, void* stream) {
    hipLaunchKernelGGL(split_planes_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, x, planes, n);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
// requires M % 256 == 0, N % 256 == 0, K % 16 == 0 (prototype: no tails)
int proto_gemm(const unsigned short* Ap, const unsigned short* Bp, float* C, int M, int N, int K, void* stream) {
    const char* e = getenv("PROTO_MODE");
    const int mode = e ? atoi(e) : 0;
    if (M % BM || N % BN || K % BK) return 2;
    static bool done = false;
    if (!done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_planes_dma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                NSTAGE * STAGE) != hipSuccess) return 3;
        done = true;
    }
    hipLaunchKernelGGL(gemm_planes_dma_kernel, dim3((M / BM) * (N / BN)), dim3(512), NSTAGE * STAGE, (hipStream_t)stream, Ap, Bp, C, M, N, K, mode);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
}
