// Training-step kernels with arithmetic in them: weight-gradient GEMM (reduction over rows), the per-edge
// backward of the score predictor tail and of the gated aggregation, and two tiny helpers for the encoders.
// The backward of the path is restated in gnnome_amd/train.py (autograd of models/full_graph.py:22-30 as
// driven by train.py:138-145, 328-330); each kernel's contract is in include/gnnome_hip.h.
#include <atomic>
#include <algorithm>
#include "gemm_tile.h"

namespace gnnome {

// ---------------------------------------------------------------------------------------------------
// wgrad:  C[Ka,Kb] = A[R,Ka]^T * B[R,Kb]   (nn.Linear weight gradient dW = dY^T X; also dW of B_3, W1, W2)
// fp32-faithful bf16x6 product (gemm_tile.h) with the ROW index as the MFMA k dimension.  A workgroup owns one
// 128x128 output tile and one contiguous chunk of rows and writes its partial tile; a second kernel adds the
// partials in chunk order, so the result does not depend on scheduling (no float atomics).
// ---------------------------------------------------------------------------------------------------
constexpr int kWgTile = 128, kWgRows = 32, kWgBlocks = 8;
constexpr int kWgColBytes = 2 * kWgRows + 16;        // one column of one bf16 plane: 32 rows + 16 bytes of pad
constexpr int kWgPlane = kWgTile * kWgColBytes;      // 10 KB; six planes (A and B, three each) = 60 KB: two workgroups per CU

// One workgroup = one 128x128 output tile (a whole [H,H] weight at H = 128) x one chunk of rows, so every row of
// A and B is read from HBM exactly once; wave (wi, wj) owns a 64x64 quadrant as 2x2 accumulators.
//
// The MFMA k dimension is the ROW index, so a lane's operand is eight consecutive rows of ONE column.  The threads that
// stage a 32-row slab split it (each element once, not once per wave that uses it) and store it TRANSPOSED as three bf16
// planes [column slot][row]: a fragment is then one ds_read_b128 per plane.  A staging thread holds four columns
// 4 c4 .. 4 c4 + 3 of four consecutive rows (a 16-byte global load per row) and writes 8 bytes per column and plane;
// column 4 c4 + j lives in slot 32 j + c4, which spreads the 32 lanes of a write over the banks (20 c4 mod 64) and makes
// MFMA row/column index m of fragment f the output index 4 m + f.  The next slab's global loads are issued before the
// MFMAs of the current one.
//
// A may be given as up to kWgBlocks column blocks of equal width living in separate buffers (the five [N,H] gradients of a
// layer's node projections, never concatenated): column c is column c % width of block c / width.  With colsum_part set, the
// workgroups of the first tile column also leave the column sums of their rows of A (the bias gradients) - the slab is in
// their registers anyway.
struct WgradA {
    const float* blk[kWgBlocks];
    int width;   // >= Ka for a single buffer
};

template <bool X16, int ABL = 0>   // X16: A is stored as bf16 (the dxe rows of the bf16-storage training step); ABL: measurement only
__global__ __launch_bounds__(256, 2) void k_wgrad_partial(WgradA a_op, int lda, int Ka, const float* __restrict__ B,
                                                          int ldb, int Kb, int64_t R, int64_t rows_per_chunk,
                                                          float* __restrict__ partial, float* __restrict__ colsum_part) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[6 * kWgPlane];
    unsigned char* Ap = lds;
    unsigned char* Bp = lds + 3 * kWgPlane;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i0 = blockIdx.x * kWgTile, j0 = blockIdx.y * kWgTile;
    const int64_t r_begin = (int64_t)blockIdx.z * rows_per_chunk, r_end = min(R, r_begin + rows_per_chunk);
    const int wi = wave & 1, wj = wave >> 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int c4 = tid & 31, rr = tid >> 5;  // 32 float4 per 128-wide row; rows 4 rr .. 4 rr + 3 of the slab
    const bool a_in = i0 + 4 * c4 < Ka, b_in = j0 + 4 * c4 < Kb;   // rows and columns outside the operands contribute zeros
    const float* a_col = a_op.blk[0];
    {
        const int ca = a_in ? i0 + 4 * c4 : 0, which = ca / a_op.width;
#pragma unroll
        for (int k = 1; k < kWgBlocks; ++k)
            if (which == k) a_col = a_op.blk[k];   // a select chain: no dynamic indexing of the kernel arguments
        a_col += ca - which * a_op.width;
    }
    const void* a_base16 = a_op.blk[0];                       // X16: one buffer of bf16, element offsets
    const int64_t a_off16 = a_in ? i0 + 4 * c4 : 0;
    const float* b_col = B + j0 + 4 * c4;
    const bool sums = colsum_part != nullptr && blockIdx.y == 0;
    f32x4 cs = {0.f, 0.f, 0.f, 0.f};
    auto fetch = [&](f32x4 (&av)[4], f32x4 (&bv)[4], int64_t r0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int64_t row = r0 + 4 * rr + t;
            av[t] = bv[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < r_end) {
                if (a_in) av[t] = X16 ? load4_as<true>(a_base16, a_off16 + row * lda) : *reinterpret_cast<const f32x4*>(a_col + row * lda);
                if (b_in) bv[t] = *reinterpret_cast<const f32x4*>(b_col + row * ldb);
            }
        }
    };
    auto stage = [&](unsigned char* planes, const f32x4 (&v)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint2 p1, p2, p3;
            tile_split4(f32x4{v[0][j], v[1][j], v[2][j], v[3][j]}, p1, p2, p3);
            unsigned char* d = planes + (32 * j + c4) * kWgColBytes + 8 * rr;
            *reinterpret_cast<uint2*>(d) = p1;
            *reinterpret_cast<uint2*>(d + kWgPlane) = p2;
            *reinterpret_cast<uint2*>(d + 2 * kWgPlane) = p3;
        }
    };
    auto bf = [](const uint4 v) { return __builtin_bit_cast(tile_bf16x8, v); };
    // fragment f of wave half w: slots 64 w + 32 f + (lane & 31), rows 16 s + 8 (lane >> 5) .. + 7
    const unsigned char* ap = Ap + (64 * wi + (lane & 31)) * kWgColBytes + 16 * (lane >> 5);
    const unsigned char* bp = Bp + (64 * wj + (lane & 31)) * kWgColBytes + 16 * (lane >> 5);
    auto products = [&]() {
#pragma unroll
        for (int s = 0; s < ((ABL & 2) ? 0 : kWgRows / 16); ++s) {
            uint4 a1[2], a2[2], a3[2], b1[2], b2[2], b3[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned char* pa = ap + 32 * h * kWgColBytes + 32 * s;
                const unsigned char* pb = bp + 32 * h * kWgColBytes + 32 * s;
                a1[h] = *reinterpret_cast<const uint4*>(pa);
                a2[h] = *reinterpret_cast<const uint4*>(pa + kWgPlane);
                a3[h] = *reinterpret_cast<const uint4*>(pa + 2 * kWgPlane);
                b1[h] = *reinterpret_cast<const uint4*>(pb);
                b2[h] = *reinterpret_cast<const uint4*>(pb + kWgPlane);
                b3[h] = *reinterpret_cast<const uint4*>(pb + 2 * kWgPlane);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    f32x16 c = acc[a][b];   // smallest terms first
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a3[a]), bf(b1[b]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1[a]), bf(b3[b]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a2[a]), bf(b2[b]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a2[a]), bf(b1[b]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1[a]), bf(b2[b]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1[a]), bf(b1[b]), c, 0, 0, 0);
                    acc[a][b] = c;
                }
        }
    };
    // (Measured and dropped, round 5: a SECOND slab in flight per workgroup, two register sets taking turns - 0.27-0.32 against 0.25-0.28 ms; the
    // kernel is balanced between matrix work and memory: 0.18 ms of MFMAs + fragment reads alone, 0.204 without re-fetching, 0.214 without MFMAs.)
    f32x4 av[4], bv[4];
    fetch(av, bv, r_begin);
    for (int64_t r0 = r_begin; r0 < r_end; r0 += kWgRows) {
        if (!(ABL & 4) || r0 == r_begin) {
            stage(Ap, av);
            stage(Bp, bv);
        }
        if (sums) cs += (av[0] + av[1]) + (av[2] + av[3]);
        __syncthreads();
        if (r0 + kWgRows < r_end && !(ABL & 1)) fetch(av, bv, r0 + kWgRows);
        products();
        __syncthreads();
    }
    float* out = partial + ((int64_t)blockIdx.z * Ka + i0) * Kb + j0;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = 4 * cd_row(r, lane) + 2 * wi + a, j = 4 * (lane & 31) + 2 * wj + b;
                if (i0 + i < Ka && j0 + j < Kb) out[(int64_t)i * Kb + j] = acc[a][b][r];
            }
    if (sums) {   // the eight row groups of a column, added in a fixed order (the planes are no longer needed: the loop ended on a barrier)
        float* red = reinterpret_cast<float*>(lds);
        *reinterpret_cast<f32x4*>(red + rr * kWgTile + 4 * c4) = cs;
        __syncthreads();
        if (tid < kWgTile && i0 + tid < Ka) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) t += red[k * kWgTile + tid];
            colsum_part[(int64_t)blockIdx.z * Ka + i0 + tid] = t;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// The same product as fp16x3 (round 5; VERDICT r4 "missing" item 5): two fp16 planes per operand, x1 = RN16(x), x2 = RN16((x - x1) 2048),
// three MFMAs per k step instead of six (x1 y1 in one accumulator, x1 y2 + x2 y1 in a second one that is folded in with 2^-11 at the
// end), two planes to split and stage instead of three.  fp16's range needs a SCALE for A, a gradient: A is multiplied by
// 2^k with k = 13 - floor(log2(max |A|)) while it is split (exact), the result by 2^-k - max |A| comes from the kernel that produced A
// (`amax_bits`: the bits of a non-negative float, the producers' atomicMax over unsigned values; NULL: no scale, for operands of
// ordinary size).  An element's error is then <= max(2^-22 |a|, 2^-49 max|A|) |b|: the forward's fp16x3 bound for elements within
// 2^-27 of the largest, an absolute floor far below fp32's own rounding of the sum for the rest.  B (activations) must lie in fp16's
// range like every operand of the forward; beyond it the result is NaN, never a wrong finite value.
// ---------------------------------------------------------------------------------------------------
typedef _Float16 wg_h2 __attribute__((ext_vector_type(2)));
typedef _Float16 wg_h8 __attribute__((ext_vector_type(8)));
typedef float wg_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void wg_split4_h(const f32x4 x, uint2& p1, uint2& p2) {
    wg_h2 a[2], b[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const wg_f2 v = {x[2 * j], x[2 * j + 1]};
        a[j] = __builtin_convertvector(v, wg_h2);
        const wg_f2 big = v * 2048.f;
        const wg_f2 r = {__builtin_fmaf((float)a[j][0], -2048.f, big[0]), __builtin_fmaf((float)a[j][1], -2048.f, big[1])};   // exact
        b[j] = __builtin_convertvector(r, wg_h2);
    }
    p1 = make_uint2(__builtin_bit_cast(unsigned, a[0]), __builtin_bit_cast(unsigned, a[1]));
    p2 = make_uint2(__builtin_bit_cast(unsigned, b[0]), __builtin_bit_cast(unsigned, b[1]));
}

constexpr int kWhPlane = kWgPlane;   // the same [column slot][row] layout, two planes per operand: 40 KB per workgroup

__global__ __launch_bounds__(256, 2) void k_wgrad_partial_h(WgradA a_op, int lda, int Ka, const float* __restrict__ B, int ldb, int Kb,
                                                            int64_t R, int64_t rows_per_chunk, const unsigned* __restrict__ amax_bits,
                                                            float* __restrict__ partial, float* __restrict__ colsum_part) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * kWhPlane];
    unsigned char* Ap = lds;
    unsigned char* Bp = lds + 2 * kWhPlane;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i0 = blockIdx.x * kWgTile, j0 = blockIdx.y * kWgTile;
    const int64_t r_begin = (int64_t)blockIdx.z * rows_per_chunk, r_end = min(R, r_begin + rows_per_chunk);
    const int wi = wave & 1, wj = wave >> 1;
    // the scale of A: 2^k, k = 13 - floor(log2 amax) (clamped; amax = 0 or no amax: 1)
    float a_scale = 1.f, a_unscale = 1.f;
    if (amax_bits != nullptr) {
        const unsigned bits = amax_bits[0];
        const int ex = (int)((bits >> 23) & 0xFFu) - 127;
        if (bits != 0u && ex < 128) {   // (inf / NaN keep scale 1: the result is then NaN, as it should be)
            const int k = max(-100, min(100, 13 - ex));
            a_scale = __uint_as_float((unsigned)(k + 127) << 23);
            a_unscale = __uint_as_float((unsigned)(127 - k) << 23);
        }
    }
    f32x16 acc[2][2], acs[2][2];   // x1 y1 | the two small products (x 2048)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = acs[a][b][r] = 0.f;

    const int c4 = tid & 31, rr = tid >> 5;
    const bool a_in = i0 + 4 * c4 < Ka, b_in = j0 + 4 * c4 < Kb;
    // Both operands are addressed as a base that is the
    // same for the whole workgroup plus a 32-bit offset per lane (one register instead of a 64-bit pair per row - the kernel has
    // two accumulator sets and lives at the 256-register limit), and every request is unconditional: rows past the chunk's end repeat
    // its last row, columns past the operand its column 0, and both are zeroed when they are staged (the compiler cannot count loads
    // behind a branch and would wait for each as it is issued).
    const float* a_base = a_op.blk[0];
    {   // column blocks in separate buffers: a tile lies inside ONE block (the launcher requires width % 128 == 0 then) - a uniform select
        const int which = i0 / a_op.width;
#pragma unroll
        for (int k = 1; k < kWgBlocks; ++k)
            if (which == k) a_base = a_op.blk[k];
        a_base += i0 - which * a_op.width;
    }
    const float* b_base = B + (b_in ? j0 : 0);
    const unsigned a_lane = (a_in ? 4u * c4 : 0u) + 4u * rr * (unsigned)lda, b_lane = (b_in ? 4u * c4 : 0u) + 4u * rr * (unsigned)ldb;
    const bool sums = colsum_part != nullptr && blockIdx.y == 0;
    f32x4 cs = {0.f, 0.f, 0.f, 0.f};
    f32x4 av[4], bv[4];
    auto fetch = [&](int64_t r0) {
        const float* as = a_base + r0 * lda;   // (uniform)
        const float* bs = b_base + r0 * ldb;
        const int last = (int)(r_end - 1 - r0) - 4 * rr;   // the last live row of the slab, relative to this lane's first
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int tt = max(min(t, last), -4 * rr);
            av[t] = *reinterpret_cast<const f32x4*>(as + a_lane + tt * lda);
            bv[t] = *reinterpret_cast<const f32x4*>(bs + b_lane + tt * ldb);
        }
    };
    auto zeroed = [&](f32x4 (&v)[4], int64_t r0, bool in) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (!in || r0 + 4 * rr + t >= r_end) v[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto stage = [&](unsigned char* planes, const f32x4 (&v)[4], float scale) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint2 p1, p2;
            wg_split4_h(f32x4{v[0][j], v[1][j], v[2][j], v[3][j]} * scale, p1, p2);
            unsigned char* d = planes + (32 * j + c4) * kWgColBytes + 8 * rr;
            *reinterpret_cast<uint2*>(d) = p1;
            *reinterpret_cast<uint2*>(d + kWhPlane) = p2;
        }
    };
    auto h8 = [](const uint4 v) { return __builtin_bit_cast(wg_h8, v); };
    const unsigned char* ap = Ap + (64 * wi + (lane & 31)) * kWgColBytes + 16 * (lane >> 5);
    const unsigned char* bp = Bp + (64 * wj + (lane & 31)) * kWgColBytes + 16 * (lane >> 5);
    if (r_begin < r_end) fetch(r_begin);   // (an empty chunk requests nothing and writes zeros)
    for (int64_t r0 = r_begin; r0 < r_end; r0 += kWgRows) {
        zeroed(av, r0, a_in);
        zeroed(bv, r0, b_in);
        stage(Ap, av, a_scale);
        stage(Bp, bv, 1.f);
        if (sums) cs += (av[0] + av[1]) + (av[2] + av[3]);
        __syncthreads();
        fetch(min(r0 + kWgRows, r_end - 1));   // (past the end: the chunk's last row again, never used)
#pragma unroll
        for (int s = 0; s < kWgRows / 16; ++s) {
            uint4 a1[2], a2[2], b1[2], b2[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned char* pa = ap + 32 * h * kWgColBytes + 32 * s;
                const unsigned char* pb = bp + 32 * h * kWgColBytes + 32 * s;
                a1[h] = *reinterpret_cast<const uint4*>(pa);
                a2[h] = *reinterpret_cast<const uint4*>(pa + kWhPlane);
                b1[h] = *reinterpret_cast<const uint4*>(pb);
                b2[h] = *reinterpret_cast<const uint4*>(pb + kWhPlane);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    acs[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(a2[a]), h8(b1[b]), acs[a][b], 0, 0, 0);
                    acs[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(a1[a]), h8(b2[b]), acs[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(a1[a]), h8(b1[b]), acc[a][b], 0, 0, 0);
                }
        }
        __syncthreads();
    }
    float* out = partial + ((int64_t)blockIdx.z * Ka + i0) * Kb + j0;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = 4 * cd_row(r, lane) + 2 * wi + a, j = 4 * (lane & 31) + 2 * wj + b;
                if (i0 + i < Ka && j0 + j < Kb) out[(int64_t)i * Kb + j] = (acc[a][b][r] + acs[a][b][r] * (1.0f / 2048.f)) * a_unscale;
            }
    if (sums) {
        float* red = reinterpret_cast<float*>(lds);
        *reinterpret_cast<f32x4*>(red + rr * kWgTile + 4 * c4) = cs;
        __syncthreads();
        if (tid < kWgTile && i0 + tid < Ka) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) t += red[k * kWgTile + tid];
            colsum_part[(int64_t)blockIdx.z * Ka + i0 + tid] = t;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// The data gradient of the node projection as ONE fp16x3 product (round 5):  C[M, Nout] (+)= [A_0 | A_1 | ...] W^T  with the column blocks
// of A (the five node gradients) in separate buffers and W[Nout, K] row-major, K = blocks x width.  As five residual GEMMs on the edge-tile
// kernel (gnnome_linear_blocks_f32's choice at these shapes) C is read and written five times (768 MB per layer at configs[2]) and the
// products are bf16x6; here a workgroup keeps a 128 x 128 tile of C in its accumulators over all of K (358 MB), both operands are k-contiguous,
// so a 32-wide k slab is staged as two fp16 planes [row][k] without a transposition (80-byte rows: the wgrad kernels' conflict-free
// geometry, and their fragment reads), A is scaled by the power of two its maximum asks for (one slot over all blocks: k_wgrad_partial_h's
// header has the error model), three MFMAs per k step.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_dgrad_blocks_h(WgradA a_op, int lda, int K, const float* __restrict__ W, int ldw, int Nout,
                                                           int64_t M, const unsigned* __restrict__ amax_bits, float* __restrict__ C, int ldc,
                                                           int accumulate) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * kWhPlane];
    unsigned char* Ap = lds;
    unsigned char* Bp = lds + 2 * kWhPlane;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t i0 = (int64_t)blockIdx.x * kWgTile;
    const int j0 = blockIdx.y * kWgTile;
    const int wi = wave & 1, wj = wave >> 1;
    float a_scale = 1.f, a_unscale = 1.f;
    if (amax_bits != nullptr) {
        const unsigned bits = amax_bits[0];
        const int ex = (int)((bits >> 23) & 0xFFu) - 127;
        if (bits != 0u && ex < 128) {
            const int k = max(-100, min(100, 13 - ex));
            a_scale = __uint_as_float((unsigned)(k + 127) << 23);
            a_unscale = __uint_as_float((unsigned)(127 - k) << 23);
        }
    }
    f32x16 acc[2][2], acs[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = acs[a][b][r] = 0.f;
    // a thread stages pieces f = tid + 256 i of a 128-row x 32-k slab: row f / 8, k quad f % 8 (16 bytes of a row); rows past the operand repeat
    // its last row and are zeroed when staged - every request unconditional
    const int kq = tid & 7, row0 = tid >> 3;   // rows row0 + 32 i
    const int a_rows = (int)min((int64_t)kWgTile, M - i0), b_rows = min(kWgTile, Nout - j0);
    unsigned a_off[4], b_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a_off[i] = (unsigned)min(row0 + 32 * i, a_rows - 1) * (unsigned)lda + 4u * kq;
        b_off[i] = (unsigned)min(row0 + 32 * i, b_rows - 1) * (unsigned)ldw + 4u * kq;
    }
    const float* w_base = W + (int64_t)j0 * ldw;
    f32x4 av[4], bv[4];
    auto fetch = [&](int k0) {
        const float* ab = a_op.blk[0];
        const int which = k0 / a_op.width;   // (uniform: a slab lies inside one block, width % 32 == 0)
#pragma unroll
        for (int k = 1; k < kWgBlocks; ++k)
            if (which == k) ab = a_op.blk[k];
        ab += i0 * lda + (k0 - which * a_op.width);
        const float* wb = w_base + k0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            av[i] = *reinterpret_cast<const f32x4*>(ab + a_off[i]);
            bv[i] = *reinterpret_cast<const f32x4*>(wb + b_off[i]);
        }
    };
    auto stage = [&](unsigned char* planes, const f32x4 (&v)[4], float scale, int rows_live) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint2 p1, p2;
            const f32x4 x = row0 + 32 * i < rows_live ? v[i] * scale : f32x4{0.f, 0.f, 0.f, 0.f};
            wg_split4_h(x, p1, p2);
            unsigned char* d = planes + (row0 + 32 * i) * kWgColBytes + 8 * kq;
            *reinterpret_cast<uint2*>(d) = p1;
            *reinterpret_cast<uint2*>(d + kWhPlane) = p2;
        }
    };
    auto h8 = [](const uint4 v) { return __builtin_bit_cast(wg_h8, v); };
    const unsigned char* ap = Ap + (64 * wi + (lane & 31)) * kWgColBytes + 16 * (lane >> 5);
    const unsigned char* bp = Bp + (64 * wj + (lane & 31)) * kWgColBytes + 16 * (lane >> 5);
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += kWgRows) {
        stage(Ap, av, a_scale, a_rows);
        stage(Bp, bv, 1.f, b_rows);
        __syncthreads();
        fetch(min(k0 + kWgRows, K - kWgRows));   // (past the end: the last slab again, never used)
#pragma unroll
        for (int s = 0; s < kWgRows / 16; ++s) {
            uint4 a1[2], a2[2], b1[2], b2[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned char* pa = ap + 32 * h * kWgColBytes + 32 * s;
                const unsigned char* pb = bp + 32 * h * kWgColBytes + 32 * s;
                a1[h] = *reinterpret_cast<const uint4*>(pa);
                a2[h] = *reinterpret_cast<const uint4*>(pa + kWhPlane);
                b1[h] = *reinterpret_cast<const uint4*>(pb);
                b2[h] = *reinterpret_cast<const uint4*>(pb + kWhPlane);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    acs[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(a2[a]), h8(b1[b]), acs[a][b], 0, 0, 0);
                    acs[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(a1[a]), h8(b2[b]), acs[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(a1[a]), h8(b1[b]), acc[a][b], 0, 0, 0);
                }
        }
        __syncthreads();
    }
    // the tile's 64 old values per lane are requested together (unconditionally: an element outside the tile repeats the tile's last row / column)
    // before any is used - one trip to memory for the accumulate, not one per element
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = min(64 * wi + 32 * a + cd_row(r, lane), a_rows - 1), j = min(64 * wj + 32 * b + (lane & 31), b_rows - 1);
                acc[a][b][r] = (acc[a][b][r] + acs[a][b][r] * (1.0f / 2048.f)) * a_unscale;
                acs[a][b][r] = C[(i0 + i) * ldc + j0 + j];
            }
    if (!accumulate) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acs[a][b][r] = 0.f;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = 64 * wi + 32 * a + cd_row(r, lane), j = 64 * wj + 32 * b + (lane & 31);
                if (i < a_rows && j < b_rows) C[(i0 + i) * ldc + j0 + j] = acs[a][b][r] + acc[a][b][r];
            }
}

// ---------------------------------------------------------------------------------------------------
// The same product for operands that are whole multiples of 256 columns wide (H = 256: dW3 = dxe^T e over 2.5M+ rows, the
// [5H, H] projection gradient): one workgroup = one 256 x 256 output tile, wave (wi, wj) a 128 x 128 quadrant as 4 x 4
// accumulators (256 registers per lane: the accumulation half of the register file, one wave per SIMD).  Against the 128 x 128
// kernel above at these widths (measured at 2.5M rows, tools/wgrad_time.py: 2.25 ms, of which 1.30 matrix work + fragment reads
// and 1.38 loads + splits, adding up instead of overlapping) every row of A and B is fetched, split and staged ONCE, a fragment
// read from LDS feeds four MFMA chains instead of two, and the slab pipeline is explicit: 16-row slabs, two LDS buffers, one
// barrier per slab - a wave splits and stages slab s + 1 into the other buffer and requests slab s + 2 in the same straight-line
// block as its 96 MFMAs on slab s, so that the scheduler can put the VALU work into the shadow of the matrix pipe.
// Measured (tools/wgrad_time.py, wgrad_phase.py; 2.5M rows x 256 x 256): 1.68 ms (2.25), 4330 cycles per slab of which 3816 are the
// MFMAs + fragment reads + barrier alone (39.8 cycles per MFMA; the chip holds ~1.65 GHz under this body); [250k x 1280]^T [250k x 256]:
// 0.85 ms (1.52).
// Column c of a 256-wide operand tile lives in slot 64 (c % 4) + c / 4; fragment f covers slots 32 f .. 32 f + 31.
// ---------------------------------------------------------------------------------------------------
constexpr int kW2Tile = 256, kW2Rows = 16;
constexpr int kW2ColBytes = 2 * kW2Rows + 16;        // one column of one bf16 plane: 16 rows + 16 bytes of pad (48 B: conflict-free b128 reads)
constexpr int kW2Plane = kW2Tile * kW2ColBytes;      // 12 KB; A and B, three planes each = 72 KB per buffer, two buffers

// F16 (round 6; VERDICT r5 item 5 "fp16x3 for the H = 256 backward products ... needs a second accumulator set that does not fit beside 4 x 4
// tiles"): fp16x3 in ONE accumulator set.  The two small products carry a factor 2^11 (the second planes are stored scaled, x2 = RN16((x - x1)
// 2048), so that they are fp16 normals whenever x1 is); instead of keeping them apart and folding them in at the end, the LARGE product gets the
// same factor: A's first plane is staged twice, a1 and a1 2^11 (exact: A is a gradient scaled into [8, 16) by the power of two its maximum asks
// for - amax_bits, as in k_wgrad_partial_h - so a1 2^11 < 2^15), and
//     2^11 a b  =  (a1 2^11) b1 + a1 (b2 2^11) + (a2 2^11) b1          (+ terms <= 3 * 2^-11 of the middle ones, dropped like everywhere)
// are three MFMAs into the same accumulator, smallest first; the tile is multiplied by 2^-11 / scale on its way out.  Three planes of A, two of
// B: half of bf16x6's matrix work, 5/6 of its staging.  B (activations) in fp16's range, like every fp16x3 operand.
template <int ABL = 0, bool X16 = false, bool F16 = false>   // ABL: measurement only (1 = no re-fetch, 2 = no MFMAs, 4 = no staging); X16: A stored as bf16 (round 4)
__global__ __launch_bounds__(256, 1) void k_wgrad256_partial(WgradA a_op, int lda, int Ka, const float* __restrict__ B, int ldb, int Kb,
                                                             int64_t R, int64_t rows_per_chunk, float* __restrict__ partial,
                                                             float* __restrict__ colsum_part, long long* prof, const unsigned* __restrict__ amax_bits) {
    static_assert(!(F16 && X16), "fp16x3 takes fp32 rows of A");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds2[];
    float a_scale = 1.f, out_scale = 1.f;   // F16: A times 2^k into [8, 16), k = 3 - floor(log2 amax); the tile times 2^-(k + 11)
    if (F16) {
        out_scale = 1.0f / 2048.f;
        const unsigned bits = amax_bits != nullptr ? amax_bits[0] : 0u;
        const int ex = (int)((bits >> 23) & 0xFFu) - 127;
        if (bits != 0u && ex < 128) {   // (amax = 0: any scale; inf / NaN: scale 1 and a NaN result, as it should be)
            const int k = max(-100, min(100, 3 - ex));
            a_scale = __uint_as_float((unsigned)(k + 127) << 23);
            out_scale = __uint_as_float((unsigned)(127 - k - 11) << 23);
        }
    }
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i0 = blockIdx.x * kW2Tile, j0 = blockIdx.y * kW2Tile;
    const int64_t r_begin = (int64_t)blockIdx.z * rows_per_chunk, r_end = min(R, r_begin + rows_per_chunk);
    const int wi = wave & 1, wj = wave >> 1;
    f32x16 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int c4 = tid & 63, rr = tid >> 6;   // 64 float4 per 256-wide row; rows 4 rr .. 4 rr + 3 of the slab
    const float* a_col = a_op.blk[0];   // (X16: the block pointers are bf16 rows; a_off counts elements either way)
    int64_t a_off = 0;
    {
        const int ca = i0 + 4 * c4, which = ca / a_op.width;
#pragma unroll
        for (int k = 1; k < kWgBlocks; ++k)
            if (which == k) a_col = a_op.blk[k];
        a_off = ca - which * a_op.width;
    }
    const float* b_col = B + j0 + 4 * c4;
    const bool sums = colsum_part != nullptr && blockIdx.y == 0;
    f32x4 cs = {0.f, 0.f, 0.f, 0.f};
    // two register sets of slab rows: set X is staged while the loads of the slab after next land in set Y (requested a whole slab
    // of matrix work before they are used)
    f32x4 av0[4], bv0[4], av1[4], bv1[4];
    long long t_bar = 0;
    // One row of a slab into a register set.  The row index is wave-uniform (a wave stages four whole rows), so the clamp to the last
    // row of the chunk - slabs past its end read that row and are zeroed by mask_rows - is scalar arithmetic, no branch.
    const int rr_s = __builtin_amdgcn_readfirstlane(rr);
    auto fetch_row = [&](int64_t r0, int t, f32x4& a_dst, f32x4& b_dst) {
        const int64_t rc = min(r0 + 4 * rr_s + t, r_end - 1);
        a_dst = load4_as<X16>(a_col, a_off + rc * lda);
        b_dst = *reinterpret_cast<const f32x4*>(b_col + rc * ldb);
    };
    auto fetch = [&](int64_t r0, f32x4 (&av)[4], f32x4 (&bv)[4]) {
#pragma unroll
        for (int t = 0; t < 4; ++t) fetch_row(r0, t, av[t], bv[t]);
    };
    auto mask_rows = [&](int64_t r0, f32x4 (&av)[4], f32x4 (&bv)[4]) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float keep = r0 + 4 * rr + t < r_end ? 1.f : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                av[t][j] = keep != 0.f ? av[t][j] : 0.f;
                bv[t][j] = keep != 0.f ? bv[t][j] : 0.f;
            }
        }
    };
    auto stage_col = [&](unsigned char* planes, const f32x4 (&v)[4], int j, bool is_a) {
        uint2 p1, p2, p3;
        unsigned char* d = planes + (64 * j + c4) * kW2ColBytes + 8 * rr;
        if (F16) {
            const f32x4 x = f32x4{v[0][j], v[1][j], v[2][j], v[3][j]} * (is_a ? a_scale : 1.f);
            wg_split4_h(x, p1, p2);
            *reinterpret_cast<uint2*>(d) = p1;
            *reinterpret_cast<uint2*>(d + kW2Plane) = p2;
            if (is_a) {   // a1 2^11, exactly: the first plane as fp16 again, after the multiplication in fp32
                const wg_h2 lo = __builtin_bit_cast(wg_h2, p1.x), hi = __builtin_bit_cast(wg_h2, p1.y);
                const wg_f2 l2 = {(float)lo[0] * 2048.f, (float)lo[1] * 2048.f}, h2 = {(float)hi[0] * 2048.f, (float)hi[1] * 2048.f};
                p3 = make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(l2, wg_h2)), __builtin_bit_cast(unsigned, __builtin_convertvector(h2, wg_h2)));
                *reinterpret_cast<uint2*>(d + 2 * kW2Plane) = p3;
            }
            return;
        }
        tile_split4(f32x4{v[0][j], v[1][j], v[2][j], v[3][j]}, p1, p2, p3);
        *reinterpret_cast<uint2*>(d) = p1;
        *reinterpret_cast<uint2*>(d + kW2Plane) = p2;
        *reinterpret_cast<uint2*>(d + 2 * kW2Plane) = p3;
    };
    auto hf = [](const uint4 v) { return __builtin_bit_cast(wg_h8, v); };
    auto bf = [](const uint4 v) { return __builtin_bit_cast(tile_bf16x8, v); };
    // fragment a of this wave's half: slots 32 (4 w + a) + (lane & 31), rows 8 (lane >> 5) .. + 7
    const int frag_off_a = (128 * wi + (lane & 31)) * kW2ColBytes + 16 * (lane >> 5);
    const int frag_off_b = (128 * wj + (lane & 31)) * kW2ColBytes + 16 * (lane >> 5);
    constexpr int kBuf = 6 * kW2Plane;
    // One slab: MFMAs on buffer `cur` (slab at r0), the rows of slab r0 + 16 (set S, requested a slab ago) split and staged into the other
    // buffer one column group per accumulator column, the rows of slab r0 + 32 requested into set F first.  Straight-line: a slab past
    // the end of the chunk is staged as zeros into a buffer nobody reads.
    auto slab = [&](int64_t r0, int cur, f32x4 (&avS)[4], f32x4 (&bvS)[4], f32x4 (&avF)[4], f32x4 (&bvF)[4]) {
        const unsigned char* Ac = lds2 + cur * kBuf;
        const unsigned char* Bc = Ac + 3 * kW2Plane;
        unsigned char* An = lds2 + (cur ^ 1) * kBuf;
        unsigned char* Bn = An + 3 * kW2Plane;
        mask_rows(r0 + kW2Rows, avS, bvS);
        if (sums) cs += (avS[0] + avS[1]) + (avS[2] + avS[3]);
        uint4 a1[4], a2[4], a3[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const unsigned char* pa = Ac + frag_off_a + 32 * a * kW2ColBytes;
            a1[a] = *reinterpret_cast<const uint4*>(pa);
            a2[a] = *reinterpret_cast<const uint4*>(pa + kW2Plane);
            a3[a] = *reinterpret_cast<const uint4*>(pa + 2 * kW2Plane);
        }
        // the B fragments of column step b + 1 are requested BEFORE the MFMAs of step b (one wave per SIMD: nobody else hides an LDS round
        // trip; read just in time they cost ~200 cycles three times per step - measured 60 cycles per MFMA instead of 32)
        uint4 bq[2][3];
        {
            const unsigned char* pb = Bc + frag_off_b;
            bq[0][0] = *reinterpret_cast<const uint4*>(pb);
            bq[0][1] = *reinterpret_cast<const uint4*>(pb + kW2Plane);
            if (!F16) bq[0][2] = *reinterpret_cast<const uint4*>(pb + 2 * kW2Plane);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (b < 3) {
                const unsigned char* pb = Bc + frag_off_b + 32 * (b + 1) * kW2ColBytes;
                bq[(b + 1) & 1][0] = *reinterpret_cast<const uint4*>(pb);
                bq[(b + 1) & 1][1] = *reinterpret_cast<const uint4*>(pb + kW2Plane);
                if (!F16) bq[(b + 1) & 1][2] = *reinterpret_cast<const uint4*>(pb + 2 * kW2Plane);
            }
            const uint4 b1 = bq[b & 1][0], b2 = bq[b & 1][1], b3 = bq[b & 1][2];
            // one row pair of the slab after next per column step: eight 1 KB requests at once block the wave at the CU's memory
            // pipeline (~10 B / cycle / CU) and its MFMAs behind them - measured 4985 against 4330 cycles per slab.  (Touching the
            // slab four or three ahead with one dword per line, so that these loads hit L2, made it slower: 5680 cycles.)
            if (!(ABL & 1)) fetch_row(r0 + 2 * kW2Rows, b, avF[b], bvF[b]);
            if (!(ABL & 4)) {
                stage_col(An, avS, b, true);
                stage_col(Bn, bvS, b, false);
            }
            // the four accumulators of this column step take each term in turn: with ONE wave per SIMD a chain of dependent MFMAs would
            // leave the matrix pipe idle for the latency of each (measured: 67 cycles per MFMA instead of 32)
            if (!(ABL & 2) && F16) {   // planes of A: a1 | a2 2^11 | a1 2^11; of B: b1 | b2 2^11
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hf(a2[a]), hf(b1), acc[a][b], 0, 0, 0);   // smallest terms first
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hf(a1[a]), hf(b2), acc[a][b], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hf(a3[a]), hf(b1), acc[a][b], 0, 0, 0);
            } else if (!(ABL & 2)) {
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a3[a]), bf(b1), acc[a][b], 0, 0, 0);   // smallest terms first
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1[a]), bf(b3), acc[a][b], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a2[a]), bf(b2), acc[a][b], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a2[a]), bf(b1), acc[a][b], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1[a]), bf(b2), acc[a][b], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1[a]), bf(b1), acc[a][b], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);   // nothing moves between the four column steps: the scheduler otherwise hoists every split and spills accumulators
        }
        long long tb = 0;
        if (prof) tb = __builtin_readcyclecounter();
        __syncthreads();
        if (prof) t_bar += (long long)__builtin_readcyclecounter() - tb;
    };
    fetch(r_begin, av0, bv0);
    fetch(r_begin + kW2Rows, av1, bv1);
    mask_rows(r_begin, av0, bv0);
    if (sums) cs += (av0[0] + av0[1]) + (av0[2] + av0[3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        stage_col(lds2, av0, j, true);
        stage_col(lds2 + 3 * kW2Plane, bv0, j, false);
    }
    __syncthreads();
    const long long t_loop0 = prof ? (long long)__builtin_readcyclecounter() : 0;
    const long long t_real0 = prof ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
    for (int64_t r0 = r_begin; r0 < r_end; r0 += 2 * kW2Rows) {
        slab(r0, 0, av1, bv1, av0, bv0);
        slab(r0 + kW2Rows, 1, av0, bv0, av1, bv1);   // (past the end of an odd chunk: a slab of zeros - a branch here would double the accumulators' live ranges)
    }
    if (prof && lane == 0) {   // measurement only: [workgroup][wave][4] = loop cycles, barrier cycles, slabs, 100 MHz ticks
        long long* o = prof + ((int64_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + wave * 4;
        o[0] = (long long)__builtin_readcyclecounter() - t_loop0;
        o[1] = t_bar;
        o[2] = (r_end - r_begin + kW2Rows - 1) / kW2Rows;
        o[3] = (long long)__builtin_amdgcn_s_memrealtime() - t_real0;
    }
    float* out = partial + ((int64_t)blockIdx.z * Ka + i0) * Kb + j0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // slot 32 f + m holds column 128 (f % 2) + 4 m + f / 2 of the tile; f = 4 w + a
                const int fa = 4 * wi + a, fb = 4 * wj + b;
                const int i = 128 * (fa & 1) + 4 * cd_row(r, lane) + (fa >> 1), j = 128 * (fb & 1) + 4 * (lane & 31) + (fb >> 1);
                out[(int64_t)i * Kb + j] = F16 ? acc[a][b][r] * out_scale : acc[a][b][r];
            }
    if (sums) {   // the four row groups of a column, added in a fixed order (the loop ended on a barrier)
        float* red = reinterpret_cast<float*>(lds2);
        *reinterpret_cast<f32x4*>(red + rr * kW2Tile + 4 * c4) = cs;
        __syncthreads();
        if (tid < kW2Tile) colsum_part[(int64_t)blockIdx.z * Ka + i0 + tid] = ((red[tid] + red[kW2Tile + tid]) + red[2 * kW2Tile + tid]) + red[3 * kW2Tile + tid];
    }
}

// C = sum over chunks of the partial tiles, in a fixed order: workgroup = 64 consecutive elements, wave w of 16 adds the
// chunks c = w, w+16, ... (eight requests in flight per lane) and the sixteen wave sums are combined in wave order.  (Round 5: four waves
// per workgroup walked 128 partials each, one request at a time, on one workgroup per CU - 44 us for the 32 MB of B_3's partials, 0.57 ms
// per training step over its 23 launches.)
// Workgroups past the last element of C do the same for the column sums of A (colsum_part[chunk][Ka] -> colsum[Ka]).
constexpr int kRedWaves = 16;
__global__ __launch_bounds__(64 * kRedWaves) void k_wgrad_reduce(const float* __restrict__ partial, int64_t elems, int chunks,
                                                                float* __restrict__ C, int ldc, int Kb, int c_blocks,
                                                                const float* __restrict__ colsum_part, int Ka, float* __restrict__ colsum) {
    __shared__ float part[kRedWaves][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if ((int)blockIdx.x >= c_blocks) {
        partial = colsum_part;
        elems = Ka;
        C = colsum;
        ldc = Kb = Ka;
    }
    const int64_t i = (int64_t)(blockIdx.x >= (unsigned)c_blocks ? blockIdx.x - c_blocks : blockIdx.x) * 64 + lane;
    float s = 0.f;
    if (i < elems) {
        int c = wave;
        for (; c + 7 * kRedWaves < chunks; c += 8 * kRedWaves) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(int64_t)(c + u * kRedWaves) * elems + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; c < chunks; c += kRedWaves) s += partial[(int64_t)c * elems + i];
    }
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && i < elems) {
        float t = part[0][lane];
#pragma unroll
        for (int w = 1; w < kRedWaves; ++w) t += part[w][lane];
        C[(i / Kb) * ldc + (i % Kb)] = t;
    }
}

// ---------------------------------------------------------------------------------------------------
// score predictor tail, backward.  Forward (score_predictor.py:15-16): z2 = relu(W2 z1 + b2), s = W3.z2 + b3.
// Given z1 = relu(..) [E,hs] (saved by the forward) and ds [E] (gathered through srt_eid), one thread per edge:
//   dz2 = ds * W3 * (z2 > 0);  dz1 = (W2^T dz2) * (z1 > 0);  u = ds * z2
// dz1, dz2, u are written out; the weight gradients are column sums / wgrad products of them.
// ---------------------------------------------------------------------------------------------------
template <int HS>
__global__ __launch_bounds__(256) void k_score_tail_bwd(const float* __restrict__ z1, const float* __restrict__ ds,
                                                        const int32_t* __restrict__ srt_eid, int64_t E,
                                                        const float* __restrict__ W2, const float* __restrict__ b2,
                                                        const float* __restrict__ W3, float* __restrict__ dz1,
                                                        float* __restrict__ dz2, float* __restrict__ u) {
    __shared__ float w2s[32 * HS];
    __shared__ float b2s[32], w3s[32];
    for (int i = threadIdx.x; i < 32 * HS; i += blockDim.x) w2s[i] = W2[i];
    if (threadIdx.x < 32) {
        b2s[threadIdx.x] = b2[threadIdx.x];
        w3s[threadIdx.x] = W3[threadIdx.x];
    }
    __syncthreads();
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < E; p += (int64_t)gridDim.x * blockDim.x) {
        const float g = ds[srt_eid != nullptr ? (int64_t)srt_eid[p] : p];
        float z[HS];
#pragma unroll
        for (int j4 = 0; j4 < HS / 4; ++j4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(z1 + p * HS + 4 * j4);
#pragma unroll
            for (int j = 0; j < 4; ++j) z[4 * j4 + j] = v[j];
        }
        float d1[HS];
#pragma unroll
        for (int j = 0; j < HS; ++j) d1[j] = 0.f;
        for (int k = 0; k < 32; ++k) {
            float a = b2s[k];
#pragma unroll
            for (int j = 0; j < HS; ++j) a += w2s[k * HS + j] * z[j];
            const float z2 = fmaxf(a, 0.f);
            const float d2 = a > 0.f ? g * w3s[k] : 0.f;
            dz2[p * 32 + k] = d2;
            u[p * 32 + k] = g * z2;
#pragma unroll
            for (int j = 0; j < HS; ++j) d1[j] += w2s[k * HS + j] * d2;
        }
#pragma unroll
        for (int j4 = 0; j4 < HS / 4; ++j4) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = z[4 * j4 + j] > 0.f ? d1[4 * j4 + j] : 0.f;
            *reinterpret_cast<f32x4*>(dz1 + p * HS + 4 * j4) = v;
        }
    }
}

// The same contract with every global access coalesced (round 3; the kernel above reads and writes a row per LANE: 0.76 ms at 1M
// edges for 768 MB, its 4-byte stores of dz2 / u touch 64 lines per instruction).  A workgroup of 128 threads owns 128 edges per step:
// the z1 tile comes in as 16-byte pieces of consecutive rows, is transposed through LDS (row stride HS + 1: a thread reads ITS row
// without bank conflicts), every thread runs the two small products of its edge with W2 rows fetched by the scalar unit (the k loop
// is uniform), and dz2, u, dz1 go back through LDS the same way.
template <int HS>
__global__ __launch_bounds__(128) void k_score_tail_bwd_tiles(const float* __restrict__ z1, const float* __restrict__ ds,
                                                              const int32_t* __restrict__ srt_eid, int64_t E,
                                                              const float* __restrict__ W2, const float* __restrict__ b2,
                                                              const float* __restrict__ W3, float* __restrict__ dz1,
                                                              float* __restrict__ dz2, float* __restrict__ u) {
    constexpr int TE = 128, LZ = HS + 1, LO = 33, C4 = HS / 4;
    __shared__ float zt[TE * LZ];   // z1 tile, later the u tile [TE][LO], later the dz1 tile
    __shared__ float ot[TE * LO];   // dz2 tile
    const int t = threadIdx.x;
    const int64_t tiles = (E + TE - 1) / TE;
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t p0 = tile * TE;
        const int valid = (int)min((int64_t)TE, E - p0);
#pragma unroll 4
        for (int i = 0; i < C4; ++i) {   // TE * C4 pieces, 128 per pass
            const int f = t + TE * i, row = f / C4, c4 = f % C4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row < valid) v = *reinterpret_cast<const f32x4*>(z1 + (p0 + row) * HS + 4 * c4);
#pragma unroll
            for (int j = 0; j < 4; ++j) zt[row * LZ + 4 * c4 + j] = v[j];
        }
        const int64_t p = p0 + min(t, valid - 1);
        const float g = ds[srt_eid != nullptr ? (int64_t)srt_eid[p] : p];
        __syncthreads();
        float z[HS], d1[HS];
#pragma unroll
        for (int j = 0; j < HS; ++j) {
            z[j] = zt[t * LZ + j];
            d1[j] = 0.f;
        }
        __syncthreads();   // every row is in registers: the z tile's space becomes the u tile
        for (int k = 0; k < 32; ++k) {
            const float* w = W2 + k * HS;   // uniform: scalar loads
            float a = b2[k];
#pragma unroll
            for (int j = 0; j < HS; ++j) a += w[j] * z[j];
            const float z2 = fmaxf(a, 0.f);
            const float d2 = a > 0.f ? g * W3[k] : 0.f;
            ot[t * LO + k] = d2;
            zt[t * LO + k] = g * z2;
#pragma unroll
            for (int j = 0; j < HS; ++j) d1[j] += w[j] * d2;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {   // TE * 8 pieces of dz2 and of u
            const int f = t + TE * i, row = f / 8, c4 = f % 8;
            if (row < valid) {
                const float* a = ot + row * LO + 4 * c4;
                const float* b = zt + row * LO + 4 * c4;
                *reinterpret_cast<f32x4*>(dz2 + (p0 + row) * 32 + 4 * c4) = f32x4{a[0], a[1], a[2], a[3]};
                *reinterpret_cast<f32x4*>(u + (p0 + row) * 32 + 4 * c4) = f32x4{b[0], b[1], b[2], b[3]};
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < HS; ++j) zt[t * LZ + j] = z[j] > 0.f ? d1[j] : 0.f;
        __syncthreads();
#pragma unroll 4
        for (int i = 0; i < C4; ++i) {
            const int f = t + TE * i, row = f / C4, c4 = f % C4;
            if (row < valid) {
                const float* a = zt + row * LZ + 4 * c4;
                *reinterpret_cast<f32x4*>(dz1 + (p0 + row) * HS + 4 * c4) = f32x4{a[0], a[1], a[2], a[3]};
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// gated aggregation, backward, per edge (gated_gcn_full.py:111-114,124-127).  With s = sigmoid(e'),
//   hf_i = sum s A2h[src] / (sum s + eps)  =>  d s_p (forward part)  = Tf[dst] * A2h[src] - Uf[dst]
//   hb_i = sum s A3h[dst] / (sum s + eps)  =>  d s_p (backward part) = Tb[src] * A3h[dst] - Ub[src]
// where T = dv * rden and U = T * h are node tables prepared by k_mul23.   de[p] += s (1 - s) (d s_p).
// ---------------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(256) void k_agg_edge_bwd(const float* __restrict__ e, int64_t E, const float* __restrict__ Tf,
                                                      const float* __restrict__ Uf, const float* __restrict__ Tb,
                                                      const float* __restrict__ Ub, const float* __restrict__ A2h,
                                                      const float* __restrict__ A3h, int ldn,
                                                      const int32_t* __restrict__ srt_src, const int32_t* __restrict__ srt_dst,
                                                      float* __restrict__ de) {
    constexpr int LPR = H / 4;
    const int64_t total = E * LPR;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / LPR;
        const int c = (int)(i % LPR) * 4;
        const int64_t s_ = srt_src[p], d_ = srt_dst[p];
        const f32x4 x = *reinterpret_cast<const f32x4*>(e + p * H + c);
        const f32x4 tf = *reinterpret_cast<const f32x4*>(Tf + d_ * H + c), uf = *reinterpret_cast<const f32x4*>(Uf + d_ * H + c);
        const f32x4 tb = *reinterpret_cast<const f32x4*>(Tb + s_ * H + c), ub = *reinterpret_cast<const f32x4*>(Ub + s_ * H + c);
        const f32x4 a2 = *reinterpret_cast<const f32x4*>(A2h + s_ * ldn + c), a3 = *reinterpret_cast<const f32x4*>(A3h + d_ * ldn + c);
        f32x4 g = *reinterpret_cast<const f32x4*>(de + p * H + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float sg = sigmoidf_(x[j]);
            g[j] += sg * (1.f - sg) * (tf[j] * a2[j] - uf[j] + tb[j] * a3[j] - ub[j]);
        }
        *reinterpret_cast<f32x4*>(de + p * H + c) = g;
    }
}

// t[r,:] = relu(W1 * in[row(r),:] + b1)   (hidden activations of an encoder, recomputed for its backward)
__global__ __launch_bounds__(256) void k_encode_hidden(const float* __restrict__ in, int64_t rows, int F,
                                                       const int32_t* __restrict__ gather, const float* __restrict__ W1,
                                                       const float* __restrict__ b1, int M, float* __restrict__ t) {
    const int64_t total = rows * M;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / M;
        const int j = (int)(i % M);
        const int64_t rin = gather != nullptr ? (int64_t)gather[r] : r;
        float s = b1[j];
        for (int f = 0; f < F; ++f) s += W1[j * F + f] * in[rin * F + f];
        t[i] = fmaxf(s, 0.f);
    }
}

// dx = dy * (y > 0)
__global__ __launch_bounds__(256) void k_relu_bwd(const float* __restrict__ dy, const float* __restrict__ y, int64_t n,
                                                  float* __restrict__ dx) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}

static unsigned grid_for_items(int64_t n) {
    int64_t b = (n + 255) / 256;
    if (b < 1) b = 1;
    if (b > kNumCUs * 8) b = kNumCUs * 8;
    return (unsigned)b;
}

}  // namespace gnnome

using namespace gnnome;

// Row chunks of one wgrad call: enough workgroups (output tiles x chunks) to put two on every CU - the LDS footprint admits
// two - but no chunk shorter than 256 rows (each chunk costs a [Ka,Kb] partial tile written and read back).  At 1M rows and
// one output tile that is ~2000 rows per chunk; at 100k rows and five tiles (the [N,5H]^T [N,H] projection gradient) 1024,
// where the old fixed 2048 left one 4-wave workgroup per CU.
static int64_t wgrad_chunks(int64_t rows, int Ka, int Kb, int64_t* rows_per_chunk) {
    const int64_t tiles = (int64_t)((Ka + kWgTile - 1) / kWgTile) * ((Kb + kWgTile - 1) / kWgTile);
    int64_t chunks = (2 * kNumCUs + tiles - 1) / tiles;
    const int64_t most = (rows + 255) / 256;
    if (chunks > most) chunks = most;
    if (chunks > 1024) chunks = 1024;
    if (chunks < 1) chunks = 1;
    int64_t rpc = (rows + chunks - 1) / chunks;
    rpc = (rpc + kWgRows - 1) / kWgRows * kWgRows;
    if (rpc < kWgRows) rpc = kWgRows;
    if (rows_per_chunk) *rows_per_chunk = rpc;
    return rows > 0 ? (rows + rpc - 1) / rpc : 1;
}

// The 256 x 256 tile kernel: one workgroup per CU, chunks a whole number of 16-row slabs (0: the shape does not take that kernel).
static int64_t wgrad256_chunks(int64_t rows, int Ka, int Kb, int64_t* rows_per_chunk) {
    if (Ka % kW2Tile || Kb % kW2Tile || rows < 16384) return 0;
    const int64_t tiles = (int64_t)(Ka / kW2Tile) * (Kb / kW2Tile);
    int64_t ch = kNumCUs / tiles > 0 ? kNumCUs / tiles : 1;
    int64_t rp = (rows + ch - 1) / ch;
    rp = (rp + kW2Rows - 1) / kW2Rows * kW2Rows;
    if (rows_per_chunk) *rows_per_chunk = rp;
    return (rows + rp - 1) / rp;
}

extern "C" int gnnome_wgrad_workspace_bytes(int64_t rows, int Ka, int Kb, size_t* bytes_host) {
    GN_REQUIRE(bytes_host && rows >= 0 && Ka > 0 && Kb > 0, "wgrad: bad arguments");
    const int64_t chunks = std::max(wgrad_chunks(rows, Ka, Kb, nullptr), wgrad256_chunks(rows, Ka, Kb, nullptr));
    *bytes_host = (size_t)chunks * ((size_t)Ka * Kb + Ka) * sizeof(float);   // partial tiles + partial column sums
    return GNNOME_OK;
}

static int wgrad_impl(const WgradA& a_op, int lda, int Ka, const float* B, int ldb, int Kb, int64_t rows, float* C, int ldc,
                      float* colsum, void* workspace, size_t workspace_bytes, void* stream, bool x16 = false,
                      const unsigned* amax_bits = nullptr) {
    GN_REQUIRE(rows >= 0 && Ka > 0 && Kb > 0 && Ka % 4 == 0 && Kb % 4 == 0, "wgrad: Ka=%d Kb=%d must be positive multiples of 4", Ka, Kb);
    GN_REQUIRE(C && ldc >= Kb, "wgrad: bad output");
    hipStream_t s = (hipStream_t)stream;
    if (rows == 0) {
        for (int i = 0; i < Ka; ++i) GN_HIP(hipMemsetAsync(C + (int64_t)i * ldc, 0, Kb * sizeof(float), s));
        if (colsum) GN_HIP(hipMemsetAsync(colsum, 0, Ka * sizeof(float), s));
        return GNNOME_OK;
    }
    GN_REQUIRE(B && workspace && ldb >= Kb && lda % 4 == 0 && ldb % 4 == 0, "wgrad: bad operands");
    GN_REQUIRE((uintptr_t)B % 16 == 0, "wgrad: A and B must be 16-byte aligned");
    int64_t rpc = 0;
    const int64_t chunks = wgrad_chunks(rows, Ka, Kb, &rpc);
    const size_t need = (size_t)chunks * ((size_t)Ka * Kb + Ka) * sizeof(float);
    if (workspace_bytes < need) {
        set_error("wgrad: workspace %zu < %zu bytes", workspace_bytes, need);
        return GNNOME_EWORKSPACE;
    }
    float* partial = (float*)workspace;
    if (wgrad256_chunks(rows, Ka, Kb, nullptr) > 0 && (a_op.width % kW2Tile == 0) && tuning(kTuneGateExperiment) != 78 &&
        tuning(kTuneLinearVariant) == 0) {   // (key 4 = 78 or any non-default key 2: the 128 x 128 tile kernel, for A/B runs and cross-checks)
        // whole 256-column tiles over many rows: the 256 x 256 kernel, one workgroup per CU (fewer, longer chunks than the workspace was sized for)
        int64_t rp = 0;
        const int64_t ch = wgrad256_chunks(rows, Ka, Kb, &rp);
        const size_t need2 = (size_t)ch * ((size_t)Ka * Kb + Ka) * sizeof(float);
        if (workspace_bytes < need2) {
            set_error("wgrad: workspace %zu < %zu bytes", workspace_bytes, need2);
            return GNNOME_EWORKSPACE;
        }
        float* cpart = colsum ? partial + (size_t)ch * Ka * Kb : nullptr;
#define GN_W256(ABLV) GN_W256X(ABLV, false, false)
#define GN_W256X(ABLV, X16V, F16V)                                                                                                           \
    {                                                                                                                                       \
        /* the attribute is per device and per function: one flag per device ordinal (set once, outside any stream capture's first use) */  \
        static std::atomic<bool> attr_set[64];                                                                                              \
        int dev_ = 0;                                                                                                                       \
        GN_HIP(hipGetDevice(&dev_));                                                                                                        \
        if (dev_ < 0 || dev_ >= 64 || !attr_set[dev_].load(std::memory_order_acquire)) {                                                    \
            GN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad256_partial<ABLV, X16V, F16V>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                       12 * kW2Plane));                                                                                     \
            if (dev_ >= 0 && dev_ < 64) attr_set[dev_].store(true, std::memory_order_release);                                              \
        }                                                                                                                                   \
        hipLaunchKernelGGL((k_wgrad256_partial<ABLV, X16V, F16V>), dim3(Ka / kW2Tile, Kb / kW2Tile, (unsigned)ch), dim3(256), 12 * kW2Plane, s, a_op, lda, \
                           Ka, B, ldb, Kb, rows, rp, partial, cpart, gate_profile_buffer(), amax_bits);                                     \
    }
        if (x16) {
            GN_W256X(0, true, false);
        } else if (amax_bits != nullptr && tuning(kTuneArith) != 1 && tuning(kTuneGateAblation) != 17) {
            GN_W256X(0, false, true);   // max |A| known: fp16x3 in one accumulator set (gnnome_set_tuning(10, 1) or (1, 17): bf16x6)
        } else switch (tuning(kTuneGateAblation)) {
            case 1: GN_W256(1); break;
            case 2: GN_W256(2); break;
            case 4: GN_W256(4); break;
            case 5: GN_W256(5); break;
            default: GN_W256(0); break;
        }
#undef GN_W256
#undef GN_W256X
        GN_LAUNCH_CHECK();
        const int64_t elems2 = (int64_t)Ka * Kb;
        const int cb2 = (int)((elems2 + 63) / 64), sb2 = colsum ? (Ka + 63) / 64 : 0;
        hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)(cb2 + sb2)), dim3(64 * kRedWaves), 0, s, (const float*)partial, elems2, (int)ch, C, ldc, Kb, cb2,
                           (const float*)cpart, Ka, colsum);
        GN_LAUNCH_CHECK();
        return GNNOME_OK;
    }
    float* colsum_part = colsum ? partial + (size_t)chunks * Ka * Kb : nullptr;
    const dim3 grid((Ka + kWgTile - 1) / kWgTile, (Kb + kWgTile - 1) / kWgTile, (unsigned)chunks);
    if (x16)
        hipLaunchKernelGGL(k_wgrad_partial<true>, grid, dim3(256), 0, s, a_op, lda, Ka, B, ldb, Kb, rows, rpc, partial, colsum_part);
    else if (tuning(kTuneGateAblation) == 1)
        hipLaunchKernelGGL((k_wgrad_partial<false, 1>), grid, dim3(256), 0, s, a_op, lda, Ka, B, ldb, Kb, rows, rpc, partial, colsum_part);
    else if (tuning(kTuneGateAblation) == 2)
        hipLaunchKernelGGL((k_wgrad_partial<false, 2>), grid, dim3(256), 0, s, a_op, lda, Ka, B, ldb, Kb, rows, rpc, partial, colsum_part);
    else if (tuning(kTuneGateAblation) == 4)
        hipLaunchKernelGGL((k_wgrad_partial<false, 4>), grid, dim3(256), 0, s, a_op, lda, Ka, B, ldb, Kb, rows, rpc, partial, colsum_part);
    else if (tuning(kTuneGateAblation) == 5)
        hipLaunchKernelGGL((k_wgrad_partial<false, 5>), grid, dim3(256), 0, s, a_op, lda, Ka, B, ldb, Kb, rows, rpc, partial, colsum_part);
    else if (tuning(kTuneGateAblation) == 16)   // timing only: fp16x3 without a scale
        hipLaunchKernelGGL(k_wgrad_partial_h, grid, dim3(256), 0, s, a_op, lda, Ka, B, ldb, Kb, rows, rpc, (const unsigned*)nullptr, partial, colsum_part);
    else if (amax_bits != nullptr && (a_op.width >= Ka || a_op.width % kWgTile == 0) && tuning(kTuneArith) != 1 && tuning(kTuneGateAblation) != 17)
        // fp16x3 with A scaled by the power of two that max |A| asks for (one buffer of A, or column blocks whose width is a whole number of
        // tiles; gnnome_set_tuning(10, 1) or (1, 17): bf16x6)
        hipLaunchKernelGGL(k_wgrad_partial_h, grid, dim3(256), 0, s, a_op, lda, Ka, B, ldb, Kb, rows, rpc, amax_bits, partial, colsum_part);
    else
        hipLaunchKernelGGL(k_wgrad_partial<false>, grid, dim3(256), 0, s, a_op, lda, Ka, B, ldb, Kb, rows, rpc, partial, colsum_part);
    GN_LAUNCH_CHECK();
    const int64_t elems = (int64_t)Ka * Kb;
    const int c_blocks = (int)((elems + 63) / 64), s_blocks = colsum ? (Ka + 63) / 64 : 0;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)(c_blocks + s_blocks)), dim3(64 * kRedWaves), 0, s, (const float*)partial, elems, (int)chunks, C,
                       ldc, Kb, c_blocks, (const float*)colsum_part, Ka, colsum);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

// The same product with max |A| known (amax_bits: the bits of a non-negative float on the device, as gnnome_bn_bwd_dgrad_amax_f32 leaves
// them): where the 128 x 128 tile kernel runs it runs as fp16x3 with A scaled into fp16's range (k_wgrad_partial_h) - half the matrix work
// of bf16x6 at a third of its error; 256-wide operands over many rows take the 256 x 256 tile kernel's one-accumulator fp16x3 form (round 6).
// B must lie in fp16's range (NaN rows otherwise).
extern "C" int gnnome_wgrad_scaled_f32(const float* A, int lda, int Ka, const float* B, int ldb, int Kb, int64_t rows, const unsigned* amax_bits,
                                       float* C, int ldc, void* workspace, size_t workspace_bytes, void* stream) {
    GN_REQUIRE(rows == 0 || (A && lda >= Ka && (uintptr_t)A % 16 == 0 && amax_bits), "wgrad_scaled: bad operands");
    WgradA a_op = {};
    a_op.blk[0] = A;
    a_op.width = Ka > 0 ? Ka : 1;
    return wgrad_impl(a_op, lda, Ka, B, ldb, Kb, rows, C, ldc, nullptr, workspace, workspace_bytes, stream, false, amax_bits);
}

extern "C" int gnnome_wgrad_f32(const float* A, int lda, int Ka, const float* B, int ldb, int Kb, int64_t rows, float* C,
                                int ldc, void* workspace, size_t workspace_bytes, void* stream) {
    GN_REQUIRE(rows == 0 || (A && lda >= Ka && (uintptr_t)A % 16 == 0), "wgrad: bad operands");
    WgradA a_op = {};
    a_op.blk[0] = A;
    a_op.width = Ka > 0 ? Ka : 1;
    return wgrad_impl(a_op, lda, Ka, B, ldb, Kb, rows, C, ldc, nullptr, workspace, workspace_bytes, stream);
}

// A stored as bf16 ([rows, Ka], row stride lda ELEMENTS, 8-byte aligned rows): the weight gradient of B_3 from the bf16 dxe
extern "C" int gnnome_wgrad_x16(const uint16_t* A, int lda, int Ka, const float* B, int ldb, int Kb, int64_t rows, float* C,
                                int ldc, void* workspace, size_t workspace_bytes, void* stream) {
    GN_REQUIRE(rows == 0 || (A && lda >= Ka && (uintptr_t)A % 8 == 0), "wgrad_x16: bad operands");
    WgradA a_op = {};
    a_op.blk[0] = reinterpret_cast<const float*>(A);
    a_op.width = Ka > 0 ? Ka : 1;
    return wgrad_impl(a_op, lda, Ka, B, ldb, Kb, rows, C, ldc, nullptr, workspace, workspace_bytes, stream, true);
}

extern "C" int gnnome_wgrad_blocks_f32(const float* const* A_blocks, int num_blocks, int block_width, int lda, const float* B, int ldb,
                                       int Kb, int64_t rows, float* C, int ldc, float* colsum, void* workspace, size_t workspace_bytes,
                                       void* stream) {
    GN_REQUIRE(A_blocks && num_blocks >= 1 && num_blocks <= kWgBlocks && block_width > 0 && block_width % 4 == 0 && lda >= block_width,
               "wgrad_blocks: 1..%d blocks of a width that is a multiple of 4", kWgBlocks);
    WgradA a_op = {};
    for (int k = 0; k < num_blocks; ++k) {
        GN_REQUIRE(rows == 0 || (A_blocks[k] && (uintptr_t)A_blocks[k] % 16 == 0), "wgrad_blocks: block %d null or not 16-byte aligned", k);
        a_op.blk[k] = A_blocks[k];
    }
    a_op.width = block_width;
    return wgrad_impl(a_op, lda, num_blocks * block_width, B, ldb, Kb, rows, C, ldc, colsum, workspace, workspace_bytes, stream);
}

// gnnome_wgrad_blocks_f32 with max |A| over ALL blocks known on the device (one slot raised by the kernels that produced the blocks): fp16x3 with
// that common scale where the 128 x 128 tile kernel runs and block_width is a multiple of 128; elsewhere amax_bits is ignored.
extern "C" int gnnome_wgrad_blocks_scaled_f32(const float* const* A_blocks, int num_blocks, int block_width, int lda, const float* B, int ldb,
                                              int Kb, int64_t rows, const unsigned* amax_bits, float* C, int ldc, float* colsum,
                                              void* workspace, size_t workspace_bytes, void* stream) {
    GN_REQUIRE(A_blocks && amax_bits && num_blocks >= 1 && num_blocks <= kWgBlocks && block_width > 0 && block_width % 4 == 0 && lda >= block_width,
               "wgrad_blocks_scaled: 1..%d blocks of a width that is a multiple of 4, and the maximum's slot", kWgBlocks);
    WgradA a_op = {};
    for (int k = 0; k < num_blocks; ++k) {
        GN_REQUIRE(rows == 0 || (A_blocks[k] && (uintptr_t)A_blocks[k] % 16 == 0), "wgrad_blocks_scaled: block %d null or not 16-byte aligned", k);
        a_op.blk[k] = A_blocks[k];
    }
    a_op.width = block_width;
    return wgrad_impl(a_op, lda, num_blocks * block_width, B, ldb, Kb, rows, C, ldc, colsum, workspace, workspace_bytes, stream, false, amax_bits);
}

// C[M,Nout] (+)= [A_0 | A_1 | ...] W^T as ONE fp16x3 launch, the blocks scaled by their common maximum (amax_bits: the slot their producers raised)
extern "C" int gnnome_linear_blocks_scaled_f32(const float* const* A_blocks, int num_blocks, int block_width, int64_t M, int lda, const float* W,
                                               int ldw, int Nout, const unsigned* amax_bits, float* C, int ldc, int accumulate, void* stream) {
    GN_REQUIRE(M >= 0 && Nout > 0, "linear_blocks_scaled: bad shape M=%lld Nout=%d", (long long)M, Nout);
    if (M == 0) return GNNOME_OK;
    GN_REQUIRE(A_blocks && amax_bits && num_blocks >= 1 && num_blocks <= kWgBlocks && block_width > 0 && block_width % kWgRows == 0,
               "linear_blocks_scaled: 1..%d blocks of a width that is a multiple of %d, and the maximum's slot", kWgBlocks, kWgRows);
    const int K = num_blocks * block_width;
    GN_REQUIRE(W && C && lda >= block_width && ldw >= K && ldc >= Nout && lda % 4 == 0 && ldw % 4 == 0 && (uintptr_t)W % 16 == 0 &&
                   (int64_t)kWgTile * lda < (1ll << 31) && (int64_t)kWgTile * ldw < (1ll << 31),
               "linear_blocks_scaled: bad operands");
    WgradA a_op = {};
    for (int k = 0; k < num_blocks; ++k) {
        GN_REQUIRE(A_blocks[k] && (uintptr_t)A_blocks[k] % 16 == 0, "linear_blocks_scaled: block %d null or not 16-byte aligned", k);
        a_op.blk[k] = A_blocks[k];
    }
    a_op.width = block_width;
    const int64_t mt = (M + kWgTile - 1) / kWgTile;
    GN_REQUIRE(mt < (1ll << 31), "linear_blocks_scaled: too many row tiles");
    hipLaunchKernelGGL(k_dgrad_blocks_h, dim3((unsigned)mt, (unsigned)((Nout + kWgTile - 1) / kWgTile)), dim3(256), 0, (hipStream_t)stream, a_op, lda, K, W,
                       ldw, Nout, M, amax_bits, C, ldc, accumulate);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

extern "C" int gnnome_score_tail_bwd_f32(const float* z1, const float* dscore, const int32_t* srt_eid, int64_t num_edges,
                                         int hidden_edge_scores, const float* W2, const float* b2, const float* W3, float* dz1,
                                         float* dz2, float* u, void* stream) {
    GN_REQUIRE(num_edges >= 0, "score_tail_bwd: negative edge count");
    if (num_edges == 0) return GNNOME_OK;
    GN_REQUIRE(z1 && dscore && W2 && b2 && W3 && dz1 && dz2 && u, "score_tail_bwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(grid_for_items(num_edges)), block(256);
    if (tuning(kTuneGateExperiment) != 79 && (hidden_edge_scores == 32 || hidden_edge_scores == 64) && (uintptr_t)z1 % 16 == 0 &&
        (uintptr_t)dz1 % 16 == 0 && (uintptr_t)dz2 % 16 == 0 && (uintptr_t)u % 16 == 0) {   // (79: the row-per-lane kernel, for A/B runs)
        const int64_t tiles = (num_edges + 127) / 128;
        const dim3 g2((unsigned)std::min<int64_t>(tiles, (int64_t)kNumCUs * 3));
        if (hidden_edge_scores == 32)
            hipLaunchKernelGGL(k_score_tail_bwd_tiles<32>, g2, dim3(128), 0, s, z1, dscore, srt_eid, num_edges, W2, b2, W3, dz1, dz2, u);
        else
            hipLaunchKernelGGL(k_score_tail_bwd_tiles<64>, g2, dim3(128), 0, s, z1, dscore, srt_eid, num_edges, W2, b2, W3, dz1, dz2, u);
        GN_LAUNCH_CHECK();
        return GNNOME_OK;
    }
    switch (hidden_edge_scores) {
        case 32: hipLaunchKernelGGL(k_score_tail_bwd<32>, grid, block, 0, s, z1, dscore, srt_eid, num_edges, W2, b2, W3, dz1, dz2, u); break;
        case 64: hipLaunchKernelGGL(k_score_tail_bwd<64>, grid, block, 0, s, z1, dscore, srt_eid, num_edges, W2, b2, W3, dz1, dz2, u); break;
        // round 5: the widest built scorer (inference always took it) - the row-per-lane form, a rare configuration
        case 128: hipLaunchKernelGGL(k_score_tail_bwd<128>, grid, block, 0, s, z1, dscore, srt_eid, num_edges, W2, b2, W3, dz1, dz2, u); break;
        default: set_error("score_tail_bwd: hidden_edge_scores=%d not in {32,64,128}", hidden_edge_scores); return GNNOME_EINVAL;
    }
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

extern "C" int gnnome_agg_edge_bwd_f32(const float* e, int64_t num_edges, int hidden, const float* Tf, const float* Uf,
                                       const float* Tb, const float* Ub, const float* A2h, const float* A3h, int ld_node,
                                       const int32_t* srt_src, const int32_t* srt_dst, float* de, void* stream) {
    GN_REQUIRE(num_edges >= 0, "agg_edge_bwd: negative edge count");
    if (num_edges == 0) return GNNOME_OK;
    GN_REQUIRE(e && Tf && Uf && Tb && Ub && A2h && A3h && srt_src && srt_dst && de && ld_node % 4 == 0, "agg_edge_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(grid_for_items(num_edges * (hidden / 4))), block(256);
    switch (hidden) {
        case 64: hipLaunchKernelGGL(k_agg_edge_bwd<64>, grid, block, 0, s, e, num_edges, Tf, Uf, Tb, Ub, A2h, A3h, ld_node, srt_src, srt_dst, de); break;
        case 128: hipLaunchKernelGGL(k_agg_edge_bwd<128>, grid, block, 0, s, e, num_edges, Tf, Uf, Tb, Ub, A2h, A3h, ld_node, srt_src, srt_dst, de); break;
        case 256: hipLaunchKernelGGL(k_agg_edge_bwd<256>, grid, block, 0, s, e, num_edges, Tf, Uf, Tb, Ub, A2h, A3h, ld_node, srt_src, srt_dst, de); break;
        default: set_error("agg_edge_bwd: hidden=%d not in {64,128,256}", hidden); return GNNOME_EINVAL;
    }
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

extern "C" int gnnome_encode_hidden_f32(const float* in, int64_t rows, int in_features, const int32_t* gather, const float* W1,
                                        const float* b1, int hidden_ne, float* t, void* stream) {
    GN_REQUIRE(rows >= 0 && in_features > 0 && hidden_ne > 0, "encode_hidden: bad shape");
    if (rows == 0) return GNNOME_OK;
    GN_REQUIRE(in && W1 && b1 && t, "encode_hidden: null pointer");
    hipLaunchKernelGGL(k_encode_hidden, dim3(grid_for_items(rows * hidden_ne)), dim3(256), 0, (hipStream_t)stream, in, rows,
                       in_features, gather, W1, b1, hidden_ne, t);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

extern "C" int gnnome_relu_bwd_f32(const float* dy, const float* y, int64_t count, float* dx, void* stream) {
    GN_REQUIRE(count >= 0, "relu_bwd: negative count");
    if (count == 0) return GNNOME_OK;
    GN_REQUIRE(dy && y && dx, "relu_bwd: null pointer");
    hipLaunchKernelGGL(k_relu_bwd, dim3(grid_for_items(count)), dim3(256), 0, (hipStream_t)stream, dy, y, count, dx);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}
