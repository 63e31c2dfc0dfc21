// gnnome_linear_f32: C[M,Nout] = A[M,K] * W[Nout,K]^T + bias on the fp32 matrix cores.
// Stands in for the nn.Linear calls on node rows - A_1,A_2,A_3,B_1,B_2 (gated_gcn_full.py:91-96,
// run as ONE GEMM over the row-concatenated weights) and the node halves of predictor.W1
// (score_predictor.py:13-14).  Bound: MFMA (2*M*K*Nout flop against 157.3 TF fp32).
#include "gemm_tile.h"

namespace gnnome {

template <int NB>
__global__ __launch_bounds__(kGemmThreads) void k_linear(const float* __restrict__ A, int64_t M, int K, int lda,
                                                         const float* __restrict__ W, int ldw,
                                                         const float* __restrict__ bias, int Nout,
                                                         float* __restrict__ C, int ldc, int n_tiles, int total_tiles,
                                                         int accumulate) {
    __shared__ __attribute__((aligned(16))) float lds[tile_lds_floats<NB>()];
    float* As = lds;
    float* Ws = lds + kTileM * kLdk;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // consecutive tile ids share the same A rows; keep them on one XCD's L2
    const int tile = xcd_remap(blockIdx.x, total_tiles);
    const int64_t row0 = (int64_t)(tile / n_tiles) * kTileM;
    const int col0 = (tile % n_tiles) * 32 * NB;

    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int col = col0 + 32 * nb + (lane & 31);
        const float b = (bias != nullptr && col < Nout) ? bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = b;
    }
    tile_gemm<NB>(acc, A, row0, M, lda, W, col0, Nout, ldw, K, As, Ws, tid);

#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int col = col0 + 32 * nb + (lane & 31);
        if (col < Nout) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = row0 + 32 * wave + cd_row(r, lane);
                if (row < M) C[row * ldc + col] = accumulate ? C[row * ldc + col] + acc[nb][r] : acc[nb][r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Weight-stationary persistent form, used when K is 64 or 128 and Nout is a multiple of the chunk width.
// A workgroup of 8 waves keeps one 32*CB-column chunk of W (all of K) in LDS and walks a contiguous run
// of 32*RB-row tiles of A: wave (rb, cb) owns one 32x32 block (one MFMA accumulator, K in one sweep).
// The tile after next is fetched into registers while the current one is multiplied; outputs go through
// an LDS tile so that they leave as whole dwordx4 row segments instead of the dword-per-lane pattern of
// the MFMA C/D layout.  Compared with k_linear: no per-tile re-staging of W, 16-byte stores.
// ---------------------------------------------------------------------------------------------------
template <int K, int CB, int RB>
struct LinWS {
    static constexpr int TM = 32 * RB, NC = 32 * CB, NW = CB * RB, NT = 64 * NW, LDK = K + 4, LDY = NC + 4;
    static constexpr int kAPieces = TM * (K / 4) / NT, kWPieces = NC * (K / 4) / NT, kYPieces = TM * (NC / 4) / NT;
    static constexpr int kLdsFloats = (NC + TM) * LDK + TM * LDY;
};

template <int K, int CB, int RB>
__global__ __launch_bounds__(64 * CB * RB) void k_linear_ws(const float* __restrict__ A, int64_t M, int lda,
                                                            const float* __restrict__ W, int ldw,
                                                            const float* __restrict__ bias, float* __restrict__ C, int ldc,
                                                            int num_tiles, int tiles_per_group, int groups, int accumulate) {
    using P = LinWS<K, CB, RB>;
    constexpr int TM = P::TM, NC = P::NC, NT = P::NT, LDK = P::LDK, LDY = P::LDY, NA = P::kAPieces, NY = P::kYPieces;
    __shared__ __attribute__((aligned(16))) float lds[P::kLdsFloats];
    float* Ws = lds;
    float* As = lds + NC * LDK;
    float* Ys = As + TM * LDK;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rb = wave % RB, cb = wave / RB, cl = lane & 31, half = lane >> 5;
    const int col = 32 * cb + cl;
    const int group = blockIdx.x, ck = blockIdx.y;  // ck: which 32*CB-column chunk of the output
    const int t0 = group * tiles_per_group;
    const int t_end = min(num_tiles, t0 + tiles_per_group);
    if (t0 >= t_end) return;
    const int col0 = ck * NC;

#pragma unroll
    for (int it = 0; it < P::kWPieces; ++it) {
        const int f = tid + NT * it, row = f / (K / 4), c4 = f % (K / 4);
        *reinterpret_cast<f32x4*>(Ws + row * LDK + 4 * c4) =
            *reinterpret_cast<const f32x4*>(W + (int64_t)(col0 + row) * ldw + 4 * c4);
    }
    const float bv = bias != nullptr ? bias[col0 + col] : 0.f;

    auto tile_valid = [&](int t) { return (int)min((int64_t)TM, M - (int64_t)t * TM); };
    auto load_a = [&](int t, f32x4 (&r)[NA]) {
        const int64_t row0 = (int64_t)t * TM;
        const int valid = tile_valid(t);
#pragma unroll
        for (int it = 0; it < NA; ++it) {
            const int f = tid + NT * it, row = min(f / (K / 4), valid - 1), c4 = f % (K / 4);
            r[it] = *reinterpret_cast<const f32x4*>(A + (row0 + row) * lda + 4 * c4);
        }
    };
    auto put_a = [&](const f32x4 (&r)[NA]) {
#pragma unroll
        for (int it = 0; it < NA; ++it) {
            const int f = tid + NT * it, row = f / (K / 4), c4 = f % (K / 4);
            *reinterpret_cast<f32x4*>(As + row * LDK + 4 * c4) = r[it];
        }
    };

    f32x4 stage[NA];
    load_a(t0, stage);
    put_a(stage);
    if (t0 + 1 < t_end) load_a(t0 + 1, stage);
    __syncthreads();

    for (int t = t0; t < t_end; ++t) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bv;
        const float* ap = As + (32 * rb + cl) * LDK + 4 * half;
        const float* wp = Ws + col * LDK + 4 * half;
#pragma unroll
        for (int q = 0; q < K / 8; ++q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(ap + 8 * q);
            const f32x4 b = *reinterpret_cast<const f32x4*>(wp + 8 * q);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[k], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) Ys[(32 * rb + cd_row(r, lane)) * LDY + col] = acc[r];
        // claim the prefetched tile before this iteration's stores are issued (see k_edge_gate_staged)
#pragma unroll
        for (int it = 0; it < NA; ++it) asm volatile("" : "+v"(stage[it]));
        __syncthreads();

        {
            const int valid = tile_valid(t);
            float* out = C + (int64_t)t * TM * ldc + col0;
#pragma unroll
            for (int it = 0; it < NY; ++it) {
                const int f = tid + NT * it, row = f / (NC / 4), c4 = f % (NC / 4);
                f32x4 y = *reinterpret_cast<const f32x4*>(Ys + row * LDY + 4 * c4);
                if (row < valid) {
                    f32x4* dst = reinterpret_cast<f32x4*>(out + (int64_t)row * ldc + 4 * c4);
                    if (accumulate) y += *dst;
                    *dst = y;
                }
            }
        }
        if (t + 1 < t_end) {
            put_a(stage);
            if (t + 2 < t_end) load_a(t + 2, stage);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// The same weight-stationary walk with the product on the bf16 matrix cores as an fp32-faithful three-way split
// (edge_gate_bf.hip explains the arithmetic: x = x1 + x2 + x3 exactly, six of the nine partial products, what is
// dropped is of the size of one fp32 rounding).  The W chunk is split ONCE per workgroup into three bf16 planes in
// LDS; every wave splits its A fragment in registers (~44 VALU operations per K = 16) - with two waves per SIMD one
// wave's split runs under the other's MFMAs.  6 x 32 cycles per K = 16 instead of 8 x 64.
// ---------------------------------------------------------------------------------------------------
typedef __bf16 lin_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void lin_split8(const f32x4 lo4, const f32x4 hi4, uint4& p1, uint4& p2, uint4& p3) {
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = j < 4 ? lo4[j] : hi4[j - 4];
        h[j] = __float_as_uint(x) & 0xFFFF0000u;
        const float r = x - __uint_as_float(h[j]);           // exact
        m[j] = __float_as_uint(r) & 0xFFFF0000u;
        l[j] = __float_as_uint(r - __uint_as_float(m[j]));   // exact, a bf16
    }
    p1 = make_uint4(__builtin_amdgcn_perm(h[1], h[0], 0x07060302u), __builtin_amdgcn_perm(h[3], h[2], 0x07060302u),
                    __builtin_amdgcn_perm(h[5], h[4], 0x07060302u), __builtin_amdgcn_perm(h[7], h[6], 0x07060302u));
    p2 = make_uint4(__builtin_amdgcn_perm(m[1], m[0], 0x07060302u), __builtin_amdgcn_perm(m[3], m[2], 0x07060302u),
                    __builtin_amdgcn_perm(m[5], m[4], 0x07060302u), __builtin_amdgcn_perm(m[7], m[6], 0x07060302u));
    p3 = make_uint4(__builtin_amdgcn_perm(l[1], l[0], 0x07060302u), __builtin_amdgcn_perm(l[3], l[2], 0x07060302u),
                    __builtin_amdgcn_perm(l[5], l[4], 0x07060302u), __builtin_amdgcn_perm(l[7], l[6], 0x07060302u));
}

template <int K, int CB, int RB>
struct LinBF {
    static constexpr int TM = 32 * RB, NC = 32 * CB, NW = CB * RB, NT = 64 * NW, LDK = K + 4, LDY = NC + 4, PLD = 2 * K + 16;
    static constexpr int kAPieces = TM * (K / 4) / NT, kWPieces = NC * (K / 8) / NT, kYPieces = TM * (NC / 4) / NT;
    static constexpr int kPlaneBytes = NC * PLD;
    static constexpr int kLdsBytes = 3 * kPlaneBytes + 4 * (TM * LDK + TM * LDY);
    static_assert(NC * (K / 8) % NT == 0 && TM * (K / 4) % NT == 0 && TM * (NC / 4) % NT == 0, "piece counts");
};

template <int K, int CB, int RB>
__global__ __launch_bounds__(64 * CB * RB) void k_linear_bf(const float* __restrict__ A, int64_t M, int lda,
                                                            const float* __restrict__ W, int ldw,
                                                            const float* __restrict__ bias, float* __restrict__ C, int ldc,
                                                            int num_tiles, int tiles_per_group, int groups, int accumulate) {
    using P = LinBF<K, CB, RB>;
    constexpr int TM = P::TM, NC = P::NC, NT = P::NT, LDK = P::LDK, LDY = P::LDY, NA = P::kAPieces, NY = P::kYPieces, PLD = P::PLD,
                  PB = P::kPlaneBytes, KS = K / 16;
    __shared__ __attribute__((aligned(16))) unsigned char lds[P::kLdsBytes];
    unsigned char* Wp = lds;                                             // three planes [NC][PLD]
    float* As = reinterpret_cast<float*>(lds + 3 * PB);                  // [TM][LDK] fp32
    float* Ys = As + TM * LDK;                                           // [TM][LDY]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rb = wave % RB, cb = wave / RB, cl = lane & 31, half = lane >> 5;
    const int col = 32 * cb + cl;
    const int group = blockIdx.x, ck = blockIdx.y;
    const int t0 = group * tiles_per_group;
    const int t_end = min(num_tiles, t0 + tiles_per_group);
    if (t0 >= t_end) return;
    const int col0 = ck * NC;

#pragma unroll
    for (int it = 0; it < P::kWPieces; ++it) {   // eight consecutive k of one W row per piece
        const int f = tid + NT * it, row = f / (K / 8), c8 = f % (K / 8);
        const float* src = W + (int64_t)(col0 + row) * ldw + 8 * c8;
        uint4 p1, p2, p3;
        lin_split8(*reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + 4), p1, p2, p3);
        unsigned char* dst = Wp + row * PLD + 16 * c8;
        *reinterpret_cast<uint4*>(dst) = p1;
        *reinterpret_cast<uint4*>(dst + PB) = p2;
        *reinterpret_cast<uint4*>(dst + 2 * PB) = p3;
    }
    const float bv = bias != nullptr ? bias[col0 + col] : 0.f;

    auto tile_valid = [&](int t) { return (int)min((int64_t)TM, M - (int64_t)t * TM); };
    auto load_a = [&](int t, f32x4 (&r)[NA]) {
        const int64_t row0 = (int64_t)t * TM;
        const int valid = tile_valid(t);
#pragma unroll
        for (int it = 0; it < NA; ++it) {
            const int f = tid + NT * it, row = min(f / (K / 4), valid - 1), c4 = f % (K / 4);
            r[it] = *reinterpret_cast<const f32x4*>(A + (row0 + row) * lda + 4 * c4);
        }
    };
    auto put_a = [&](const f32x4 (&r)[NA]) {
#pragma unroll
        for (int it = 0; it < NA; ++it) {
            const int f = tid + NT * it, row = f / (K / 4), c4 = f % (K / 4);
            *reinterpret_cast<f32x4*>(As + row * LDK + 4 * c4) = r[it];
        }
    };

    f32x4 stage[NA];
    load_a(t0, stage);
    put_a(stage);
    if (t0 + 1 < t_end) load_a(t0 + 1, stage);
    __syncthreads();

    for (int t = t0; t < t_end; ++t) {
        f32x16 acc, acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[r] = bv;
            acc2[r] = 0.f;
        }
        const float* ap = As + (32 * rb + cl) * LDK + 8 * half;         // + 16 q
        const unsigned char* wp = Wp + col * PLD + 16 * half;           // + 32 q, + plane * PB
        f32x4 x0 = *reinterpret_cast<const f32x4*>(ap), x1 = *reinterpret_cast<const f32x4*>(ap + 4);
#pragma unroll
        for (int q = 0; q < KS; ++q) {
            const int qn = q + 1 < KS ? q + 1 : q;
            const f32x4 n0 = *reinterpret_cast<const f32x4*>(ap + 16 * qn), n1 = *reinterpret_cast<const f32x4*>(ap + 16 * qn + 4);
            const uint4 w1 = *reinterpret_cast<const uint4*>(wp + 32 * q), w2 = *reinterpret_cast<const uint4*>(wp + 32 * q + PB),
                        w3 = *reinterpret_cast<const uint4*>(wp + 32 * q + 2 * PB);
            uint4 a1, a2, a3;
            lin_split8(x0, x1, a1, a2, a3);
            auto bf = [](const uint4 v) { return __builtin_bit_cast(lin_bf16x8, v); };
            // the 2^-16 terms into one chain, the leading ones into the other
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a3), bf(w1), acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a2), bf(w1), acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1), bf(w3), acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1), bf(w2), acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a2), bf(w2), acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1), bf(w1), acc, 0, 0, 0);
            x0 = n0;
            x1 = n1;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) Ys[(32 * rb + cd_row(r, lane)) * LDY + col] = acc[r] + acc2[r];
        // claim the prefetched tile before this iteration's stores are issued (see k_edge_gate_staged)
#pragma unroll
        for (int it = 0; it < NA; ++it) asm volatile("" : "+v"(stage[it]));
        __syncthreads();

        {
            const int valid = tile_valid(t);
            float* out = C + (int64_t)t * TM * ldc + col0;
#pragma unroll
            for (int it = 0; it < NY; ++it) {
                const int f = tid + NT * it, row = f / (NC / 4), c4 = f % (NC / 4);
                f32x4 y = *reinterpret_cast<const f32x4*>(Ys + row * LDY + 4 * c4);
                if (row < valid) {
                    f32x4* dst = reinterpret_cast<f32x4*>(out + (int64_t)row * ldc + 4 * c4);
                    if (accumulate) y += *dst;
                    *dst = y;
                }
            }
        }
        if (t + 1 < t_end) {
            put_a(stage);
            if (t + 2 < t_end) load_a(t + 2, stage);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// Barrier-free streaming form of the above (the shipped default).  The W chunk (64 columns, three bf16 planes) is all
// that lives in LDS; every wave owns 32 rows x 64 columns of the output and streams its A rows STRAIGHT FROM GLOBAL
// MEMORY INTO MFMA FRAGMENTS - a lane's share of a 32x32x16 operand is 8 consecutive k of one row, 32 contiguous
// bytes - so there is no A staging, no per-tile barrier and no output detour through LDS (a 32-lane half wave stores
// 128 contiguous bytes of one output row).  Latency is hidden by occupancy: ~150 VGPRs and 52 KB of LDS allow three
// 4-wave workgroups (12 waves) per CU.
// ---------------------------------------------------------------------------------------------------
template <int K, int NW_ = 4>
struct LinBF2 {
    static constexpr int NW = NW_, NT = 64 * NW, TM = 32 * NW, NC = 64, PLD = 2 * K + 16, kPlaneBytes = NC * PLD;
    static constexpr int kWPieces = NC * (K / 8) / NT;
    static_assert(NC * (K / 8) % NT == 0, "piece count");
};

// ROWST: the output leaves through a wave-private 32 x 32 LDS tile and is stored ROW-major, 16 bytes per lane (eight lanes to a
// 128-byte row segment): 8 vector-memory stores per 32 x 64 block instead of 32 dword ones.  The projection writes 5 x the
// bytes it reads, and a CU's vector-memory queue takes instructions, not bytes (round 3: edge_gate_pl256.hip's header) - at
// N = 100k the 4-byte form issues 1M store instructions next to 0.5M loads.  Costs 18 KB of LDS per workgroup (two per CU
// instead of three at K = 128).  C rows must be 16-byte aligned (ldc % 4 == 0).
template <int K, int NW = 4, int ROWST = 0>
__global__ __launch_bounds__(64 * NW) void k_linear_bf2(const float* __restrict__ A, int64_t M, int lda, const float* __restrict__ W,
                                                    int ldw, const float* __restrict__ bias, float* __restrict__ C, int ldc,
                                                    int num_tiles, int tiles_per_group, int accumulate) {
    using P = LinBF2<K, NW>;
    constexpr int PLD = P::PLD, PB = P::kPlaneBytes, KS = K / 16, HS = 2;   // fragments are fetched two K = 16 steps at a time
    constexpr int TLD = 36;   // floats per row of the epilogue tile: 16-byte aligned rows, the two half waves' rows 4 apart
    __shared__ __attribute__((aligned(16))) unsigned char Wp[3 * PB];
    __shared__ __attribute__((aligned(16))) float Tt[ROWST == 1 ? NW * 32 * TLD : 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cl = lane & 31, half = lane >> 5;
    const int t0 = blockIdx.x * tiles_per_group, t_end = min(num_tiles, t0 + tiles_per_group);
    if (t0 >= t_end) return;
    const int col0 = blockIdx.y * P::NC;
#pragma unroll
    for (int it = 0; it < P::kWPieces; ++it) {   // split the W chunk once: eight consecutive k of one W row per piece
        const int f = tid + P::NT * it, row = f / (K / 8), c8 = f % (K / 8);
        const float* src = W + (int64_t)(col0 + row) * ldw + 8 * c8;
        uint4 p1, p2, p3;
        lin_split8(*reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + 4), p1, p2, p3);
        unsigned char* dst = Wp + row * PLD + 16 * c8;
        *reinterpret_cast<uint4*>(dst) = p1;
        *reinterpret_cast<uint4*>(dst + PB) = p2;
        *reinterpret_cast<uint4*>(dst + 2 * PB) = p3;
    }
    const float b0 = bias != nullptr ? bias[col0 + cl] : 0.f, b1 = bias != nullptr ? bias[col0 + 32 + cl] : 0.f;
    __syncthreads();   // the only barrier

    auto bf = [](const uint4 v) { return __builtin_bit_cast(lin_bf16x8, v); };
    const unsigned char* wp = Wp + cl * PLD + 16 * half;   // + 32 * 32 rows for the second column block, + 32 q, + plane * PB
    for (int t = t0; t < t_end; ++t) {
        const int64_t row0 = (int64_t)t * P::TM + 32 * wave;
        if (row0 >= M) continue;
        const int64_t arow = min(row0 + cl, M - 1);   // rows past the end read the last row (never stored)
        const float* ap = A + arow * lda + 8 * half;  // + 16 q
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc0[r] = b0;
            acc1[r] = b1;
        }
        // fragments are fetched HS K = 16 steps at a time, one batch AHEAD of the MFMAs that consume them
        f32x4 x[HS][2];
#pragma unroll
        for (int q = 0; q < HS; ++q) {
            x[q][0] = *reinterpret_cast<const f32x4*>(ap + 16 * q);
            x[q][1] = *reinterpret_cast<const f32x4*>(ap + 16 * q + 4);
        }
#pragma unroll
        for (int hq = 0; hq < KS; hq += HS) {
            f32x4 nx[HS][2];
            const int hn = hq + HS < KS ? hq + HS : hq;
#pragma unroll
            for (int q = 0; q < HS; ++q) {
                nx[q][0] = *reinterpret_cast<const f32x4*>(ap + 16 * (hn + q));
                nx[q][1] = *reinterpret_cast<const f32x4*>(ap + 16 * (hn + q) + 4);
            }
            __builtin_amdgcn_sched_barrier(0);   // the next batch's loads are in flight before this batch's MFMAs start
#pragma unroll
            for (int q = 0; q < HS; ++q) {
                uint4 a1, a2, a3;
                lin_split8(x[q][0], x[q][1], a1, a2, a3);
                const unsigned char* w = wp + 32 * (hq + q);
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const uint4 w1 = *reinterpret_cast<const uint4*>(w + cb * 32 * PLD), w2 = *reinterpret_cast<const uint4*>(w + cb * 32 * PLD + PB),
                                w3 = *reinterpret_cast<const uint4*>(w + cb * 32 * PLD + 2 * PB);
                    f32x16 c = cb == 0 ? acc0 : acc1;   // smallest terms first
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a3), bf(w1), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1), bf(w3), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a2), bf(w2), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a2), bf(w1), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1), bf(w2), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1), bf(w1), c, 0, 0, 0);
                    if (cb == 0) {
                        acc0 = c;
                    } else {
                        acc1 = c;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < HS; ++q) {
                x[q][0] = nx[q][0];
                x[q][1] = nx[q][1];
            }
        }
        if (ROWST == 2) {
            // the same row-major stores WITHOUT LDS: 4 x 4 transposes inside lane quads (two butterfly stages of v_mov_dpp quad_perm
            // + v_cndmask, 16 VALU operations per four registers): lane (quad q, j) ends up with row 8 g + 4 half + j, columns 4 q .. + 3
            const int j = lane & 3, q4 = 4 * ((lane & 31) >> 2);
            const bool odd = j & 1, hi = j & 2;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = cb == 0 ? acc0[4 * g + i] : acc1[4 * g + i];
#pragma unroll
                    for (int i = 0; i < 4; i += 2) {   // exchange with lane ^ 1
                        const float send = odd ? v[i] : v[i + 1];
                        const float recv = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, true));
                        v[i] = odd ? recv : v[i];
                        v[i + 1] = odd ? v[i + 1] : recv;
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i) {      // exchange with lane ^ 2
                        const float send = hi ? v[i] : v[i + 2];
                        const float recv = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), 0x4E, 0xF, 0xF, true));
                        v[i] = hi ? recv : v[i];
                        v[i + 2] = hi ? v[i + 2] : recv;
                    }
                    const int64_t row = row0 + 8 * g + 4 * half + j;
                    if (row < M) {
                        f32x4* o = reinterpret_cast<f32x4*>(C + row * ldc + col0 + 32 * cb + q4);
                        const f32x4 y = {v[0], v[1], v[2], v[3]};
                        *o = accumulate ? *o + y : y;
                    }
                }
            }
            continue;
        }
        if (ROWST == 1) {
            float* tile = Tt + wave * 32 * TLD;
            const int er = lane >> 3, ec = 4 * (lane & 7);   // row-major: lane -> rows er + 8 it, columns ec .. ec + 3
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) tile[cd_row(r, lane) * TLD + cl] = cb == 0 ? acc0[r] : acc1[r];
                __builtin_amdgcn_wave_barrier();   // the tile is this wave's own: LDS keeps a wave's accesses in order
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(tile + (er + 8 * it) * TLD + ec);
                    if (row0 + er + 8 * it < M) {
                        f32x4* o = reinterpret_cast<f32x4*>(C + (row0 + er + 8 * it) * ldc + col0 + 32 * cb + ec);
                        *o = accumulate ? *o + v : v;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            continue;
        }
        float* out = C + row0 * ldc + col0 + cl;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lr = cd_row(r, lane);
            if (row0 + lr < M) {
                float* o = out + (int64_t)lr * ldc;
                if (accumulate) {
                    o[0] += acc0[r];
                    o[32] += acc1[r];
                } else {
                    o[0] = acc0[r];
                    o[32] = acc1[r];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// A-stationary form for wide outputs (the node projection [N,H] -> [N,5H]).  The streaming kernel above keeps a W chunk per
// workgroup and re-reads - and re-splits - the A rows for every 64-column chunk (ten times at Nout = 5H, in 32-byte pieces that
// thrash L1).  Here a wave loads its 32 rows x K ONCE, splits them ONCE into registers (3 K / 16 x 4 VGPRs) and walks over ALL
// column chunks; the W chunks (L2-resident, Nout x K x 4 bytes) stream through one LDS buffer, split by the threads that
// stage them, the next chunk's rows fetched into registers while the MFMAs of the current one run.  Two barriers per chunk.
// ---------------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256, 2) void k_linear_as(const float* __restrict__ A, int64_t M, int lda, const float* __restrict__ W,
                                                      int ldw, const float* __restrict__ bias, int Nout, float* __restrict__ C, int ldc,
                                                      int accumulate) {
    using P = LinBF2<K, 4>;
    constexpr int PLD = P::PLD, PB = P::kPlaneBytes, KS = K / 16, NP = P::kWPieces;
    __shared__ __attribute__((aligned(16))) unsigned char Wp[3 * PB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cl = lane & 31, half = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * P::TM + 32 * wave;
    const bool live = row0 < M;   // a wave past the end still stages W and meets the barriers
    auto bf = [](const uint4 v) { return __builtin_bit_cast(lin_bf16x8, v); };

    uint4 a1[KS], a2[KS], a3[KS];
    {
        const int64_t arow = min(row0 + cl, M - 1);   // rows past the end read the last row (never stored)
        const float* ap = A + arow * lda + 8 * half;
        f32x4 x[KS][2];
#pragma unroll
        for (int q = 0; q < KS; ++q) {
            x[q][0] = *reinterpret_cast<const f32x4*>(ap + 16 * q);
            x[q][1] = *reinterpret_cast<const f32x4*>(ap + 16 * q + 4);
        }
#pragma unroll
        for (int q = 0; q < KS; ++q) lin_split8(x[q][0], x[q][1], a1[q], a2[q], a3[q]);
    }
    f32x4 wr[NP][2];
    auto fetch_w = [&](int chunk) {   // eight consecutive k of one W row per piece
#pragma unroll
        for (int it = 0; it < NP; ++it) {
            const int f = tid + 256 * it, row = f / (K / 8), c8 = f % (K / 8);
            const float* src = W + (int64_t)(chunk * P::NC + row) * ldw + 8 * c8;
            wr[it][0] = *reinterpret_cast<const f32x4*>(src);
            wr[it][1] = *reinterpret_cast<const f32x4*>(src + 4);
        }
    };
    const int n_chunks = Nout / P::NC;
    const unsigned char* wp = Wp + cl * PLD + 16 * half;
    fetch_w(0);
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        if (chunk > 0) __syncthreads();   // every wave is done with the previous chunk's planes
#pragma unroll
        for (int it = 0; it < NP; ++it) {
            const int f = tid + 256 * it, row = f / (K / 8), c8 = f % (K / 8);
            uint4 p1, p2, p3;
            lin_split8(wr[it][0], wr[it][1], p1, p2, p3);
            unsigned char* dst = Wp + row * PLD + 16 * c8;
            *reinterpret_cast<uint4*>(dst) = p1;
            *reinterpret_cast<uint4*>(dst + PB) = p2;
            *reinterpret_cast<uint4*>(dst + 2 * PB) = p3;
        }
        __syncthreads();
        if (chunk + 1 < n_chunks) fetch_w(chunk + 1);
        if (!live) continue;
        const int col0 = chunk * P::NC;
        const float b0 = bias != nullptr ? bias[col0 + cl] : 0.f, b1 = bias != nullptr ? bias[col0 + 32 + cl] : 0.f;
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc0[r] = b0;
            acc1[r] = b1;
        }
#pragma unroll
        for (int q = 0; q < KS; ++q) {
            const unsigned char* w = wp + 32 * q;
            const uint4 u1 = *reinterpret_cast<const uint4*>(w), u2 = *reinterpret_cast<const uint4*>(w + PB),
                        u3 = *reinterpret_cast<const uint4*>(w + 2 * PB);
            const uint4 v1 = *reinterpret_cast<const uint4*>(w + 32 * PLD), v2 = *reinterpret_cast<const uint4*>(w + 32 * PLD + PB),
                        v3 = *reinterpret_cast<const uint4*>(w + 32 * PLD + 2 * PB);
            // smallest terms first; the two column blocks alternate, so consecutive MFMAs never wait on each other
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a3[q]), bf(u1), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a3[q]), bf(v1), acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1[q]), bf(u3), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1[q]), bf(v3), acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a2[q]), bf(u2), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a2[q]), bf(v2), acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a2[q]), bf(u1), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a2[q]), bf(v1), acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1[q]), bf(u2), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1[q]), bf(v2), acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1[q]), bf(u1), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1[q]), bf(v1), acc1, 0, 0, 0);
        }
        float* out = C + row0 * ldc + col0 + cl;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lr = cd_row(r, lane);
            if (row0 + lr < M) {
                float* o = out + (int64_t)lr * ldc;
                if (accumulate) {
                    o[0] += acc0[r];
                    o[32] += acc1[r];
                } else {
                    o[0] = acc0[r];
                    o[32] = acc1[r];
                }
            }
        }
    }
}

constexpr int64_t kAStationaryRows = 400000;   // ~6 row blocks of 128 per workgroup slot (2 per CU)

template <int K>
static int launch_linear_as(const float* A, int64_t M, int lda, const float* W, int ldw, const float* bias, int Nout, float* C,
                            int ldc, hipStream_t s, int accumulate) {
    const int64_t blocks = (M + 127) / 128;
    GN_REQUIRE(blocks < (1ll << 31), "linear: too many tiles");
    hipLaunchKernelGGL(k_linear_as<K>, dim3((unsigned)blocks), dim3(256), 0, s, A, M, lda, W, ldw, bias, Nout, C, ldc, accumulate);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

template <int K, int NW = 4, int WGS_PER_CU = 3, int ROWST = 0>
static int launch_linear_bf2(const float* A, int64_t M, int lda, const float* W, int ldw, const float* bias, int Nout, float* C,
                             int ldc, hipStream_t s, int accumulate) {
    using P = LinBF2<K, NW>;
    const int n_chunks = Nout / P::NC;
    const int64_t tiles = (M + P::TM - 1) / P::TM;
    GN_REQUIRE(tiles < (1ll << 31), "linear: too many tiles");
    int groups = WGS_PER_CU * persistent_grid() / n_chunks;   // resident workgroups per CU (LDS: 52 KB of W planes at K = 128, 101 KB at 256); all
                                                   // chunks of a group share an XCD (see launch_linear_ws)
    if (groups >= kXcds) groups -= groups % kXcds;
    if (groups < 1) groups = 1;
    if (groups > tiles) groups = (int)tiles;
    const int tpg = (int)((tiles + groups - 1) / groups);
    hipLaunchKernelGGL((k_linear_bf2<K, NW, ROWST>), dim3(groups, n_chunks), dim3(P::NT), 0, s, A, M, lda, W, ldw, bias, C, ldc, (int)tiles, tpg,
                       accumulate);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

template <int K, int CB, int RB>
static int launch_linear_bf(const float* A, int64_t M, int lda, const float* W, int ldw, const float* bias, int Nout,
                            float* C, int ldc, hipStream_t s, int accumulate) {
    using P = LinBF<K, CB, RB>;
    const int n_chunks = Nout / P::NC;
    const int64_t tiles = (M + P::TM - 1) / P::TM;
    GN_REQUIRE(tiles < (1ll << 31), "linear: too many tiles");
    int groups = persistent_grid() / n_chunks;   // see launch_linear_ws
    if (groups >= kXcds) groups -= groups % kXcds;
    if (groups < 1) groups = 1;
    const int tpg = (int)((tiles + groups - 1) / groups);
    hipLaunchKernelGGL((k_linear_bf<K, CB, RB>), dim3(groups, n_chunks), dim3(P::NT), 0, s, A, M, lda, W, ldw, bias, C, ldc,
                       (int)tiles, tpg, groups, accumulate);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

template <int K, int CB, int RB>
static int launch_linear_ws(const float* A, int64_t M, int lda, const float* W, int ldw, const float* bias, int Nout,
                            float* C, int ldc, hipStream_t s, int accumulate) {
    using P = LinWS<K, CB, RB>;
    const int n_chunks = Nout / P::NC;
    const int64_t tiles = (M + P::TM - 1) / P::TM;
    GN_REQUIRE(tiles < (1ll << 31), "linear: too many tiles");
    // linear block id = chunk * groups + group and block b runs on XCD b % 8: with groups a multiple of 8 all
    // chunks of one group (which read the same A tiles) share one XCD's L2
    int groups = persistent_grid() / n_chunks;
    if (groups >= kXcds) groups -= groups % kXcds;
    if (groups < 1) groups = 1;
    const int tpg = (int)((tiles + groups - 1) / groups);
    hipLaunchKernelGGL((k_linear_ws<K, CB, RB>), dim3(groups, n_chunks), dim3(P::NT), 0, s, A, M, lda, W, ldw, bias, C, ldc,
                       (int)tiles, tpg, groups, accumulate);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

template <int NB>
static int launch_linear(const float* A, int64_t M, int K, int lda, const float* W, int ldw, const float* bias, int Nout,
                         float* C, int ldc, hipStream_t s, int accumulate) {
    const int n_tiles = (Nout + 32 * NB - 1) / (32 * NB);
    const int64_t m_tiles = (M + kTileM - 1) / kTileM;
    const int64_t total = m_tiles * n_tiles;
    GN_REQUIRE(total < (1ll << 31), "linear: too many tiles");
    hipLaunchKernelGGL(k_linear<NB>, dim3((unsigned)total), dim3(kGemmThreads), 0, s, A, M, K, lda, W, ldw, bias, Nout, C,
                       ldc, n_tiles, (int)total, accumulate);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

// k_linear with A as column blocks (tile_gemm_blocks), 128 output columns per tile, C (+)= A W^T
__global__ __launch_bounds__(kGemmThreads) void k_linear_blocks(ABlocks ab, int64_t M, int K, int lda, const float* __restrict__ W,
                                                                int ldw, int Nout, float* __restrict__ C, int ldc, int n_tiles,
                                                                int total_tiles, int accumulate) {
    constexpr int NB = 4;
    __shared__ __attribute__((aligned(16))) float lds[tile_lds_floats<NB>()];
    float* As = lds;
    float* Ws = lds + kTileM * kLdk;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tile = xcd_remap(blockIdx.x, total_tiles);
    const int64_t row0 = (int64_t)(tile / n_tiles) * kTileM;
    const int col0 = (tile % n_tiles) * 32 * NB;
    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    tile_gemm_blocks<NB>(acc, ab, row0, M, lda, W, col0, Nout, ldw, K, As, Ws, tid);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int col = col0 + 32 * nb + (lane & 31);
        if (col < Nout) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = row0 + 32 * wave + cd_row(r, lane);
                if (row < M) C[row * ldc + col] = accumulate ? C[row * ldc + col] + acc[nb][r] : acc[nb][r];
            }
        }
    }
}

}  // namespace gnnome

static int linear_impl(const float* A, int64_t M, int K, int lda, const float* W, int ldw, const float* bias, int Nout,
                       float* C, int ldc, void* stream, int accumulate);

// C[M,Nout] (+)= [A_0 | A_1 | ...] W^T with the column blocks of A in separate buffers (all [M, block_width], row stride lda)
extern "C" int gnnome_linear_blocks_f32(const float* const* A_blocks, int num_blocks, int block_width, int64_t M, int lda, const float* W,
                                        int ldw, int Nout, float* C, int ldc, int accumulate, void* stream) {
    using namespace gnnome;
    GN_REQUIRE(M >= 0 && Nout > 0, "linear_blocks: bad shape M=%lld Nout=%d", (long long)M, Nout);
    if (M == 0) return GNNOME_OK;
    GN_REQUIRE(A_blocks && num_blocks >= 1 && num_blocks <= kABlocks && block_width > 0 && block_width % kKC == 0,
               "linear_blocks: 1..%d blocks of a width that is a multiple of %d", kABlocks, kKC);
    const int K = num_blocks * block_width;
    GN_REQUIRE(W && C && lda >= block_width && ldw >= K && ldc >= Nout && lda % 4 == 0 && ldw % 4 == 0 && (uintptr_t)W % 16 == 0,
               "linear_blocks: bad operands");
    ABlocks ab = {};
    for (int k = 0; k < num_blocks; ++k) {
        GN_REQUIRE(A_blocks[k] && (uintptr_t)A_blocks[k] % 16 == 0, "linear_blocks: block %d null or not 16-byte aligned", k);
        ab.blk[k] = A_blocks[k];
    }
    ab.width = block_width;
    if (tuning(kTuneLinearVariant) == 0 && block_width == Nout && (Nout == 128 || Nout == 256) && lda == Nout && ldc == Nout && (uintptr_t)C % 16 == 0 &&
        M >= 32768) {
        // square blocks over many rows (dh += sum_b dP_b W_b at H = 128 / 256): one residual GEMM per block on the wave-specialised edge-tile
        // kernels (W_b in registers, A_b split once per tile) - C is read and written once per block (2.5 x this kernel's bytes) and it is
        // still ahead: 0.18 -> see DESIGN 4b
        for (int k = 0; k < num_blocks; ++k) {
            const int rc = linear_impl(ab.blk[k], M, block_width, lda, W + (int64_t)k * block_width, ldw, nullptr, Nout, C, ldc, stream, k ? 1 : accumulate);
            if (rc != GNNOME_OK) return rc;
        }
        return GNNOME_OK;
    }
    const int n_tiles = (Nout + 127) / 128;
    const int64_t total = (M + kTileM - 1) / kTileM * n_tiles;
    GN_REQUIRE(total < (1ll << 31), "linear_blocks: too many tiles");
    hipLaunchKernelGGL(k_linear_blocks, dim3((unsigned)total), dim3(kGemmThreads), 0, (hipStream_t)stream, ab, M, K, lda, W, ldw, Nout, C,
                       ldc, n_tiles, (int)total, accumulate);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

static int linear_impl(const float* A, int64_t M, int K, int lda, const float* W, int ldw, const float* bias, int Nout,
                       float* C, int ldc, void* stream, int accumulate) {
    using namespace gnnome;
    GN_REQUIRE(M >= 0 && Nout > 0, "linear: bad shape M=%lld Nout=%d", (long long)M, Nout);
    if (M == 0) return GNNOME_OK;
    GN_REQUIRE(A && W && C, "linear: null pointer");
    GN_REQUIRE(K > 0 && K % kKC == 0, "linear: K=%d must be a positive multiple of %d", K, kKC);
    GN_REQUIRE(lda >= K && ldw >= K && ldc >= Nout && lda % 4 == 0 && ldw % 4 == 0, "linear: bad strides");
    GN_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)W % 16 == 0), "linear: A and W must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const bool aligned_out = ((uintptr_t)C % 16 == 0) && ldc % 4 == 0;
    if (tuning(kTuneLinearVariant) == 0 && accumulate && bias == nullptr && K == Nout && (K == 64 || K == 128) && lda == K &&
        ldc == K && aligned_out && M >= 32768)   // edge-sized square residual GEMM: the wave-specialised edge-tile kernel
        return ws_linear_acc(A, M, K, W, ldw, C, s);
    if (tuning(kTuneLinearVariant) == 0 && accumulate && bias == nullptr && K == 256 && Nout == 256 && lda == 256 && ldc == 256 && aligned_out &&
        ldw % 4 == 0 && M >= 32768 && A != C)   // the same at H = 256: the streaming edge-tile kernel (4.67 -> see DESIGN 4b)
    {
        if (tuning(kTuneGateVariant) == 9) return stream_linear_acc_256(A, M, W, ldw, C, s);   // round 2's streaming kernel
        GateBfArgs g = {};
        g.e_in = A; g.e_out = C; g.E = M; g.B1h = C; g.ldn = 256; g.W3 = W; g.ldw = ldw;
        return gate_pl256_launch(2, g, s);   // the plane form as a residual GEMM (edge_gate_pl256.hip)
    }
    if ((tuning(kTuneLinearVariant) == 0 || tuning(kTuneLinearVariant) == 10) && !accumulate && K == 256 && Nout % 128 == 0 && Nout / 128 <= 16 && lda % 4 == 0 && ldw % 4 == 0 && aligned_out &&
        M >= 1 && (const void*)A != (const void*)C) {   // at every row count, as at K = 128: a row's bits do not depend on how the caller cuts the node range
        // K = 256 (the node projection and the scorer's node halves of configs[3] / [4]): the plane-form edge-tile kernel with W in
        // registers as a plain GEMM (edge_gate_pl256.hip mode 4); the tile kernel it replaces: 1.03 ms at N = 250k, Nout = 1280
        GateBfArgs g = {};
        g.e_in = A; g.e_out = C; g.E = M; g.ldn = lda; g.ld_out = ldc; g.W3 = W; g.ldw = ldw; g.scale = bias; g.num_cblocks = Nout / 128;
        return gate_pl256_launch(4, g, s);
    }
    if ((tuning(kTuneLinearVariant) == 0 || tuning(kTuneLinearVariant) == 9 || tuning(kTuneLinearVariant) == 10) && !accumulate && K == 128 && Nout % 128 == 0 && Nout >= 256 && Nout / 128 <= 16 &&
        lda % 4 == 0 && ldw % 4 == 0 && aligned_out && (const void*)A != (const void*)C &&
        M >= 1) {   // at every row count (12 us at 100 rows like the streaming kernel, level with k_linear_as from 400k): one kernel, so a
                    // row's bits do not depend on how the caller cuts the node range (engine.aggregate_then_project, dist.py)
        // wide K = 128 products (the node projection [N,128] -> [N,640]): the plane-form edge-tile kernel as a plain GEMM (mode 4) -
        // W in the compute waves' registers, A split once per tile by the load waves, bias + 16-byte row stores in the store waves
        GateBfArgs g = {};
        g.e_in = A; g.e_out = C; g.E = M; g.ldn = lda; g.ld_out = ldc; g.W3 = W; g.ldw = ldw; g.scale = bias; g.num_cblocks = Nout / 128;
        return gate_bf_launch(128, 4, false, g, s);
    }
    if (tuning(kTuneLinearVariant) == 0 && ldw % 4 == 0 && Nout % 64 == 0 && Nout >= 256 && M >= kAStationaryRows) {
        // many row blocks and a wide output: A loaded and split once per row block (measured at Nout = 5H = 640: 1.08 against
        // 1.24 ms at M = 1M; at M = 100k - three row blocks per CU - the streaming kernel below wins, 0.120 against 0.133 ms)
        if (K == 128) return launch_linear_as<128>(A, M, lda, W, ldw, bias, Nout, C, ldc, s, accumulate);
        if (K == 64) return launch_linear_as<64>(A, M, lda, W, ldw, bias, Nout, C, ldc, s, accumulate);
    }
    if (tuning(kTuneLinearVariant) == 0 && ldw % 4 == 0 && Nout % 64 == 0) {   // the shipped default: bf16x6, barrier-free streaming
        if (K == 128) return launch_linear_bf2<128>(A, M, lda, W, ldw, bias, Nout, C, ldc, s, accumulate);
        // K = 64: row-major 16-byte stores through a wave-private LDS tile (0.0398 against 0.0471 ms at N = 100k, Nout = 320; the
        // same at K = 128 costs the third resident workgroup per CU and measures 0.122 against 0.121 - variant 7)
        if (K == 64 && aligned_out) return launch_linear_bf2<64, 4, 3, 1>(A, M, lda, W, ldw, bias, Nout, C, ldc, s, accumulate);
        if (K == 64) return launch_linear_bf2<64>(A, M, lda, W, ldw, bias, Nout, C, ldc, s, accumulate);
        // (K = 256: 101 KB of W planes leave one 8-wave workgroup per CU, too few waves to hide the fragment fetches:
        //  1.06 ms against 0.91 ms for the tile kernel at N = 250k, Nout = 1280 - measured, so K = 256 falls through)
    }
    if (tuning(kTuneLinearVariant) == 7 && ldw % 4 == 0 && Nout % 64 == 0 && aligned_out) {   // 7: the streaming kernel with row-major 16-byte stores
        if (K == 128) return launch_linear_bf2<128, 4, 2, 1>(A, M, lda, W, ldw, bias, Nout, C, ldc, s, accumulate);
        if (K == 64) return launch_linear_bf2<64, 4, 3, 1>(A, M, lda, W, ldw, bias, Nout, C, ldc, s, accumulate);
    }
    if (tuning(kTuneLinearVariant) == 8 && ldw % 4 == 0 && Nout % 64 == 0 && aligned_out) {   // 8: row-major stores by in-register quad transposes (no LDS)
        if (K == 128) return launch_linear_bf2<128, 4, 3, 2>(A, M, lda, W, ldw, bias, Nout, C, ldc, s, accumulate);
        if (K == 64) return launch_linear_bf2<64, 4, 3, 2>(A, M, lda, W, ldw, bias, Nout, C, ldc, s, accumulate);
    }
    if (tuning(kTuneLinearVariant) == 6 && ldw % 4 == 0 && Nout % 64 == 0) {   // 6: A-stationary (see k_linear_as)
        if (K == 128) return launch_linear_as<128>(A, M, lda, W, ldw, bias, Nout, C, ldc, s, accumulate);
        if (K == 64) return launch_linear_as<64>(A, M, lda, W, ldw, bias, Nout, C, ldc, s, accumulate);
    }
    if (tuning(kTuneLinearVariant) == 4 && ldw % 4 == 0 && Nout % 64 == 0) {   // 4: the default kernel as 8-wave workgroups, two per CU
        if (K == 128) return launch_linear_bf2<128, 8, 2>(A, M, lda, W, ldw, bias, Nout, C, ldc, s, accumulate);
        if (K == 64) return launch_linear_bf2<64, 8, 4>(A, M, lda, W, ldw, bias, Nout, C, ldc, s, accumulate);
    }
    if (tuning(kTuneLinearVariant) == 5 && ldw % 4 == 0 && Nout % 64 == 0) {   // 5: 2-wave workgroups, six per CU would need 6 x 52 KB: 3 fit
        if (K == 128) return launch_linear_bf2<128, 2, 3>(A, M, lda, W, ldw, bias, Nout, C, ldc, s, accumulate);
    }
    if (tuning(kTuneLinearVariant) == 3 && aligned_out && ldw % 4 == 0) {   // 3: bf16x6 with A staged through LDS
        if (K == 128 && Nout % 64 == 0) return launch_linear_bf<128, 2, 4>(A, M, lda, W, ldw, bias, Nout, C, ldc, s, accumulate);
        if (K == 64 && Nout % 64 == 0) return launch_linear_bf<64, 2, 4>(A, M, lda, W, ldw, bias, Nout, C, ldc, s, accumulate);
    }
    if (tuning(kTuneLinearVariant) != 1 && aligned_out) {   // 2: the exact-fp32-MFMA weight-stationary kernel
        if (K == 128 && Nout % 128 == 0) return launch_linear_ws<128, 4, 2>(A, M, lda, W, ldw, bias, Nout, C, ldc, s, accumulate);
        if (K == 64 && Nout % 64 == 0) return launch_linear_ws<64, 2, 4>(A, M, lda, W, ldw, bias, Nout, C, ldc, s, accumulate);
    }
    if (Nout > 64) return launch_linear<4>(A, M, K, lda, W, ldw, bias, Nout, C, ldc, s, accumulate);
    if (Nout > 32) return launch_linear<2>(A, M, K, lda, W, ldw, bias, Nout, C, ldc, s, accumulate);
    return launch_linear<1>(A, M, K, lda, W, ldw, bias, Nout, C, ldc, s, accumulate);
}

extern "C" int gnnome_linear_f32(const float* A, int64_t M, int K, int lda, const float* W, int ldw, const float* bias,
                                 int Nout, float* C, int ldc, void* stream) {
    return linear_impl(A, M, K, lda, W, ldw, bias, Nout, C, ldc, stream, 0);
}

// C += A * W^T + bias  (residual form, used by the backward: d e_in = d e' + dxe * W3, dh = dh_in + dP * Wcat)
extern "C" int gnnome_linear_acc_f32(const float* A, int64_t M, int K, int lda, const float* W, int ldw, const float* bias,
                                     int Nout, float* C, int ldc, void* stream) {
    return linear_impl(A, M, K, lda, W, ldw, bias, Nout, C, ldc, stream, 1);
}
