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
    __shared__ __attribute__((aligned(16))) float lds[(kTileM + 32 * NB) * kLdk];
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

template <int K, int CB, int RB>
static int launch_linear_ws(const float* A, int64_t M, int lda, const float* W, int ldw, const float* bias, int Nout,
                            float* C, int ldc, hipStream_t s, int accumulate) {
    using P = LinWS<K, CB, RB>;
    const int n_chunks = Nout / P::NC;
    const int64_t tiles = (M + P::TM - 1) / P::TM;
    GN_REQUIRE(tiles < (1ll << 31), "linear: too many tiles");
    // linear block id = chunk * groups + group and block b runs on XCD b % 8: with groups a multiple of 8 all
    // chunks of one group (which read the same A tiles) share one XCD's L2
    int groups = kNumCUs / n_chunks;
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

}  // namespace gnnome

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
    if (tuning(kTuneLinearVariant) != 1 && aligned_out) {
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
