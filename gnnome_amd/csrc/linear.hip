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
                                                         float* __restrict__ C, int ldc, int n_tiles, int total_tiles) {
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
                if (row < M) C[row * ldc + col] = acc[nb][r];
            }
        }
    }
}

template <int NB>
static int launch_linear(const float* A, int64_t M, int K, int lda, const float* W, int ldw, const float* bias, int Nout,
                         float* C, int ldc, hipStream_t s) {
    const int n_tiles = (Nout + 32 * NB - 1) / (32 * NB);
    const int64_t m_tiles = (M + kTileM - 1) / kTileM;
    const int64_t total = m_tiles * n_tiles;
    GN_REQUIRE(total < (1ll << 31), "linear: too many tiles");
    hipLaunchKernelGGL(k_linear<NB>, dim3((unsigned)total), dim3(kGemmThreads), 0, s, A, M, K, lda, W, ldw, bias, Nout, C,
                       ldc, n_tiles, (int)total);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

}  // namespace gnnome

extern "C" int gnnome_linear_f32(const float* A, int64_t M, int K, int lda, const float* W, int ldw, const float* bias,
                                 int Nout, float* C, int ldc, void* stream) {
    using namespace gnnome;
    GN_REQUIRE(M >= 0 && Nout > 0, "linear: bad shape M=%lld Nout=%d", (long long)M, Nout);
    if (M == 0) return GNNOME_OK;
    GN_REQUIRE(A && W && C, "linear: null pointer");
    GN_REQUIRE(K > 0 && K % kKC == 0, "linear: K=%d must be a positive multiple of %d", K, kKC);
    GN_REQUIRE(lda >= K && ldw >= K && ldc >= Nout && lda % 4 == 0 && ldw % 4 == 0, "linear: bad strides");
    GN_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)W % 16 == 0), "linear: A and W must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (Nout > 64) return launch_linear<4>(A, M, K, lda, W, ldw, bias, Nout, C, ldc, s);
    if (Nout > 32) return launch_linear<2>(A, M, K, lda, W, ldw, bias, Nout, C, ldc, s);
    return launch_linear<1>(A, M, K, lda, W, ldw, bias, Nout, C, ldc, s);
}
