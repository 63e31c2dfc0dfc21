// Exact-fp32 MFMA tile GEMM building blocks shared by linear / edge_gate / edge_score.
//
// A workgroup is 4 waves (256 threads) and owns a 128-row x (32*NB)-column output tile
// C = A[128,K] * W[32*NB,K]^T.  Wave w owns rows [32w, 32w+32) and all NB column blocks, one
// v_mfma_f32_32x32x2_f32 accumulator (16 VGPRs) per block.  K is streamed through LDS in chunks of
// KC = 64: both operands are staged row-major with a 4-float pad (row stride 68 floats = 272 B) so
// that the ds_read_b128 fragment reads - 16-lane groups whose rows are distinct mod 16, each at the
// same column - fall on 16 distinct 16-byte slots of the 256-byte bank row (MI355X_MICROARCH.md, LDS).
//
// Fragment mapping (cdna_hip_programming.md section 3): for 32x32x2, lane l supplies A[i=l&31][k=l>>5] and
// B[k=l>>5][j=l&31].  A lane reads 4 consecutive k of its row at column 8q + 4*(l>>5) and feeds them
// to 4 consecutive MFMAs, so MFMA t of step q consumes k = 8q+t (lanes 0-31) and 8q+4+t (lanes
// 32-63): every k exactly once, A and W always paired on the same k.  The sum over k is an fmaf
// chain in that order - exact fp32, no reduced-precision path.
// C/D: acc[nb][r] holds row (r&3) + 8*(r>>2) + 4*(l>>5), column 32*nb + (l&31).
#pragma once
#include "common.h"

namespace gnnome {

constexpr int kTileM = 128;
constexpr int kKC = 64;
constexpr int kLdk = kKC + 4;  // padded LDS row stride in floats
constexpr int kGemmThreads = 256;

__device__ __forceinline__ int cd_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// Global -> registers for one 64-wide K chunk of `ROWS` rows (ROWS % 16 == 0).  Thread t fetches the
// 16-byte piece (t & 15) of rows (t >> 4) + 16*it: 16 lanes cover one 256-byte row segment.
template <int ROWS>
struct ChunkRegs {
    static constexpr int kIters = ROWS / 16;
    f32x4 v[kIters];

    // `tile` points at the tile's first row; rows >= rows_valid (>= 1) are clamped to the last valid
    // row: their products only reach output rows / columns that are never stored.
    __device__ __forceinline__ void load(const float* __restrict__ tile, int rows_valid, int ld, int kcol, int tid) {
        const int c4 = tid & 15, r0 = tid >> 4;
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            const int row = min(r0 + 16 * it, rows_valid - 1);
            v[it] = *reinterpret_cast<const f32x4*>(tile + (uint32_t)(row * ld + kcol + 4 * c4));
        }
    }
    __device__ __forceinline__ void store(float* lds, int tid) const {
        const int c4 = tid & 15, r0 = tid >> 4;
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            *reinterpret_cast<f32x4*>(lds + (r0 + 16 * it) * kLdk + 4 * c4) = v[it];
        }
    }
};

// acc[nb] += A_lds[32*wave .. +32, 0..64) * W_lds[32*nb .. +32, 0..64)^T
template <int NB>
__device__ __forceinline__ void mma_chunk(f32x16 (&acc)[NB], const float* As, const float* Ws, int wave, int lane) {
    const float* ap = As + (32 * wave + (lane & 31)) * kLdk + 4 * (lane >> 5);
    const float* wp = Ws + (lane & 31) * kLdk + 4 * (lane >> 5);
#pragma unroll
    for (int q = 0; q < kKC / 8; ++q) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(ap + 8 * q);
        f32x4 b[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) b[nb] = *reinterpret_cast<const f32x4*>(wp + 32 * nb * kLdk + 8 * q);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[nb][t], acc[nb], 0, 0, 0);
            }
        }
    }
}

// Full K loop with one chunk of register prefetch:  acc += A[row0.., :K] * W[wrow0.., :K]^T.
// As/Ws are the workgroup's LDS tiles ([128][68] and [32*NB][68]).  Ends with a barrier, so the
// caller may reuse As/Ws immediately.
template <int NB>
__device__ __forceinline__ void tile_gemm(f32x16 (&acc)[NB], const float* __restrict__ A, int64_t row0, int64_t M, int lda,
                                          const float* __restrict__ W, int wrow0, int Nout, int ldw, int K, float* As,
                                          float* Ws, int tid) {
    const int wave = tid >> 6, lane = tid & 63;
    ChunkRegs<kTileM> ra;
    ChunkRegs<32 * NB> rw;
    const float* a_tile = A + row0 * lda;
    const float* w_tile = W + (int64_t)wrow0 * ldw;
    const int a_valid = (int)min((int64_t)kTileM, M - row0);
    const int w_valid = min(32 * NB, Nout - wrow0);
    ra.load(a_tile, a_valid, lda, 0, tid);
    rw.load(w_tile, w_valid, ldw, 0, tid);
    for (int kc = 0; kc < K; kc += kKC) {
        ra.store(As, tid);
        rw.store(Ws, tid);
        __syncthreads();
        if (kc + kKC < K) {
            ra.load(a_tile, a_valid, lda, kc + kKC, tid);
            rw.load(w_tile, w_valid, ldw, kc + kKC, tid);
        }
        mma_chunk<NB>(acc, As, Ws, wave, lane);
        __syncthreads();
    }
}

}  // namespace gnnome
