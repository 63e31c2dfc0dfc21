// MFMA tile GEMM building blocks shared by linear / edge_gate / edge_score (the shapes the weight-stationary and
// edge-tile kernels do not take: K = 256, odd output widths, LayerNorm, the scorer).
//
// A workgroup is 4 waves (256 threads) and owns a 128-row x (32*NB)-column output tile
// C = A[128,K] * W[32*NB,K]^T.  Wave w owns rows [32w, 32w+32) and all NB column blocks, one accumulator
// (16 VGPRs) per block.  The product runs on the bf16 matrix cores as the fp32-faithful three-way split
// ("bf16x6": x = x1 + x2 + x3 exactly, six of the nine partial products, what is dropped is of the size of one fp32
// rounding - edge_gate_bf.hip and DESIGN.md explain and measure it).
//
// K is streamed through LDS in chunks of KC = 32:
//   * A rows are staged as fp32 ([128][36]: 4-float pad, conflict-free ds_read_b128 fragments); a wave's 32 rows are its
//     own, so it splits its fragment in registers (44 VALU operations per K = 16, hidden under the 6*NB MFMAs);
//   * W rows are shared by the four waves, so they are split ONCE, by the threads that stage them, into three bf16
//     planes ([32*NB][80 bytes] each: 64 bytes of data + 16 of pad).
// Fragment mapping of v_mfma_f32_32x32x16_bf16: lane l supplies A[i = l & 31][k = 8 (l >> 5) .. + 7] and
// B[k = 8 (l >> 5) .. + 7][j = l & 31]; C/D: acc[nb][r] holds row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column
// 32 nb + (l & 31).
#pragma once
#include "common.h"

namespace gnnome {

constexpr int kTileM = 128;
constexpr int kKC = 32;
constexpr int kLdk = kKC + 4;          // A tile: padded row stride in floats
constexpr int kWRowBytes = 2 * kKC + 16;   // one bf16 plane row
constexpr int kWRowFloats = 3 * kWRowBytes / 4;   // a W row across its three planes, in floats of LDS
constexpr int kGemmThreads = 256;

// LDS of one tile GEMM: As = lds, Ws = lds + kTileM * kLdk
template <int NB>
constexpr int tile_lds_floats() { return kTileM * kLdk + 32 * NB * kWRowFloats; }

typedef __bf16 tile_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int cd_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ unsigned tile_pack_hi(unsigned odd, unsigned even) {   // (even >> 16) | (odd & 0xFFFF0000)
    return __builtin_amdgcn_perm(odd, even, 0x07060302u);
}

// exact three-way bf16 split of one float4 -> three 8-byte groups
__device__ __forceinline__ void tile_split4(const f32x4 x, uint2& p1, uint2& p2, uint2& p3) {
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h[j] = __float_as_uint(x[j]) & 0xFFFF0000u;
        const float r = x[j] - __uint_as_float(h[j]);        // exact
        m[j] = __float_as_uint(r) & 0xFFFF0000u;
        l[j] = __float_as_uint(r - __uint_as_float(m[j]));   // exact, a bf16
    }
    p1 = make_uint2(tile_pack_hi(h[1], h[0]), tile_pack_hi(h[3], h[2]));
    p2 = make_uint2(tile_pack_hi(m[1], m[0]), tile_pack_hi(m[3], m[2]));
    p3 = make_uint2(tile_pack_hi(l[1], l[0]), tile_pack_hi(l[3], l[2]));
}

// Global -> registers for one 32-wide K chunk of `ROWS` rows (ROWS % 32 == 0).  Thread t fetches the
// 16-byte piece (t & 7) of rows (t >> 3) + 32*it: 8 lanes cover one 128-byte row segment.
template <int ROWS>
struct ChunkRegs {
    static constexpr int kIters = ROWS / 32;
    f32x4 v[kIters];

    // `tile` points at the tile's first row; rows >= rows_valid (>= 1) are clamped to the last valid
    // row: their products only reach output rows / columns that are never stored.
    __device__ __forceinline__ void load(const float* __restrict__ tile, int rows_valid, int ld, int kcol, int tid) {
        const int c4 = tid & 7, r0 = tid >> 3;
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            const int row = min(r0 + 32 * it, rows_valid - 1);
            v[it] = *reinterpret_cast<const f32x4*>(tile + (uint32_t)(row * ld + kcol + 4 * c4));
        }
    }
    // A operand: fp32 rows
    __device__ __forceinline__ void store(float* lds, int tid) const {
        const int c4 = tid & 7, r0 = tid >> 3;
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            *reinterpret_cast<f32x4*>(lds + (r0 + 32 * it) * kLdk + 4 * c4) = v[it];
        }
    }
    // W operand: split here, once for the four waves -> three bf16 planes of ROWS rows each
    __device__ __forceinline__ void store_planes(float* lds, int tid) const {
        const int c4 = tid & 7, r0 = tid >> 3;
        unsigned char* base = reinterpret_cast<unsigned char*>(lds);
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            uint2 p1, p2, p3;
            tile_split4(v[it], p1, p2, p3);
            unsigned char* dst = base + (r0 + 32 * it) * kWRowBytes + 8 * c4;
            *reinterpret_cast<uint2*>(dst) = p1;
            *reinterpret_cast<uint2*>(dst + ROWS * kWRowBytes) = p2;
            *reinterpret_cast<uint2*>(dst + 2 * ROWS * kWRowBytes) = p3;
        }
    }
};

// acc[nb] += A_lds[32*wave .. +32, 0..32) * W_planes[32*nb .. +32, 0..32)^T
template <int NB>
__device__ __forceinline__ void mma_chunk(f32x16 (&acc)[NB], const float* As, const float* Ws, int wave, int lane) {
    constexpr int PB = 32 * NB * kWRowBytes;   // bytes per W plane
    const float* ap = As + (32 * wave + (lane & 31)) * kLdk + 8 * (lane >> 5);
    const unsigned char* wp = reinterpret_cast<const unsigned char*>(Ws) + (lane & 31) * kWRowBytes + 16 * (lane >> 5);
    auto bf = [](const uint4 v) { return __builtin_bit_cast(tile_bf16x8, v); };
#pragma unroll
    for (int q = 0; q < kKC / 16; ++q) {
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(ap + 16 * q), x1 = *reinterpret_cast<const f32x4*>(ap + 16 * q + 4);
        uint2 l1, l2, l3, h1, h2, h3;
        tile_split4(x0, l1, l2, l3);
        tile_split4(x1, h1, h2, h3);
        const uint4 a1 = make_uint4(l1.x, l1.y, h1.x, h1.y), a2 = make_uint4(l2.x, l2.y, h2.x, h2.y),
                    a3 = make_uint4(l3.x, l3.y, h3.x, h3.y);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const unsigned char* w = wp + 32 * nb * kWRowBytes + 32 * q;
            const uint4 w1 = *reinterpret_cast<const uint4*>(w), w2 = *reinterpret_cast<const uint4*>(w + PB),
                        w3 = *reinterpret_cast<const uint4*>(w + 2 * PB);
            // smallest terms first
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a3), bf(w1), acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1), bf(w3), acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a2), bf(w2), acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a2), bf(w1), acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1), bf(w2), acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1), bf(w1), acc[nb], 0, 0, 0);
        }
    }
}

// Full K loop with one chunk of register prefetch:  acc += A[row0.., :K] * W[wrow0.., :K]^T.
// As/Ws are the workgroup's LDS tiles (tile_lds_floats<NB>() floats from As).  Ends with a barrier, so the caller
// may reuse the LDS immediately.
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
        rw.store_planes(Ws, tid);
        __syncthreads();
        if (kc + kKC < K) {
            ra.load(a_tile, a_valid, lda, kc + kKC, tid);
            rw.load(w_tile, w_valid, ldw, kc + kKC, tid);
        }
        mma_chunk<NB>(acc, As, Ws, wave, lane);
        __syncthreads();
    }
}

// The same with A given as column blocks of equal width in separate buffers (K = blocks x width, width % kKC == 0): the K chunk
// kc comes from block kc / width.  Used by the backward's dh += [dA1 | dA2 | dA3 | dB1 | dB2] * Wcat without concatenating.
constexpr int kABlocks = 8;
struct ABlocks {
    const float* blk[kABlocks];
    int width;
};

template <int NB>
__device__ __forceinline__ void tile_gemm_blocks(f32x16 (&acc)[NB], const ABlocks& ab, int64_t row0, int64_t M, int lda,
                                                 const float* __restrict__ W, int wrow0, int Nout, int ldw, int K, float* As,
                                                 float* Ws, int tid) {
    const int wave = tid >> 6, lane = tid & 63;
    ChunkRegs<kTileM> ra;
    ChunkRegs<32 * NB> rw;
    const float* w_tile = W + (int64_t)wrow0 * ldw;
    const int a_valid = (int)min((int64_t)kTileM, M - row0);
    const int w_valid = min(32 * NB, Nout - wrow0);
    auto load_a = [&](int kc) {
        const int which = kc / ab.width;
        const float* base = ab.blk[0];
#pragma unroll
        for (int k = 1; k < kABlocks; ++k)
            if (which == k) base = ab.blk[k];
        ra.load(base + row0 * lda, a_valid, lda, kc - which * ab.width, tid);
    };
    load_a(0);
    rw.load(w_tile, w_valid, ldw, 0, tid);
    for (int kc = 0; kc < K; kc += kKC) {
        ra.store(As, tid);
        rw.store_planes(Ws, tid);
        __syncthreads();
        if (kc + kKC < K) {
            load_a(kc + kKC);
            rw.load(w_tile, w_valid, ldw, kc + kKC, tid);
        }
        mma_chunk<NB>(acc, As, Ws, wave, lane);
        __syncthreads();
    }
}

}  // namespace gnnome
