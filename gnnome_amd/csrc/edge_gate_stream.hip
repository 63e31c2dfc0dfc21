// Edge gate for H = 256, where W3 fits neither the registers of four compute waves (edge_gate_bf.hip: 3 x 128 KB as bf16
// planes) nor, whole, LDS.  Same barrier-free streaming scheme as k_linear_bf2: a workgroup of 8 waves keeps the three bf16
// planes of ONE 64-column chunk of W3 in LDS (101 KB) and walks 256-edge tiles; every wave owns 32 edges x 64 output
// columns, takes its e rows straight from global memory into MFMA fragments (bf16x6 product, gemm_tile.h), and applies the
// gate epilogue in the accumulator layout: G = B1h[src] + B2h[dst] gathered as dwords (issued before the MFMA loop, added
// after it), bn + relu + residual, 128-byte row-segment stores.  The four column chunks of a row run in different
// workgroups, so the update CANNOT be in place: e_out must be a different buffer (the host ping-pongs two at H = 256).
// With K = 256 a 32 x 64 output block carries 192 MFMAs against ~160 vector-memory instructions: the dword accesses
// that sink the 128-edge tile kernel are a small part of the work here.
#include "gemm_tile.h"

namespace gnnome {

constexpr int kMaxTilesPerGroup = 64;   // tiles a workgroup walks per launch, see gate_stream_launch

template <int K>
struct GateStream {
    static constexpr int NW = 8, NT = 64 * NW, TM = 32 * NW, NC = 64, PLD = 2 * K + 16, kPlaneBytes = NC * PLD;
    static constexpr int kWPieces = NC * (K / 4) / NT;
    static constexpr int TLD = 40;   // floats per row of the epilogue tile: the two half waves' rows (4 apart) land 32 banks apart
    static_assert(NC * (K / 4) % NT == 0, "piece count");
};

// MODE 0: the gate.  MODE 2: the residual GEMM e_out += e_in W3^T (no gathers, no norm; e_out read and written by the lane that owns
// the element, so it is updated in place; e_in is a different tensor) - gnnome_linear_acc_f32 at K = Nout = 256, the backward's d e_in.
template <int K, int MODE = 0>
__global__ __launch_bounds__(512) void k_edge_gate_stream(const float* __restrict__ e_in, float* __restrict__ e_out, int64_t E,
                                                          const float* __restrict__ B1h, const float* __restrict__ B2h, int ldn,
                                                          const int32_t* __restrict__ srt_src, const int32_t* __restrict__ srt_dst,
                                                          const float* __restrict__ W3, int ldw, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int num_tiles, int tiles_per_group) {
    using P = GateStream<K>;
    constexpr int PLD = P::PLD, PB = P::kPlaneBytes, KS = K / 16, H = K, NB = KS / 4;
    __shared__ __attribute__((aligned(16))) unsigned char Wp[3 * PB];
    __shared__ __attribute__((aligned(16))) float T[P::NW][32 * P::TLD];   // per-wave epilogue tile (see below)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cl = lane & 31, half = lane >> 5;
    const int t0 = blockIdx.x * tiles_per_group, t_end = min(num_tiles, t0 + tiles_per_group);
    if (t0 >= t_end) return;
    const int col0 = blockIdx.y * P::NC;
#pragma unroll
    for (int it = 0; it < P::kWPieces; ++it) {   // split the W3 chunk once
        const int f = tid + P::NT * it, row = f / (K / 4), c4 = f % (K / 4);
        uint2 p1, p2, p3;
        tile_split4(*reinterpret_cast<const f32x4*>(W3 + (int64_t)(col0 + row) * ldw + 4 * c4), p1, p2, p3);
        unsigned char* dst = Wp + row * PLD + 8 * c4;
        *reinterpret_cast<uint2*>(dst) = p1;
        *reinterpret_cast<uint2*>(dst + PB) = p2;
        *reinterpret_cast<uint2*>(dst + 2 * PB) = p3;
    }
    __syncthreads();   // the only barrier

    auto bf = [](const uint4 v) { return __builtin_bit_cast(tile_bf16x8, v); };
    // k numbering of the matrix-core steps (the same for both operands, see the A fragments below): step 4 b + q of the lower /
    // upper half wave takes k in [64 b + 32 half + 8 q, + 8)
    const unsigned char* wp = Wp + cl * PLD + 64 * half;   // + 128 b + 16 q
    // Epilogue layout: the accumulators (a column per lane, 16 rows in registers) go through a wave-private 32 x 32 LDS tile and
    // come back ROW-major - lane l holds columns 4 (l % 8) .. + 3 of rows l / 8 + 8 it (it < 4) - so that the gathers of
    // B1h[src] / B2h[dst], the residual and the stores are 16-byte pieces, eight lanes to a 128-byte row segment: 16 + 8 + 8
    // vector-memory instructions per 32 x 64 block where the accumulator layout needed 64 + 32 + 32 dword ones.
    float* tile = &T[wave][0];
    const int er = lane >> 3, ec = 4 * (lane & 7);
    f32x4 sc[2], sh[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        if (MODE == 0) {
            sc[cb] = *reinterpret_cast<const f32x4*>(scale + col0 + 32 * cb + ec);
            sh[cb] = *reinterpret_cast<const f32x4*>(shift + col0 + 32 * cb + ec);
        }
    }
    for (int t = t0; t < t_end; ++t) {
        const int64_t row0 = (int64_t)t * P::TM + 32 * wave;
        if (row0 >= E) continue;
        const int64_t arow = min(row0 + cl, E - 1);   // rows past the end read the last row (never stored)
        // the gathers of this tile go out first; they are consumed after the MFMA loop
        const int my_s = MODE == 0 ? srt_src[arow] : 0, my_d = MODE == 0 ? srt_dst[arow] : 0;
        f32x4 b1[2][4], b2[2][4];
#pragma unroll
        for (int it = 0; MODE == 0 && it < 4; ++it) {
            const int64_t so = (int64_t)__shfl(my_s, er + 8 * it) * ldn + col0 + ec, dof = (int64_t)__shfl(my_d, er + 8 * it) * ldn + col0 + ec;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                b1[cb][it] = *reinterpret_cast<const f32x4*>(B1h + so + 32 * cb);
                b2[cb][it] = *reinterpret_cast<const f32x4*>(B2h + dof + 32 * cb);
            }
        }
        // A fragments.  The k index a lane feeds into a matrix-core step is free as long as the W operand uses the same one (the
        // product is a sum over k), so the steps are numbered such that a lane's share of FOUR consecutive steps is one whole
        // 128-byte line of its row, fetched with eight back-to-back 16-byte loads.
        const float* ap = e_in + arow * H + 32 * half;  // + 64 b
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc0[r] = 0.f;
            acc1[r] = 0.f;
        }
        f32x4 x[8];   // one batch AHEAD of the MFMAs that consume it
        f32x4 res[2][4];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = *reinterpret_cast<const f32x4*>(ap + 4 * i);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            f32x4 nx[8];
            if (b + 1 < NB) {
#pragma unroll
                for (int i = 0; i < 8; ++i) nx[i] = *reinterpret_cast<const f32x4*>(ap + 64 * (b + 1) + 4 * i);
            } else {
                // last batch: nothing left to prefetch for the product - the registers of the look-ahead batch take the
                // residual values of the epilogue instead, so that their latency runs under these 48 MFMAs
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const float* rp = (MODE == 2 ? e_out : e_in) + min(row0 + er + 8 * it, E - 1) * H + col0 + ec;   // mode 2: the old C
                    res[0][it] = *reinterpret_cast<const f32x4*>(rp);
                    res[1][it] = *reinterpret_cast<const f32x4*>(rp + 32);
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // the next batch's loads are in flight before this batch's MFMAs start
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint2 l1, l2, l3, h1, h2, h3;
                tile_split4(x[2 * q], l1, l2, l3);
                tile_split4(x[2 * q + 1], h1, h2, h3);
                const uint4 a1 = make_uint4(l1.x, l1.y, h1.x, h1.y), a2 = make_uint4(l2.x, l2.y, h2.x, h2.y),
                            a3 = make_uint4(l3.x, l3.y, h3.x, h3.y);
                const unsigned char* w = wp + 128 * b + 16 * q;
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
            if (b + 1 < NB) {
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = nx[i];
            }
        }
        // epilogue: e' = relu((e W3^T + G) * scale + shift) + e   (gated_gcn_full.py:104-110)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) tile[cd_row(r, lane) * P::TLD + cl] = cb == 0 ? acc0[r] : acc1[r];
            __builtin_amdgcn_wave_barrier();   // the tile is this wave's own: LDS keeps a wave's accesses in order
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(tile + (er + 8 * it) * P::TLD + ec);
                f32x4 y;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    y[j] = MODE == 2 ? a[j] + res[cb][it][j]
                                     : fmaxf((a[j] + (b1[cb][it][j] + b2[cb][it][j])) * sc[cb][j] + sh[cb][j], 0.f) + res[cb][it][j];
                if (row0 + er + 8 * it < E) *reinterpret_cast<f32x4*>(e_out + (row0 + er + 8 * it) * H + col0 + 32 * cb + ec) = y;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

template <int MODE>
static int stream_launch(const float* e_in, float* e_out, int64_t E, const float* B1h, const float* B2h, int ldn, const int32_t* ss,
                         const int32_t* sd, const float* W3, int ldw, const float* scale, const float* shift, hipStream_t s) {
    using P = GateStream<256>;
    const int n_chunks = 256 / P::NC;
    const int64_t tiles = (E + P::TM - 1) / P::TM;
    GN_REQUIRE(tiles < (1ll << 31), "edge_gate: too many tiles");
    int groups = persistent_grid() / n_chunks;   // one resident workgroup per CU (LDS); the four chunks of a group share an XCD's L2
    if (groups >= kXcds) groups -= groups % kXcds;
    if (groups > tiles) groups = (int)tiles;
    if (groups < 1) groups = 1;
    // The four workgroups that produce the four column chunks of the same rows read those rows at about the same time - three of
    // the four reads are L2 hits - only while they stay in step; nothing synchronises them, and over a long walk they drift
    // apart.  The launch is therefore cut into pieces of at most kMaxTilesPerGroup tiles per workgroup: every piece starts
    // the four in step again (cost: the W3 chunk is split once per piece).  Measured at E = 20M (1220 tiles per workgroup in one
    // launch): 21.95 ms uncut, 18.68 with 40 or 80 tiles per piece, 18.98 with 160, 20.25 with 320; at E = 2.5M 2.38 -> 2.36.
    const int cap = tuning(kTuneGateExperiment) > 0 ? tuning(kTuneGateExperiment) : kMaxTilesPerGroup;
    const int64_t tiles_per_launch = (int64_t)groups * cap;
    for (int64_t first = 0; first < tiles; first += tiles_per_launch) {
        const int64_t n_t = tiles - first < tiles_per_launch ? tiles - first : tiles_per_launch;
        const int64_t row_off = first * P::TM, rows = (E - row_off) < n_t * P::TM ? (E - row_off) : n_t * P::TM;
        int g = groups > n_t ? (int)n_t : groups;
        const int tpg = (int)((n_t + g - 1) / g);
        hipLaunchKernelGGL((k_edge_gate_stream<256, MODE>), dim3(g, n_chunks), dim3(P::NT), 0, s, e_in + row_off * 256, e_out + row_off * 256, rows,
                           B1h, B2h, ldn, MODE == 0 ? ss + row_off : nullptr, MODE == 0 ? sd + row_off : nullptr, W3, ldw, scale, shift, (int)n_t, tpg);
    }
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

int gate_stream_launch(const float* e_in, float* e_out, int64_t E, const float* B1h, const float* B2h, int ldn, const int32_t* ss,
                       const int32_t* sd, const float* W3, int ldw, const float* scale, const float* shift, hipStream_t s) {
    return stream_launch<0>(e_in, e_out, E, B1h, B2h, ldn, ss, sd, W3, ldw, scale, shift, s);
}

// C[M,256] += A[M,256] W^T (W = [256,256], row stride ldw; A and C dense, distinct): the streaming kernel as a residual GEMM
int stream_linear_acc_256(const float* A, int64_t M, const float* W, int ldw, float* C, hipStream_t s) {
    return stream_launch<2>(A, C, M, nullptr, nullptr, 0, nullptr, nullptr, W, ldw, nullptr, nullptr, s);
}

}  // namespace gnnome
