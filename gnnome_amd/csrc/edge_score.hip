// gnnome_edge_score_f32: GNNome's ScorePredictor fused into one pass over e[E,H].
//
//   z1        = relu(Ps[src_p,:] + Qd[dst_p,:] + e[p,:] * W1e^T)          [hs]
//   z2        = relu(W2 * z1 + b2)                                         [32]
//   logits[.] = W3 . z2 + b3
//
// Reference lines replaced: score_predictor.py:13 (two DGL edge gathers + torch.cat into a
// materialised [E,3H] tensor), :14-16 (W1, relu, W2, relu, W3).  predictor.W1 is split by column
// block, W1 * [x_src | x_dst | e] = W1[:, :H] x_src + W1[:, H:2H] x_dst + W1[:, 2H:] e: the two node
// terms are N-row GEMMs done once per node (gnnome_linear_f32 -> Ps, Qd with b1 folded into Qd) and
// only the e term is per edge.  HBM traffic: one read of e (4*H bytes per edge) + 4 bytes out.
//
// Structure: 128-edge tile per workgroup.  Stage 1 is the shared MFMA tile GEMM with the
// accumulator pre-loaded with the gathered Ps[src] + Qd[dst]; relu(z1) is parked in LDS, stage 2
// (K = hs, 32 outputs) is a second 32x32x2 MFMA chain per wave, and the final 32-long dot product
// with W3 is a half-wave shuffle reduction.  The logit is written at the edge's ORIGINAL id, so the
// caller gets DGL edge-id order back without an un-permute pass.
#include "gemm_tile.h"

namespace gnnome {

template <int NBH /*H/32*/, int NBS /*hs/32*/>
__global__ __launch_bounds__(kGemmThreads) void k_edge_score(
    const float* __restrict__ e, int64_t E, const float* __restrict__ Ps, const float* __restrict__ Qd, int ldn,
    const int32_t* __restrict__ srt_src, const int32_t* __restrict__ srt_dst, const int32_t* __restrict__ srt_eid,
    const float* __restrict__ W1e, int ldw1, const float* __restrict__ W2, const float* __restrict__ b2,
    const float* __restrict__ W3, const float* __restrict__ b3, float* __restrict__ logits, int total_tiles,
    float* __restrict__ z1_out) {
    constexpr int H = 32 * NBH, HS = 32 * NBS, LDZ = HS + 4;
    constexpr int kGemmFloats = tile_lds_floats<NBS>();
    constexpr int kStage2Floats = (kTileM + 32) * LDZ;
    constexpr int kLdsFloats = kGemmFloats > kStage2Floats ? kGemmFloats : kStage2Floats;
    __shared__ __attribute__((aligned(16))) float lds[kLdsFloats + 2 * kTileM];
    float* As = lds;
    float* Ws = lds + kTileM * kLdk;
    float* Zs = lds;                   // stage 2 reuses the GEMM tiles
    float* W2s = lds + kTileM * LDZ;
    int* s_src = reinterpret_cast<int*>(lds + kLdsFloats);
    int* s_dst = s_src + kTileM;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, cl = lane & 31;
    const int tile = xcd_remap(blockIdx.x, total_tiles);
    const int64_t row0 = (int64_t)tile * kTileM;
    const int valid = (int)min((int64_t)kTileM, E - row0);

    if (tid < kTileM) {
        const int r = min(tid, valid - 1);
        s_src[tid] = srt_src[row0 + r];
        s_dst[tid] = srt_dst[row0 + r];
    }
    __syncthreads();

    f32x16 acc[NBS];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int lr = 32 * wave + cd_row(r, lane);
        const float* p1 = Ps + (int64_t)s_src[lr] * ldn + cl;
        const float* p2 = Qd + (int64_t)s_dst[lr] * ldn + cl;
#pragma unroll
        for (int nb = 0; nb < NBS; ++nb) acc[nb][r] = p1[32 * nb] + p2[32 * nb];
    }
    tile_gemm<NBS>(acc, e, row0, E, H, W1e, 0, HS, ldw1, H, As, Ws, tid);

    // park relu(z1) and W2 in LDS (tile_gemm ended with a barrier)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int lr = 32 * wave + cd_row(r, lane);
#pragma unroll
        for (int nb = 0; nb < NBS; ++nb) Zs[lr * LDZ + 32 * nb + cl] = fmaxf(acc[nb][r], 0.f);
    }
    for (int i = tid; i < 32 * HS; i += kGemmThreads) W2s[(i / HS) * LDZ + (i % HS)] = W2[i];
    __syncthreads();
    if (z1_out != nullptr) {  // training: keep relu(z1) for the backward of the tail
        for (int i = tid; i < kTileM * (HS / 4); i += kGemmThreads) {
            const int row = i / (HS / 4), q4 = i % (HS / 4);
            if (row < valid) *reinterpret_cast<f32x4*>(z1_out + (row0 + row) * HS + 4 * q4) = *reinterpret_cast<const f32x4*>(Zs + row * LDZ + 4 * q4);
        }
    }

    f32x16 acc2;
    const float bias2 = b2[cl];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = bias2;
    const float* zp = Zs + (32 * wave + cl) * LDZ + 4 * (lane >> 5);
    const float* wp = W2s + cl * LDZ + 4 * (lane >> 5);
#pragma unroll
    for (int q = 0; q < HS / 8; ++q) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(zp + 8 * q);
        const f32x4 b = *reinterpret_cast<const f32x4*>(wp + 8 * q);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc2, 0, 0, 0);
    }

    const float w3 = W3[cl], bias3 = b3[0];
    float mine = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float s = half_wave_sum(fmaxf(acc2[r], 0.f) * w3);
        if (cl == r) mine = s;
    }
    if (cl < 16) {
        const int lr = 32 * wave + cd_row(cl, lane);
        if (lr < valid) {
            const int64_t p = row0 + lr;
            const int64_t eid = srt_eid != nullptr ? (int64_t)srt_eid[p] : p;
            logits[eid] = mine + bias3;
        }
    }
}

template <int NBH, int NBS>
static int launch_score(const float* e, int64_t E, const float* Ps, const float* Qd, int ldn, const int32_t* ss,
                        const int32_t* sd, const int32_t* se, const float* W1e, int ldw1, const float* W2, const float* b2,
                        const float* W3, const float* b3, float* logits, hipStream_t s, float* z1_out) {
    const int64_t tiles = (E + kTileM - 1) / kTileM;
    GN_REQUIRE(tiles < (1ll << 31), "edge_score: too many tiles");
    hipLaunchKernelGGL((k_edge_score<NBH, NBS>), dim3((unsigned)tiles), dim3(kGemmThreads), 0, s, e, E, Ps, Qd, ldn, ss, sd,
                       se, W1e, ldw1, W2, b2, W3, b3, logits, (int)tiles, z1_out);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

template <int NBH>
static int dispatch_hs(int hs, const float* e, int64_t E, const float* Ps, const float* Qd, int ldn, const int32_t* ss,
                       const int32_t* sd, const int32_t* se, const float* W1e, int ldw1, const float* W2, const float* b2,
                       const float* W3, const float* b3, float* logits, hipStream_t s, float* z1_out) {
    switch (hs) {
        case 32: return launch_score<NBH, 1>(e, E, Ps, Qd, ldn, ss, sd, se, W1e, ldw1, W2, b2, W3, b3, logits, s, z1_out);
        case 64: return launch_score<NBH, 2>(e, E, Ps, Qd, ldn, ss, sd, se, W1e, ldw1, W2, b2, W3, b3, logits, s, z1_out);
        case 128: return launch_score<NBH, 4>(e, E, Ps, Qd, ldn, ss, sd, se, W1e, ldw1, W2, b2, W3, b3, logits, s, z1_out);
        default: set_error("edge_score: hidden_edge_scores=%d not in {32,64,128}", hs); return GNNOME_EINVAL;
    }
}

}  // namespace gnnome

extern "C" int gnnome_edge_score_f32(const float* e, int64_t num_edges, int hidden, int hidden_edge_scores,
                                     const float* Ps, const float* Qd, int ld_node, const int32_t* srt_src,
                                     const int32_t* srt_dst, const int32_t* srt_eid, const float* W1e, int ldw1,
                                     const float* W2, const float* b2, const float* W3, const float* b3, float* logits,
                                     float* z1_out, void* stream) {
    using namespace gnnome;
    GN_REQUIRE(num_edges >= 0, "edge_score: negative edge count");
    if (num_edges == 0) return GNNOME_OK;
    GN_REQUIRE(e && Ps && Qd && srt_src && srt_dst && W1e && W2 && b2 && W3 && b3 && logits, "edge_score: null pointer");
    GN_REQUIRE(ld_node >= hidden_edge_scores && ldw1 >= hidden && ldw1 % 4 == 0, "edge_score: bad strides");
    GN_REQUIRE(((uintptr_t)e % 16 == 0) && ((uintptr_t)W1e % 16 == 0), "edge_score: e and W1e must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    switch (hidden) {
        case 64: return dispatch_hs<2>(hidden_edge_scores, e, num_edges, Ps, Qd, ld_node, srt_src, srt_dst, srt_eid, W1e, ldw1, W2, b2, W3, b3, logits, s, z1_out);
        case 128: return dispatch_hs<4>(hidden_edge_scores, e, num_edges, Ps, Qd, ld_node, srt_src, srt_dst, srt_eid, W1e, ldw1, W2, b2, W3, b3, logits, s, z1_out);
        case 256: return dispatch_hs<8>(hidden_edge_scores, e, num_edges, Ps, Qd, ld_node, srt_src, srt_dst, srt_eid, W1e, ldw1, W2, b2, W3, b3, logits, s, z1_out);
        default: set_error("edge_score: hidden=%d not in {64,128,256}", hidden); return GNNOME_EINVAL;
    }
}
