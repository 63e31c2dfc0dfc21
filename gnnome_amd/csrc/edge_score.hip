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

// ---------------------------------------------------------------------------------------------------
// Weight-stationary streaming form (hs = 64, H in {64,128}; the inference path).  The tile kernel above re-stages and
// re-splits W1e for every 128-edge tile and synchronises the workgroup eight times per tile.  Here W1e lives in LDS as
// three bf16 planes for the whole launch (52 KB at H = 128); every wave owns 32 edges at a time and streams their e rows
// STRAIGHT FROM GLOBAL MEMORY INTO MFMA FRAGMENTS (each e row is read exactly once: one 64-column chunk), parks e W1e^T in a
// wave-private LDS tile, adds Ps[src] + Qd[dst] there row-wise (16-byte gathers, 16 lanes per row), and runs the 64 -> 32 -> 1
// tail from that tile.  One barrier per launch; 8 waves per workgroup, one workgroup per CU.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void score_split8(const f32x4 lo4, const f32x4 hi4, uint4& p1, uint4& p2, uint4& p3) {
    uint2 l1, l2, l3, h1, h2, h3;
    tile_split4(lo4, l1, l2, l3);
    tile_split4(hi4, h1, h2, h3);
    p1 = make_uint4(l1.x, l1.y, h1.x, h1.y);
    p2 = make_uint4(l2.x, l2.y, h2.x, h2.y);
    p3 = make_uint4(l3.x, l3.y, h3.x, h3.y);
}

// fp16x3 (round 4; see edge_tile_f16.hip's header): two fp16 planes per operand, three f16 MFMAs per k step, the two small products in a second
// accumulator folded in with 2^-11.  F16 = true also runs the 64 -> 32 product on the f16 matrix cores (it was 32 exact-fp32 MFMAs = 2048 matrix-pipe
// cycles per 32-edge tile, now 12 f16 MFMAs = 384) and lets W1e fit LDS at K = 256 (two planes: 68 KB), so that width streams too.
typedef _Float16 sc_h2 __attribute__((ext_vector_type(2)));
typedef _Float16 sc_h8 __attribute__((ext_vector_type(8)));
typedef float sc_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void score_split8h(const f32x4 lo, const f32x4 hi, uint4& p1, uint4& p2) {
    unsigned a[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const sc_f2 v = j < 2 ? sc_f2{lo[2 * j], lo[2 * j + 1]} : sc_f2{hi[2 * j - 4], hi[2 * j - 3]};
        const sc_h2 x1 = __builtin_convertvector(v, sc_h2);
        const sc_f2 big = v * 2048.f;
        const sc_f2 r = {__builtin_fmaf((float)x1[0], -2048.f, big[0]), __builtin_fmaf((float)x1[1], -2048.f, big[1])};   // exact
        const sc_h2 x2 = __builtin_convertvector(r, sc_h2);
        a[j] = __builtin_bit_cast(unsigned, x1);
        b[j] = __builtin_bit_cast(unsigned, x2);
    }
    p1 = make_uint4(a[0], a[1], a[2], a[3]);
    p2 = make_uint4(b[0], b[1], b[2], b[3]);
}

template <int K, bool F16 = false>
__global__ __launch_bounds__(512) void k_edge_score_ws(
    const float* __restrict__ e, int64_t E, const float* __restrict__ Ps, const float* __restrict__ Qd, int ldn,
    const int32_t* __restrict__ srt_src, const int32_t* __restrict__ srt_dst, const int32_t* __restrict__ srt_eid,
    const float* __restrict__ W1e, int ldw1, const float* __restrict__ W2, const float* __restrict__ b2,
    const float* __restrict__ W3, const float* __restrict__ b3, float* __restrict__ logits, int num_tiles, int tiles_per_group) {
    constexpr int NW = 8, HS = 64, PLD = 2 * K + 16, PB = HS * PLD, KS = K / 16, LDZ = HS + 4, BATCH = 2;
    static_assert(F16 || K <= 128, "three bf16 planes of W1e fit LDS up to K = 128");
    __shared__ __attribute__((aligned(16))) unsigned char Wp[(F16 ? 2 : 3) * PB];
    __shared__ __attribute__((aligned(16))) float W2s[32 * LDZ];
    __shared__ __attribute__((aligned(16))) float Zall[NW * 32 * LDZ];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cl = lane & 31, half = lane >> 5;
    for (int f = tid; f < HS * (K / 8); f += 64 * NW) {   // split W1e once: eight consecutive k of one row per piece
        const int row = f / (K / 8), c8 = f % (K / 8);
        const float* src = W1e + (int64_t)row * ldw1 + 8 * c8;
        uint4 p1, p2, p3;
        unsigned char* dst = Wp + row * PLD + 16 * c8;
        if (F16) {
            score_split8h(*reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + 4), p1, p2);
        } else {
            score_split8(*reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + 4), p1, p2, p3);
            *reinterpret_cast<uint4*>(dst + 2 * PB) = p3;
        }
        *reinterpret_cast<uint4*>(dst) = p1;
        *reinterpret_cast<uint4*>(dst + PB) = p2;
    }
    for (int i = tid; i < 32 * HS; i += 64 * NW) W2s[(i / HS) * LDZ + (i % HS)] = W2[i];
    const float bias2 = b2[cl], w3 = W3[cl], bias3 = b3[0];
    __syncthreads();   // the only barrier
    // F16: this lane's B fragments of W2 (output column cl, k = 16 q + 8 half .. + 7) as two fp16 planes, for the whole launch
    uint4 w2a[HS / 16], w2b[HS / 16];
    if (F16) {
#pragma unroll
        for (int q = 0; q < HS / 16; ++q) {
            const float* wr = W2s + cl * LDZ + 16 * q + 8 * half;
            score_split8h(*reinterpret_cast<const f32x4*>(wr), *reinterpret_cast<const f32x4*>(wr + 4), w2a[q], w2b[q]);
        }
    }
    auto h8 = [](const uint4 v) { return __builtin_bit_cast(sc_h8, v); };

    auto bf = [](const uint4 v) { return __builtin_bit_cast(tile_bf16x8, v); };
    float* Zs = Zall + wave * 32 * LDZ;
    const unsigned char* wp = Wp + cl * PLD + 16 * half;
    const int t0 = blockIdx.x * tiles_per_group, t_end = min(num_tiles, t0 + tiles_per_group);
    for (int t = t0 + wave; t < t_end; t += NW) {
        const int64_t row0 = (int64_t)t * 32;
        const int valid = (int)min((int64_t)32, E - row0);
        // Ps[src] + Qd[dst], row-wise: lane l takes rows 4 i + (l >> 4), columns 4 (l & 15) .. + 3; issued first, consumed last
        const int grow = lane >> 4, gc4 = lane & 15;
        int si[8], di[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t row = row0 + min(4 * i + grow, valid - 1);
            si[i] = srt_src[row];
            di[i] = srt_dst[row];
        }
        // e rows -> fragments (8 consecutive k of one row per lane and K = 16 step), BATCH steps in flight ahead of the MFMAs
        const float* ap = e + (row0 + min(cl, valid - 1)) * K + 8 * half;
        f32x16 acc0, acc1, acc0c, acc1c;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = acc0c[r] = acc1c[r] = 0.f;
        f32x4 x[BATCH][2];
#pragma unroll
        for (int q = 0; q < BATCH; ++q) {
            x[q][0] = *reinterpret_cast<const f32x4*>(ap + 16 * q);
            x[q][1] = *reinterpret_cast<const f32x4*>(ap + 16 * q + 4);
        }
#pragma unroll
        for (int hq = 0; hq < KS; hq += BATCH) {
            f32x4 nx[BATCH][2];
            const int hn = hq + BATCH < KS ? hq + BATCH : hq;
#pragma unroll
            for (int q = 0; q < BATCH; ++q) {
                nx[q][0] = *reinterpret_cast<const f32x4*>(ap + 16 * (hn + q));
                nx[q][1] = *reinterpret_cast<const f32x4*>(ap + 16 * (hn + q) + 4);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < BATCH; ++q) {
                uint4 a1, a2, a3;
                const unsigned char* w = wp + 32 * (hq + q);
                if (F16) {
                    score_split8h(x[q][0], x[q][1], a1, a2);
                    const uint4 u1 = *reinterpret_cast<const uint4*>(w), u2 = *reinterpret_cast<const uint4*>(w + PB);
                    const uint4 v1 = *reinterpret_cast<const uint4*>(w + 32 * PLD), v2 = *reinterpret_cast<const uint4*>(w + 32 * PLD + PB);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(a1), h8(u1), acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(a1), h8(v1), acc1, 0, 0, 0);
                    acc0c = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(a1), h8(u2), acc0c, 0, 0, 0);
                    acc1c = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(a1), h8(v2), acc1c, 0, 0, 0);
                    acc0c = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(a2), h8(u1), acc0c, 0, 0, 0);
                    acc1c = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(a2), h8(v1), acc1c, 0, 0, 0);
                    continue;
                }
                score_split8(x[q][0], x[q][1], a1, a2, a3);
                const uint4 u1 = *reinterpret_cast<const uint4*>(w), u2 = *reinterpret_cast<const uint4*>(w + PB),
                            u3 = *reinterpret_cast<const uint4*>(w + 2 * PB);
                const uint4 v1 = *reinterpret_cast<const uint4*>(w + 32 * PLD), v2 = *reinterpret_cast<const uint4*>(w + 32 * PLD + PB),
                            v3 = *reinterpret_cast<const uint4*>(w + 32 * PLD + 2 * PB);
                // smallest terms first; the two column blocks alternate
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a3), bf(u1), acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a3), bf(v1), acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1), bf(u3), acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1), bf(v3), acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a2), bf(u2), acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a2), bf(v2), acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a2), bf(u1), acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a2), bf(v1), acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1), bf(u2), acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1), bf(v2), acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1), bf(u1), acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1), bf(v1), acc1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < BATCH; ++q) {
                x[q][0] = nx[q][0];
                x[q][1] = nx[q][1];
            }
        }
        // the gathers (their indices arrived under the GEMM)
        f32x4 g[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            g[i] = *reinterpret_cast<const f32x4*>(Ps + (int64_t)si[i] * ldn + 4 * gc4) + *reinterpret_cast<const f32x4*>(Qd + (int64_t)di[i] * ldn + 4 * gc4);
        if (F16) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc0[r] += acc0c[r] * (1.0f / 2048.f);
                acc1[r] += acc1c[r] * (1.0f / 2048.f);
            }
        }
        // e W1e^T -> the wave's LDS tile (accumulator layout), then relu(. + G) row-wise in place
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lr = cd_row(r, lane);
            Zs[lr * LDZ + cl] = acc0[r];
            Zs[lr * LDZ + 32 + cl] = acc1[r];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float* zr = Zs + (4 * i + grow) * LDZ + 4 * gc4;
            f32x4 z = *reinterpret_cast<const f32x4*>(zr) + g[i];
            // F16: an operand beyond fp16's range (an e element, or a z1 value below) leaves the matrix cores as inf / NaN and must not come out of
            // the relu as 0 - fmaxf(NaN, 0) = 0 would hand back a finite, wrong logit; t - t is 0 iff t is finite
#pragma unroll
            for (int j = 0; j < 4; ++j) z[j] = (!F16 || z[j] - z[j] == 0.f) ? fmaxf(z[j], 0.f) : __builtin_nanf("");
            *reinterpret_cast<f32x4*>(zr) = z;
        }
        // z2 = relu(W2 z1 + b2) on the exact-fp32 matrix cores (K = 64), logit = W3 . z2 + b3
        f32x16 acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[r] = bias2;
        if (F16) {
            f32x16 acc2c;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2c[r] = 0.f;
            const float* zq = Zs + cl * LDZ + 8 * half;
#pragma unroll
            for (int q = 0; q < HS / 16; ++q) {
                uint4 z1, z2;
                score_split8h(*reinterpret_cast<const f32x4*>(zq + 16 * q), *reinterpret_cast<const f32x4*>(zq + 16 * q + 4), z1, z2);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(z1), h8(w2a[q]), acc2, 0, 0, 0);
                acc2c = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(z1), h8(w2b[q]), acc2c, 0, 0, 0);
                acc2c = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(z2), h8(w2a[q]), acc2c, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[r] += acc2c[r] * (1.0f / 2048.f);
        } else {
            const float* zp = Zs + cl * LDZ + 4 * half;
            const float* w2p = W2s + cl * LDZ + 4 * half;
#pragma unroll
            for (int q = 0; q < HS / 8; ++q) {
                const f32x4 za = *reinterpret_cast<const f32x4*>(zp + 8 * q);
                const f32x4 wb = *reinterpret_cast<const f32x4*>(w2p + 8 * q);
#pragma unroll
                for (int k = 0; k < 4; ++k) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(za[k], wb[k], acc2, 0, 0, 0);
            }
        }
        float mine = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float t2 = acc2[r];
            const float sum = half_wave_sum(((!F16 || t2 - t2 == 0.f) ? fmaxf(t2, 0.f) : __builtin_nanf("")) * w3);
            if (cl == r) mine = sum;
        }
        if (cl < 16) {
            const int lr = cd_row(cl, lane);
            if (lr < valid) {
                const int64_t p = row0 + lr;
                const int64_t eid = srt_eid != nullptr ? (int64_t)srt_eid[p] : p;
                logits[eid] = mine + bias3;
            }
        }
    }
}

template <int K, bool F16 = false>
static int launch_score_ws(const float* e, int64_t E, const float* Ps, const float* Qd, int ldn, const int32_t* ss,
                           const int32_t* sd, const int32_t* se, const float* W1e, int ldw1, const float* W2, const float* b2,
                           const float* W3, const float* b3, float* logits, hipStream_t s) {
    const int64_t tiles = (E + 31) / 32;
    GN_REQUIRE(tiles < (1ll << 31), "edge_score: too many tiles");
    int groups = persistent_grid();
    if (groups > (tiles + 7) / 8) groups = (int)((tiles + 7) / 8);
    const int tpg = (int)((tiles + groups - 1) / groups);
    hipLaunchKernelGGL((k_edge_score_ws<K, F16>), dim3(groups), dim3(512), 0, s, e, E, Ps, Qd, ldn, ss, sd, se, W1e, ldw1, W2, b2, W3, b3, logits,
                       (int)tiles, tpg);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
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
    // inference at hs = 64: the weight-stationary streaming kernel (gnnome_set_tuning(2, 1) = the tile kernels keeps the old one)
    const bool f16 = tuning(kTuneArith) == 0;   // fp16x3 (round 4): at H = 128 / 256; gnnome_set_tuning(10, 1) = bf16x6 (H <= 128)
    if (z1_out == nullptr && hidden_edge_scores == 64 && (hidden == 64 || hidden == 128 || (hidden == 256 && f16)) && ld_node % 4 == 0 &&
        ((uintptr_t)Ps % 16 == 0) && ((uintptr_t)Qd % 16 == 0) && tuning(kTuneLinearVariant) != 1) {
        if (hidden == 256) return launch_score_ws<256, true>(e, num_edges, Ps, Qd, ld_node, srt_src, srt_dst, srt_eid, W1e, ldw1, W2, b2, W3, b3, logits, s);
        if (hidden == 128)
            return f16 ? launch_score_ws<128, true>(e, num_edges, Ps, Qd, ld_node, srt_src, srt_dst, srt_eid, W1e, ldw1, W2, b2, W3, b3, logits, s)
                       : launch_score_ws<128>(e, num_edges, Ps, Qd, ld_node, srt_src, srt_dst, srt_eid, W1e, ldw1, W2, b2, W3, b3, logits, s);
        return f16 ? launch_score_ws<64, true>(e, num_edges, Ps, Qd, ld_node, srt_src, srt_dst, srt_eid, W1e, ldw1, W2, b2, W3, b3, logits, s)
                   : launch_score_ws<64>(e, num_edges, Ps, Qd, ld_node, srt_src, srt_dst, srt_eid, W1e, ldw1, W2, b2, W3, b3, logits, s);
    }
    switch (hidden) {
        case 64: return dispatch_hs<2>(hidden_edge_scores, e, num_edges, Ps, Qd, ld_node, srt_src, srt_dst, srt_eid, W1e, ldw1, W2, b2, W3, b3, logits, s, z1_out);
        case 128: return dispatch_hs<4>(hidden_edge_scores, e, num_edges, Ps, Qd, ld_node, srt_src, srt_dst, srt_eid, W1e, ldw1, W2, b2, W3, b3, logits, s, z1_out);
        case 256: return dispatch_hs<8>(hidden_edge_scores, e, num_edges, Ps, Qd, ld_node, srt_src, srt_dst, srt_eid, W1e, ldw1, W2, b2, W3, b3, logits, s, z1_out);
        default: set_error("edge_score: hidden=%d not in {64,128,256}", hidden); return GNNOME_EINVAL;
    }
}
