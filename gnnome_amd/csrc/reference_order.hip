// Reference-ORDER fp32 kernels: the dense products of the path evaluated exactly as the reference's CPU libraries
// evaluate them, so that the result is the reference's bit for bit rather than "another correct fp32 rounding".
//
// Why they exist.  torch's CPU nn.Linear (MKL sgemm behind addmm) computes every output element as ONE k-ascending
// chain of fused multiply-adds that starts from zero, and adds the bias afterwards with one more rounding
//     y[r][c] = fl( fma(a[K-1], w[K-1], ... fma(a[1], w[1], fl(a[0] * w[0])) ...) + b[c] )
// (checked bit-exactly against torch 2.10 / MKL 2024.2 for K = 2 ... 192: tests/test_reference_order.py), and its
// eval-mode BatchNorm1d is fma(x, alpha, beta) with alpha = gamma * (1 / sqrt(var + eps)), beta = fma(-mean, alpha, bias).
// A reordered product (the bf16x6 split on the matrix cores, or any blocked fp32 sum) is just as close to the exact
// result, but it is a DIFFERENT fp32 number, and the shipped checkpoint (weights/weights.pt) has a layer that magnifies
// such differences: layer 0's bn_e has channels with running_var ~ 5e-5, i.e. alpha up to 135, so reorder noise of
// 1e-6 in B1h[src] + B2h[dst] + B_3(e) comes out of the normalisation as 1e-4 and is what separates two fp32
// evaluations of the model on an E. coli-sized graph (1.2e-4 in edge probability between the torch oracle and a plain
// C loop; 2e-6 once layer 0 is evaluated in the reference's order - DESIGN.md section 2).  The host (engine.py) therefore runs
// a layer through these kernels when its eval-BatchNorm gain is large, and through the bf16x6 kernels otherwise.
//
// Structure: one ROW per lane.  A lane keeps its row of the left operand in K registers; the weight row is the same
// for all lanes, so it is fetched with SCALAR loads (s_load_dwordx8/x16 into SGPRs, no LDS, no cross-lane traffic)
// and every product is a v_fmac_f32 with an SGPR operand.  Eight output columns are in flight per lane (eight
// independent dependency chains).  The fp32 VALU rate equals the fp32 MFMA rate on gfx950 (157 TF), so at H = 64 this
// costs about what the matrix-core kernels cost; it is the ORDER that the matrix cores cannot give (a
// v_mfma_f32_32x32x2_f32 is itself a k-ordered fma chain, but its operand layout interleaves k = 0,4,1,5,...).
#include "common.h"

namespace gnnome {

constexpr int kRefThreads = 256;
constexpr int kRefCC = 8;   // output columns per lane per pass (two float4 stores)

// acc[c] = fma-chain over k of a[k] * W[(c0 + c) * ldw + k], k ascending from 0, starting at 0.  NC columns at a time:
// the weight rows live in SGPRs (~100 per wave), which is what bounds how many chains can be in flight.
template <int K, int NC>
__device__ __forceinline__ void chain_cols(const float (&a)[K], const float* __restrict__ W, int ldw, int c0, float* acc) {
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = __builtin_fmaf(a[k], W[(int64_t)(c0 + c) * ldw + k], acc[c]);
    }
}
template <int K>
__device__ __forceinline__ void chain8(const float (&a)[K], const float* __restrict__ W, int ldw, int c0, float (&acc)[kRefCC]) {
    if (K <= 64) {
        chain_cols<K, 8>(a, W, ldw, c0, acc);
    } else {
        chain_cols<K, 4>(a, W, ldw, c0, acc);
        chain_cols<K, 4>(a, W, ldw, c0 + 4, acc + 4);
    }
}

// C[M,Nout] = chain(A W^T) + bias   (nn.Linear on the CPU: gated_gcn_full.py:91-96, models/full_graph.py:26-27)
template <int K>
__global__ __launch_bounds__(kRefThreads) void k_linear_ref(const float* __restrict__ A, int64_t M, int lda,
                                                           const float* __restrict__ W, int ldw,
                                                           const float* __restrict__ bias, int Nout, float* __restrict__ C,
                                                           int ldc) {
    const int64_t row = (int64_t)blockIdx.x * kRefThreads + threadIdx.x;
    const bool live = row < M;
    const int64_t r = live ? row : M - 1;
    float a[K];
#pragma unroll
    for (int k = 0; k < K; k += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(A + r * lda + k);
        a[k] = v[0], a[k + 1] = v[1], a[k + 2] = v[2], a[k + 3] = v[3];
    }
#pragma unroll 1
    for (int c0 = 0; c0 < Nout; c0 += kRefCC) {
        float acc[kRefCC];
        chain8<K>(a, W, ldw, c0, acc);
        if (bias != nullptr) {
#pragma unroll
            for (int c = 0; c < kRefCC; ++c) acc[c] = acc[c] + bias[c0 + c];
        }
        if (live) {
            float* out = C + r * ldc + c0;
            *reinterpret_cast<f32x4*>(out) = f32x4{acc[0], acc[1], acc[2], acc[3]};
            *reinterpret_cast<f32x4*>(out + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
        }
    }
}

// The edge gate in the reference's order (gated_gcn_full.py:97,104-110), one sorted position per lane:
//   B3e = chain(e W3^T) + b3            :97
//   x   = fl(fl(B1h[src] + B2h[dst]) + B3e)   :104-105
//   e'  = fl(max(fma(x, alpha, beta), 0) + e) :106-110  (eval BatchNorm1d = fma(x, alpha, beta) in torch)
// ENC: e = chain(relu(chain(raw W1^T) + b1) W2^T) + b2 is computed in registers from the raw edge features
// (models/full_graph.py:27), exactly as torch evaluates the two small nn.Linear calls.
template <int H, bool ENC>
__global__ __launch_bounds__(kRefThreads) void k_edge_gate_ref(const float* e_in, float* e_out, int64_t E,
                                                              const float* __restrict__ B1h, const float* __restrict__ B2h,
                                                              int ldn, const int32_t* __restrict__ srt_src,
                                                              const int32_t* __restrict__ srt_dst, const float* __restrict__ W3,
                                                              int ldw, const float* __restrict__ b3,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              GateEnc enc) {
    const int64_t row = (int64_t)blockIdx.x * kRefThreads + threadIdx.x;
    const bool live = row < E;
    const int64_t p = live ? row : E - 1;
    float a[H];
    if (ENC) {
        const int64_t eid = enc.srt_eid[p];
        const float x0 = enc.e_raw[2 * eid], x1 = enc.e_raw[2 * eid + 1];
        float t[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) t[j] = fmaxf(__builtin_fmaf(x1, enc.W1[2 * j + 1], x0 * enc.W1[2 * j]) + enc.b1[j], 0.f);
#pragma unroll
        for (int k = 0; k < H; ++k) {
            float s = t[0] * enc.W2[k * 16];
#pragma unroll
            for (int j = 1; j < 16; ++j) s = __builtin_fmaf(t[j], enc.W2[k * 16 + j], s);
            a[k] = s + enc.b2[k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < H; k += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(e_in + p * H + k);
            a[k] = v[0], a[k + 1] = v[1], a[k + 2] = v[2], a[k + 3] = v[3];
        }
    }
    const float* g1 = B1h + (int64_t)srt_src[p] * ldn;
    const float* g2 = B2h + (int64_t)srt_dst[p] * ldn;
    // e_out may alias e_in: a lane owns its row, has read all of it above, and writes each column once
#pragma unroll 1
    for (int c0 = 0; c0 < H; c0 += kRefCC) {
        float acc[kRefCC];
        chain8<H>(a, W3, ldw, c0, acc);
        const f32x4 u0 = *reinterpret_cast<const f32x4*>(g1 + c0), u1 = *reinterpret_cast<const f32x4*>(g1 + c0 + 4);
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(g2 + c0), v1 = *reinterpret_cast<const f32x4*>(g2 + c0 + 4);
        float y[kRefCC];
#pragma unroll
        for (int c = 0; c < kRefCC; ++c) {
            const float g = (c < 4 ? u0[c] : u1[c - 4]) + (c < 4 ? v0[c] : v1[c - 4]);
            const float x = g + (acc[c] + b3[c0 + c]);
            y[c] = fmaxf(__builtin_fmaf(x, scale[c0 + c], shift[c0 + c]), 0.f);
        }
        // the residual: this row's own columns c0..c0+7 (a[] cannot be indexed by the runtime c0 without going to scratch)
        if (ENC) {
            float t[16];
            const int64_t eid = enc.srt_eid[p];
            const float x0 = enc.e_raw[2 * eid], x1 = enc.e_raw[2 * eid + 1];
#pragma unroll
            for (int j = 0; j < 16; ++j) t[j] = fmaxf(__builtin_fmaf(x1, enc.W1[2 * j + 1], x0 * enc.W1[2 * j]) + enc.b1[j], 0.f);
#pragma unroll
            for (int c = 0; c < kRefCC; ++c) {
                float s = t[0] * enc.W2[(c0 + c) * 16];
#pragma unroll
                for (int j = 1; j < 16; ++j) s = __builtin_fmaf(t[j], enc.W2[(c0 + c) * 16 + j], s);
                y[c] = y[c] + (s + enc.b2[c0 + c]);
            }
        } else {
            const f32x4 r0 = *reinterpret_cast<const f32x4*>(e_in + p * H + c0), r1 = *reinterpret_cast<const f32x4*>(e_in + p * H + c0 + 4);
#pragma unroll
            for (int c = 0; c < kRefCC; ++c) y[c] = y[c] + (c < 4 ? r0[c] : r1[c - 4]);
        }
        if (live) {
            float* out = e_out + p * H + c0;
            *reinterpret_cast<f32x4*>(out) = f32x4{y[0], y[1], y[2], y[3]};
            *reinterpret_cast<f32x4*>(out + 4) = f32x4{y[4], y[5], y[6], y[7]};
        }
    }
}

template <int H, bool ENC>
static int launch_gate_ref(const float* e_in, float* e_out, int64_t E, const float* B1h, const float* B2h, int ldn,
                           const int32_t* ss, const int32_t* sd, const float* W3, int ldw, const float* b3, const float* scale,
                           const float* shift, const GateEnc& enc, hipStream_t s) {
    const int64_t blocks = (E + kRefThreads - 1) / kRefThreads;
    GN_REQUIRE(blocks < (1ll << 31), "edge_gate_ref: too many edges");
    hipLaunchKernelGGL((k_edge_gate_ref<H, ENC>), dim3((unsigned)blocks), dim3(kRefThreads), 0, s, e_in, e_out, E, B1h, B2h, ldn, ss,
                       sd, W3, ldw, b3, scale, shift, enc);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

}  // namespace gnnome

extern "C" int gnnome_linear_ref_f32(const float* A, int64_t M, int K, int lda, const float* W, int ldw, const float* bias,
                                     int Nout, float* C, int ldc, void* stream) {
    using namespace gnnome;
    GN_REQUIRE(M >= 0, "linear_ref: negative row count");
    if (M == 0 || Nout == 0) return GNNOME_OK;
    GN_REQUIRE(A && W && C, "linear_ref: null pointer");
    GN_REQUIRE(Nout % kRefCC == 0, "linear_ref: Nout=%d must be a multiple of %d", Nout, kRefCC);
    GN_REQUIRE(lda % 4 == 0 && ldc % 4 == 0 && lda >= K && ldc >= Nout && ldw >= K, "linear_ref: bad strides");
    GN_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)C % 16 == 0), "linear_ref: A and C must be 16-byte aligned");
    const int64_t blocks = (M + kRefThreads - 1) / kRefThreads;
    GN_REQUIRE(blocks < (1ll << 31), "linear_ref: too many rows");
    hipStream_t s = (hipStream_t)stream;
    // the shipped default: the same chain on the fp32 matrix cores (reference_order_mfma.hip); variant 1 = the VALU form below
    if ((tuning(kTuneRefVariant) == 0 || K == 256) && (K == 64 || K == 128 || K == 256) && ((uintptr_t)W % 4 == 0))   // (K = 256: the matrix-core form only, round 4)
        return linear_refm_launch(A, M, K, lda, W, ldw, bias, Nout, C, ldc, s);
    switch (K) {
        case 64: hipLaunchKernelGGL(k_linear_ref<64>, dim3((unsigned)blocks), dim3(kRefThreads), 0, s, A, M, lda, W, ldw, bias, Nout, C, ldc); break;
        case 128: hipLaunchKernelGGL(k_linear_ref<128>, dim3((unsigned)blocks), dim3(kRefThreads), 0, s, A, M, lda, W, ldw, bias, Nout, C, ldc); break;
        default: set_error("linear_ref: K=%d not in {64,128,256}", K); return GNNOME_EINVAL;
    }
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

extern "C" int gnnome_edge_gate_ref_f32(const float* e_in, float* e_out, int64_t num_edges, int hidden, const float* B1h,
                                        const float* B2h, int ld_node, const int32_t* srt_src, const int32_t* srt_dst,
                                        const float* W3, int ldw, const float* b3, const float* norm_scale,
                                        const float* norm_shift, const float* e_raw, const int32_t* srt_eid, const float* encW1,
                                        const float* encb1, const float* encW2, const float* encb2, void* stream) {
    using namespace gnnome;
    GN_REQUIRE(num_edges >= 0, "edge_gate_ref: negative edge count");
    if (num_edges == 0) return GNNOME_OK;
    const bool enc = e_raw != nullptr;
    GN_REQUIRE(e_out && B1h && B2h && srt_src && srt_dst && W3 && b3 && norm_scale && norm_shift, "edge_gate_ref: null pointer");
    GN_REQUIRE(enc ? (srt_eid && encW1 && encb1 && encW2 && encb2) : (e_in != nullptr), "edge_gate_ref: e_in, or e_raw + srt_eid + encoder weights");
    GN_REQUIRE(ld_node % 4 == 0 && ld_node >= hidden && ldw >= hidden, "edge_gate_ref: bad strides");
    GN_REQUIRE(((uintptr_t)B1h % 16 == 0) && ((uintptr_t)B2h % 16 == 0) && ((uintptr_t)e_out % 16 == 0) && ((uintptr_t)e_in % 16 == 0),
               "edge_gate_ref: tensors must be 16-byte aligned");
    GateEnc ge{e_raw, srt_eid, encW1, encb1, encW2, encb2};
    hipStream_t s = (hipStream_t)stream;
    if ((tuning(kTuneRefVariant) == 0 && (hidden == 64 || hidden == 128)) || hidden == 256)   // the shipped default: the chain on the fp32 matrix cores (the only form at 256)
        return gate_refm_launch(hidden, enc, e_in, e_out, num_edges, B1h, B2h, ld_node, srt_src, srt_dst, W3, ldw, b3, norm_scale, norm_shift, ge, s);
    if (hidden == 64) {
        return enc ? launch_gate_ref<64, true>(e_in, e_out, num_edges, B1h, B2h, ld_node, srt_src, srt_dst, W3, ldw, b3, norm_scale, norm_shift, ge, s)
                   : launch_gate_ref<64, false>(e_in, e_out, num_edges, B1h, B2h, ld_node, srt_src, srt_dst, W3, ldw, b3, norm_scale, norm_shift, ge, s);
    }
    if (hidden == 128) {
        return enc ? launch_gate_ref<128, true>(e_in, e_out, num_edges, B1h, B2h, ld_node, srt_src, srt_dst, W3, ldw, b3, norm_scale, norm_shift, ge, s)
                   : launch_gate_ref<128, false>(e_in, e_out, num_edges, B1h, B2h, ld_node, srt_src, srt_dst, W3, ldw, b3, norm_scale, norm_shift, ge, s);
    }
    set_error("edge_gate_ref: hidden=%d not in {64,128,256}", hidden);
    return GNNOME_EINVAL;
}
