// Shared host/device helpers for libgnnome_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gnnome_hip.h"

namespace gnnome {

void set_error(const char* fmt, ...);

// Tuning knobs (gnnome_set_tuning): variant selection for A/B measurements; 0 = the shipped default.
enum { kTuneGateVariant = 0, kTuneGateAblation = 1, kTuneLinearVariant = 2, kTuneGateTileOrder = 3, kTuneGateExperiment = 4, kTuneAggLdsKiB = 5, kTuneAggHubs = 6, kTuneAggVariant = 7, kTuneRefVariant = 8, kTuneOverlapBand = 9, kTuneArith = 10, kTuneCount = 16 };
int tuning(int key);

// Layer 0 only: the e tile is not loaded but COMPUTED by the load waves from the raw edge features,
// e0[p,:] = W2e * relu(W1e * e_raw[srt_eid[p],:] + b1e) + b2e (models/full_graph.py:27, in_features = 2,
// hidden_ne = 16): the edge encoder's [E,H] output is never written to or read from HBM.
struct GateEnc {
    const float* e_raw;       // [E,2] in edge-id order
    const int32_t* srt_eid;   // sorted position -> edge id
    const float *W1, *b1, *W2, *b2;   // [16,2] [16] [H,16] [H]
    const float *W23, *b23;           // filled by the launcher for k_edge_gate_enc16: W3 * W2 [H,16] and W3 * b2 [H]
};

// Mode 3 of the edge-tile kernel: the A operand is not read but COMPUTED by the load waves from two streams,
// A = a[c] * (dy m - c1[c] - (x - mean[c]) rstd[c] c2[c]),  m = (x scale[c] + shift[c] > 0)   (BatchNorm backward through the relu),
// with dy = the old rows of C and x = the rows at e_in; A is also written to `a_out`.  One pass does what
// gnnome_bn_bwd_apply_f32 + gnnome_linear_acc_f32 did in two (train.py: dxe, then d e_in = d e' + dxe W3).
struct GateBnBwd {
    const float *a, *c1, *c2, *mean, *rstd, *scale, *shift;   // [H] each
    float* a_out;                                              // [E,H]
    int64_t n_once;   // rows [0, n_once) get the mean-subtraction terms c1, c2; the rest (replicas of rows another rank owns) do not
    unsigned* amax_bits = nullptr;   // NULL or: max |a_out| as the bits of a non-negative float, raised with atomicMax (zeroed by the launcher)
};

// Arguments of the bf16x6 edge-tile kernel (edge_gate_bf.hip); mode 0 gate, 1 raw gate + statistics, 2 C += A W^T, 3 see GateBnBwd.
struct GateBfArgs {
    const float* e_in;        // A operand rows [E,H] (mode 2: A); unused with enc
    float* e_out;             // result rows [E,H] (may alias e_in; mode 2: C)
    int64_t E;
    const float* B1h;         // mode 0/1: gathered by srt_src; mode 2: the old rows of C
    const float* B2h;         // mode 0/1: gathered by srt_dst
    int ldn;
    const int32_t* srt_src;
    const int32_t* srt_dst;
    const float* W3;          // [H,H] row-major ([out,in]), row stride ldw
    int ldw;
    const float* scale;       // mode 0: folded norm scale; mode 1: per-column centre of the statistics
    const float* shift;       // mode 0
    float* stats;             // mode 1: [kNumCUs * RB][2H] per-workgroup shifted column sums
    int num_tiles;            // filled by the launcher
    int abl;                  // measurement-only ablation mask (gnnome_set_tuning key 1), 0 in normal use
    long long* prof;          // measurement only: per-workgroup phase cycle counters [gridDim][8] (NULL in normal use)
    int xp;                   // experiment knob (key 4): producers' poll interval 0..3 = s_sleep 1/4/16/64
    int num_cblocks, ld_out;  // edge_gate_pl256.hip mode 4 (C = A W^T + bias): 128-column blocks of the output, its row stride
    GateEnc enc;              // mode 0 with the folded edge encoder
    GateBnBwd bnb;            // mode 3
};
int gate_bf_launch(int hidden, int mode, bool enc, const GateBfArgs& args, hipStream_t s, bool x16 = false, int extra = 0);
// H = 256, affine norm, e_out != e_in: barrier-free streaming gate (edge_gate_stream.hip)
int gate_stream_launch(const float* e_in, float* e_out, int64_t E, const float* B1h, const float* B2h, int ldn, const int32_t* ss,
                       const int32_t* sd, const float* W3, int ldw, const float* scale, const float* shift, hipStream_t s);

// C[M,K] += A[M,K] * W[K,K]^T for K in {64,128}, contiguous 16-byte aligned A and C: the wave-specialised
// edge-tile kernel (edge_gate.hip) in accumulate mode; linear.hip routes the backward's [E,H] dgrad here.
int ws_linear_acc(const float* A, int64_t M, int K, const float* W, int ldw, float* C, hipStream_t s);
int stream_linear_acc_256(const float* A, int64_t M, const float* W, int ldw, float* C, hipStream_t s);   // edge_gate_stream.hip
// H = 256 in the wave-specialised plane form (edge_gate_pl256.hip): modes 0 (gate), 1 (raw gate, optional statistics), 2 (C += A W^T)
int gate_pl256_launch(int mode, const GateBfArgs& a, hipStream_t s, bool x16 = false);   // x16: mode 1 (xe out as bf16), mode 3 (xe in / dxe out as bf16)
// the same tiles in fp16x3 arithmetic with LDS-DMA tile loads (edge_tile_f16.hip; modes 0, 1, 4): the default, gnnome_set_tuning(10, 1) = bf16x6
int gate_f16_launch(int mode, const GateBfArgs& a, int grid, hipStream_t s, bool x16 = false);
int gate_enc256_launch(const GateBfArgs& a, hipStream_t s);   // layer 0 at H = 256 with the edge encoder folded (edge_gate_bf.hip -> edge_tile_f16.hip mode 5)
int gate_pl256_stats_rows();
void hub_cache_invalidate();        // node_aggregate.hip: forget the hub list of the previous graph (called when views are built)
long long* gate_profile_buffer();   // gnnome_debug_gate_profile's buffer (edge_gate_bf.hip), NULL in normal use
// reference-order kernels on the fp32 matrix cores (reference_order_mfma.hip); K / hidden in {64, 128}
int linear_refm_launch(const float* A, int64_t M, int K, int lda, const float* W, int ldw, const float* bias, int Nout, float* C, int ldc,
                       hipStream_t s);
int gate_refm_launch(int hidden, bool with_enc, const float* e_in, float* e_out, int64_t E, const float* B1h, const float* B2h, int ldn,
                     const int32_t* ss, const int32_t* sd, const float* W3, int ldw, const float* b3, const float* scale, const float* shift,
                     const GateEnc& enc, hipStream_t s);

#define GN_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            ::gnnome::set_error(__VA_ARGS__); \
            return GNNOME_EINVAL;             \
        }                                     \
    } while (0)

#define GN_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t err__ = (call);                                                                \
        if (err__ != hipSuccess) {                                                                \
            ::gnnome::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(err__), __FILE__, \
                                __LINE__);                                                        \
            return GNNOME_EHIP;                                                                   \
        }                                                                                         \
    } while (0)

#define GN_LAUNCH_CHECK() GN_HIP(hipGetLastError())

constexpr int kWave = 64;       // gfx950 wavefront
constexpr int kXcds = 8;        // MI355X: 8 XCDs, block b is dispatched to XCD b % 8
constexpr int kNumCUs = 256;    // MI355X in SPX mode: also the upper bound the per-workgroup scratch buffers are sized for
// Workgroups of a persistent launch (one per CU): the current device's CU count, rounded down to a multiple of kXcds and
// capped at kNumCUs (a partitioned MI355X - CPX / DPX modes - exposes fewer CUs per device).
int persistent_grid();
constexpr float kAggEps = 1e-6f;   // gated_gcn_full.py:114,127
constexpr float kNormEps = 1e-5f;  // torch BatchNorm1d / LayerNorm default eps

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// bf16 STORAGE of training activations (the "x16" entry points): round to nearest even on the way out, exact on the way in.
// Arithmetic stays fp32; only the bytes of the pre-normalisation gate output xe and of its gradient dxe in HBM are halved.
__device__ __forceinline__ unsigned bf16_bits(float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ uint2 pack_bf16x4(const f32x4 v) {
    return make_uint2(bf16_bits(v[0]) | (bf16_bits(v[1]) << 16), bf16_bits(v[2]) | (bf16_bits(v[3]) << 16));
}
__device__ __forceinline__ f32x4 unpack_bf16x4(const uint2 p) {
    return f32x4{__uint_as_float(p.x << 16), __uint_as_float(p.x & 0xFFFF0000u), __uint_as_float(p.y << 16),
                 __uint_as_float(p.y & 0xFFFF0000u)};
}
// a row piece of four values at element offset `off` of a tensor stored as fp32 (X16 = false) or bf16 (true)
template <bool X16>
__device__ __forceinline__ f32x4 load4_as(const void* base, int64_t off) {
    if (X16) return unpack_bf16x4(*reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + off));
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + off);
}
template <bool X16>
__device__ __forceinline__ void store4_as(void* base, int64_t off, const f32x4 v) {
    if (X16)
        *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(base) + off) = pack_bf16x4(v);
    else
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + off) = v;
}


// Give each XCD (private 4 MiB L2) a contiguous range of work items so that neighbouring tiles -
// which share destination rows and, in layout-ordered assembly graphs, nearby source rows - hit
// the same L2.  Bijective for any n (cdna_hip_programming.md 5.5 T1).
__device__ __forceinline__ int xcd_remap(int bid, int n) {
    const int xcd = bid % kXcds, q = n / kXcds, r = n % kXcds;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + bid / kXcds;
}

// v_exp_f32 + v_rcp_f32 (each within 1 ulp); an IEEE division here costs ~10 more VALU operations per element and the
// aggregation kernels evaluate this E*H*2 times per layer
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// relu that keeps NaN a NaN (fmaxf(NaN, 0) = 0): a node row whose sums went non-finite - an fp16x3 operand beyond 65504 upstream - must stay loud
// all the way to the logits, where engine.forward_in_range looks.  Same value as fmaxf(t, 0) for every t that is not NaN.
__device__ __forceinline__ float relu_keep_nan(float t) { return t < 0.f ? 0.f : t; }

// max |.| of a wave's values into amax_bits[0] (the bits of a non-negative float; unsigned order = float order): one atomicMax per wave, and
// only when the value read first is smaller - a maximum does not depend on the order it is formed in, so results stay reproducible.
__device__ __forceinline__ void wave_amax_to(unsigned* amax_bits, float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0) {
        const unsigned b = __float_as_uint(v);
        if (b > __hip_atomic_load(amax_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(amax_bits, b);
    }
}

// sum over the 32 lanes that share (lane >> 5)
__device__ __forceinline__ float half_wave_sum(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 16);
    return v;
}

}  // namespace gnnome
