// gnnome_encode_f32: the two-layer input encoders, out = W2 * relu(W1 * in + b1) + b2.
// Reference lines replaced: models/full_graph.py:26 (linear2_node(relu(linear1_node(x)))) and :27
// (same for edges).  With `gather` = srt_eid the edge encoder also moves e from edge-id order into
// destination-sorted order, so the [E,H] tensor is written exactly once, already where the layer
// kernels want it.  Bound: HBM write of rows*H*4 bytes (the K=2 / K=16 products are VALU work).
//
// A row of H outputs is produced by H/4 lanes (one float4 each).  Lane i of the row first computes
// hidden unit(s) i, i+H/4, ... of relu(W1*in+b1); the units are then broadcast with __shfl while each
// lane accumulates its four output columns against W2^T held in LDS.
#include "common.h"

namespace gnnome {

constexpr int kEncThreads = 256;
constexpr int kEncMaxF = 8, kEncMaxM = 64;

template <int H>
__global__ __launch_bounds__(kEncThreads) void k_encode(const float* __restrict__ in, int64_t rows, int F,
                                                        const int32_t* __restrict__ gather, const float* __restrict__ W1,
                                                        const float* __restrict__ b1, int M, const float* __restrict__ W2,
                                                        const float* __restrict__ b2, float* __restrict__ out) {
    constexpr int LPR = H / 4, RPB = kEncThreads / LPR, KMAX = (kEncMaxM + LPR - 1) / LPR;
    __shared__ __attribute__((aligned(16))) float w2t[kEncMaxM * H];  // [M][H] = W2 transposed
    __shared__ float w1s[kEncMaxM * kEncMaxF];
    __shared__ float b1s[kEncMaxM];
    const int tid = threadIdx.x;
    for (int i = tid; i < M * H; i += kEncThreads) w2t[i] = W2[(i % H) * M + (i / H)];
    for (int i = tid; i < M * F; i += kEncThreads) w1s[i] = W1[i];
    for (int i = tid; i < M; i += kEncThreads) b1s[i] = b1[i];
    __syncthreads();

    const int li = tid % LPR, rg = tid / LPR, c = 4 * li;
    const int lane_base = (tid & 63) - li;  // first lane of this row group inside the wave
    const f32x4 bias = *reinterpret_cast<const f32x4*>(b2 + c);

    for (int64_t r0 = (int64_t)blockIdx.x * RPB; r0 < rows; r0 += (int64_t)gridDim.x * RPB) {
        const int64_t r = r0 + rg;
        const bool live = r < rows;
        const int64_t rin = live ? (gather != nullptr ? (int64_t)gather[r] : r) : 0;
        float xin[kEncMaxF];
#pragma unroll
        for (int f = 0; f < kEncMaxF; ++f) xin[f] = (live && f < F) ? in[rin * F + f] : 0.f;
        float t[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int j = li + k * LPR;
            float s = 0.f;
            if (j < M) {   // torch's nn.Linear on the CPU: one k-ascending fma chain from zero, bias added afterwards
#pragma unroll
                for (int f = 0; f < kEncMaxF; ++f)
                    if (f < F) s = __builtin_fmaf(xin[f], w1s[j * F + f], s);
                s = fmaxf(s + b1s[j], 0.f);
            }
            t[k] = s;
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int jn = min(LPR, M - k * LPR);
            for (int jj = 0; jj < jn; ++jj) {
                const float tj = __shfl(t[k], lane_base + jj);
                const f32x4 w = *reinterpret_cast<const f32x4*>(&w2t[(k * LPR + jj) * H + c]);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_fmaf(tj, w[i], acc[i]);
            }
        }
        if (live) *reinterpret_cast<f32x4*>(out + r * H + c) = acc + bias;
    }
}

// Register-only fast path for the reference's shapes (in_features = 2, hidden_ne = 16; hyperparameters.py:23-26):
// each lane keeps its four W2 rows (64 floats) in VGPRs and recomputes the 16 hidden units of its row itself
// (32 FMAs) - no LDS reads, no cross-lane traffic; the general kernel above spends 16 ds_read_b128 + 16
// ds_bpermute per row and tops out near 2 TB/s of output.
template <int H>
__global__ __launch_bounds__(kEncThreads) void k_encode_f2m16(const float* __restrict__ in, int64_t rows,
                                                              const int32_t* __restrict__ gather,
                                                              const float* __restrict__ W1, const float* __restrict__ b1,
                                                              const float* __restrict__ W2, const float* __restrict__ b2,
                                                              float* __restrict__ out) {
    constexpr int LPR = H / 4, RPB = kEncThreads / LPR, M = 16;
    const int tid = threadIdx.x, li = tid % LPR, rg = tid / LPR, c = 4 * li;
    float w1a[M], w1b[M], bb[M];
#pragma unroll
    for (int j = 0; j < M; ++j) {
        w1a[j] = W1[2 * j];
        w1b[j] = W1[2 * j + 1];
        bb[j] = b1[j];
    }
    f32x4 w2[M];  // w2[j][i] = W2[c + i][j]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j4 = 0; j4 < M / 4; ++j4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(W2 + (c + i) * M + 4 * j4);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) w2[4 * j4 + jj][i] = v[jj];
        }
    }
    const f32x4 bias = *reinterpret_cast<const f32x4*>(b2 + c);
    for (int64_t r = (int64_t)blockIdx.x * RPB + rg; r < rows; r += (int64_t)gridDim.x * RPB) {
        const int64_t rin = gather != nullptr ? (int64_t)gather[r] : r;
        const float x0 = in[2 * rin], x1 = in[2 * rin + 1];
        // the reference's order (torch nn.Linear on the CPU: k-ascending fma chain from zero, then + bias; see
        // reference_order.hip) - two more VALU operations per hidden unit than the bias-first form, nothing at this size
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < M; ++j) {
            const float t = fmaxf(__builtin_fmaf(x1, w1b[j], x0 * w1a[j]) + bb[j], 0.f);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_fmaf(t, w2[j][i], acc[i]);
        }
        *reinterpret_cast<f32x4*>(out + r * H + c) = acc + bias;
    }
}

template <int H>
static int launch_encode(const float* in, int64_t rows, int F, const int32_t* gather, const float* W1, const float* b1,
                         int M, const float* W2, const float* b2, float* out, hipStream_t s) {
    constexpr int RPB = kEncThreads / (H / 4);
    int64_t blocks = (rows + RPB - 1) / RPB;
    if (blocks > kNumCUs * 8) blocks = kNumCUs * 8;  // grid-stride beyond 8 blocks per CU
    if (F == 2 && M == 16 && ((uintptr_t)W2 % 16 == 0)) {
        hipLaunchKernelGGL(k_encode_f2m16<H>, dim3((unsigned)blocks), dim3(kEncThreads), 0, s, in, rows, gather, W1, b1, W2, b2,
                           out);
    } else {
        hipLaunchKernelGGL(k_encode<H>, dim3((unsigned)blocks), dim3(kEncThreads), 0, s, in, rows, F, gather, W1, b1, M, W2,
                           b2, out);
    }
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

__global__ __launch_bounds__(256) void k_gather_rows(const float* __restrict__ in, int ld_in,
                                                     const int32_t* __restrict__ idx, int64_t rows, int w4,
                                                     float* __restrict__ out, int ld_out) {
    const int64_t total = rows * w4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / w4;
        const int c = (int)(i % w4) * 4;
        *reinterpret_cast<f32x4*>(out + r * ld_out + c) = *reinterpret_cast<const f32x4*>(in + (int64_t)idx[r] * ld_in + c);
    }
}

// out[idx[r], :] += in[r, :]; idx holds distinct rows, so every output element has exactly one writer
__global__ __launch_bounds__(256) void k_scatter_add_rows(const float* __restrict__ in, int ld_in,
                                                          const int32_t* __restrict__ idx, int64_t rows, int w4,
                                                          float* __restrict__ out, int ld_out) {
    const int64_t total = rows * w4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / w4;
        const int c = (int)(i % w4) * 4;
        f32x4* dst = reinterpret_cast<f32x4*>(out + (int64_t)idx[r] * ld_out + c);
        const f32x4 a = *reinterpret_cast<const f32x4*>(in + r * ld_in + c);
        f32x4 b = *dst;
        b[0] += a[0]; b[1] += a[1]; b[2] += a[2]; b[3] += a[3];
        *dst = b;
    }
}

}  // namespace gnnome

extern "C" int gnnome_encode_f32(const float* in, int64_t rows, int in_features, const int32_t* gather, const float* W1,
                                 const float* b1, int hidden_ne, const float* W2, const float* b2, int hidden, float* out,
                                 void* stream) {
    using namespace gnnome;
    GN_REQUIRE(rows >= 0, "encode: negative row count");
    if (rows == 0) return GNNOME_OK;
    GN_REQUIRE(in && W1 && b1 && W2 && b2 && out, "encode: null pointer");
    GN_REQUIRE(in_features >= 1 && in_features <= kEncMaxF, "encode: in_features=%d not in [1,%d]", in_features, kEncMaxF);
    GN_REQUIRE(hidden_ne >= 1 && hidden_ne <= kEncMaxM, "encode: hidden_ne=%d not in [1,%d]", hidden_ne, kEncMaxM);
    GN_REQUIRE(((uintptr_t)out % 16 == 0) && ((uintptr_t)b2 % 16 == 0), "encode: out and b2 must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    switch (hidden) {
        case 64: return launch_encode<64>(in, rows, in_features, gather, W1, b1, hidden_ne, W2, b2, out, s);
        case 128: return launch_encode<128>(in, rows, in_features, gather, W1, b1, hidden_ne, W2, b2, out, s);
        case 256: return launch_encode<256>(in, rows, in_features, gather, W1, b1, hidden_ne, W2, b2, out, s);
        default: set_error("encode: hidden=%d not in {64,128,256}", hidden); return GNNOME_EINVAL;
    }
}

extern "C" int gnnome_gather_rows_f32(const float* in, int ld_in, const int32_t* idx, int64_t rows, int width, float* out,
                                      int ld_out, void* stream) {
    using namespace gnnome;
    GN_REQUIRE(rows >= 0, "gather_rows: negative row count");
    if (rows == 0) return GNNOME_OK;
    GN_REQUIRE(in && idx && out, "gather_rows: null pointer");
    GN_REQUIRE(width > 0 && width % 4 == 0 && ld_in % 4 == 0 && ld_out % 4 == 0 && ld_in >= width && ld_out >= width,
               "gather_rows: width and strides must be multiples of 4");
    GN_REQUIRE(((uintptr_t)in % 16 == 0) && ((uintptr_t)out % 16 == 0), "gather_rows: 16-byte alignment required");
    const int w4 = width / 4;
    int64_t blocks = (rows * w4 + 255) / 256;
    if (blocks > kNumCUs * 8) blocks = kNumCUs * 8;
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, ld_in, idx, rows, w4,
                       out, ld_out);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

extern "C" int gnnome_scatter_add_rows_f32(const float* in, int ld_in, const int32_t* idx, int64_t rows, int width,
                                           float* out, int ld_out, void* stream) {
    using namespace gnnome;
    GN_REQUIRE(rows >= 0, "scatter_add_rows: negative row count");
    if (rows == 0) return GNNOME_OK;
    GN_REQUIRE(in && idx && out, "scatter_add_rows: null pointer");
    GN_REQUIRE(width > 0 && width % 4 == 0 && ld_in % 4 == 0 && ld_out % 4 == 0 && ld_in >= width && ld_out >= width,
               "scatter_add_rows: width and strides must be multiples of 4");
    GN_REQUIRE(((uintptr_t)in % 16 == 0) && ((uintptr_t)out % 16 == 0), "scatter_add_rows: 16-byte alignment required");
    const int w4 = width / 4;
    int64_t blocks = (rows * w4 + 255) / 256;
    if (blocks > kNumCUs * 8) blocks = kNumCUs * 8;
    hipLaunchKernelGGL(k_scatter_add_rows, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, ld_in, idx, rows,
                       w4, out, ld_out);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}
