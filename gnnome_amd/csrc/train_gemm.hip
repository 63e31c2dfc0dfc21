// Training-step kernels with arithmetic in them: weight-gradient GEMM (reduction over rows), the per-edge
// backward of the score predictor tail and of the gated aggregation, and two tiny helpers for the encoders.
// The backward of the path is restated in gnnome_amd/train.py (autograd of models/full_graph.py:22-30 as
// driven by train.py:138-145, 328-330); each kernel's contract is in include/gnnome_hip.h.
#include "gemm_tile.h"

namespace gnnome {

// ---------------------------------------------------------------------------------------------------
// wgrad:  C[Ka,Kb] = A[R,Ka]^T * B[R,Kb]   (nn.Linear weight gradient dW = dY^T X; also dW of B_3, W1, W2)
// fp32-faithful bf16x6 product (gemm_tile.h) with the ROW index as the MFMA k dimension.  A workgroup owns one
// 128x128 output tile and one contiguous chunk of rows and writes its partial tile; a second kernel adds the
// partials in chunk order, so the result does not depend on scheduling (no float atomics).
// ---------------------------------------------------------------------------------------------------
constexpr int kWgTile = 128, kWgRows = 32, kWgBlocks = 8;
constexpr int kWgColBytes = 2 * kWgRows + 16;        // one column of one bf16 plane: 32 rows + 16 bytes of pad
constexpr int kWgPlane = kWgTile * kWgColBytes;      // 10 KB; six planes (A and B, three each) = 60 KB: two workgroups per CU

// One workgroup = one 128x128 output tile (a whole [H,H] weight at H = 128) x one chunk of rows, so every row of
// A and B is read from HBM exactly once; wave (wi, wj) owns a 64x64 quadrant as 2x2 accumulators.
//
// The MFMA k dimension is the ROW index, so a lane's operand is eight consecutive rows of ONE column.  The threads that
// stage a 32-row slab split it (each element once, not once per wave that uses it) and store it TRANSPOSED as three bf16
// planes [column slot][row]: a fragment is then one ds_read_b128 per plane.  A staging thread holds four columns
// 4 c4 .. 4 c4 + 3 of four consecutive rows (a 16-byte global load per row) and writes 8 bytes per column and plane;
// column 4 c4 + j lives in slot 32 j + c4, which spreads the 32 lanes of a write over the banks (20 c4 mod 64) and makes
// MFMA row/column index m of fragment f the output index 4 m + f.  The next slab's global loads are issued before the
// MFMAs of the current one.
//
// A may be given as up to kWgBlocks column blocks of equal width living in separate buffers (the five [N,H] gradients of a
// layer's node projections, never concatenated): column c is column c % width of block c / width.  With colsum_part set, the
// workgroups of the first tile column also leave the column sums of their rows of A (the bias gradients) - the slab is in
// their registers anyway.
struct WgradA {
    const float* blk[kWgBlocks];
    int width;   // >= Ka for a single buffer
};

template <bool X16, int ABL = 0>   // X16: A is stored as bf16 (the dxe rows of the bf16-storage training step); ABL: measurement only
__global__ __launch_bounds__(256, 2) void k_wgrad_partial(WgradA a_op, int lda, int Ka, const float* __restrict__ B,
                                                          int ldb, int Kb, int64_t R, int64_t rows_per_chunk,
                                                          float* __restrict__ partial, float* __restrict__ colsum_part) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[6 * kWgPlane];
    unsigned char* Ap = lds;
    unsigned char* Bp = lds + 3 * kWgPlane;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i0 = blockIdx.x * kWgTile, j0 = blockIdx.y * kWgTile;
    const int64_t r_begin = (int64_t)blockIdx.z * rows_per_chunk, r_end = min(R, r_begin + rows_per_chunk);
    const int wi = wave & 1, wj = wave >> 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int c4 = tid & 31, rr = tid >> 5;  // 32 float4 per 128-wide row; rows 4 rr .. 4 rr + 3 of the slab
    const bool a_in = i0 + 4 * c4 < Ka, b_in = j0 + 4 * c4 < Kb;   // rows and columns outside the operands contribute zeros
    const float* a_col = a_op.blk[0];
    {
        const int ca = a_in ? i0 + 4 * c4 : 0, which = ca / a_op.width;
#pragma unroll
        for (int k = 1; k < kWgBlocks; ++k)
            if (which == k) a_col = a_op.blk[k];   // a select chain: no dynamic indexing of the kernel arguments
        a_col += ca - which * a_op.width;
    }
    const void* a_base16 = a_op.blk[0];                       // X16: one buffer of bf16, element offsets
    const int64_t a_off16 = a_in ? i0 + 4 * c4 : 0;
    const float* b_col = B + j0 + 4 * c4;
    const bool sums = colsum_part != nullptr && blockIdx.y == 0;
    f32x4 cs = {0.f, 0.f, 0.f, 0.f};
    f32x4 av[4], bv[4];
    auto fetch = [&](int64_t r0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int64_t row = r0 + 4 * rr + t;
            av[t] = bv[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < r_end) {
                if (a_in) av[t] = X16 ? load4_as<true>(a_base16, a_off16 + row * lda) : *reinterpret_cast<const f32x4*>(a_col + row * lda);
                if (b_in) bv[t] = *reinterpret_cast<const f32x4*>(b_col + row * ldb);
            }
        }
    };
    auto stage = [&](unsigned char* planes, const f32x4 (&v)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint2 p1, p2, p3;
            tile_split4(f32x4{v[0][j], v[1][j], v[2][j], v[3][j]}, p1, p2, p3);
            unsigned char* d = planes + (32 * j + c4) * kWgColBytes + 8 * rr;
            *reinterpret_cast<uint2*>(d) = p1;
            *reinterpret_cast<uint2*>(d + kWgPlane) = p2;
            *reinterpret_cast<uint2*>(d + 2 * kWgPlane) = p3;
        }
    };
    auto bf = [](const uint4 v) { return __builtin_bit_cast(tile_bf16x8, v); };
    // fragment f of wave half w: slots 64 w + 32 f + (lane & 31), rows 16 s + 8 (lane >> 5) .. + 7
    const unsigned char* ap = Ap + (64 * wi + (lane & 31)) * kWgColBytes + 16 * (lane >> 5);
    const unsigned char* bp = Bp + (64 * wj + (lane & 31)) * kWgColBytes + 16 * (lane >> 5);
    fetch(r_begin);
    for (int64_t r0 = r_begin; r0 < r_end; r0 += kWgRows) {
        if (!(ABL & 4) || r0 == r_begin) {
            stage(Ap, av);
            stage(Bp, bv);
        }
        if (sums) cs += (av[0] + av[1]) + (av[2] + av[3]);
        __syncthreads();
        if (r0 + kWgRows < r_end && !(ABL & 1)) fetch(r0 + kWgRows);
#pragma unroll
        for (int s = 0; s < ((ABL & 2) ? 0 : kWgRows / 16); ++s) {
            uint4 a1[2], a2[2], a3[2], b1[2], b2[2], b3[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned char* pa = ap + 32 * h * kWgColBytes + 32 * s;
                const unsigned char* pb = bp + 32 * h * kWgColBytes + 32 * s;
                a1[h] = *reinterpret_cast<const uint4*>(pa);
                a2[h] = *reinterpret_cast<const uint4*>(pa + kWgPlane);
                a3[h] = *reinterpret_cast<const uint4*>(pa + 2 * kWgPlane);
                b1[h] = *reinterpret_cast<const uint4*>(pb);
                b2[h] = *reinterpret_cast<const uint4*>(pb + kWgPlane);
                b3[h] = *reinterpret_cast<const uint4*>(pb + 2 * kWgPlane);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    f32x16 c = acc[a][b];   // smallest terms first
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a3[a]), bf(b1[b]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1[a]), bf(b3[b]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a2[a]), bf(b2[b]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a2[a]), bf(b1[b]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1[a]), bf(b2[b]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a1[a]), bf(b1[b]), c, 0, 0, 0);
                    acc[a][b] = c;
                }
        }
        __syncthreads();
    }
    float* out = partial + ((int64_t)blockIdx.z * Ka + i0) * Kb + j0;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = 4 * cd_row(r, lane) + 2 * wi + a, j = 4 * (lane & 31) + 2 * wj + b;
                if (i0 + i < Ka && j0 + j < Kb) out[(int64_t)i * Kb + j] = acc[a][b][r];
            }
    if (sums) {   // the eight row groups of a column, added in a fixed order (the planes are no longer needed: the loop ended on a barrier)
        float* red = reinterpret_cast<float*>(lds);
        *reinterpret_cast<f32x4*>(red + rr * kWgTile + 4 * c4) = cs;
        __syncthreads();
        if (tid < kWgTile && i0 + tid < Ka) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) t += red[k * kWgTile + tid];
            colsum_part[(int64_t)blockIdx.z * Ka + i0 + tid] = t;
        }
    }
}

// C = sum over chunks of the partial tiles, in a fixed order: workgroup = 64 consecutive elements, wave w adds the
// chunks c = w, w+4, ... and the four wave sums are combined in wave order.
// Workgroups past the last element of C do the same for the column sums of A (colsum_part[chunk][Ka] -> colsum[Ka]).
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ partial, int64_t elems, int chunks,
                                                      float* __restrict__ C, int ldc, int Kb, int c_blocks,
                                                      const float* __restrict__ colsum_part, int Ka, float* __restrict__ colsum) {
    __shared__ float part[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if ((int)blockIdx.x >= c_blocks) {
        partial = colsum_part;
        elems = Ka;
        C = colsum;
        ldc = Kb = Ka;
    }
    const int64_t i = (int64_t)(blockIdx.x >= (unsigned)c_blocks ? blockIdx.x - c_blocks : blockIdx.x) * 64 + lane;
    float s = 0.f;
    if (i < elems)
        for (int c = wave; c < chunks; c += 4) s += partial[(int64_t)c * elems + i];
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && i < elems) C[(i / Kb) * ldc + (i % Kb)] = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
}

// ---------------------------------------------------------------------------------------------------
// score predictor tail, backward.  Forward (score_predictor.py:15-16): z2 = relu(W2 z1 + b2), s = W3.z2 + b3.
// Given z1 = relu(..) [E,hs] (saved by the forward) and ds [E] (gathered through srt_eid), one thread per edge:
//   dz2 = ds * W3 * (z2 > 0);  dz1 = (W2^T dz2) * (z1 > 0);  u = ds * z2
// dz1, dz2, u are written out; the weight gradients are column sums / wgrad products of them.
// ---------------------------------------------------------------------------------------------------
template <int HS>
__global__ __launch_bounds__(256) void k_score_tail_bwd(const float* __restrict__ z1, const float* __restrict__ ds,
                                                        const int32_t* __restrict__ srt_eid, int64_t E,
                                                        const float* __restrict__ W2, const float* __restrict__ b2,
                                                        const float* __restrict__ W3, float* __restrict__ dz1,
                                                        float* __restrict__ dz2, float* __restrict__ u) {
    __shared__ float w2s[32 * HS];
    __shared__ float b2s[32], w3s[32];
    for (int i = threadIdx.x; i < 32 * HS; i += blockDim.x) w2s[i] = W2[i];
    if (threadIdx.x < 32) {
        b2s[threadIdx.x] = b2[threadIdx.x];
        w3s[threadIdx.x] = W3[threadIdx.x];
    }
    __syncthreads();
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < E; p += (int64_t)gridDim.x * blockDim.x) {
        const float g = ds[srt_eid != nullptr ? (int64_t)srt_eid[p] : p];
        float z[HS];
#pragma unroll
        for (int j4 = 0; j4 < HS / 4; ++j4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(z1 + p * HS + 4 * j4);
#pragma unroll
            for (int j = 0; j < 4; ++j) z[4 * j4 + j] = v[j];
        }
        float d1[HS];
#pragma unroll
        for (int j = 0; j < HS; ++j) d1[j] = 0.f;
        for (int k = 0; k < 32; ++k) {
            float a = b2s[k];
#pragma unroll
            for (int j = 0; j < HS; ++j) a += w2s[k * HS + j] * z[j];
            const float z2 = fmaxf(a, 0.f);
            const float d2 = a > 0.f ? g * w3s[k] : 0.f;
            dz2[p * 32 + k] = d2;
            u[p * 32 + k] = g * z2;
#pragma unroll
            for (int j = 0; j < HS; ++j) d1[j] += w2s[k * HS + j] * d2;
        }
#pragma unroll
        for (int j4 = 0; j4 < HS / 4; ++j4) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = z[4 * j4 + j] > 0.f ? d1[4 * j4 + j] : 0.f;
            *reinterpret_cast<f32x4*>(dz1 + p * HS + 4 * j4) = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// gated aggregation, backward, per edge (gated_gcn_full.py:111-114,124-127).  With s = sigmoid(e'),
//   hf_i = sum s A2h[src] / (sum s + eps)  =>  d s_p (forward part)  = Tf[dst] * A2h[src] - Uf[dst]
//   hb_i = sum s A3h[dst] / (sum s + eps)  =>  d s_p (backward part) = Tb[src] * A3h[dst] - Ub[src]
// where T = dv * rden and U = T * h are node tables prepared by k_mul23.   de[p] += s (1 - s) (d s_p).
// ---------------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(256) void k_agg_edge_bwd(const float* __restrict__ e, int64_t E, const float* __restrict__ Tf,
                                                      const float* __restrict__ Uf, const float* __restrict__ Tb,
                                                      const float* __restrict__ Ub, const float* __restrict__ A2h,
                                                      const float* __restrict__ A3h, int ldn,
                                                      const int32_t* __restrict__ srt_src, const int32_t* __restrict__ srt_dst,
                                                      float* __restrict__ de) {
    constexpr int LPR = H / 4;
    const int64_t total = E * LPR;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / LPR;
        const int c = (int)(i % LPR) * 4;
        const int64_t s_ = srt_src[p], d_ = srt_dst[p];
        const f32x4 x = *reinterpret_cast<const f32x4*>(e + p * H + c);
        const f32x4 tf = *reinterpret_cast<const f32x4*>(Tf + d_ * H + c), uf = *reinterpret_cast<const f32x4*>(Uf + d_ * H + c);
        const f32x4 tb = *reinterpret_cast<const f32x4*>(Tb + s_ * H + c), ub = *reinterpret_cast<const f32x4*>(Ub + s_ * H + c);
        const f32x4 a2 = *reinterpret_cast<const f32x4*>(A2h + s_ * ldn + c), a3 = *reinterpret_cast<const f32x4*>(A3h + d_ * ldn + c);
        f32x4 g = *reinterpret_cast<const f32x4*>(de + p * H + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float sg = sigmoidf_(x[j]);
            g[j] += sg * (1.f - sg) * (tf[j] * a2[j] - uf[j] + tb[j] * a3[j] - ub[j]);
        }
        *reinterpret_cast<f32x4*>(de + p * H + c) = g;
    }
}

// t[r,:] = relu(W1 * in[row(r),:] + b1)   (hidden activations of an encoder, recomputed for its backward)
__global__ __launch_bounds__(256) void k_encode_hidden(const float* __restrict__ in, int64_t rows, int F,
                                                       const int32_t* __restrict__ gather, const float* __restrict__ W1,
                                                       const float* __restrict__ b1, int M, float* __restrict__ t) {
    const int64_t total = rows * M;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / M;
        const int j = (int)(i % M);
        const int64_t rin = gather != nullptr ? (int64_t)gather[r] : r;
        float s = b1[j];
        for (int f = 0; f < F; ++f) s += W1[j * F + f] * in[rin * F + f];
        t[i] = fmaxf(s, 0.f);
    }
}

// dx = dy * (y > 0)
__global__ __launch_bounds__(256) void k_relu_bwd(const float* __restrict__ dy, const float* __restrict__ y, int64_t n,
                                                  float* __restrict__ dx) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}

static unsigned grid_for_items(int64_t n) {
    int64_t b = (n + 255) / 256;
    if (b < 1) b = 1;
    if (b > kNumCUs * 8) b = kNumCUs * 8;
    return (unsigned)b;
}

}  // namespace gnnome

using namespace gnnome;

// Row chunks of one wgrad call: enough workgroups (output tiles x chunks) to put two on every CU - the LDS footprint admits
// two - but no chunk shorter than 256 rows (each chunk costs a [Ka,Kb] partial tile written and read back).  At 1M rows and
// one output tile that is ~2000 rows per chunk; at 100k rows and five tiles (the [N,5H]^T [N,H] projection gradient) 1024,
// where the old fixed 2048 left one 4-wave workgroup per CU.
static int64_t wgrad_chunks(int64_t rows, int Ka, int Kb, int64_t* rows_per_chunk) {
    const int64_t tiles = (int64_t)((Ka + kWgTile - 1) / kWgTile) * ((Kb + kWgTile - 1) / kWgTile);
    int64_t chunks = (2 * kNumCUs + tiles - 1) / tiles;
    const int64_t most = (rows + 255) / 256;
    if (chunks > most) chunks = most;
    if (chunks > 1024) chunks = 1024;
    if (chunks < 1) chunks = 1;
    int64_t rpc = (rows + chunks - 1) / chunks;
    rpc = (rpc + kWgRows - 1) / kWgRows * kWgRows;
    if (rpc < kWgRows) rpc = kWgRows;
    if (rows_per_chunk) *rows_per_chunk = rpc;
    return rows > 0 ? (rows + rpc - 1) / rpc : 1;
}

extern "C" int gnnome_wgrad_workspace_bytes(int64_t rows, int Ka, int Kb, size_t* bytes_host) {
    GN_REQUIRE(bytes_host && rows >= 0 && Ka > 0 && Kb > 0, "wgrad: bad arguments");
    *bytes_host = (size_t)wgrad_chunks(rows, Ka, Kb, nullptr) * ((size_t)Ka * Kb + Ka) * sizeof(float);   // partial tiles + partial column sums
    return GNNOME_OK;
}

static int wgrad_impl(const WgradA& a_op, int lda, int Ka, const float* B, int ldb, int Kb, int64_t rows, float* C, int ldc,
                      float* colsum, void* workspace, size_t workspace_bytes, void* stream, bool x16 = false) {
    GN_REQUIRE(rows >= 0 && Ka > 0 && Kb > 0 && Ka % 4 == 0 && Kb % 4 == 0, "wgrad: Ka=%d Kb=%d must be positive multiples of 4", Ka, Kb);
    GN_REQUIRE(C && ldc >= Kb, "wgrad: bad output");
    hipStream_t s = (hipStream_t)stream;
    if (rows == 0) {
        for (int i = 0; i < Ka; ++i) GN_HIP(hipMemsetAsync(C + (int64_t)i * ldc, 0, Kb * sizeof(float), s));
        if (colsum) GN_HIP(hipMemsetAsync(colsum, 0, Ka * sizeof(float), s));
        return GNNOME_OK;
    }
    GN_REQUIRE(B && workspace && ldb >= Kb && lda % 4 == 0 && ldb % 4 == 0, "wgrad: bad operands");
    GN_REQUIRE((uintptr_t)B % 16 == 0, "wgrad: A and B must be 16-byte aligned");
    int64_t rpc = 0;
    const int64_t chunks = wgrad_chunks(rows, Ka, Kb, &rpc);
    const size_t need = (size_t)chunks * ((size_t)Ka * Kb + Ka) * sizeof(float);
    if (workspace_bytes < need) {
        set_error("wgrad: workspace %zu < %zu bytes", workspace_bytes, need);
        return GNNOME_EWORKSPACE;
    }
    float* partial = (float*)workspace;
    float* colsum_part = colsum ? partial + (size_t)chunks * Ka * Kb : nullptr;
    const dim3 grid((Ka + kWgTile - 1) / kWgTile, (Kb + kWgTile - 1) / kWgTile, (unsigned)chunks);
    if (x16)
        hipLaunchKernelGGL(k_wgrad_partial<true>, grid, dim3(256), 0, s, a_op, lda, Ka, B, ldb, Kb, rows, rpc, partial, colsum_part);
    else if (tuning(kTuneGateAblation) == 1)
        hipLaunchKernelGGL((k_wgrad_partial<false, 1>), grid, dim3(256), 0, s, a_op, lda, Ka, B, ldb, Kb, rows, rpc, partial, colsum_part);
    else if (tuning(kTuneGateAblation) == 2)
        hipLaunchKernelGGL((k_wgrad_partial<false, 2>), grid, dim3(256), 0, s, a_op, lda, Ka, B, ldb, Kb, rows, rpc, partial, colsum_part);
    else if (tuning(kTuneGateAblation) == 4)
        hipLaunchKernelGGL((k_wgrad_partial<false, 4>), grid, dim3(256), 0, s, a_op, lda, Ka, B, ldb, Kb, rows, rpc, partial, colsum_part);
    else if (tuning(kTuneGateAblation) == 5)
        hipLaunchKernelGGL((k_wgrad_partial<false, 5>), grid, dim3(256), 0, s, a_op, lda, Ka, B, ldb, Kb, rows, rpc, partial, colsum_part);
    else
        hipLaunchKernelGGL(k_wgrad_partial<false>, grid, dim3(256), 0, s, a_op, lda, Ka, B, ldb, Kb, rows, rpc, partial, colsum_part);
    GN_LAUNCH_CHECK();
    const int64_t elems = (int64_t)Ka * Kb;
    const int c_blocks = (int)((elems + 63) / 64), s_blocks = colsum ? (Ka + 63) / 64 : 0;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)(c_blocks + s_blocks)), dim3(256), 0, s, (const float*)partial, elems, (int)chunks, C,
                       ldc, Kb, c_blocks, (const float*)colsum_part, Ka, colsum);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

extern "C" int gnnome_wgrad_f32(const float* A, int lda, int Ka, const float* B, int ldb, int Kb, int64_t rows, float* C,
                                int ldc, void* workspace, size_t workspace_bytes, void* stream) {
    GN_REQUIRE(rows == 0 || (A && lda >= Ka && (uintptr_t)A % 16 == 0), "wgrad: bad operands");
    WgradA a_op = {};
    a_op.blk[0] = A;
    a_op.width = Ka > 0 ? Ka : 1;
    return wgrad_impl(a_op, lda, Ka, B, ldb, Kb, rows, C, ldc, nullptr, workspace, workspace_bytes, stream);
}

// A stored as bf16 ([rows, Ka], row stride lda ELEMENTS, 8-byte aligned rows): the weight gradient of B_3 from the bf16 dxe
extern "C" int gnnome_wgrad_x16(const uint16_t* A, int lda, int Ka, const float* B, int ldb, int Kb, int64_t rows, float* C,
                                int ldc, void* workspace, size_t workspace_bytes, void* stream) {
    GN_REQUIRE(rows == 0 || (A && lda >= Ka && (uintptr_t)A % 8 == 0), "wgrad_x16: bad operands");
    WgradA a_op = {};
    a_op.blk[0] = reinterpret_cast<const float*>(A);
    a_op.width = Ka > 0 ? Ka : 1;
    return wgrad_impl(a_op, lda, Ka, B, ldb, Kb, rows, C, ldc, nullptr, workspace, workspace_bytes, stream, true);
}

extern "C" int gnnome_wgrad_blocks_f32(const float* const* A_blocks, int num_blocks, int block_width, int lda, const float* B, int ldb,
                                       int Kb, int64_t rows, float* C, int ldc, float* colsum, void* workspace, size_t workspace_bytes,
                                       void* stream) {
    GN_REQUIRE(A_blocks && num_blocks >= 1 && num_blocks <= kWgBlocks && block_width > 0 && block_width % 4 == 0 && lda >= block_width,
               "wgrad_blocks: 1..%d blocks of a width that is a multiple of 4", kWgBlocks);
    WgradA a_op = {};
    for (int k = 0; k < num_blocks; ++k) {
        GN_REQUIRE(rows == 0 || (A_blocks[k] && (uintptr_t)A_blocks[k] % 16 == 0), "wgrad_blocks: block %d null or not 16-byte aligned", k);
        a_op.blk[k] = A_blocks[k];
    }
    a_op.width = block_width;
    return wgrad_impl(a_op, lda, num_blocks * block_width, B, ldb, Kb, rows, C, ldc, colsum, workspace, workspace_bytes, stream);
}

extern "C" int gnnome_score_tail_bwd_f32(const float* z1, const float* dscore, const int32_t* srt_eid, int64_t num_edges,
                                         int hidden_edge_scores, const float* W2, const float* b2, const float* W3, float* dz1,
                                         float* dz2, float* u, void* stream) {
    GN_REQUIRE(num_edges >= 0, "score_tail_bwd: negative edge count");
    if (num_edges == 0) return GNNOME_OK;
    GN_REQUIRE(z1 && dscore && W2 && b2 && W3 && dz1 && dz2 && u, "score_tail_bwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(grid_for_items(num_edges)), block(256);
    switch (hidden_edge_scores) {
        case 32: hipLaunchKernelGGL(k_score_tail_bwd<32>, grid, block, 0, s, z1, dscore, srt_eid, num_edges, W2, b2, W3, dz1, dz2, u); break;
        case 64: hipLaunchKernelGGL(k_score_tail_bwd<64>, grid, block, 0, s, z1, dscore, srt_eid, num_edges, W2, b2, W3, dz1, dz2, u); break;
        default: set_error("score_tail_bwd: hidden_edge_scores=%d not in {32,64}", hidden_edge_scores); return GNNOME_EINVAL;
    }
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

extern "C" int gnnome_agg_edge_bwd_f32(const float* e, int64_t num_edges, int hidden, const float* Tf, const float* Uf,
                                       const float* Tb, const float* Ub, const float* A2h, const float* A3h, int ld_node,
                                       const int32_t* srt_src, const int32_t* srt_dst, float* de, void* stream) {
    GN_REQUIRE(num_edges >= 0, "agg_edge_bwd: negative edge count");
    if (num_edges == 0) return GNNOME_OK;
    GN_REQUIRE(e && Tf && Uf && Tb && Ub && A2h && A3h && srt_src && srt_dst && de && ld_node % 4 == 0, "agg_edge_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(grid_for_items(num_edges * (hidden / 4))), block(256);
    switch (hidden) {
        case 64: hipLaunchKernelGGL(k_agg_edge_bwd<64>, grid, block, 0, s, e, num_edges, Tf, Uf, Tb, Ub, A2h, A3h, ld_node, srt_src, srt_dst, de); break;
        case 128: hipLaunchKernelGGL(k_agg_edge_bwd<128>, grid, block, 0, s, e, num_edges, Tf, Uf, Tb, Ub, A2h, A3h, ld_node, srt_src, srt_dst, de); break;
        case 256: hipLaunchKernelGGL(k_agg_edge_bwd<256>, grid, block, 0, s, e, num_edges, Tf, Uf, Tb, Ub, A2h, A3h, ld_node, srt_src, srt_dst, de); break;
        default: set_error("agg_edge_bwd: hidden=%d not in {64,128,256}", hidden); return GNNOME_EINVAL;
    }
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

extern "C" int gnnome_encode_hidden_f32(const float* in, int64_t rows, int in_features, const int32_t* gather, const float* W1,
                                        const float* b1, int hidden_ne, float* t, void* stream) {
    GN_REQUIRE(rows >= 0 && in_features > 0 && hidden_ne > 0, "encode_hidden: bad shape");
    if (rows == 0) return GNNOME_OK;
    GN_REQUIRE(in && W1 && b1 && t, "encode_hidden: null pointer");
    hipLaunchKernelGGL(k_encode_hidden, dim3(grid_for_items(rows * hidden_ne)), dim3(256), 0, (hipStream_t)stream, in, rows,
                       in_features, gather, W1, b1, hidden_ne, t);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

extern "C" int gnnome_relu_bwd_f32(const float* dy, const float* y, int64_t count, float* dx, void* stream) {
    GN_REQUIRE(count >= 0, "relu_bwd: negative count");
    if (count == 0) return GNNOME_OK;
    GN_REQUIRE(dy && y && dx, "relu_bwd: null pointer");
    hipLaunchKernelGGL(k_relu_bwd, dim3(grid_for_items(count)), dim3(256), 0, (hipStream_t)stream, dy, y, count, dx);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}
