// The callers' arithmetic either side of the model call, on the device so that a
// training / scoring harness never leaves it between feature preparation and the optimizer step:
//   * degree features       inference.py:416-420, train.py:112-122
//   * edge features         utils/data_utils.py:31-41 (preprocess_graph)
//   * BCE-with-logits(pos_weight) and the symmetry loss   train.py:103-109, 138-145
//   * TP / TN / FP / FN      utils/metrics.py:6-12
// All HBM-bound single passes; every reduction is deterministic (integer atomics, or per-block partials summed
// in a fixed order).
#include "common.h"

namespace gnnome {
namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 1024;

__device__ __forceinline__ double block_sum(double v, double* scratch /*[kThreads/64]*/) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < kThreads / 64; ++w) t += scratch[w];  // same order in every thread
    return t;
}

__device__ __forceinline__ unsigned long long block_sum_u64(unsigned long long v, unsigned long long* scratch) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    unsigned long long t = 0;
    for (int w = 0; w < kThreads / 64; ++w) t += scratch[w];
    return t;
}

// ---- degree features -------------------------------------------------------------------------------
// sums[0..3] = sum in_deg, sum in_deg^2, sum out_deg, sum out_deg^2: integers, so the atomics are exact and the
// result does not depend on their order.
__global__ __launch_bounds__(kThreads) void k_degree_sums(const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ out_ptr,
                                                          int64_t N, unsigned long long* __restrict__ sums) {
    __shared__ unsigned long long scratch[kThreads / 64];
    unsigned long long s[4] = {0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < N; i += (int64_t)gridDim.x * kThreads) {
        const unsigned long long di = (unsigned long long)(in_ptr[i + 1] - in_ptr[i]);
        const unsigned long long dout = (unsigned long long)(out_ptr[i + 1] - out_ptr[i]);
        s[0] += di;
        s[1] += di * di;
        s[2] += dout;
        s[3] += dout * dout;
    }
    for (int k = 0; k < 4; ++k) {
        const unsigned long long t = block_sum_u64(s[k], scratch);
        if (threadIdx.x == 0 && t) atomicAdd(&sums[k], t);
    }
}

__global__ __launch_bounds__(kThreads) void k_degree_zscore(const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ out_ptr,
                                                            int64_t N, const unsigned long long* __restrict__ sums, int reverse,
                                                            float* __restrict__ x) {
    const double n = (double)N;
    const double mi = (double)sums[0] / n, mo = (double)sums[2] / n;
    // torch.std: unbiased (N-1); N == 1 gives nan there and here
    const double vi = ((double)sums[1] - n * mi * mi) / (n - 1.0), vo = ((double)sums[3] - n * mo * mo) / (n - 1.0);
    const float mean_in = (float)mi, mean_out = (float)mo;
    const float std_in = (float)sqrt(vi > 0.0 ? vi : (vi == vi ? 0.0 : vi)), std_out = (float)sqrt(vo > 0.0 ? vo : (vo == vo ? 0.0 : vo));
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < N; i += (int64_t)gridDim.x * kThreads) {
        const float zi = ((float)(in_ptr[i + 1] - in_ptr[i]) - mean_in) / std_in;
        const float zo = ((float)(out_ptr[i + 1] - out_ptr[i]) - mean_out) / std_out;
        reinterpret_cast<float2*>(x)[i] = reverse ? make_float2(zo, zi) : make_float2(zi, zo);
    }
}

// ---- edge features ---------------------------------------------------------------------------------
// partial[b] = (sum (v - c), sum (v - c)^2) over block b's rows, c = v[0]: shifted sums do not cancel
__global__ __launch_bounds__(kThreads) void k_shifted_partials(const float* __restrict__ v, int64_t E, double* __restrict__ partial) {
    __shared__ double scratch[kThreads / 64];
    const double c = (double)v[0];
    double s1 = 0.0, s2 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < E; i += (int64_t)gridDim.x * kThreads) {
        const double d = (double)v[i] - c;
        s1 += d;
        s2 += d * d;
    }
    const double t1 = block_sum(s1, scratch), t2 = block_sum(s2, scratch);
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = t1;
        partial[2 * blockIdx.x + 1] = t2;
    }
}

// one block: partials -> (mean, unbiased std) in stats[0..1]
__global__ __launch_bounds__(kThreads) void k_finish_stats(const float* __restrict__ v, int64_t E, const double* __restrict__ partial,
                                                           int blocks, float* __restrict__ stats) {
    if (threadIdx.x != 0) return;
    double s1 = 0.0, s2 = 0.0;
    for (int b = 0; b < blocks; ++b) {
        s1 += partial[2 * b];
        s2 += partial[2 * b + 1];
    }
    const double n = (double)E, m = s1 / n;
    const double var = (s2 - n * m * m) / (n - 1.0);
    stats[0] = (float)((double)v[0] + m);
    stats[1] = (float)sqrt(var > 0.0 ? var : (var == var ? 0.0 : var));
}

__global__ __launch_bounds__(kThreads) void k_edge_features(const float* __restrict__ len, const float* __restrict__ sim, int64_t E,
                                                            const float* __restrict__ stats, float* __restrict__ e) {
    const float mean = stats[0], sd = stats[1];
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < E; i += (int64_t)gridDim.x * kThreads)
        reinterpret_cast<float2*>(e)[i] = make_float2((len[i] - mean) / sd, sim[i]);
}

// ---- loss + confusion counts -----------------------------------------------------------------------
// F.binary_cross_entropy_with_logits(x, y, pos_weight=pw, reduction='none'), torch's own stable form:
//   (1 - y) x + (1 + (pw - 1) y) (log1p(exp(-|x|)) + max(-x, 0));   d/dx = (1 + (pw - 1) y) sigmoid(x) - pw y
__device__ __forceinline__ float bce_term(float x, float y, float pw, float& grad) {
    const float w = 1.f + (pw - 1.f) * y;
    const float sig = 1.f / (1.f + expf(-x));
    grad = w * sig - pw * y;
    return (1.f - y) * x + w * (log1pf(expf(-fabsf(x))) + fmaxf(-x, 0.f));
}

template <bool SYM>
__global__ __launch_bounds__(kThreads) void k_edge_loss(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ y,
                                                        int64_t E, const float* __restrict__ pos_weight, float alpha, float gscale,
                                                        float* __restrict__ da, float* __restrict__ db, double* __restrict__ partial,
                                                        unsigned long long* __restrict__ tfpn) {
    __shared__ double scratch[kThreads / 64];
    __shared__ unsigned long long scratch_u[kThreads / 64];
    const float pw = pos_weight[0];
    double loss = 0.0;
    unsigned long long cnt[4] = {0, 0, 0, 0};  // TP, TN, FP, FN of `a`
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < E; i += (int64_t)gridDim.x * kThreads) {
        const float xa = a[i], yi = y[i];
        float ga;
        float l = bce_term(xa, yi, pw, ga);
        if (SYM) {
            const float xb = b[i];
            float gb;
            l += bce_term(xb, yi, pw, gb);
            const float d = xa - xb;
            l += alpha * fabsf(d);
            const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);   // torch.abs' subgradient: sign(0) = 0
            ga += alpha * sgn;
            gb -= alpha * sgn;
            if (db) db[i] = gb * gscale;
        }
        if (da) da[i] = ga * gscale;
        loss += (double)l;
        if (tfpn) {
            // torch.round(torch.sigmoid(x)): half-to-even sends exactly 0.5 to 0, so the prediction is 1 iff the
            // fp32 sigmoid is strictly above one half
            const bool pred = 1.f / (1.f + expf(-xa)) > 0.5f;
            const bool pos = yi == 1.f, neg = yi == 0.f;
            cnt[0] += pred && pos;
            cnt[1] += !pred && neg;
            cnt[2] += pred && neg;
            cnt[3] += !pred && pos;
        }
    }
    const double t = block_sum(loss, scratch);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
    if (tfpn) {
        for (int k = 0; k < 4; ++k) {
            const unsigned long long c = block_sum_u64(cnt[k], scratch_u);
            if (threadIdx.x == 0 && c) atomicAdd(&tfpn[k], c);
        }
    }
}

// one wave: lane l adds partials l, l + 64, ... and the 64 lane sums are folded by a butterfly - a fixed order (one thread walking the
// list alone took 65 us of dependent loads per training step)
__global__ void k_finish_loss(const double* __restrict__ partial, int blocks, int64_t E, float* __restrict__ loss) {
    if (blockIdx.x != 0 || threadIdx.x >= 64) return;
    double s = 0.0;
    for (int b = threadIdx.x; b < blocks; b += 64) s += partial[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) loss[0] = (float)(s / (double)E);
}

int grid_for(int64_t rows) {
    const int64_t b = (rows + kThreads - 1) / kThreads;
    return (int)(b < 1 ? 1 : (b > kMaxBlocks ? kMaxBlocks : b));
}

}  // namespace
}  // namespace gnnome

extern "C" int gnnome_closure_workspace_bytes(size_t* bytes_host) {
    using namespace gnnome;
    GN_REQUIRE(bytes_host != nullptr, "closure_workspace_bytes: null pointer");
    *bytes_host = (size_t)(2 * kMaxBlocks + 8) * sizeof(double);
    return GNNOME_OK;
}

extern "C" int gnnome_degree_features_f32(const int32_t* in_ptr, const int32_t* out_ptr, int64_t num_nodes, int reverse, float* x,
                                          void* workspace, size_t workspace_bytes, void* stream) {
    using namespace gnnome;
    GN_REQUIRE(num_nodes >= 0, "degree_features: negative node count");
    if (num_nodes == 0) return GNNOME_OK;
    GN_REQUIRE(in_ptr && out_ptr && x && workspace, "degree_features: null pointer");
    GN_REQUIRE(workspace_bytes >= 4 * sizeof(unsigned long long) && (uintptr_t)workspace % 8 == 0 && (uintptr_t)x % 8 == 0,
               "degree_features: workspace too small or misaligned");
    hipStream_t s = (hipStream_t)stream;
    unsigned long long* sums = (unsigned long long*)workspace;
    GN_HIP(hipMemsetAsync(sums, 0, 4 * sizeof(unsigned long long), s));
    const int grid = grid_for(num_nodes);
    hipLaunchKernelGGL(k_degree_sums, dim3(grid), dim3(kThreads), 0, s, in_ptr, out_ptr, num_nodes, sums);
    GN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_degree_zscore, dim3(grid), dim3(kThreads), 0, s, in_ptr, out_ptr, num_nodes, sums, reverse, x);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

extern "C" int gnnome_edge_features_f32(const float* overlap_length, const float* overlap_similarity, int64_t num_edges, float* e,
                                        void* workspace, size_t workspace_bytes, void* stream) {
    using namespace gnnome;
    GN_REQUIRE(num_edges >= 0, "edge_features: negative edge count");
    if (num_edges == 0) return GNNOME_OK;
    GN_REQUIRE(overlap_length && overlap_similarity && e && workspace, "edge_features: null pointer");
    size_t need = 0;
    gnnome_closure_workspace_bytes(&need);
    GN_REQUIRE(workspace_bytes >= need && (uintptr_t)workspace % 8 == 0 && (uintptr_t)e % 8 == 0,
               "edge_features: workspace too small or misaligned (gnnome_closure_workspace_bytes)");
    hipStream_t s = (hipStream_t)stream;
    double* partial = (double*)workspace;
    float* stats = (float*)(partial + 2 * kMaxBlocks);
    const int grid = grid_for(num_edges);
    hipLaunchKernelGGL(k_shifted_partials, dim3(grid), dim3(kThreads), 0, s, overlap_length, num_edges, partial);
    GN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_finish_stats, dim3(1), dim3(kThreads), 0, s, overlap_length, num_edges, partial, grid, stats);
    GN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_edge_features, dim3(grid), dim3(kThreads), 0, s, overlap_length, overlap_similarity, num_edges, stats, e);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

extern "C" int gnnome_edge_loss_f32(const float* logits, const float* logits_rev, const float* labels, int64_t num_edges,
                                    const float* pos_weight, float alpha, float grad_scale, float* loss, float* dlogits,
                                    float* dlogits_rev, int64_t* tfpn, void* workspace, size_t workspace_bytes, void* stream) {
    using namespace gnnome;
    GN_REQUIRE(num_edges > 0, "edge_loss: needs at least one edge (the mean of an empty loss is undefined)");
    GN_REQUIRE(logits && labels && pos_weight && loss && workspace, "edge_loss: null pointer");
    GN_REQUIRE(logits_rev != nullptr || dlogits_rev == nullptr, "edge_loss: dlogits_rev without logits_rev");
    size_t need = 0;
    gnnome_closure_workspace_bytes(&need);
    GN_REQUIRE(workspace_bytes >= need && (uintptr_t)workspace % 8 == 0, "edge_loss: workspace too small or misaligned");
    hipStream_t s = (hipStream_t)stream;
    double* partial = (double*)workspace;
    if (tfpn) GN_HIP(hipMemsetAsync(tfpn, 0, 4 * sizeof(int64_t), s));
    const int grid = grid_for(num_edges);
    if (logits_rev) {
        hipLaunchKernelGGL((k_edge_loss<true>), dim3(grid), dim3(kThreads), 0, s, logits, logits_rev, labels, num_edges, pos_weight,
                           alpha, grad_scale, dlogits, dlogits_rev, partial, (unsigned long long*)tfpn);
    } else {
        hipLaunchKernelGGL((k_edge_loss<false>), dim3(grid), dim3(kThreads), 0, s, logits, logits_rev, labels, num_edges, pos_weight,
                           alpha, grad_scale, dlogits, dlogits_rev, partial, (unsigned long long*)tfpn);
    }
    GN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_finish_loss, dim3(1), dim3(64), 0, s, partial, grid, num_edges, loss);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}
