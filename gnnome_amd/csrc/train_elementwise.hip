// Training-step support kernels that are pure streaming work (HBM-bound): BatchNorm statistics and
// application, BatchNorm backward, segment sums over the graph views.
//
// Reference semantics: torch.nn.BatchNorm1d in training mode as used at gated_gcn_full.py:106,119 (bn_e
// over all E edge rows, called twice per layer) and :132 (bn_h over all N node rows); autograd through
// relu / residual (:107-110, :134-137).  The layer's backward is restated in gnnome_amd/train.py; these
// kernels are its building blocks.  All reductions over rows are per-channel column sums.
#include "common.h"

namespace gnnome {

constexpr int kEwThreads = 256;

// Column reduction skeleton: thread t owns float4 column group (t % (H/4)) and walks rows r = r0 + k*stride;
// per-thread partials are combined through LDS into one value per column per workgroup, parked in the workspace
// ([NACC][H][kColMaxBlocks]); k_col_finish then adds the workgroups' values up in a FIXED order.  No floating-point
// atomics: their order, and with it the last bits of every BatchNorm statistic and bias gradient, would change from
// launch to launch.
constexpr int kColMaxBlocks = 1024;
constexpr size_t kColWorkspaceBytes = sizeof(float) * kColMaxBlocks * 2 * 256;

// The workgroup's column sums of the threads' accumulators -> part[a][c][workgroup] (thread t holds columns 4 (t % (H/4)) .. + 3 of its rows)
template <int NACC>
__device__ __forceinline__ void column_fold(const f32x4 (&acc)[NACC], int H, float* part) {
    __shared__ float red[NACC][kEwThreads * 4];
    const int lpr = H / 4, tid = threadIdx.x;
    const int c4 = tid % lpr, rsub = tid / lpr, rpb = kEwThreads / lpr;
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[a][(4 * c4 + j) * rpb + rsub] = acc[a][j];
    __syncthreads();
    for (int c = tid; c < H; c += kEwThreads) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            float s = 0.f;
            for (int k = 0; k < rpb; ++k) s += red[a][c * rpb + k];
            part[((int64_t)a * H + c) * kColMaxBlocks + blockIdx.x] = s;   // [a][c][workgroup]: the finish reads along b
        }
    }
}

// The rows thread t of workgroup b visits: first row, end, stride (see column_reduce)
struct RowWalk { int64_t r, r1, step; };
__device__ __forceinline__ RowWalk column_walk(int64_t rows, int H) {
    const int lpr = H / 4, rsub = threadIdx.x / lpr, rpb = kEwThreads / lpr;
    if (gridDim.x % kXcds == 0) {
        const int64_t per_xcd = ((rows + kXcds - 1) / kXcds + rpb - 1) / rpb * rpb;
        const int64_t r0 = (int64_t)(blockIdx.x % kXcds) * per_xcd;
        return RowWalk{r0 + (int64_t)(blockIdx.x / kXcds) * rpb + rsub, min(rows, r0 + per_xcd), (int64_t)(gridDim.x / kXcds) * rpb};
    }
    return RowWalk{(int64_t)blockIdx.x * rpb + rsub, rows, (int64_t)gridDim.x * rpb};
}

template <int NACC, class F>
__device__ __forceinline__ void column_reduce(int64_t rows, int H, float* part, F&& per_row) {
    const int lpr = H / 4, tid = threadIdx.x;
    const int c4 = tid % lpr;
    f32x4 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Round 5: every XCD (workgroup b runs on XCD b % 8, each with a private L2) walks its OWN contiguous eighth of the rows.  With the plain
    // grid-stride walk the 8 row groups of a sweep went to 8 different XCDs, so a kernel that gathers node rows by the rows' endpoints
    // (k_agg_edge_bwd_stats: six tables) pulled every table row into up to eight L2s: 3.15 GB fetched per launch at configs[2] against
    // 1.85 GB of operands (profiles/r05_train_pmc_c2.json).  The partial sums stay per workgroup and are added in workgroup order as before.
    const RowWalk w = column_walk(rows, H);
    for (int64_t r = w.r; r < w.r1; r += w.step) per_row(r, 4 * c4, acc);
    column_fold<NACC>(acc, H, part);
}

// out[a][c] = sum_b part[a][c][b] in a FIXED order: one wave per (a, c) pair, lane l adds b = l, l + 64, ... and
// the 64 lane sums are folded by a butterfly (the same tree on every launch).
__global__ __launch_bounds__(256) void k_col_finish(const float* __restrict__ part, int nblocks, int H, float* __restrict__ s1,
                                                    float* __restrict__ s2) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);   // flattened (a, c)
    const float* p = part + (int64_t)j * kColMaxBlocks;
    float s = 0.f;
    int b = lane;
    for (; b + 7 * 64 < nblocks; b += 8 * 64) {   // eight requests in flight, added in the order of the plain loop (same bits)
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[b + 64 * u];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; b < nblocks; b += 64) s += p[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) {
        float* out = j < H ? s1 : s2;
        out[j < H ? j : j - H] = s;
    }
}

// s1[c] = sum_r x'[r,c];  s2[c] = sum_r x'[r,c] * y'[r,c],  x' = x - center[c] (center NULL: 0), y' likewise when
// y aliases x.  With center = the column mean this is the second pass of a two-pass variance: sum (x-mean)^2 has
// none of the cancellation of sum x^2 / n - mean^2 (the edge state reaches |e| ~ 500 with a spread of a few units).
__global__ __launch_bounds__(kEwThreads) void k_colsum2(const float* __restrict__ x, const float* __restrict__ y, int64_t rows,
                                                        int H, const float* __restrict__ center, float* __restrict__ part) {
    const bool same = x == y;
    column_reduce<2>(rows, H, part, [&](int64_t r, int c, f32x4 (&acc)[2]) {
        f32x4 xv = *reinterpret_cast<const f32x4*>(x + r * H + c);
        f32x4 yv = *reinterpret_cast<const f32x4*>(y + r * H + c);
        if (center != nullptr) {
            const f32x4 cv = *reinterpret_cast<const f32x4*>(center + c);
            xv -= cv;
            if (same) yv -= cv;
        }
        acc[0] += xv;
        acc[1] += xv * yv;
    });
}

// BatchNorm backward statistics through the relu:  m = (x*scale + shift > 0), the forward's own expression
// (recovering the mask from out - res would lose activations smaller than an ulp of the residual)
//   s1[c] += sum_r dy*m ;  s2[c] += sum_r dy*m*(x - mean[c])
__global__ __launch_bounds__(kEwThreads) void k_bn_bwd_stats(const float* __restrict__ dy, const float* __restrict__ x,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             const float* __restrict__ mean, int64_t rows, int H,
                                                             float* __restrict__ part) {
    column_reduce<2>(rows, H, part, [&](int64_t r, int c, f32x4 (&acc)[2]) {
        const f32x4 d = *reinterpret_cast<const f32x4*>(dy + r * H + c);
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + r * H + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float g = (xv[j] * scale[c + j] + shift[c + j] > 0.f) ? d[j] : 0.f;
            acc[0][j] += g;
            acc[1][j] += g * (xv[j] - mean[c + j]);
        }
    });
}

// dx = a[c] * (dy*m - c1[c] - (x - mean[c]) * rstd[c] * c2[c]),  m = (x*scale + shift > 0)
__global__ __launch_bounds__(kEwThreads) void k_bn_bwd_apply(const float* __restrict__ dy, const float* __restrict__ x,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             int64_t rows, int H, const float* __restrict__ a,
                                                             const float* __restrict__ c1, const float* __restrict__ c2,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             float* __restrict__ dx) {
    const int64_t total = rows * (H / 4);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % (H / 4)) * 4;
        const int64_t off = (i / (H / 4)) * H + c;
        const f32x4 d = *reinterpret_cast<const f32x4*>(dy + off);
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + off);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float g = (xv[j] * scale[c + j] + shift[c + j] > 0.f) ? d[j] : 0.f;
            o[j] = a[c + j] * (g - c1[c + j] - (xv[j] - mean[c + j]) * rstd[c + j] * c2[c + j]);
        }
        *reinterpret_cast<f32x4*>(dx + off) = o;
    }
}

// k_bn_bwd_apply for the node BatchNorm of a layer WITH the four node tables of the aggregation's backward (the two k_mul23 launches that
// followed it, each reading dx again):  Tf = dx rdf, Uf = Tf hf, Tb = dx rdb, Ub = Tb hb  - one pass over the [rows, H] tensors.
__global__ __launch_bounds__(kEwThreads) void k_bn_bwd_apply_tables(const float* __restrict__ dy, const float* __restrict__ x,
                                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                                    int64_t rows, int H, const float* __restrict__ a,
                                                                    const float* __restrict__ c1, const float* __restrict__ c2,
                                                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                    const float* __restrict__ rdf, const float* __restrict__ hf,
                                                                    const float* __restrict__ rdb, const float* __restrict__ hb,
                                                                    float* __restrict__ dx, float* __restrict__ Tf, float* __restrict__ Uf,
                                                                    float* __restrict__ Tb, float* __restrict__ Ub, unsigned* __restrict__ amax_bits) {
    float amax = 0.f;   // (of dx, for the consumers that scale it into fp16's range)
    const int64_t total = rows * (H / 4);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % (H / 4)) * 4;
        const int64_t off = (i / (H / 4)) * H + c;
        const f32x4 d = *reinterpret_cast<const f32x4*>(dy + off);
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + off);
        const f32x4 r1 = *reinterpret_cast<const f32x4*>(rdf + off), h1 = *reinterpret_cast<const f32x4*>(hf + off);
        const f32x4 r2 = *reinterpret_cast<const f32x4*>(rdb + off), h2 = *reinterpret_cast<const f32x4*>(hb + off);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float g = (xv[j] * scale[c + j] + shift[c + j] > 0.f) ? d[j] : 0.f;
            o[j] = a[c + j] * (g - c1[c + j] - (xv[j] - mean[c + j]) * rstd[c + j] * c2[c + j]);
        }
        const f32x4 t1 = o * r1, t2 = o * r2;
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
        *reinterpret_cast<f32x4*>(dx + off) = o;
        *reinterpret_cast<f32x4*>(Tf + off) = t1;
        *reinterpret_cast<f32x4*>(Uf + off) = t1 * h1;
        *reinterpret_cast<f32x4*>(Tb + off) = t2;
        *reinterpret_cast<f32x4*>(Ub + off) = t2 * h2;
    }
    if (amax_bits != nullptr) wave_amax_to(amax_bits, amax);
}

// out = relu(x * scale + shift) + res
template <bool X16>
__global__ __launch_bounds__(kEwThreads) void k_bn_relu_res(const void* __restrict__ x, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, const float* __restrict__ res,
                                                            int64_t rows, int H, float* __restrict__ out) {
    const int64_t total = rows * (H / 4);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % (H / 4)) * 4;
        const int64_t off = (i / (H / 4)) * H + c;
        const f32x4 xv = load4_as<X16>(x, off);
        const f32x4 rv = *reinterpret_cast<const f32x4*>(res + off);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = fmaxf(xv[j] * scale[c + j] + shift[c + j], 0.f) + rv[j];
        *reinterpret_cast<f32x4*>(out + off) = o;
    }
}

// ---- LayerNorm (normalization='layer', gated_gcn_full.py:40-42): per-ROW statistics, so training needs no global
// reduction in the forward; the backward's d gamma / d beta are column sums like BatchNorm's.
// A row is held by H/4 consecutive lanes (one float4 each); lpr_sum adds over them.
__device__ __forceinline__ float lpr_sum(float v, int lpr) {
    for (int m = 1; m < lpr; m <<= 1) v += __shfl_xor(v, m);
    return v;
}

// out = relu(LayerNorm(x) * gamma + beta) + res ; LayerNorm over the H entries of each row, biased variance, eps 1e-5
__global__ __launch_bounds__(kEwThreads) void k_ln_relu_res(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* __restrict__ res,
                                                            int64_t rows, int H, int W, float* __restrict__ out) {
    // W <= H: the statistics run over the first W channels (a zero-padded narrower model: the padded channels hold exact zeros)
    const int lpr = H / 4, c = (threadIdx.x % lpr) * 4, rpb = kEwThreads / lpr;
    const float inv_w = 1.0f / (float)W;
    // whole rows only: a row never straddles two passes, idle lanes past the end still take part in the shuffles
    const int64_t passes = (rows + rpb - 1) / rpb;
    for (int64_t pass = blockIdx.x; pass < passes; pass += gridDim.x) {
        const int64_t r = pass * rpb + threadIdx.x / lpr;
        const bool live = r < rows;
        const int64_t off = (live ? r : 0) * H + c;
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + off);
        const float mean = lpr_sum(xv[0] + xv[1] + xv[2] + xv[3], lpr) * inv_w;
        float s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) s2 += (c + j < W) ? (xv[j] - mean) * (xv[j] - mean) : 0.f;
        const float rstd = rsqrtf(lpr_sum(s2, lpr) * inv_w + kNormEps);
        if (live) {
            const f32x4 rv = *reinterpret_cast<const f32x4*>(res + off);
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = fmaxf((xv[j] - mean) * rstd * gamma[c + j] + beta[c + j], 0.f) + rv[j];
            *reinterpret_cast<f32x4*>(out + off) = o;
        }
    }
}

// Backward of out = relu(LayerNorm(x) * gamma + beta) + res with respect to x, gamma, beta:
//   xhat = (x - mean_row) rstd_row,  m = (xhat gamma + beta > 0),  g = dy m gamma
//   dx = rstd_row (g - mean_row(g) - xhat mean_row(g xhat));   d beta[c] += sum_r dy m,  d gamma[c] += sum_r dy m xhat
// (column sums deterministic: per-workgroup partials + k_col_finish, as for BatchNorm)
__global__ __launch_bounds__(kEwThreads) void k_ln_bwd(const float* __restrict__ dy, const float* __restrict__ x,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, int64_t rows,
                                                       int H, int W, float* __restrict__ dx, float* __restrict__ part) {
    __shared__ float red[2][kEwThreads * 4];
    const int lpr = H / 4, tid = threadIdx.x, c4 = tid % lpr, rsub = tid / lpr, rpb = kEwThreads / lpr, c = 4 * c4;
    const float inv_w = 1.0f / (float)W;   // (W <= H: see k_ln_relu_res; a padded channel gets dx = 0)
    f32x4 acc_b = {0.f, 0.f, 0.f, 0.f}, acc_g = acc_b;
    const int64_t passes = (rows + rpb - 1) / rpb;
    for (int64_t pass = blockIdx.x; pass < passes; pass += gridDim.x) {
        const int64_t r = pass * rpb + rsub;
        const bool live = r < rows;
        const int64_t off = (live ? r : 0) * H + c;
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + off);
        const f32x4 dv = *reinterpret_cast<const f32x4*>(dy + off);
        const float mean = lpr_sum(xv[0] + xv[1] + xv[2] + xv[3], lpr) * inv_w;
        float s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) s2 += (c + j < W) ? (xv[j] - mean) * (xv[j] - mean) : 0.f;
        const float rstd = rsqrtf(lpr_sum(s2, lpr) * inv_w + kNormEps);
        f32x4 xh, g;
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            xh[j] = (c + j < W) ? (xv[j] - mean) * rstd : 0.f;
            const float dm = (xh[j] * gamma[c + j] + beta[c + j] > 0.f) ? dv[j] : 0.f;
            g[j] = dm * gamma[c + j];
            sg += g[j];
            sgx += g[j] * xh[j];
            if (live) {
                acc_b[j] += dm;
                acc_g[j] += dm * xh[j];
            }
        }
        const float mg = lpr_sum(sg, lpr) * inv_w, mgx = lpr_sum(sgx, lpr) * inv_w;
        if (live) {
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (c + j < W) ? rstd * (g[j] - mg - xh[j] * mgx) : 0.f;
            *reinterpret_cast<f32x4*>(dx + off) = o;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        red[0][(c + j) * rpb + rsub] = acc_b[j];
        red[1][(c + j) * rpb + rsub] = acc_g[j];
    }
    __syncthreads();
    for (int cc = tid; cc < H; cc += kEwThreads) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            float sum = 0.f;
            for (int k = 0; k < rpb; ++k) sum += red[a][cc * rpb + k];
            part[((int64_t)a * H + cc) * kColMaxBlocks + blockIdx.x] = sum;
        }
    }
}

// out = a + b
__global__ __launch_bounds__(kEwThreads) void k_add(const float* __restrict__ a, const float* __restrict__ b, int64_t n4,
                                                    float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
        reinterpret_cast<f32x4*>(out)[i] = reinterpret_cast<const f32x4*>(a)[i] + reinterpret_cast<const f32x4*>(b)[i];
}

// o1 = a*b ; o2 = a*b*c   (node-sized helper of the aggregation backward)
__global__ __launch_bounds__(kEwThreads) void k_mul23(const float* __restrict__ a, const float* __restrict__ b,
                                                      const float* __restrict__ c, int64_t n4, float* __restrict__ o1,
                                                      float* __restrict__ o2) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 av = reinterpret_cast<const f32x4*>(a)[i], bv = reinterpret_cast<const f32x4*>(b)[i];
        const f32x4 cv = reinterpret_cast<const f32x4*>(c)[i];
        const f32x4 t = av * bv;
        reinterpret_cast<f32x4*>(o1)[i] = t;
        reinterpret_cast<f32x4*>(o2)[i] = t * cv;
    }
}

// out[i,:] = sum_{q in [ptr[i], ptr[i+1])} X[pos ? pos[q] : q, :]   one wave per node, W/4 lanes per row
template <int W>
__global__ __launch_bounds__(256) void k_segment_sum(const float* __restrict__ X, const int32_t* __restrict__ ptr,
                                                     const int32_t* __restrict__ pos, int64_t n_nodes, float* __restrict__ out,
                                                     int ld_out) {
    constexpr int LPR = W / 4, G = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int64_t node = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (node >= n_nodes) return;
    const int group = lane / LPR, c = (lane % LPR) * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int b = ptr[node], e = ptr[node + 1];
    for (int q = b + group; q < e; q += G) {
        const int64_t p = pos != nullptr ? pos[q] : q;
        acc += *reinterpret_cast<const f32x4*>(X + p * W + c);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int m = LPR; m < 64; m <<= 1) acc[j] += __shfl_xor(acc[j], m);
    if (group == 0) *reinterpret_cast<f32x4*>(out + node * ld_out + c) = acc;
}

// Both segment sums of one [E,W] tensor in one launch: out_in[i,:] = sum over the in-edge rows of node i (a contiguous run of
// sorted positions), out_out[i,:] = sum over its out-edge rows (reached through out_pos).  One wave per node, as the
// aggregation kernel walks the same two lists: the out-edge pass finds most of its rows in L2, where the in-edge pass of
// a neighbouring node has just put them - two separate launches stream X from HBM twice.
template <int W, bool X16>
__global__ __launch_bounds__(256) void k_segment_sum2(const void* __restrict__ X, const int32_t* __restrict__ in_ptr,
                                                      const int32_t* __restrict__ out_ptr, const int32_t* __restrict__ out_pos,
                                                      int64_t n_nodes, float* __restrict__ out_in, int ld_in, float* __restrict__ out_out,
                                                      int ld_out, int total_blocks, unsigned* __restrict__ amax_bits) {
    constexpr int LPR = W / 4, G = 64 / LPR, U = 4;
    const int lane = threadIdx.x & 63;
    const int64_t node = (int64_t)xcd_remap(blockIdx.x, total_blocks) * 4 + (threadIdx.x >> 6);
    if (node >= n_nodes) return;
    const int group = lane / LPR, c = (lane % LPR) * 4;
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    // all four CSR pointers and the positions of the first 64 out-edges (one per lane) are requested up front: the out-edge pass
    // then starts without its own pointer -> position -> row chain of round trips (same sums in the same order)
    const int b_in = in_ptr[node], e_in = in_ptr[node + 1], b_out = out_ptr[node], e_out = out_ptr[node + 1];
    const int my_pos = b_out + lane < e_out ? out_pos[b_out + lane] : 0;
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const int b = side == 0 ? b_in : b_out, e = side == 0 ? e_in : e_out;
        for (int q0 = b; q0 < e; q0 += G * U) {
            f32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = q0 + u * G + group;
                int64_t p = 0;
                if (side == 0) {
                    p = q < e ? q : 0;
                } else {
                    const int k = q - b;   // position in the node's out-list
                    const int near = __shfl(my_pos, k < 64 ? k : 0);
                    p = q < e ? (k < 64 ? near : out_pos[q]) : 0;
                }
                v[u] = q < e ? load4_as<X16>(X, p * W + c) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc[side] += v[u];
        }
    }
#pragma unroll
    for (int side = 0; side < 2; ++side)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int m = LPR; m < 64; m <<= 1) acc[side][j] += __shfl_xor(acc[side][j], m);
    if (group == 0) {
        *reinterpret_cast<f32x4*>(out_in + node * ld_in + c) = acc[0];
        *reinterpret_cast<f32x4*>(out_out + node * ld_out + c) = acc[1];
    }
    if (amax_bits != nullptr) {   // max |.| of both outputs, for the consumers that scale them into fp16's range
        float amax = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fmaxf(fabsf(acc[0][j]), fabsf(acc[1][j])));
        wave_amax_to(amax_bits, amax);
    }
}

static unsigned ew_grid(int64_t work_items) {
    int64_t b = (work_items + kEwThreads - 1) / kEwThreads;
    if (b < 1) b = 1;
    if (b > kNumCUs * 8) b = kNumCUs * 8;
    return (unsigned)b;
}

static bool ok_width(int H) { return H == 16 || H == 32 || H == 64 || H == 128 || H == 256; }

}  // namespace gnnome

using namespace gnnome;

static unsigned col_grid(int64_t rows, int hidden) {
    const int rpb = kEwThreads / (hidden / 4);
    unsigned g = ew_grid((rows + rpb - 1) / rpb * kEwThreads / 4);
    g = g > (unsigned)kColMaxBlocks ? (unsigned)kColMaxBlocks : g;
    return g >= 2 * kXcds ? g / kXcds * kXcds : g;   // whole multiples of the XCD count: column_reduce gives every XCD its own row range
}

extern "C" int gnnome_colsum_workspace_bytes(size_t* bytes_host) {
    GN_REQUIRE(bytes_host != nullptr, "colsum_workspace_bytes: null pointer");
    *bytes_host = kColWorkspaceBytes;
    return GNNOME_OK;
}

extern "C" int gnnome_colsum2_f32(const float* x, const float* y, int64_t rows, int hidden, const float* center, float* s1,
                                  float* s2, void* workspace, size_t workspace_bytes, void* stream) {
    GN_REQUIRE(rows >= 0 && ok_width(hidden), "colsum2: hidden=%d not in {16,32,64,128,256}", hidden);
    if (rows == 0) return GNNOME_OK;
    GN_REQUIRE(x && s1 && s2, "colsum2: null pointer");
    GN_REQUIRE(workspace && workspace_bytes >= kColWorkspaceBytes && (uintptr_t)workspace % 16 == 0,
               "colsum2: workspace too small or misaligned (gnnome_colsum_workspace_bytes)");
    const unsigned grid = col_grid(rows, hidden);
    hipLaunchKernelGGL(k_colsum2, dim3(grid), dim3(kEwThreads), 0, (hipStream_t)stream, x, y ? y : x, rows, hidden, center,
                       (float*)workspace);
    GN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_col_finish, dim3(2 * hidden / 4), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, (int)grid,
                       hidden, s1, s2);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

extern "C" int gnnome_bn_bwd_stats_f32(const float* dy, const float* x, const float* scale, const float* shift,
                                       const float* mean, int64_t rows, int hidden, float* s1, float* s2, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    GN_REQUIRE(rows >= 0 && ok_width(hidden), "bn_bwd_stats: hidden=%d not in {16,32,64,128,256}", hidden);
    if (rows == 0) return GNNOME_OK;
    GN_REQUIRE(dy && x && scale && shift && mean && s1 && s2, "bn_bwd_stats: null pointer");
    GN_REQUIRE(workspace && workspace_bytes >= kColWorkspaceBytes && (uintptr_t)workspace % 16 == 0,
               "bn_bwd_stats: workspace too small or misaligned (gnnome_colsum_workspace_bytes)");
    const unsigned grid = col_grid(rows, hidden);
    hipLaunchKernelGGL(k_bn_bwd_stats, dim3(grid), dim3(kEwThreads), 0, (hipStream_t)stream, dy, x, scale, shift, mean, rows, hidden,
                       (float*)workspace);
    GN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_col_finish, dim3(2 * hidden / 4), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, (int)grid,
                       hidden, s1, s2);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

extern "C" int gnnome_bn_bwd_apply_f32(const float* dy, const float* x, const float* scale, const float* shift, int64_t rows,
                                       int hidden, const float* a, const float* c1, const float* c2, const float* mean,
                                       const float* rstd, float* dx, void* stream) {
    GN_REQUIRE(rows >= 0 && hidden > 0 && hidden % 4 == 0, "bn_bwd_apply: bad shape");
    if (rows == 0) return GNNOME_OK;
    GN_REQUIRE(dy && x && scale && shift && a && c1 && c2 && mean && rstd && dx, "bn_bwd_apply: null pointer");
    hipLaunchKernelGGL(k_bn_bwd_apply, dim3(ew_grid(rows * (hidden / 4))), dim3(kEwThreads), 0, (hipStream_t)stream, dy, x, scale,
                       shift, rows, hidden, a, c1, c2, mean, rstd, dx);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

extern "C" int gnnome_bn_bwd_apply_tables_f32(const float* dy, const float* x, const float* scale, const float* shift, int64_t rows,
                                              int hidden, const float* a, const float* c1, const float* c2, const float* mean,
                                              const float* rstd, const float* rdf, const float* hf, const float* rdb, const float* hb,
                                              float* dx, float* Tf, float* Uf, float* Tb, float* Ub, unsigned* amax_bits, void* stream) {
    GN_REQUIRE(rows >= 0 && hidden > 0 && hidden % 4 == 0, "bn_bwd_apply_tables: bad shape");
    if (rows == 0) return GNNOME_OK;
    GN_REQUIRE(dy && x && scale && shift && a && c1 && c2 && mean && rstd && rdf && hf && rdb && hb && dx && Tf && Uf && Tb && Ub,
               "bn_bwd_apply_tables: null pointer");
    hipLaunchKernelGGL(k_bn_bwd_apply_tables, dim3(ew_grid(rows * (hidden / 4))), dim3(kEwThreads), 0, (hipStream_t)stream, dy, x, scale,
                       shift, rows, hidden, a, c1, c2, mean, rstd, rdf, hf, rdb, hb, dx, Tf, Uf, Tb, Ub, amax_bits);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

static int bn_relu_res_impl(const void* x, bool x16, const float* scale, const float* shift, const float* res, int64_t rows, int hidden,
                            float* out, void* stream) {
    GN_REQUIRE(rows >= 0 && hidden > 0 && hidden % 4 == 0, "bn_relu_res: bad shape");
    if (rows == 0) return GNNOME_OK;
    GN_REQUIRE(x && scale && shift && res && out, "bn_relu_res: null pointer");
    const dim3 grid(ew_grid(rows * (hidden / 4)));
    if (x16)
        hipLaunchKernelGGL(k_bn_relu_res<true>, grid, dim3(kEwThreads), 0, (hipStream_t)stream, x, scale, shift, res, rows, hidden, out);
    else
        hipLaunchKernelGGL(k_bn_relu_res<false>, grid, dim3(kEwThreads), 0, (hipStream_t)stream, x, scale, shift, res, rows, hidden, out);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

extern "C" int gnnome_bn_relu_res_f32(const float* x, const float* scale, const float* shift, const float* res, int64_t rows,
                                      int hidden, float* out, void* stream) {
    return bn_relu_res_impl(x, false, scale, shift, res, rows, hidden, out, stream);
}

extern "C" int gnnome_bn_relu_res_x16(const uint16_t* x, const float* scale, const float* shift, const float* res, int64_t rows,
                                      int hidden, float* out, void* stream) {
    return bn_relu_res_impl(x, true, scale, shift, res, rows, hidden, out, stream);
}

extern "C" int gnnome_ln_relu_res_f32(const float* x, const float* gamma, const float* beta, const float* res, int64_t rows,
                                      int hidden, int norm_width, float* out, void* stream) {
    GN_REQUIRE(rows >= 0 && ok_width(hidden), "ln_relu_res: hidden=%d not in {16,32,64,128,256}", hidden);
    if (norm_width <= 0) norm_width = hidden;
    GN_REQUIRE(norm_width <= hidden, "ln_relu_res: norm_width=%d exceeds hidden=%d", norm_width, hidden);
    if (rows == 0) return GNNOME_OK;
    GN_REQUIRE(x && gamma && beta && res && out, "ln_relu_res: null pointer");
    hipLaunchKernelGGL(k_ln_relu_res, dim3(ew_grid(rows * (hidden / 4))), dim3(kEwThreads), 0, (hipStream_t)stream, x, gamma, beta, res,
                       rows, hidden, norm_width, out);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

extern "C" int gnnome_ln_bwd_f32(const float* dy, const float* x, const float* gamma, const float* beta, int64_t rows, int hidden,
                                 int norm_width, float* dx, float* dbeta, float* dgamma, void* workspace, size_t workspace_bytes, void* stream) {
    GN_REQUIRE(rows >= 0 && ok_width(hidden), "ln_bwd: hidden=%d not in {16,32,64,128,256}", hidden);
    if (norm_width <= 0) norm_width = hidden;
    GN_REQUIRE(norm_width <= hidden, "ln_bwd: norm_width=%d exceeds hidden=%d", norm_width, hidden);
    if (rows == 0) return GNNOME_OK;
    GN_REQUIRE(dy && x && gamma && beta && dx && dbeta && dgamma, "ln_bwd: null pointer");
    GN_REQUIRE(workspace && workspace_bytes >= kColWorkspaceBytes && (uintptr_t)workspace % 16 == 0,
               "ln_bwd: workspace too small or misaligned (gnnome_colsum_workspace_bytes)");
    const unsigned grid = col_grid(rows, hidden);
    hipLaunchKernelGGL(k_ln_bwd, dim3(grid), dim3(kEwThreads), 0, (hipStream_t)stream, dy, x, gamma, beta, rows, hidden, norm_width, dx,
                       (float*)workspace);
    GN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_col_finish, dim3(2 * hidden / 4), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, (int)grid,
                       hidden, dbeta, dgamma);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

extern "C" int gnnome_mul23_f32(const float* a, const float* b, const float* c, int64_t count, float* o1, float* o2,
                                void* stream) {
    GN_REQUIRE(count >= 0 && count % 4 == 0, "mul23: count must be a multiple of 4");
    if (count == 0) return GNNOME_OK;
    GN_REQUIRE(a && b && c && o1 && o2, "mul23: null pointer");
    hipLaunchKernelGGL(k_mul23, dim3(ew_grid(count / 4)), dim3(kEwThreads), 0, (hipStream_t)stream, a, b, c, count / 4, o1, o2);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

extern "C" int gnnome_segment_sum_f32(const float* X, int width, const int32_t* ptr, const int32_t* pos, int64_t num_nodes,
                                      float* out, int ld_out, void* stream) {
    GN_REQUIRE(num_nodes >= 0, "segment_sum: negative node count");
    if (num_nodes == 0) return GNNOME_OK;
    GN_REQUIRE(ptr && out && ld_out >= width && ld_out % 4 == 0, "segment_sum: bad arguments");
    const dim3 grid((unsigned)((num_nodes + 3) / 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    switch (width) {
        case 64: hipLaunchKernelGGL(k_segment_sum<64>, grid, block, 0, s, X, ptr, pos, num_nodes, out, ld_out); break;
        case 128: hipLaunchKernelGGL(k_segment_sum<128>, grid, block, 0, s, X, ptr, pos, num_nodes, out, ld_out); break;
        case 256: hipLaunchKernelGGL(k_segment_sum<256>, grid, block, 0, s, X, ptr, pos, num_nodes, out, ld_out); break;
        case 32: hipLaunchKernelGGL(k_segment_sum<32>, grid, block, 0, s, X, ptr, pos, num_nodes, out, ld_out); break;
        default: set_error("segment_sum: width=%d not in {16,32,64,128,256}", width); return GNNOME_EINVAL;
    }
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

extern "C" int gnnome_add_f32(const float* a, const float* b, int64_t count, float* out, void* stream) {
    GN_REQUIRE(count >= 0 && count % 4 == 0, "add: count must be a multiple of 4");
    if (count == 0) return GNNOME_OK;
    GN_REQUIRE(a && b && out, "add: null pointer");
    hipLaunchKernelGGL(k_add, dim3(ew_grid(count / 4)), dim3(kEwThreads), 0, (hipStream_t)stream, a, b, count / 4, out);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// Per-channel vector work of a train-mode BatchNorm1d in ONE launch (it used to be ~15 [H]-sized torch kernels per call,
// 16 calls per step): from the shifted column sums of the batch (d1 = sum(x - c), d2 = sum((x - c)^2), c = `center`)
//   mean = c + d1 / rows,  var = max(d2 / rows - (d1 / rows)^2, 0)                       (biased, what normalises)
//   rstd = 1 / sqrt(var + eps),  scale = gamma * rstd,  shift = beta - mean * scale       (gated_gcn_full.py:106,119,132)
// and nn.BatchNorm1d's buffer update, applied `updates` times (bn_e is called twice per layer on identical inputs):
//   running_mean = (1 - m) running_mean + m mean,  running_var = (1 - m) running_var + m var rows / (rows - 1),
//   num_batches_tracked += updates.
namespace gnnome {
__global__ __launch_bounds__(256) void k_bn_train_finish(const float* __restrict__ d1, const float* __restrict__ d2,
                                                         const float* __restrict__ center, double rows, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* running_mean, float* running_var,
                                                         int64_t* num_batches_tracked, float momentum, float eps, int updates, int H,
                                                         float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                         float* __restrict__ scale_out, float* __restrict__ shift_out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c == 0 && num_batches_tracked != nullptr) *num_batches_tracked += updates;
    if (c >= H) return;
    const float inv = (float)(1.0 / rows);
    const float m1 = d1[c] * inv;
    const float mean = (center != nullptr ? center[c] : 0.f) + m1;
    const float var = fmaxf(d2[c] * inv - m1 * m1, 0.f);
    const float rstd = 1.0f / sqrtf(var + eps);
    const float scale = gamma[c] * rstd;
    mean_out[c] = mean;
    rstd_out[c] = rstd;
    scale_out[c] = scale;
    shift_out[c] = beta[c] - mean * scale;
    if (running_mean != nullptr) {
        const float unbiased = var * (float)(rows / (rows > 1.0 ? rows - 1.0 : 1.0));
        float rm = running_mean[c], rv = running_var[c];
        for (int u = 0; u < updates; ++u) {
            rm = rm * (1.0f - momentum) + momentum * mean;
            rv = rv * (1.0f - momentum) + momentum * unbiased;
        }
        running_mean[c] = rm;
        running_var[c] = rv;
    }
}

// Row 0 of the raw gate, x0 = B1h[src_0] + B2h[dst_0] + e[0,:] W3^T: the centre the batch statistics of the gate are
// shifted by (any row is a sample of the distribution; the unshifted one-pass variance cancels catastrophically here).
__global__ __launch_bounds__(256) void k_gate_center(const float* __restrict__ e, const float* __restrict__ B1h, const float* __restrict__ B2h,
                                                     const int32_t* __restrict__ srt_src, const int32_t* __restrict__ srt_dst,
                                                     const float* __restrict__ W3, int ldw, int ldn, int H, float* __restrict__ center) {
    // one wave per channel j, the lanes share the dot product (a thread per channel walked a strided row of W3 alone: 17 us per layer)
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= H) return;
    float acc = 0.f;
    for (int k = lane; k < H; k += 64) acc = fmaf(e[k], W3[(int64_t)j * ldw + k], acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) center[j] = B1h[(int64_t)srt_src[0] * ldn + j] + B2h[(int64_t)srt_dst[0] * ldn + j] + acc;
}
}  // namespace gnnome

extern "C" int gnnome_bn_train_finish_f32(const float* d1, const float* d2, const float* center, int64_t rows, int hidden,
                                          const float* gamma, const float* beta, float* running_mean, float* running_var,
                                          int64_t* num_batches_tracked, float momentum, float eps, int updates, float* mean,
                                          float* rstd, float* scale, float* shift, void* stream) {
    using namespace gnnome;
    GN_REQUIRE(rows >= 1 && hidden >= 1 && updates >= 0, "bn_train_finish: bad sizes");
    GN_REQUIRE(d1 && d2 && gamma && beta && mean && rstd && scale && shift, "bn_train_finish: null pointer");
    GN_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_train_finish: running_mean and running_var go together");
    hipLaunchKernelGGL(k_bn_train_finish, dim3((hidden + 255) / 256), dim3(256), 0, (hipStream_t)stream, d1, d2, center, (double)rows, gamma,
                       beta, running_mean, running_var, num_batches_tracked, momentum, eps, updates, hidden, mean, rstd, scale, shift);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

extern "C" int gnnome_gate_center_f32(const float* e, int64_t num_edges, int hidden, const float* B1h, const float* B2h, int ld_node,
                                      const int32_t* srt_src, const int32_t* srt_dst, const float* W3, int ldw, float* center,
                                      void* stream) {
    using namespace gnnome;
    GN_REQUIRE(num_edges >= 1 && hidden >= 1, "gate_center: needs at least one edge");
    GN_REQUIRE(e && B1h && B2h && srt_src && srt_dst && W3 && center && ld_node >= hidden && ldw >= hidden, "gate_center: bad arguments");
    hipLaunchKernelGGL(k_gate_center, dim3((hidden + 3) / 4), dim3(256), 0, (hipStream_t)stream, e, B1h, B2h, srt_src, srt_dst, W3, ldw,
                       ld_node, hidden, center);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// Per-edge gradient of both gated aggregations (gnnome_agg_edge_bwd_f32, train_gemm.hip) WITH the BatchNorm-backward
// statistics of the result gathered in the same pass: de' = de + s(1-s)(...) is final for this layer's e' once this kernel
// has added its term, and bn_e's backward needs  s1 = sum_p de' m,  s2 = sum_p de' m (xe - mean)  over exactly these rows
// (m = relu mask rebuilt from the forward's expression).  One read of xe here replaces gnnome_bn_bwd_stats_f32's reads of de'
// and xe.
namespace gnnome {
template <int H, bool X16, bool BATCH = true, bool NT = false>
__global__ __launch_bounds__(kEwThreads) void k_agg_edge_bwd_stats(const float* __restrict__ e, int64_t E, const float* __restrict__ Tf,
                                                                   const float* __restrict__ Uf, const float* __restrict__ Tb,
                                                                   const float* __restrict__ Ub, const float* __restrict__ A2h,
                                                                   const float* __restrict__ A3h, int ldn,
                                                                   const int32_t* __restrict__ srt_src, const int32_t* __restrict__ srt_dst,
                                                                   float* __restrict__ de, const void* __restrict__ xe,
                                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                                   const float* __restrict__ mean, float* __restrict__ part) {
    column_reduce<2>(E, H, part, [&](int64_t p, int c, f32x4 (&acc)[2]) {
        // the three row loads whose addresses need no index go out with the index loads; the six gathers follow when the indices are in
        const int64_t s_ = srt_src[p], d_ = srt_dst[p];
        // NT: the three streams (each row read once by the whole launch) go past the L2's replacement order, which then keeps the six tables' rows
        const f32x4 x = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(e + p * H + c)) : *reinterpret_cast<const f32x4*>(e + p * H + c);
        const f32x4 xv = (NT && !X16) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(xe) + p * H + c))
                                      : load4_as<X16>(xe, p * H + c);
        f32x4 g = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(de + p * H + c)) : *reinterpret_cast<const f32x4*>(de + p * H + c);
        const f32x4 tf = *reinterpret_cast<const f32x4*>(Tf + d_ * H + c), uf = *reinterpret_cast<const f32x4*>(Uf + d_ * H + c);
        const f32x4 tb = *reinterpret_cast<const f32x4*>(Tb + s_ * H + c), ub = *reinterpret_cast<const f32x4*>(Ub + s_ * H + c);
        const f32x4 a2 = *reinterpret_cast<const f32x4*>(A2h + s_ * ldn + c), a3 = *reinterpret_cast<const f32x4*>(A3h + d_ * ldn + c);
        // Left alone, the compiler sinks every load to its first use to save registers - one memory round trip after the other
        // (seven `s_waitcnt vmcnt(0)` per row in the ISA).  Nothing may cross this line: all nine loads are in flight together.
        if (BATCH) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float sg = sigmoidf_(x[j]);
            g[j] += sg * (1.f - sg) * (tf[j] * a2[j] - uf[j] + tb[j] * a3[j] - ub[j]);
            const float gm = (xv[j] * scale[c + j] + shift[c + j] > 0.f) ? g[j] : 0.f;
            acc[0][j] += gm;
            acc[1][j] += gm * (xv[j] - mean[c + j]);
        }
        *reinterpret_cast<f32x4*>(de + p * H + c) = g;
    });
}

// ---------------------------------------------------------------------------------------------------------------------
// The aggregation's WHOLE backward over the edges in one launch (round 5): the node sums of gnnome_node_aggregate_raw_f32 mode 2
//   sum_in[i] = sum_{p: dst_p = i} s_p Tb[src_p],   sum_out[i] = sum_{p: src_p = i} s_p Tf[dst_p],   s = sigmoid(e')
// AND the per-edge pass above (de' += s(1-s)(...), bn_e's backward statistics) - both read every e' row and evaluate its sigmoid;
// as two launches they stream e' from HBM twice (0.20 + 0.49 ms at configs[2]).  One wave per node, as the aggregation walks the
// lists: the in-edge rows of node i are the contiguous sorted positions in_ptr[i] .. in_ptr[i+1], so the rows of e', xe and de' of those
// edges are streamed once here, three of the per-edge pass's six gathers (Tf, Uf, A3h at dst = i) become ONE row per node, and
// the out-edge pass reads the e' rows its neighbours' in-edge passes have just brought into the L2.
// Workgroups walk the 4-node groups as column_reduce walks rows (an XCD owns a contiguous eighth, its workgroups interleave in it: many
// CUs stream adjacent rows at the same time); the statistics leave as per-workgroup partial sums in a fixed order (k_col_finish).
// A node of any degree is walked by its one wave (64 items at a time): correct for hubs, not fast.
template <int H, bool X16, int UI = 2, int UO = 4, int WPE = 4>
__global__ __launch_bounds__(kEwThreads) __attribute__((amdgpu_waves_per_eu(WPE))) void k_agg_bwd_fused(const float* __restrict__ e, int64_t n_nodes, int64_t n_edges, const float* __restrict__ Tf,
                                                              const float* __restrict__ Uf, const float* __restrict__ Tb,
                                                              const float* __restrict__ Ub, const float* __restrict__ A2h,
                                                              const float* __restrict__ A3h, int ldn, const int32_t* __restrict__ in_ptr,
                                                              const int32_t* __restrict__ srt_src, const int32_t* __restrict__ out_ptr,
                                                              const int32_t* __restrict__ out_pos, const int32_t* __restrict__ out_dst,
                                                              float* __restrict__ de, const void* __restrict__ xe,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              const float* __restrict__ mean, float* __restrict__ sum_in,
                                                              float* __restrict__ sum_out, float* __restrict__ part, unsigned* __restrict__ amax_bits) {
    constexpr int LPR = H / 4, G = 64 / LPR, WAVES = kEwThreads / 64;
    float amax = 0.f;   // (of the node sums, for the consumers that scale them into fp16's range)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int group = lane / LPR, c = (lane % LPR) * 4;
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    // the three per-channel vectors live in LDS (12 registers per lane otherwise: the kernel is held to 128 for four waves per SIMD)
    __shared__ __attribute__((aligned(16))) float cst[3][H];
    for (int i = threadIdx.x; i < H; i += kEwThreads) {
        cst[0][i] = scale[i];
        cst[1][i] = shift[i];
        cst[2][i] = mean[i];
    }
    __syncthreads();
    const int64_t groups = (n_nodes + WAVES - 1) / WAVES;
    int64_t g0, g1, gstep;
    if (gridDim.x % kXcds == 0) {
        const int64_t per_xcd = (groups + kXcds - 1) / kXcds;
        g0 = (int64_t)(blockIdx.x % kXcds) * per_xcd;
        g1 = min(groups, g0 + per_xcd);
        g0 += blockIdx.x / kXcds;
        gstep = gridDim.x / kXcds;
    } else {
        g0 = blockIdx.x, g1 = groups, gstep = gridDim.x;
    }
    // (Measured and dropped: requesting the NEXT node's list pointers at the top of a node and its index lists before the out-edge pass, so
    // that a node starts with its rows - 0.620 against 0.566 ms: inside 128 registers the values held ahead are spilled as they arrive.)
    const int64_t last = n_edges - 1;
    for (int64_t gi = g0; gi < g1; gi += gstep) {
        const int64_t node = gi * WAVES + wave;
        if (node >= n_nodes) continue;   // (no barrier inside the loop)
        const int ib = in_ptr[node], din = in_ptr[node + 1] - ib;
        const int ob = out_ptr[node], dout = out_ptr[node + 1] - ob;
        // the neighbour indices of the first 64 items of BOTH lists go out together, unconditionally (a lane past the end of a list repeats a
        // valid position): the out-list's two index loads are then under way while the in-edges are walked
        const int first_n = srt_src[min((int64_t)ib + min(lane, max(din - 1, 0)), last)];
        const int64_t oq = min((int64_t)ob + min(lane, max(dout - 1, 0)), last);
        const int first_op = out_pos[oq], first_on = out_dst[oq];
        const f32x4 tf = *reinterpret_cast<const f32x4*>(Tf + node * H + c), uf = *reinterpret_cast<const f32x4*>(Uf + node * H + c);
        const f32x4 a3 = *reinterpret_cast<const f32x4*>(A3h + node * ldn + c);
        f32x4 nf = {0.f, 0.f, 0.f, 0.f}, nb = nf;
        // ---- in-edges: rows ib .. ib + din of e', xe, de' (streamed), the tables at src (gathered)
        for (int base = 0; base < din; base += 64) {
            const int m = min(64, din - base);
            int my_n = first_n;
            if (base > 0) my_n = lane < m ? srt_src[ib + base + lane] : 0;
            for (int j0 = 0; j0 < m; j0 += G * UI) {
                f32x4 x[UI], xv[UI], g[UI], tb[UI], ub[UI], a2[UI];
                bool live[UI];
                int64_t p[UI];
#pragma unroll
                for (int u = 0; u < UI; ++u) {
                    const int item = j0 + u * G + group;
                    live[u] = item < m;
                    p[u] = ib + base + (live[u] ? item : 0);
                    // e' stays in the L2's ordinary order (the out-edge pass of a neighbour reads the row again); xe and de' are read once
                    x[u] = *reinterpret_cast<const f32x4*>(e + p[u] * H + c);
                    xv[u] = X16 ? load4_as<true>(xe, p[u] * H + c)
                                : __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(xe) + p[u] * H + c));
                    g[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(de + p[u] * H + c));
                }
#pragma unroll
                for (int u = 0; u < UI; ++u) {
                    const int64_t nn = __shfl(my_n, live[u] ? j0 + u * G + group : 0);
                    tb[u] = *reinterpret_cast<const f32x4*>(Tb + nn * H + c);
                    ub[u] = *reinterpret_cast<const f32x4*>(Ub + nn * H + c);
                    a2[u] = *reinterpret_cast<const f32x4*>(A2h + nn * ldn + c);
                }
                __builtin_amdgcn_sched_barrier(0);   // all twelve loads of the step are in flight before the first use
                const f32x4 sc = *reinterpret_cast<const f32x4*>(&cst[0][c]), sh = *reinterpret_cast<const f32x4*>(&cst[1][c]);
                const f32x4 mn = *reinterpret_cast<const f32x4*>(&cst[2][c]);
#pragma unroll
                for (int u = 0; u < UI; ++u) {
                    // a dead slot (its loads repeat the batch's first item) adds +0 everywhere: no branch around the arithmetic
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float sg = live[u] ? sigmoidf_(x[u][k]) : 0.f;
                        nf[k] += sg * tb[u][k];
                        g[u][k] += sg * (1.f - sg) * (tf[k] * a2[u][k] - uf[k] + tb[u][k] * a3[k] - ub[u][k]);
                        const float gm = (live[u] && xv[u][k] * sc[k] + sh[k] > 0.f) ? g[u][k] : 0.f;
                        acc[0][k] += gm;
                        acc[1][k] += gm * (xv[u][k] - mn[k]);
                    }
                    if (live[u]) __builtin_nontemporal_store(g[u], reinterpret_cast<f32x4*>(de + p[u] * H + c));
                }
            }
        }
        // ---- out-edges: e' rows through out_pos (rows a neighbour's in-edge pass streams at about this time), Tf at dst
        for (int base = 0; base < dout; base += 64) {
            const int m = min(64, dout - base);
            int my_p = first_op, my_n = first_on;
            if (base > 0) {
                my_p = lane < m ? out_pos[ob + base + lane] : 0;
                my_n = lane < m ? out_dst[ob + base + lane] : 0;
            }
            for (int j0 = 0; j0 < m; j0 += G * UO) {
                f32x4 x[UO], t[UO];
                bool live[UO];
#pragma unroll
                for (int u = 0; u < UO; ++u) {
                    const int item = j0 + u * G + group;
                    live[u] = item < m;
                    const int64_t pp = __shfl(my_p, live[u] ? item : 0), nn = __shfl(my_n, live[u] ? item : 0);
                    x[u] = *reinterpret_cast<const f32x4*>(e + pp * H + c);
                    t[u] = *reinterpret_cast<const f32x4*>(Tf + nn * H + c);
                }
#pragma unroll
                for (int u = 0; u < UO; ++u)
#pragma unroll
                    for (int k = 0; k < 4; ++k) nb[k] += (live[u] ? sigmoidf_(x[u][k]) : 0.f) * t[u][k];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int mm = LPR; mm < 64; mm <<= 1) {
                nf[k] += __shfl_xor(nf[k], mm);
                nb[k] += __shfl_xor(nb[k], mm);
            }
        }
        if (group == 0) {
            *reinterpret_cast<f32x4*>(sum_in + node * H + c) = nf;
            *reinterpret_cast<f32x4*>(sum_out + node * H + c) = nb;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) amax = fmaxf(amax, fmaxf(fabsf(nf[k]), fabsf(nb[k])));
    }
    if (amax_bits != nullptr) wave_amax_to(amax_bits, amax);
    column_fold<2>(acc, H, part);
}
}  // namespace gnnome

static int agg_edge_bwd_stats_impl(const float* e, int64_t num_edges, int hidden, const float* Tf, const float* Uf,
                                   const float* Tb, const float* Ub, const float* A2h, const float* A3h, int ld_node,
                                   const int32_t* srt_src, const int32_t* srt_dst, float* de, const void* xe, bool x16,
                                   const float* scale, const float* shift, const float* mean, float* s1, float* s2,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    using namespace gnnome;
    GN_REQUIRE(num_edges >= 0, "agg_edge_bwd_stats: negative edge count");
    if (num_edges == 0) return GNNOME_OK;
    GN_REQUIRE(e && Tf && Uf && Tb && Ub && A2h && A3h && srt_src && srt_dst && de && xe && scale && shift && mean && s1 && s2 && ld_node % 4 == 0,
               "agg_edge_bwd_stats: bad arguments");
    GN_REQUIRE(workspace && workspace_bytes >= kColWorkspaceBytes && (uintptr_t)workspace % 16 == 0,
               "agg_edge_bwd_stats: workspace too small or misaligned (gnnome_colsum_workspace_bytes)");
    hipStream_t s = (hipStream_t)stream;
    const unsigned grid = col_grid(num_edges, hidden);
#define GN_AEBS(HH, XX)                                                                                                              \
    do {                                                                                                                             \
        if (tuning(kTuneGateExperiment) == 77)   /* A/B: the three streams through the L2's ordinary replacement order */            \
            hipLaunchKernelGGL((k_agg_edge_bwd_stats<HH, XX>), dim3(grid), dim3(kEwThreads), 0, s, e, num_edges, Tf, Uf, Tb, Ub, A2h, A3h, \
                               ld_node, srt_src, srt_dst, de, xe, scale, shift, mean, (float*)workspace);                            \
        else                                                                                                                         \
            hipLaunchKernelGGL((k_agg_edge_bwd_stats<HH, XX, true, true>), dim3(grid), dim3(kEwThreads), 0, s, e, num_edges, Tf, Uf, Tb, Ub, \
                               A2h, A3h, ld_node, srt_src, srt_dst, de, xe, scale, shift, mean, (float*)workspace);                  \
    } while (0)
    switch (hidden) {
        case 64: if (x16) GN_AEBS(64, true); else GN_AEBS(64, false); break;
        case 128: if (x16) GN_AEBS(128, true); else GN_AEBS(128, false); break;
        case 256: if (x16) GN_AEBS(256, true); else GN_AEBS(256, false); break;
        default: set_error("agg_edge_bwd_stats: hidden=%d not in {64,128,256}", hidden); return GNNOME_EINVAL;
    }
#undef GN_AEBS
    GN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_col_finish, dim3(2 * hidden / 4), dim3(256), 0, s, (const float*)workspace, (int)grid, hidden, s1, s2);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

static int agg_bwd_fused_impl(const float* e, int64_t num_nodes, int64_t num_edges, int hidden, const float* Tf, const float* Uf, const float* Tb,
                              const float* Ub, const float* A2h, const float* A3h, int ld_node, const int32_t* in_ptr, const int32_t* srt_src,
                              const int32_t* out_ptr, const int32_t* out_pos, const int32_t* out_dst, float* de, const void* xe, bool x16,
                              const float* scale, const float* shift, const float* mean, float* sum_in, float* sum_out, float* s1, float* s2,
                              unsigned* amax_bits, void* workspace, size_t workspace_bytes, void* stream) {
    using namespace gnnome;
    GN_REQUIRE(num_nodes >= 1 && num_edges >= 0, "agg_bwd_fused: needs at least one node");
    GN_REQUIRE(Tf && Uf && Tb && Ub && A2h && A3h && in_ptr && out_ptr && scale && shift && mean && sum_in && sum_out && s1 && s2 &&
                   ld_node % 4 == 0 && ld_node >= hidden,
               "agg_bwd_fused: bad arguments");
    GN_REQUIRE(num_edges == 0 || (e && srt_src && out_pos && out_dst && de && xe), "agg_bwd_fused: null edge operand");
    GN_REQUIRE(workspace && workspace_bytes >= kColWorkspaceBytes && (uintptr_t)workspace % 16 == 0,
               "agg_bwd_fused: workspace too small or misaligned (gnnome_colsum_workspace_bytes)");
    hipStream_t s = (hipStream_t)stream;
    if (num_edges == 0) {   // nothing to walk: the sums of empty lists
        GN_HIP(hipMemsetAsync(sum_in, 0, sizeof(float) * (size_t)num_nodes * hidden, s));
        GN_HIP(hipMemsetAsync(sum_out, 0, sizeof(float) * (size_t)num_nodes * hidden, s));
        GN_HIP(hipMemsetAsync(s1, 0, sizeof(float) * hidden, s));
        GN_HIP(hipMemsetAsync(s2, 0, sizeof(float) * hidden, s));
        return GNNOME_OK;
    }
    const int64_t groups = (num_nodes + kEwThreads / 64 - 1) / (kEwThreads / 64);
    unsigned grid = (unsigned)std::min<int64_t>(groups, kColMaxBlocks);
    if (grid >= 2 * kXcds) grid = grid / kXcds * kXcds;
    // Occupancy against requests per step, measured (tools/agg_edge_bwd_ab.py; profiles/r05_train_step_ab.txt): three waves per SIMD with
    // 3 in-edge / 4 out-edge items per lane group and step beat four waves with 2 / 4 by 3-4 % at H <= 128 (0.529 against 0.548 ms), two waves
    // with 5 / 5 by 6.5 % at H = 256 (0.524 against 0.561); 4 / 4 at three waves spills (0.677).  The grid is cut to the resident waves.
    const int wpe = hidden == 256 ? 2 : 3;
    if (tuning(kTuneGateExperiment) != 88) grid = std::max(1u, grid / 4 * wpe / kXcds * kXcds);
#define GN_ABF(HH, XX)                                                                                                               \
    do {                                                                                                                             \
        if (tuning(kTuneGateExperiment) == 88)   /* A/B: four waves per SIMD, 2 / 4 items (the first form) */                        \
            hipLaunchKernelGGL((k_agg_bwd_fused<HH, XX, 2, 4, 4>), dim3(grid), dim3(kEwThreads), 0, s, e, num_nodes, num_edges, Tf, Uf, Tb, Ub, \
                               A2h, A3h, ld_node, in_ptr, srt_src, out_ptr, out_pos, out_dst, de, xe, scale, shift, mean, sum_in, sum_out, \
                               (float*)workspace, amax_bits);                                                                                   \
        else if (HH == 256)                                                                                                          \
            hipLaunchKernelGGL((k_agg_bwd_fused<HH, XX, 5, 5, 2>), dim3(grid), dim3(kEwThreads), 0, s, e, num_nodes, num_edges, Tf, Uf, Tb, Ub, \
                               A2h, A3h, ld_node, in_ptr, srt_src, out_ptr, out_pos, out_dst, de, xe, scale, shift, mean, sum_in, sum_out, \
                               (float*)workspace, amax_bits);                                                                                   \
        else                                                                                                                         \
            hipLaunchKernelGGL((k_agg_bwd_fused<HH, XX, 3, 4, 3>), dim3(grid), dim3(kEwThreads), 0, s, e, num_nodes, num_edges, Tf, Uf, Tb, Ub, \
                               A2h, A3h, ld_node, in_ptr, srt_src, out_ptr, out_pos, out_dst, de, xe, scale, shift, mean, sum_in, sum_out, \
                               (float*)workspace, amax_bits);                                                                                   \
    } while (0)
    switch (hidden) {
        case 64: if (x16) GN_ABF(64, true); else GN_ABF(64, false); break;
        case 128: if (x16) GN_ABF(128, true); else GN_ABF(128, false); break;
        case 256: if (x16) GN_ABF(256, true); else GN_ABF(256, false); break;
        default: set_error("agg_bwd_fused: hidden=%d not in {64,128,256}", hidden); return GNNOME_EINVAL;
    }
#undef GN_ABF
    GN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_col_finish, dim3(2 * hidden / 4), dim3(256), 0, s, (const float*)workspace, (int)grid, hidden, s1, s2);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

extern "C" int gnnome_agg_bwd_fused_f32(const float* e, int64_t num_nodes, int64_t num_edges, int hidden, const float* Tf, const float* Uf,
                                        const float* Tb, const float* Ub, const float* A2h, const float* A3h, int ld_node,
                                        const int32_t* in_ptr, const int32_t* srt_src, const int32_t* out_ptr, const int32_t* out_pos,
                                        const int32_t* out_dst, float* de, const float* xe, const float* scale, const float* shift,
                                        const float* mean, float* sum_in, float* sum_out, float* s1, float* s2, unsigned* amax_bits,
                                        void* workspace, size_t workspace_bytes, void* stream) {
    return agg_bwd_fused_impl(e, num_nodes, num_edges, hidden, Tf, Uf, Tb, Ub, A2h, A3h, ld_node, in_ptr, srt_src, out_ptr, out_pos, out_dst, de,
                              xe, false, scale, shift, mean, sum_in, sum_out, s1, s2, amax_bits, workspace, workspace_bytes, stream);
}

extern "C" int gnnome_agg_bwd_fused_x16(const float* e, int64_t num_nodes, int64_t num_edges, int hidden, const float* Tf, const float* Uf,
                                        const float* Tb, const float* Ub, const float* A2h, const float* A3h, int ld_node,
                                        const int32_t* in_ptr, const int32_t* srt_src, const int32_t* out_ptr, const int32_t* out_pos,
                                        const int32_t* out_dst, float* de, const uint16_t* xe, const float* scale, const float* shift,
                                        const float* mean, float* sum_in, float* sum_out, float* s1, float* s2, unsigned* amax_bits,
                                        void* workspace, size_t workspace_bytes, void* stream) {
    return agg_bwd_fused_impl(e, num_nodes, num_edges, hidden, Tf, Uf, Tb, Ub, A2h, A3h, ld_node, in_ptr, srt_src, out_ptr, out_pos, out_dst, de,
                              xe, true, scale, shift, mean, sum_in, sum_out, s1, s2, amax_bits, workspace, workspace_bytes, stream);
}

extern "C" int gnnome_agg_edge_bwd_stats_f32(const float* e, int64_t num_edges, int hidden, const float* Tf, const float* Uf,
                                             const float* Tb, const float* Ub, const float* A2h, const float* A3h, int ld_node,
                                             const int32_t* srt_src, const int32_t* srt_dst, float* de, const float* xe,
                                             const float* scale, const float* shift, const float* mean, float* s1, float* s2,
                                             void* workspace, size_t workspace_bytes, void* stream) {
    return agg_edge_bwd_stats_impl(e, num_edges, hidden, Tf, Uf, Tb, Ub, A2h, A3h, ld_node, srt_src, srt_dst, de, xe, false, scale, shift,
                                   mean, s1, s2, workspace, workspace_bytes, stream);
}

extern "C" int gnnome_agg_edge_bwd_stats_x16(const float* e, int64_t num_edges, int hidden, const float* Tf, const float* Uf,
                                             const float* Tb, const float* Ub, const float* A2h, const float* A3h, int ld_node,
                                             const int32_t* srt_src, const int32_t* srt_dst, float* de, const uint16_t* xe,
                                             const float* scale, const float* shift, const float* mean, float* s1, float* s2,
                                             void* workspace, size_t workspace_bytes, void* stream) {
    return agg_edge_bwd_stats_impl(e, num_edges, hidden, Tf, Uf, Tb, Ub, A2h, A3h, ld_node, srt_src, srt_dst, de, xe, true, scale, shift,
                                   mean, s1, s2, workspace, workspace_bytes, stream);
}

static int segment_sum2_impl(const void* X, bool x16, int width, const int32_t* in_ptr, const int32_t* out_ptr, const int32_t* out_pos,
                             int64_t num_nodes, float* out_in, int ld_in, float* out_out, int ld_out, void* stream, unsigned* amax_bits = nullptr) {
    using namespace gnnome;
    GN_REQUIRE(num_nodes >= 0, "segment_sum2: negative node count");
    if (num_nodes == 0) return GNNOME_OK;
    GN_REQUIRE(in_ptr && out_ptr && out_in && out_out && ld_in >= width && ld_out >= width && ld_in % 4 == 0 && ld_out % 4 == 0,
               "segment_sum2: bad arguments");
    const unsigned blocks = (unsigned)((num_nodes + 3) / 4);
    hipStream_t s = (hipStream_t)stream;
#define GN_SS2(WW, XX)                                                                                                                    \
    hipLaunchKernelGGL((k_segment_sum2<WW, XX>), dim3(blocks), dim3(256), 0, s, X, in_ptr, out_ptr, out_pos, num_nodes, out_in, ld_in, out_out, \
                       ld_out, (int)blocks, amax_bits)
    switch (width) {
        case 64: if (x16) GN_SS2(64, true); else GN_SS2(64, false); break;
        case 128: if (x16) GN_SS2(128, true); else GN_SS2(128, false); break;
        case 256: if (x16) GN_SS2(256, true); else GN_SS2(256, false); break;
        case 32: if (x16) GN_SS2(32, true); else GN_SS2(32, false); break;
        default: set_error("segment_sum2: width=%d not in {32,64,128,256}", width); return GNNOME_EINVAL;
    }
#undef GN_SS2
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

extern "C" int gnnome_segment_sum2_f32(const float* X, int width, const int32_t* in_ptr, const int32_t* out_ptr, const int32_t* out_pos,
                                       int64_t num_nodes, float* out_in, int ld_in, float* out_out, int ld_out, void* stream) {
    return segment_sum2_impl(X, false, width, in_ptr, out_ptr, out_pos, num_nodes, out_in, ld_in, out_out, ld_out, stream);
}

// ... raising amax_bits[0] (the bits of a non-negative float, NOT zeroed here: several producers share one slot) to max |out_in|, |out_out|
extern "C" int gnnome_segment_sum2_amax_f32(const float* X, int width, const int32_t* in_ptr, const int32_t* out_ptr, const int32_t* out_pos,
                                            int64_t num_nodes, float* out_in, int ld_in, float* out_out, int ld_out, unsigned* amax_bits,
                                            void* stream) {
    GN_REQUIRE(amax_bits != nullptr, "segment_sum2_amax: null amax_bits");
    return segment_sum2_impl(X, false, width, in_ptr, out_ptr, out_pos, num_nodes, out_in, ld_in, out_out, ld_out, stream, amax_bits);
}

extern "C" int gnnome_segment_sum2_x16(const uint16_t* X, int width, const int32_t* in_ptr, const int32_t* out_ptr, const int32_t* out_pos,
                                       int64_t num_nodes, float* out_in, int ld_in, float* out_out, int ld_out, void* stream) {
    return segment_sum2_impl(X, true, width, in_ptr, out_ptr, out_pos, num_nodes, out_in, ld_in, out_out, ld_out, stream);
}


// ---------------------------------------------------------------------------------------------------------------------
// Small per-channel / per-weight bookkeeping of the training step, each in ONE launch instead of several torch operators
// (the step replays ~480 launches from a hipGraph; every one of these cost a graph node of a few microseconds).
namespace gnnome {
// BatchNorm backward, the three per-channel vectors between the statistics pass and the apply pass (train.py `_bn_bwd`, one rank):
// s2h = rstd * s2,  c1 = s1 / rows,  c2 = s2h / rows   (the same fp32 operations torch performs for them)
__global__ __launch_bounds__(256) void k_bn_bwd_terms(const float* __restrict__ s1, const float* __restrict__ s2, const float* __restrict__ rstd,
                                                      float inv_rows, int H, float* __restrict__ s2h, float* __restrict__ c1, float* __restrict__ c2) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= H) return;
    const float t = rstd[c] * s2[c];
    s2h[c] = t;
    c1[c] = s1[c] * inv_rows;   // (torch divides a tensor by a host scalar as a product with its fp32 reciprocal)
    c2[c] = t * inv_rows;
}

struct PackLayer {
    const float* w[5];   // A_1, A_2, A_3, B_1, B_2 weights [H,H]
    const float* b[5];   // their biases [H]
    const float* w3;     // B_3 weight [H,H]
    const float* b3;     // B_3 bias [H]
};
// Wcat[5H,H] = the five weights stacked, bcat[5H] = the five biases (B_2's plus B_3's: the gate adds both), WcatT[H,5H] and W3T[H,H]
// the transposes the backward's data-gradient products read (gated_gcn_full.py:91-97 as one projection, train.py `_cat_layer`)
__global__ __launch_bounds__(256) void k_pack_layer(PackLayer p, int H, float* __restrict__ Wcat, float* __restrict__ bcat,
                                                    float* __restrict__ WcatT, float* __restrict__ W3T) {
    __shared__ float tile[32][33];
    const int blocks_per_mat = (H / 32) * (H / 32);
    const int mat = blockIdx.x / blocks_per_mat, blk = blockIdx.x % blocks_per_mat;   // mat 0..4: the stacked weights, 5: B_3
    const int r0 = (blk / (H / 32)) * 32, c0 = (blk % (H / 32)) * 32;
    const float* src = p.w[0];
#pragma unroll
    for (int k = 1; k < 5; ++k)
        if (mat == k) src = p.w[k];
    if (mat == 5) src = p.w3;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i;
        const float v = src[(int64_t)r * H + c0 + tx];
        tile[ty + 8 * i][tx] = v;
        if (mat < 5) Wcat[((int64_t)mat * H + r) * H + c0 + tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i;   // row of the transpose
        const float v = tile[tx][ty + 8 * i];
        if (mat < 5) WcatT[(int64_t)c * (5 * H) + mat * H + r0 + tx] = v;
        else W3T[(int64_t)c * H + r0 + tx] = v;
    }
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < 5 * H; i += 256) {
            const int k = i / H, c = i % H;
            const float* bsrc = p.b[0];
#pragma unroll
            for (int q = 1; q < 5; ++q)
                if (k == q) bsrc = p.b[q];
            bcat[i] = k == 4 ? bsrc[c] + p.b3[c] : bsrc[c];
        }
}
}  // namespace gnnome

extern "C" int gnnome_bn_bwd_terms_f32(const float* s1, const float* s2, const float* rstd, int64_t rows, int hidden, float* s2h, float* c1,
                                       float* c2, void* stream) {
    using namespace gnnome;
    GN_REQUIRE(s1 && s2 && rstd && s2h && c1 && c2 && rows >= 1 && hidden >= 1, "bn_bwd_terms: bad arguments");
    hipLaunchKernelGGL(k_bn_bwd_terms, dim3((hidden + 255) / 256), dim3(256), 0, (hipStream_t)stream, s1, s2, rstd, 1.0f / (float)rows, hidden, s2h, c1, c2);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

extern "C" int gnnome_pack_layer_f32(const float* const* weights5, const float* const* biases5, const float* B3_weight, const float* B3_bias,
                                     int hidden, float* Wcat, float* bcat, float* WcatT, float* W3T, void* stream) {
    using namespace gnnome;
    GN_REQUIRE(weights5 && biases5 && B3_weight && B3_bias && Wcat && bcat && WcatT && W3T && hidden >= 32 && hidden % 32 == 0,
               "pack_layer: bad arguments (hidden must be a multiple of 32)");
    PackLayer p = {};
    for (int k = 0; k < 5; ++k) {
        GN_REQUIRE(weights5[k] && biases5[k], "pack_layer: null weight %d", k);
        p.w[k] = weights5[k];
        p.b[k] = biases5[k];
    }
    p.w3 = B3_weight;
    p.b3 = B3_bias;
    const int blocks = 6 * (hidden / 32) * (hidden / 32);
    hipLaunchKernelGGL(k_pack_layer, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, hidden, Wcat, bcat, WcatT, W3T);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}
