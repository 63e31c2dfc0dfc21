// Reference-ORDER fp32 kernels on the fp32 matrix cores (round 3; the VALU forms in reference_order.hip stay as variant 1).
//
// Why the matrix cores CAN give the reference's order after all.  tools/mfma_order_probe.hip (run on the MI355X,
// profiles/r03_mfma_order.txt) shows that v_mfma_f32_32x32x2_f32 evaluates  D = fma(a[k1], b[k1], fma(a[k0], b[k0], C))  - an
// fma chain, k0 BEFORE k1, bit for bit on 204 800 random outputs (and v_mfma_f32_16x16x4_f32 its four k in ascending order).
// Which k a lane feeds is the kernel's choice: here lane (row, kk = lane / 32) hands instruction j the element k = 2 j + kk,
// so that the instructions j = 0 .. K/2-1 in program order ARE torch-CPU / MKL's k-ascending fma chain from zero
// (reference_order.hip's header has the reference facts).  Round 2 read "the operand layout interleaves k = 0,4,1,5" off a kernel
// that fed float4 pieces; that was a property of that kernel's loads, not of the instruction.
// So the high-gain layer of the shipped checkpoint runs at the fp32-MFMA rate (157 TF) instead of ~10 TF of scalar-fed VALU
// chains: k_edge_gate_ref 0.231 ms -> see DESIGN.md at the E. coli-sized graph, same bits.
//
// Structure (both kernels): a workgroup of four waves; a ROW GROUP of 32 rows is owned by H/64 waves (one at H = 64, two at
// H = 128), each holding the B operand - W rows of its 64 output columns, every second k - in registers for the whole launch
// (K/2 VGPRs per 32-column block).  The A tile goes through LDS: coalesced 16-byte loads in, ds_read_b128 + one v_cndmask per
// MFMA out (a lane needs elements kk and kk + 2 of every four).  Everything after the chain follows reference_order.hip line
// by line: + bias with one rounding, (B1h[src] + B2h[dst]) + B3e, fma(x, alpha, beta), relu, + e.
#include "common.h"

#include <algorithm>

namespace gnnome {
namespace {

constexpr int kRmThreads = 256;

template <int K>
struct RmTile {
    static constexpr int LD = K + 4;   // floats per LDS row: 16-byte aligned rows, conflict-free ds_read_b128 (16 lanes x 16 B, stride 4 banks)
};

__device__ __forceinline__ int acc_row(int r, int kk) { return (r & 3) + 8 * (r >> 2) + 4 * kk; }   // tile row of accumulator element r

// This lane's share of a weight row for the chain: elements kk, kk + 2, kk + 4, ... of W[col][0 .. K): whole 16-byte pieces are
// loaded (both half waves read the same row) and the lane keeps every second element.
template <int K>
__device__ __forceinline__ void load_w_row(const float* __restrict__ wrow, int kk, float (&w)[K / 2]) {
#pragma unroll
    for (int q = 0; q < K / 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(wrow + 4 * q);
        w[2 * q] = kk ? v[1] : v[0];
        w[2 * q + 1] = kk ? v[3] : v[2];
    }
}

// acc[cb] = the k-ascending fma chain over the LDS tile's row `cl` against the register-held weights; NCB 32-column blocks.
template <int K, int NCB>
__device__ __forceinline__ void chain_tile(const float* __restrict__ tile_row, int kk, const float (&w)[NCB][K / 2], f32x16 (&acc)[NCB]) {
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
#pragma unroll
    for (int q = 0; q < K / 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(tile_row + 4 * q);
        const float a0 = kk ? v[1] : v[0], a1 = kk ? v[3] : v[2];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, w[cb][2 * q], acc[cb], 0, 0, 0);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, w[cb][2 * q + 1], acc[cb], 0, 0, 0);
    }
}

// C[M,Nout] = chain(A W^T) + bias.  grid = (row groups of 4 x 32 rows, Nout / 64 column groups); every wave: 32 rows x 64 columns.
// K = 256 (round 4): NCB = 1 - a wave holds ONE 32-column block of W (K/2 = 128 registers), the grid has Nout / 32 column groups.
template <int K, int NCB = (K == 256 ? 1 : 2)>
__global__ __launch_bounds__(kRmThreads) void k_linear_refm(const float* __restrict__ A, int64_t M, int lda, const float* __restrict__ W, int ldw,
                                                            const float* __restrict__ bias, int Nout, float* __restrict__ C, int ldc,
                                                            int row_groups) {
    constexpr int LD = RmTile<K>::LD;
    __shared__ __attribute__((aligned(16))) float tiles[4 * 32 * LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, cl = lane & 31, kk = lane >> 5;
    const int col0 = blockIdx.y * 32 * NCB;
    const bool w_rows16 = ldw % 4 == 0 && ((uintptr_t)W % 16 == 0);   // (uniform) 16-byte pieces of the weight rows
    float w[NCB][K / 2];
    float bcol[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const int col = min(col0 + 32 * cb + cl, Nout - 1);   // (Nout % 32 != 0: the surplus columns are computed and not stored)
        const float* wr = W + (int64_t)col * ldw;
        if (w_rows16) {
            load_w_row<K>(wr, kk, w[cb]);
        } else {
#pragma unroll
            for (int j = 0; j < K / 2; ++j) w[cb][j] = wr[2 * j + kk];
        }
        bcol[cb] = bias != nullptr ? bias[col] : 0.f;
    }
    float* tile = tiles + wave * 32 * LD;
    for (int g = blockIdx.x; g < row_groups; g += gridDim.x) {
        const int64_t row0 = ((int64_t)g * 4 + wave) * 32;
        if (row0 < M) {   // (wave-uniform)
            // the wave's own 32 x K tile, coalesced: K/4 lanes per row
            constexpr int LPR = K / 4, RPI = 64 / LPR;
#pragma unroll
            for (int it = 0; it < 32 / RPI; ++it) {
                const int r = it * RPI + lane / LPR, c4 = lane % LPR;
                const int64_t row = min(row0 + r, M - 1);
                *reinterpret_cast<f32x4*>(tile + r * LD + 4 * c4) = *reinterpret_cast<const f32x4*>(A + row * lda + 4 * c4);
            }
            f32x16 acc[NCB];
            chain_tile<K, NCB>(tile + cl * LD, kk, w, acc);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const int col = col0 + 32 * cb + cl;
                if (col < Nout) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int64_t row = row0 + acc_row(r, kk);
                        if (row < M) C[row * ldc + col] = bias != nullptr ? acc[cb][r] + bcol[cb] : acc[cb][r];
                    }
                }
            }
        }
    }
}

// The edge gate in the reference's order (k_edge_gate_ref, reference_order.hip) with the chain on the matrix cores.
// WPR = H / 64 waves per row group; RG = 4 / WPR row groups per workgroup iteration.  Two LDS tiles per row group: E (the A
// operand = the residual: the e rows, or with ENC the encoder's output) and X (the chain's result), so that the epilogue runs
// ROW-major - 16-byte pieces of B1h[src] / B2h[dst] / e', sixteen lanes to a 256-byte row segment - instead of 4-byte gathers
// in the accumulator layout.
// H = 256 (round 4): NCB = 1, eight waves (512 threads) to a row group, one row group per workgroup iteration.
template <int H, bool ENC>
__global__ __launch_bounds__((H == 256 ? 512 : kRmThreads)) void k_edge_gate_refm(const float* e_in, float* e_out, int64_t E, const float* __restrict__ B1h,
                                                               const float* __restrict__ B2h, int ldn, const int32_t* __restrict__ srt_src,
                                                               const int32_t* __restrict__ srt_dst, const float* __restrict__ W3, int ldw,
                                                               const float* __restrict__ b3, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, GateEnc enc, int iterations) {
    constexpr int LD = RmTile<H>::LD, NCB = H == 256 ? 1 : 2, CW = 32 * NCB, WPR = H / CW, NW = H == 256 ? 8 : 4, RG = NW / WPR;
    __shared__ __attribute__((aligned(16))) float tiles[RG * 2 * 32 * LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, cl = lane & 31, kk = lane >> 5;
    const int rg = wave / WPR, wsub = wave % WPR;      // row group inside the workgroup, column block inside the row group
    const int col0 = CW * wsub;
    const bool w_rows16 = ldw % 4 == 0 && ((uintptr_t)W3 % 16 == 0);
    float w[NCB][H / 2];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const float* wr = W3 + (int64_t)(col0 + 32 * cb + cl) * ldw;
        if (w_rows16) {
            load_w_row<H>(wr, kk, w[cb]);
        } else {
#pragma unroll
            for (int j = 0; j < H / 2; ++j) w[cb][j] = wr[2 * j + kk];
        }
    }
    // ENC: this lane's share of the encoder - hidden units 2 j + kk of its row, W2 rows of its columns (models/full_graph.py:27)
    float w1a[8], w1b[8], b1v[8], w2[NCB][8], b2c[NCB];
    if (ENC) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            w1a[j] = enc.W1[2 * (2 * j + kk)], w1b[j] = enc.W1[2 * (2 * j + kk) + 1], b1v[j] = enc.b1[2 * j + kk];
        }
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int col = col0 + 32 * cb + cl;
#pragma unroll
            for (int j = 0; j < 8; ++j) w2[cb][j] = enc.W2[col * 16 + 2 * j + kk];
            b2c[cb] = enc.b2[col];
        }
    }
    // epilogue mapping: lane -> row er + RPP it (it < NIT), columns col0 + ec .. + 3 (CW / 4 lanes to a row segment of the wave's columns)
    constexpr int LPS = CW / 4, RPP = 64 / LPS, NIT = 32 / RPP;
    const int er = lane / LPS, ec = 4 * (lane % LPS);
    const f32x4 b3v = *reinterpret_cast<const f32x4*>(b3 + col0 + ec), alv = *reinterpret_cast<const f32x4*>(scale + col0 + ec),
                bev = *reinterpret_cast<const f32x4*>(shift + col0 + ec);
    float* tileE = tiles + rg * 2 * 32 * LD;
    float* tileX = tileE + 32 * LD;
    for (int it = 0; it < iterations; ++it) {
        const int64_t row0 = (((int64_t)it * gridDim.x + blockIdx.x) * RG + rg) * 32;
        const bool live = row0 < E;   // (uniform over the waves of a row group)
        int si_mine = 0, di_mine = 0;
        if (live) {
            const int64_t mine = min(row0 + cl, E - 1);
            si_mine = srt_src[mine], di_mine = srt_dst[mine];
            if (ENC) {
                const int64_t eid = enc.srt_eid[mine];
                const float x0 = enc.e_raw[2 * eid], x1 = enc.e_raw[2 * eid + 1];
                float t[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] = fmaxf(__builtin_fmaf(x1, w1b[j], x0 * w1a[j]) + b1v[j], 0.f);
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    f32x16 ev;
#pragma unroll
                    for (int r = 0; r < 16; ++r) ev[r] = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) ev = __builtin_amdgcn_mfma_f32_32x32x2f32(t[j], w2[cb][j], ev, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 16; ++r) tileE[acc_row(r, kk) * LD + col0 + 32 * cb + cl] = ev[r] + b2c[cb];
                }
            } else {
                // the row group's 32 x H tile, coalesced, by its WPR waves together
                constexpr int LPR = H / 4, RPI = 64 * WPR / LPR;
                const int gl = wsub * 64 + lane;
#pragma unroll
                for (int i = 0; i < 32 / RPI; ++i) {
                    const int r = i * RPI + gl / LPR, c4 = gl % LPR;
                    const int64_t row = min(row0 + r, E - 1);
                    *reinterpret_cast<f32x4*>(tileE + r * LD + 4 * c4) = *reinterpret_cast<const f32x4*>(e_in + row * H + 4 * c4);
                }
            }
        }
        __syncthreads();   // the E tile is complete (and, in place, every global read of these rows has happened)
        if (live) {
            // the gathers of this tile go out before the chain; they are consumed after it
            f32x4 g1[NIT], g2[NIT];
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const int tr = er + RPP * i;
                const int64_t so = (int64_t)__shfl(si_mine, tr) * ldn + col0 + ec, dof = (int64_t)__shfl(di_mine, tr) * ldn + col0 + ec;
                g1[i] = *reinterpret_cast<const f32x4*>(B1h + so);
                g2[i] = *reinterpret_cast<const f32x4*>(B2h + dof);
            }
            f32x16 acc[NCB];
            chain_tile<H, NCB>(tileE + cl * LD, kk, w, acc);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) tileX[acc_row(r, kk) * LD + col0 + 32 * cb + cl] = acc[cb][r];
            __builtin_amdgcn_wave_barrier();   // the wave's own 64 columns of X: LDS keeps a wave's accesses in order
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const int tr = er + RPP * i;
                const f32x4 xv = *reinterpret_cast<const f32x4*>(tileX + tr * LD + col0 + ec);
                const f32x4 ev = *reinterpret_cast<const f32x4*>(tileE + tr * LD + col0 + ec);
                f32x4 y;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float g = g1[i][j] + g2[i][j];
                    const float x = g + (xv[j] + b3v[j]);
                    y[j] = fmaxf(__builtin_fmaf(x, alv[j], bev[j]), 0.f) + ev[j];
                }
                if (row0 + tr < E) *reinterpret_cast<f32x4*>(e_out + (row0 + tr) * H + col0 + ec) = y;
            }
        }
        __syncthreads();   // before the next iteration overwrites the tiles
    }
}

}  // namespace

int linear_refm_launch(const float* A, int64_t M, int K, int lda, const float* W, int ldw, const float* bias, int Nout, float* C, int ldc,
                       hipStream_t s) {
    const int64_t row_groups = (M + 127) / 128;
    GN_REQUIRE(row_groups < (1ll << 31), "linear_ref: too many rows");
    const int col_groups = K == 256 ? (Nout + 31) / 32 : (Nout + 63) / 64;
    // a few row groups per workgroup amortise the weight load (K/2 x 2 registers per lane from L2)
    const int gx = (int)std::min<int64_t>(row_groups, std::max<int64_t>(1, (int64_t)persistent_grid() * 8 / col_groups));
    if (K == 64)
        hipLaunchKernelGGL(k_linear_refm<64>, dim3(gx, col_groups), dim3(kRmThreads), 0, s, A, M, lda, W, ldw, bias, Nout, C, ldc, (int)row_groups);
    else if (K == 256)
        hipLaunchKernelGGL(k_linear_refm<256>, dim3(gx, col_groups), dim3(kRmThreads), 0, s, A, M, lda, W, ldw, bias, Nout, C, ldc, (int)row_groups);
    else
        hipLaunchKernelGGL(k_linear_refm<128>, dim3(gx, col_groups), dim3(kRmThreads), 0, s, A, M, lda, W, ldw, bias, Nout, C, ldc, (int)row_groups);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

template <int H, bool ENC>
static int launch_gate_refm(const float* e_in, float* e_out, int64_t E, const float* B1h, const float* B2h, int ldn, const int32_t* ss,
                            const int32_t* sd, const float* W3, int ldw, const float* b3, const float* scale, const float* shift,
                            const GateEnc& enc, hipStream_t s) {
    constexpr int RG = H == 256 ? 1 : 4 / (H / 64), NT = H == 256 ? 512 : kRmThreads;
    const int64_t groups = (E + 32 * RG - 1) / (32 * RG);   // workgroup iterations in total
    GN_REQUIRE(groups < (1ll << 31), "edge_gate_ref: too many edges");
    const int grid = (int)std::min<int64_t>(groups, (int64_t)persistent_grid() * 4);
    const int iterations = (int)((groups + grid - 1) / grid);
    hipLaunchKernelGGL((k_edge_gate_refm<H, ENC>), dim3(grid), dim3(NT), 0, s, e_in, e_out, E, B1h, B2h, ldn, ss, sd, W3, ldw, b3, scale,
                       shift, enc, iterations);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

int gate_refm_launch(int hidden, bool with_enc, const float* e_in, float* e_out, int64_t E, const float* B1h, const float* B2h, int ldn,
                     const int32_t* ss, const int32_t* sd, const float* W3, int ldw, const float* b3, const float* scale, const float* shift,
                     const GateEnc& enc, hipStream_t s) {
    if (hidden == 64)
        return with_enc ? launch_gate_refm<64, true>(e_in, e_out, E, B1h, B2h, ldn, ss, sd, W3, ldw, b3, scale, shift, enc, s)
                        : launch_gate_refm<64, false>(e_in, e_out, E, B1h, B2h, ldn, ss, sd, W3, ldw, b3, scale, shift, enc, s);
    if (hidden == 256)
        return with_enc ? launch_gate_refm<256, true>(e_in, e_out, E, B1h, B2h, ldn, ss, sd, W3, ldw, b3, scale, shift, enc, s)
                        : launch_gate_refm<256, false>(e_in, e_out, E, B1h, B2h, ldn, ss, sd, W3, ldw, b3, scale, shift, enc, s);
    return with_enc ? launch_gate_refm<128, true>(e_in, e_out, E, B1h, B2h, ldn, ss, sd, W3, ldw, b3, scale, shift, enc, s)
                    : launch_gate_refm<128, false>(e_in, e_out, E, B1h, B2h, ldn, ss, sd, W3, ldw, b3, scale, shift, enc, s);
}

}  // namespace gnnome
