// gnnome_edge_gate_f32: the per-edge gate of SymGatedGCN in one pass over e[E,H] (sorted order).
//
//   e_out[p,:] = relu(norm_e(B1h[src_p,:] + B2h[dst_p,:] + e_in[p,:] * W3^T)) + e_in[p,:]
//
// Reference lines replaced: gated_gcn_full.py:97 (B_3(e), the path's dominant GEMM, 2*E*H^2 flop),
// :104 (DGL gsddmm u_add_v), :105-110 (add, bn_e, relu, residual); the reversed-graph copy at
// :117-122 evaluates the same expression with the operands of the first add commuted, so it is not
// recomputed.  The reference spends ~13 reads + 8 writes of [E,H] here; this kernel reads e once
// and writes e' once.
//
// Structure: 128-edge tile per workgroup; the B_3 GEMM runs on v_mfma_f32_32x32x2_f32 with the
// accumulator pre-loaded with the gathered B1h[src] + B2h[dst] (MFMA computes D = A*B + C, so the
// u_add_v and the "+ B3e" cost nothing); norm / relu / residual are applied to the accumulator
// registers and e' is stored straight from them (each store instruction covers two 128-byte row
// segments).  Bound: fp32 MFMA (2*H^2 flop per edge against 8*H bytes per edge; AI = H/4 flop/B vs a
// machine balance of ~20-25 flop/B, so H >= 128 is matrix-bound, H = 64 HBM-bound).
#include "gemm_tile.h"

namespace gnnome {

template <int NB, int NORM>
__global__ __launch_bounds__(kGemmThreads) void k_edge_gate(const float* e_in, float* e_out, int64_t E,
                                                            const float* __restrict__ B1h,
                                                            const float* __restrict__ B2h, int ldn,
                                                            const int32_t* __restrict__ srt_src,
                                                            const int32_t* __restrict__ srt_dst,
                                                            const float* __restrict__ W3, int ldw,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift, int total_tiles) {
    constexpr int H = 32 * NB;
    __shared__ __attribute__((aligned(16))) float lds[(kTileM + H) * kLdk + 2 * kTileM];
    float* As = lds;
    float* Ws = lds + kTileM * kLdk;
    int* s_src = reinterpret_cast<int*>(lds + (kTileM + H) * kLdk);
    int* s_dst = s_src + kTileM;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tile = xcd_remap(blockIdx.x, total_tiles);
    const int64_t row0 = (int64_t)tile * kTileM;
    const int valid = (int)min((int64_t)kTileM, E - row0);

    if (tid < kTileM) {
        const int r = min(tid, valid - 1);
        s_src[tid] = srt_src[row0 + r];
        s_dst[tid] = srt_dst[row0 + r];
    }
    __syncthreads();

    // accumulator <- B1h[src] + B2h[dst] in the MFMA C/D layout
    const int cl = lane & 31;
    f32x16 acc[NB];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int lr = 32 * wave + cd_row(r, lane);
        const float* p1 = B1h + (int64_t)s_src[lr] * ldn + cl;
        const float* p2 = B2h + (int64_t)s_dst[lr] * ldn + cl;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb][r] = p1[32 * nb] + p2[32 * nb];
    }

    tile_gemm<NB>(acc, e_in, row0, E, H, W3, 0, H, ldw, H, As, Ws, tid);

    // epilogue on the accumulator registers
    float sc[NB], sh[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        sc[nb] = scale[32 * nb + cl];
        sh[nb] = shift[32 * nb + cl];
    }
    const float* ein_tile = e_in + row0 * H;
    float* eout_tile = e_out + row0 * H;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int lr = 32 * wave + cd_row(r, lane);
        float v[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) v[nb] = acc[nb][r];
        if (NORM == GNNOME_NORM_LAYER) {
            // a row lives in the NB registers of the 32 lanes sharing (lane >> 5)
            float s1 = 0.f;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) s1 += v[nb];
            const float mean = half_wave_sum(s1) * (1.0f / H);
            float s2 = 0.f;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) s2 += (v[nb] - mean) * (v[nb] - mean);
            const float rstd = rsqrtf(half_wave_sum(s2) * (1.0f / H) + kNormEps);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) v[nb] = (v[nb] - mean) * rstd;
        }
        if (lr < valid) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const uint32_t off = (uint32_t)(lr * H + 32 * nb + cl);
                const float y = fmaxf(v[nb] * sc[nb] + sh[nb], 0.f) + ein_tile[off];
                eout_tile[off] = y;
            }
        }
    }
}

template <int NB>
static int launch_gate(const float* e_in, float* e_out, int64_t E, const float* B1h, const float* B2h, int ldn,
                       const int32_t* ss, const int32_t* sd, const float* W3, int ldw, int norm, const float* scale,
                       const float* shift, hipStream_t s) {
    const int64_t tiles = (E + kTileM - 1) / kTileM;
    GN_REQUIRE(tiles < (1ll << 31), "edge_gate: too many tiles");
    if (norm == GNNOME_NORM_AFFINE) {
        hipLaunchKernelGGL((k_edge_gate<NB, GNNOME_NORM_AFFINE>), dim3((unsigned)tiles), dim3(kGemmThreads), 0, s, e_in,
                           e_out, E, B1h, B2h, ldn, ss, sd, W3, ldw, scale, shift, (int)tiles);
    } else {
        hipLaunchKernelGGL((k_edge_gate<NB, GNNOME_NORM_LAYER>), dim3((unsigned)tiles), dim3(kGemmThreads), 0, s, e_in,
                           e_out, E, B1h, B2h, ldn, ss, sd, W3, ldw, scale, shift, (int)tiles);
    }
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

}  // namespace gnnome

extern "C" int gnnome_edge_gate_f32(const float* e_in, float* e_out, int64_t num_edges, int hidden, const float* B1h,
                                    const float* B2h, int ld_node, const int32_t* srt_src, const int32_t* srt_dst,
                                    const float* W3, int ldw, int norm_kind, const float* norm_scale,
                                    const float* norm_shift, void* stream) {
    using namespace gnnome;
    GN_REQUIRE(num_edges >= 0, "edge_gate: negative edge count");
    if (num_edges == 0) return GNNOME_OK;
    GN_REQUIRE(e_in && e_out && B1h && B2h && srt_src && srt_dst && W3 && norm_scale && norm_shift,
               "edge_gate: null pointer");
    GN_REQUIRE(norm_kind == GNNOME_NORM_AFFINE || norm_kind == GNNOME_NORM_LAYER, "edge_gate: bad norm_kind %d", norm_kind);
    GN_REQUIRE(ld_node >= hidden && ldw >= hidden && ldw % 4 == 0, "edge_gate: bad strides");
    GN_REQUIRE(((uintptr_t)e_in % 16 == 0) && ((uintptr_t)W3 % 16 == 0), "edge_gate: e_in and W3 must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    switch (hidden) {
        case 64: return launch_gate<2>(e_in, e_out, num_edges, B1h, B2h, ld_node, srt_src, srt_dst, W3, ldw, norm_kind, norm_scale, norm_shift, s);
        case 128: return launch_gate<4>(e_in, e_out, num_edges, B1h, B2h, ld_node, srt_src, srt_dst, W3, ldw, norm_kind, norm_scale, norm_shift, s);
        case 256: return launch_gate<8>(e_in, e_out, num_edges, B1h, B2h, ld_node, srt_src, srt_dst, W3, ldw, norm_kind, norm_scale, norm_shift, s);
        default: set_error("edge_gate: hidden=%d not in {64,128,256}", hidden); return GNNOME_EINVAL;
    }
}
