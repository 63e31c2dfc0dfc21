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
                                                            const float* __restrict__ shift, int total_tiles, int norm_width) {
    constexpr int H = 32 * NB;
    __shared__ __attribute__((aligned(16))) float lds[tile_lds_floats<NB>() + 2 * kTileM];
    float* As = lds;
    float* Ws = lds + kTileM * kLdk;
    int* s_src = reinterpret_cast<int*>(lds + tile_lds_floats<NB>());
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
        sc[nb] = NORM == 2 ? 1.f : scale[32 * nb + cl];
        sh[nb] = NORM == 2 ? 0.f : shift[32 * nb + cl];
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
            // norm_width < H: a model narrower than the built width runs zero-padded - the statistics are those of its own channels
            // (the padded ones hold exact zeros: they add nothing to the sum and are left out of the centred second moment)
            float s1 = 0.f;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) s1 += v[nb];
            const float inv_w = 1.0f / (float)norm_width;
            const float mean = half_wave_sum(s1) * inv_w;
            float s2 = 0.f;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) s2 += (32 * nb + cl < norm_width) ? (v[nb] - mean) * (v[nb] - mean) : 0.f;
            const float rstd = rsqrtf(half_wave_sum(s2) * inv_w + kNormEps);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) v[nb] = (v[nb] - mean) * rstd;
        }
        if (lr < valid) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const uint32_t off = (uint32_t)(lr * H + 32 * nb + cl);
                // NORM == 2 ("raw"): the pre-normalisation sum B1h[src] + B2h[dst] + e*W3^T, for train-mode BatchNorm
                const float y = NORM == 2 ? v[nb] : fmaxf(v[nb] * sc[nb] + sh[nb], 0.f) + ein_tile[off];
                eout_tile[off] = y;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Wave-specialised form with the exact-fp32 MFMA (gnnome_set_tuning(0, 5 | 6); the shipped default is its bf16x6
// successor in edge_gate_bf.hip).
//
// What three earlier single-stream variants taught - persistent weight-stationary, software-pipelined, LDS-staged;
// their measurements are in profiles/r01_c2_persistent_gate.* and r01_c2_pipelined_gate.*, their code in the
// repository history: the MFMA side alone needs 0.27 ms and the memory side alone 0.25 ms per launch
// (E = 1M, H = 128), but in one instruction stream they add up (0.42 ms) instead of overlapping.  Two reasons.  (1) A wave's vector loads return IN ORDER, so any
// wait on a young L2 gather also waits for the older HBM tile prefetch: one tile (~4 us) is all the
// latency a single stream tolerates, and that is about what HBM needs under this load.  (2) hipcc cannot
// count outstanding loads across a loop back-edge and waits vmcnt(0) at the first cross-iteration use.
// Both disappear if no wave both computes and loads:
//
//   * 4 COMPUTE waves, one per SIMD: wave (rb, cb) keeps its 32 columns of W3 in 64 VGPRs for its whole
//     life (no LDS reads for the B operand), sweeps K with one accumulator - a single dependent MFMA chain
//     already saturates the SIMD's matrix pipe - and weaves the epilogue of the previous tile between the
//     MFMAs.  It issues stores but NEVER a load, so it never waits on memory.
//   * 8 LOAD waves in 4 groups.  Group g owns every 4th tile: it fetches the tile's e rows and the
//     B1h[src] / B2h[dst] rows as float4 row pieces (16 bytes per lane), and only FOUR iterations later
//     sums the gathers and writes both tiles into the LDS ring.  A group has exactly one batch in flight,
//     so the compiler's conservative vmcnt(0) is exact, and 4 x 48 KiB are in flight per CU.
//   * LDS is a 4-deep ring of (e tile, G tile) slots; one workgroup barrier per tile.
//
//   iteration i:   load group (i+1)%4 : registers -> slot (i+1)%4, then issue the loads of tile i+5
//                  compute waves      : MFMA sweep of tile i (slot i%4) + epilogue of tile i-1 (slot (i-1)%4)
// ---------------------------------------------------------------------------------------------------
template <int CB, int RB>
struct GateWS {
    static_assert(CB * RB == 4, "four compute waves");
    static constexpr int H = 32 * CB, TM = 32 * RB, LDK = H + 4, RING = 4, NGROUPS = 4, LWAVES = 2;
    static constexpr int NT = 64 * (4 + NGROUPS * LWAVES);           // 768 threads
    static constexpr int NP = TM * (H / 4) / (64 * LWAVES);          // float4 pieces per load lane per tile (= 8)
    static constexpr int kSlotFloats = TM * LDK;
    static constexpr int kLdsFloats = RING * 2 * kSlotFloats;
};

// One compute wave, one tile: acc = A_tile[32 rows] * W3[32 cols]^T over K = H, with the epilogue of the PREVIOUS tile
// (product accp, G rows Gp, residual rows Ap, all per-lane base pointers) spread between the MFMAs.
// FULL = every row of the previous tile exists: unconditional stores, so a k-step is one basic block and the
// sched_group_barrier pattern below can place the epilogue's LDS reads / VALU / store in the issue gaps that the
// dependent MFMA chain leaves (an MFMA blocks only the next MFMA; anything else issues under it).
// MODE 0: y = relu((acc + G) * sc + sh) + residual (the gate);  MODE >= 1: y = acc + G (raw: the input of a train-mode
// BatchNorm, or C += A * W^T with G = the old rows of C), with the shifted column sums s1 += y - sc, s2 += (y - sc)^2
// for the batch statistics when STATS (sc then carries the column's centre, sh is unused).
template <int H, bool FULL, int ABL, int MODE = 0, bool STATS = false>
__device__ __forceinline__ void gate_ws_sweep(f32x16& acc, const float* ap, const f32x4 (&wv)[H / 8], const float (&accp)[16],
                                              const float* Gp, const float* Ap, float* out_p, uint32_t lane_glb, int valid_p,
                                              float sc, float sh, float& s1, float& s2) {
    constexpr int LDK = H + 4, QS = H / 8, EPQ = 16 / QS;
    auto crow = [](int r) { return (r & 3) + 8 * (r >> 2); };
    f32x4 a_cur = *reinterpret_cast<const f32x4*>(ap);  // (ABL: measurement-only ablation mask, 0 in the shipped kernel)
    float g_cur[EPQ], r_cur[EPQ];
#pragma unroll
    for (int j = 0; j < EPQ; ++j) {
        g_cur[j] = Gp[crow(j) * LDK];
        r_cur[j] = MODE == 0 ? Ap[crow(j) * LDK] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < QS; ++q) {
        // operands of the NEXT k-step: A fragment and the epilogue's LDS values
        const int qn = q + 1 < QS ? q + 1 : q;
        const f32x4 a_nxt = *reinterpret_cast<const f32x4*>(ap + 8 * qn);
        float g_nxt[EPQ], r_nxt[EPQ];
#pragma unroll
        for (int j = 0; j < EPQ; ++j) {
            g_nxt[j] = Gp[crow(qn * EPQ + j) * LDK];
            r_nxt[j] = MODE == 0 ? Ap[crow(qn * EPQ + j) * LDK] : 0.f;
        }
        if (!(ABL & 8)) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[k], wv[q][k], acc, 0, 0, 0);
        } else {
            asm volatile("" ::"v"(a_cur));
        }
#pragma unroll
        for (int j = 0; j < EPQ; ++j) {
            const int r = q * EPQ + j;
            const float y = MODE == 0 ? fmaxf((accp[r] + g_cur[j]) * sc + sh, 0.f) + r_cur[j] : accp[r] + g_cur[j];
            if (STATS && (FULL || crow(r) < valid_p)) {
                const float d = y - sc;
                s1 += d;
                s2 = fmaf(d, d, s2);
            }
            if (ABL & 2) {
                asm volatile("" ::"v"(y));
            } else if (FULL || crow(r) < valid_p) {
                if (ABL & 16) {
                    __builtin_nontemporal_store(y, out_p + crow(r) * H + lane_glb);
                } else {
                    (out_p + crow(r) * H)[lane_glb] = y;
                }
            }
        }
        if (FULL) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);            // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1 + (MODE == 0 ? 2 : 1) * EPQ, 0);  // LDS reads for the next step
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4 * EPQ, 0);      // epilogue VALU
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x040, EPQ, 0);          // e' store
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        a_cur = a_nxt;
#pragma unroll
        for (int j = 0; j < EPQ; ++j) {
            g_cur[j] = g_nxt[j];
            r_cur[j] = r_nxt[j];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Slot hand-over without workgroup barriers (FLAGS variant): two monotone counters per ring slot in LDS.
//   full[s]: +1 by each of the two load waves of the slot's group once its share of a tile is written (2 per tile)
//   done[s]: +1 by each compute wave once it has finished BOTH uses of a tile (MFMA operand, then residual/G rows
//            in the next iteration's epilogue) (4 per tile)
// A wave's DS instructions execute in order, so a counter bump issued after the data accesses is ordered behind
// them without any s_waitcnt - in particular a compute wave never waits for its e' stores here, which a
// workgroup-scope release fence (vmcnt(0)) would make it do every tile.
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ void flag_wait(unsigned addr, unsigned want, int nap = 3) {
    unsigned v, spins = 0;
    for (;;) {
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
        if (__builtin_amdgcn_readfirstlane(v) >= want) break;
        if (++spins > (1u << 26)) __builtin_trap();  // a lost hand-over must end the launch, not hang the queue
        // consumers (nap 3) poll tightly: their wait is on the critical path.  Producers run a whole slot ahead and
        // poll rarely (nap 0 = 64 x 64 cycles): their ds_reads compete with the compute waves' operand reads.
        if (nap == 0) {
            __builtin_amdgcn_s_sleep(64);
        } else if (nap == 1) {
            __builtin_amdgcn_s_sleep(16);
        } else if (nap == 2) {
            __builtin_amdgcn_s_sleep(4);
        } else {
            __builtin_amdgcn_s_sleep(1);
        }
    }
}
__device__ __forceinline__ void flag_bump(unsigned addr, int lane) {
    if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(1u) : "memory");
}

// MODE 0: the gate.  MODE 1: raw gate x = B1h[src] + B2h[dst] + e W3^T (train mode), with per-workgroup shifted
// column sums for the BatchNorm batch statistics written to `stats` ([gridDim.x * RB][2H]; `scale` = the centre).
// MODE 2: C += A W^T on [E,H] rows (the backward's d e_in = d e' + dxe W3): G = the old rows of C, passed as B1h
// with ldn = H; srt_src / srt_dst / B2h unused.
template <int CB, int RB, int ABL, bool ENC, bool FLAGS, int MODE = 0>
__global__ __launch_bounds__(768) void k_edge_gate_ws(
    const float* e_in, float* e_out, int64_t E, const float* B1h, const float* __restrict__ B2h, int ldn,
    const int32_t* __restrict__ srt_src, const int32_t* __restrict__ srt_dst, const float* __restrict__ W3, int ldw,
    const float* __restrict__ scale, const float* __restrict__ shift, int num_tiles, int tiles_per_block, int interleave,
    GateEnc enc, int xp, float* __restrict__ stats = nullptr) {
    static_assert(MODE == 0 || (FLAGS && !ENC && ABL == 0), "raw modes: counter hand-over only");
    using P = GateWS<CB, RB>;
    constexpr int H = P::H, TM = P::TM, LDK = P::LDK, QS = H / 8, NP = P::NP, SLOT = P::kSlotFloats;
    constexpr int kEncFloats = ENC ? 16 * H + H + 48 : 0;  // W2^T [16][H], b2 [H], W1 [32], b1 [16]
    __shared__ __attribute__((aligned(16))) float lds[P::kLdsFloats + kEncFloats];
    float* Aring = lds;                       // [RING][TM][LDK]  e tiles
    float* Gring = lds + P::RING * SLOT;      // [RING][TM][LDK]  B1h[src] + B2h[dst]
    float* w2t = lds + P::kLdsFloats;         // ENC only
    float* b2s = w2t + 16 * H;
    float* w1s = b2s + H;
    float* b1s = w1s + 32;
    __shared__ unsigned flags[8];             // FLAGS only: full[4], done[4]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned full0 = lds_addr(&flags[0]), done0 = lds_addr(&flags[4]);
    // Tile order.  chunked: workgroup b owns the contiguous run [b*tpb, (b+1)*tpb).  interleaved: in round r the
    // whole chip works on ONE contiguous window of gridDim.x tiles (DRAM row locality for the e stream in and
    // out), and inside the window every XCD (blocks b % 8) gets a contiguous sub-window so that neighbouring
    // tiles - which share B2h[dst] rows and nearby B1h[src] rows - meet in the same L2.
    const int per_xcd = gridDim.x / kXcds;
    const int first = interleave ? (int)(blockIdx.x % kXcds) * per_xcd + (int)(blockIdx.x / kXcds) : (int)blockIdx.x * tiles_per_block;
    const int stride = interleave ? (int)gridDim.x : 1;
    int n;  // tiles of this workgroup
    if (interleave) {
        n = first < num_tiles ? (num_tiles - first + stride - 1) / stride : 0;
    } else {
        n = min(num_tiles, first + tiles_per_block) - first;
    }
    if (n <= 0) return;
    auto tile_of = [&](int r) { return first + r * stride; };
    auto tile_valid = [&](int r) { return (int)min((int64_t)TM, E - (int64_t)tile_of(r) * TM); };
    if (ENC) {  // encoder weights -> LDS, by the whole workgroup, before the roles split
        for (int i = tid; i < 16 * H; i += P::NT) w2t[i] = enc.W2[(i % H) * 16 + (i / H)];
        for (int i = tid; i < H; i += P::NT) b2s[i] = enc.b2[i];
        if (tid < 32) w1s[tid] = enc.W1[tid];
        if (tid < 16) b1s[tid] = enc.b1[tid];
    }
    if (FLAGS && tid < 8) flags[tid] = 0;
    if (ENC || FLAGS) __syncthreads();

    if (wave < 4) {
        // ------------------------------------------------------------------ compute wave
        const int rb = wave % RB, cb = wave / RB, cl = lane & 31, half = lane >> 5;
        const int col = 32 * cb + cl;
        f32x4 wv[QS];  // this lane's B operands: W3[col][8q + 4*half .. +3]
#pragma unroll
        for (int q = 0; q < QS; ++q) wv[q] = *reinterpret_cast<const f32x4*>(W3 + (int64_t)col * ldw + 8 * q + 4 * half);
        const float sc = MODE == 2 ? 0.f : scale[col], sh = MODE == 0 ? shift[col] : 0.f;   // MODE 1: sc = the column's centre
        float s1 = 0.f, s2 = 0.f;
        float accp[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) accp[r] = 0.f;

        // row / column of this lane's accumulator elements: element r sits in tile row lrow + crow(r)
        const int lrow = 32 * rb + 4 * half;
        const int lane_lds = lrow * LDK + col;
        const uint32_t lane_glb = (uint32_t)(lrow * H + col);
        auto crow = [](int r) { return (r & 3) + 8 * (r >> 2); };

        if (!FLAGS) __syncthreads();  // iteration -1: the load waves hand over tile 0
        if (xp & 1) __builtin_amdgcn_s_setprio(3);
        for (int i = 0; i < n; ++i) {
            if (FLAGS) flag_wait(full0 + 4 * (i & 3), 2u * ((unsigned)(i >> 2) + 1u));
            const float* As = Aring + (i & 3) * SLOT;
            // previous tile (i-1): its G rows and its e rows (the residual) are still in the ring
            const float* Gp = Gring + ((i - 1) & 3) * SLOT + lane_lds;
            const float* Ap = Aring + ((i - 1) & 3) * SLOT + lane_lds;
            const int valid_p = (i >= 1 ? tile_valid(i - 1) : 0) - lrow;
            float* out_p = e_out + (int64_t)tile_of(max(i - 1, 0)) * TM * H;

            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* ap = As + (32 * rb + cl) * LDK + 4 * half;
            if (valid_p + lrow >= TM) {  // wave-uniform: every row of the previous tile exists
                gate_ws_sweep<H, true, ABL, MODE, MODE == 1>(acc, ap, wv, accp, Gp, Ap, out_p, lane_glb, valid_p, sc, sh, s1, s2);
            } else {                     // first iteration (nothing pending) or the ragged last tile
                gate_ws_sweep<H, false, ABL, MODE, MODE == 1>(acc, ap, wv, accp, Gp, Ap, out_p, lane_glb, valid_p, sc, sh, s1, s2);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) accp[r] = acc[r];
            if (FLAGS) {
                if (i >= 1) flag_bump(done0 + 4 * ((i - 1) & 3), lane);   // tile i-1: operand and residual reads are over
            } else {
                __syncthreads();
            }
        }
        // drain: epilogue of the last tile
        {
            const float* Gp = Gring + ((n - 1) & 3) * SLOT + lane_lds;
            const float* Ap = Aring + ((n - 1) & 3) * SLOT + lane_lds;
            const int valid_p = tile_valid(n - 1) - lrow;
            float* out_p = e_out + (int64_t)tile_of(n - 1) * TM * H;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (crow(r) < valid_p) {
                    const float x = accp[r] + Gp[crow(r) * LDK];
                    if (MODE == 0) {
                        (out_p + crow(r) * H)[lane_glb] = fmaxf(x * sc + sh, 0.f) + Ap[crow(r) * LDK];
                    } else {
                        (out_p + crow(r) * H)[lane_glb] = x;
                        if (MODE == 1) {
                            s1 += x - sc;
                            s2 = fmaf(x - sc, x - sc, s2);
                        }
                    }
                }
            }
        }
        if (MODE == 1) {  // lanes l and l + 32 hold different rows of the same column
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (half == 0) {
                float* dst = stats + ((int64_t)blockIdx.x * RB + rb) * 2 * H;
                dst[col] = s1;
                dst[H + col] = s2;
            }
        }
    } else {
        // ------------------------------------------------------------------ load wave
        const int group = (wave - 4) / P::LWAVES;
        const int gl = ((wave - 4) % P::LWAVES) * 64 + lane;  // lane index inside the group, 0..127
        // piece p of this lane: row r0 + p * RSTEP, float4 column c4 (the group's 128 lanes cover RSTEP whole rows)
        constexpr int RSTEP = 64 * P::LWAVES / (H / 4);
        const int r0 = gl / (H / 4), c4 = gl % (H / 4);
        f32x4 a[NP], g1[NP], g2[NP];
        float raw0[NP], raw1[NP];  // ENC: the two raw edge features of each of this lane's rows
        if (ABL & 5) {
#pragma unroll
            for (int p = 0; p < NP; ++p) a[p] = g1[p] = g2[p] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        auto issue = [&](int r) {  // start fetching relative tile r
            const int64_t row0 = (int64_t)tile_of(r) * TM;
            const int valid = tile_valid(r);
            int si[NP], di[NP], ei[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int64_t row = row0 + min(r0 + p * RSTEP, valid - 1);
                if (MODE != 2) {
                    si[p] = srt_src[row];
                    di[p] = srt_dst[row];
                }
                if (ENC) ei[p] = enc.srt_eid[row];
            }
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int64_t row = row0 + min(r0 + p * RSTEP, valid - 1);
                if (ENC) {
                    raw0[p] = enc.e_raw[2 * (int64_t)ei[p]];
                    raw1[p] = enc.e_raw[2 * (int64_t)ei[p] + 1];
                } else if (!(ABL & 4)) {
                    const f32x4* src = reinterpret_cast<const f32x4*>(e_in + row * H + 4 * c4);
                    a[p] = (ABL & 16) ? __builtin_nontemporal_load(src) : *src;  // 16: streamed once, keep it out of L1
                }
                if (MODE == 2) {
                    g1[p] = *reinterpret_cast<const f32x4*>(B1h + row * ldn + 4 * c4);   // the old rows of C
                } else if (!(ABL & 1)) {
                    g1[p] = *reinterpret_cast<const f32x4*>(B1h + (int64_t)si[p] * ldn + 4 * c4);
                    g2[p] = *reinterpret_cast<const f32x4*>(B2h + (int64_t)di[p] * ldn + 4 * c4);
                }
            }
        };
        // ENC: turn the raw features of the pending tile into its encoded e rows.  Done in an iteration in which this
        // group has nothing else to do (two iterations before its hand-over), so that no iteration carries both a
        // hand-over and ~900 VALU operations per lane on the barrier's critical path.
        auto encode_pending = [&]() {
            // in the reference's order (torch nn.Linear on the CPU = k-ascending fma chain from zero, then + bias)
#pragma unroll
            for (int p = 0; p < NP; ++p) a[p] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int j = 0; j < 16; ++j) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(w2t + j * H + 4 * c4);
                const float wa = w1s[2 * j], wb = w1s[2 * j + 1], bj = b1s[j];
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const float t = fmaxf(__builtin_fmaf(raw1[p], wb, raw0[p] * wa) + bj, 0.f);
#pragma unroll
                    for (int i = 0; i < 4; ++i) a[p][i] = __builtin_fmaf(t, w[i], a[p][i]);
                }
            }
            {
                const f32x4 b2v = *reinterpret_cast<const f32x4*>(b2s + 4 * c4);
#pragma unroll
                for (int p = 0; p < NP; ++p) a[p] += b2v;
            }
        };
        if (group < n) {
            issue(group);
            if (ENC) encode_pending();
        }
        if (FLAGS) {
            // this group's tiles: group, group + 4, ...; each waits only for its own slot to be free
            for (int r = group; r < n; r += 4) {
                flag_wait(done0 + 4 * group, 4u * (unsigned)(r >> 2), (xp >> 1) & 3);
                float* As = Aring + group * SLOT;
                float* Gs = Gring + group * SLOT;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    *reinterpret_cast<f32x4*>(As + (r0 + p * RSTEP) * LDK + 4 * c4) = a[p];
                    *reinterpret_cast<f32x4*>(Gs + (r0 + p * RSTEP) * LDK + 4 * c4) = MODE == 2 ? g1[p] : g1[p] + g2[p];
                }
                flag_bump(full0 + 4 * group, lane);
                if (r + 4 < n) {
                    issue(r + 4);
                    if (ENC) encode_pending();
                }
            }
            return;
        }
        for (int i = -1; i < n; ++i) {
            if (ENC && i + 3 >= 4 && i + 3 < n && ((i + 3) & 3) == group) encode_pending();   // tile i+3, handed over at i+2
            const int r = i + 1;  // tile to hand over this iteration
            if (r < n && (r & 3) == group) {
                float* As = Aring + (r & 3) * SLOT;
                float* Gs = Gring + (r & 3) * SLOT;
#pragma unroll
                for (int p = 0; p < NP; ++p) {   // (ENC: a[] was encoded two iterations ago, see below)
                    *reinterpret_cast<f32x4*>(As + (r0 + p * RSTEP) * LDK + 4 * c4) = a[p];
                    *reinterpret_cast<f32x4*>(Gs + (r0 + p * RSTEP) * LDK + 4 * c4) = g1[p] + g2[p];
                }
                if (r + 4 < n) issue(r + 4);
            }
            __syncthreads();
        }
    }
}

template <int CB, int RB>
static int launch_gate_ws(const float* e_in, float* e_out, int64_t E, const float* B1h, const float* B2h, int ldn,
                          const int32_t* ss, const int32_t* sd, const float* W3, int ldw, const float* scale,
                          const float* shift, hipStream_t s, const GateEnc* enc = nullptr) {
    using P = GateWS<CB, RB>;
    const int64_t tiles = (E + P::TM - 1) / P::TM;
    GN_REQUIRE(tiles < (1ll << 31), "edge_gate: too many tiles");
    const int tpb = (int)((tiles + persistent_grid() - 1) / persistent_grid());  // one resident workgroup per CU (LDS-limited)
    const int interleave = tuning(kTuneGateTileOrder) == 1 ? 0 : 1;
    const int grid = interleave ? persistent_grid() : (int)((tiles + tpb - 1) / tpb);
    const bool flags = tuning(kTuneGateVariant) != 6;   // 6 = hand-over through workgroup barriers (the earlier form)
    const int xp = tuning(kTuneGateExperiment);
    if (enc != nullptr) {
        if (flags) {
            hipLaunchKernelGGL((k_edge_gate_ws<CB, RB, 0, true, true>), dim3(grid), dim3(P::NT), 0, s, e_in, e_out, E, B1h, B2h, ldn,
                               ss, sd, W3, ldw, scale, shift, (int)tiles, tpb, interleave, *enc, xp);
        } else {
            hipLaunchKernelGGL((k_edge_gate_ws<CB, RB, 0, true, false>), dim3(grid), dim3(P::NT), 0, s, e_in, e_out, E, B1h, B2h, ldn,
                               ss, sd, W3, ldw, scale, shift, (int)tiles, tpb, interleave, *enc, xp);
        }
        GN_LAUNCH_CHECK();
        return GNNOME_OK;
    }
    const GateEnc none = {};
#define GN_WS_ABL(M)                                                                                                              \
    case M:                                                                                                                       \
        if (flags) {                                                                                                              \
            hipLaunchKernelGGL((k_edge_gate_ws<CB, RB, M, false, true>), dim3(grid), dim3(P::NT), 0, s, e_in, e_out, E, B1h, B2h, \
                               ldn, ss, sd, W3, ldw, scale, shift, (int)tiles, tpb, interleave, none, xp);                            \
        } else {                                                                                                                  \
            hipLaunchKernelGGL((k_edge_gate_ws<CB, RB, M, false, false>), dim3(grid), dim3(P::NT), 0, s, e_in, e_out, E, B1h,     \
                               B2h, ldn, ss, sd, W3, ldw, scale, shift, (int)tiles, tpb, interleave, none, xp);                       \
        }                                                                                                                         \
        break;
    switch (tuning(kTuneGateAblation)) {
        GN_WS_ABL(1) GN_WS_ABL(2) GN_WS_ABL(4) GN_WS_ABL(8) GN_WS_ABL(7) GN_WS_ABL(15) GN_WS_ABL(16)
        default: GN_WS_ABL(0)
    }
#undef GN_WS_ABL
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

// train mode: raw gate + per-workgroup shifted column sums (see k_edge_gate_ws MODE 1)
template <int CB, int RB>
static int launch_ws_raw_stats(const float* e_in, float* x_out, int64_t E, const float* B1h, const float* B2h, int ldn,
                               const int32_t* ss, const int32_t* sd, const float* W3, int ldw, const float* center, float* stats,
                               hipStream_t s) {
    using P = GateWS<CB, RB>;
    const int64_t tiles = (E + P::TM - 1) / P::TM;
    GN_REQUIRE(tiles < (1ll << 31), "edge_gate_raw_stats: too many tiles");
    const int tpb = (int)((tiles + persistent_grid() - 1) / persistent_grid());
    GN_HIP(hipMemsetAsync(stats, 0, sizeof(float) * kNumCUs * RB * 2 * P::H, s));   // idle workgroups leave zeros
    const GateEnc none = {};
    hipLaunchKernelGGL((k_edge_gate_ws<CB, RB, 0, false, true, 1>), dim3(persistent_grid()), dim3(P::NT), 0, s, e_in, x_out, E, B1h, B2h, ldn, ss,
                       sd, W3, ldw, center, nullptr, (int)tiles, tpb, 1, none, tuning(kTuneGateExperiment), stats);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

// C[M,H] += A[M,H] * W[H,H]^T through the same kernel (MODE 2): the backward's d e_in = d e' + dxe * W3
template <int CB, int RB>
static int launch_ws_acc(const float* A, float* C, int64_t M, const float* W, int ldw, hipStream_t s) {
    using P = GateWS<CB, RB>;
    const int64_t tiles = (M + P::TM - 1) / P::TM;
    GN_REQUIRE(tiles < (1ll << 31), "linear_acc: too many tiles");
    const int tpb = (int)((tiles + persistent_grid() - 1) / persistent_grid());
    const GateEnc none = {};
    hipLaunchKernelGGL((k_edge_gate_ws<CB, RB, 0, false, true, 2>), dim3(persistent_grid()), dim3(P::NT), 0, s, A, C, M, C, nullptr, P::H, nullptr,
                       nullptr, W, ldw, nullptr, nullptr, (int)tiles, tpb, 1, none, tuning(kTuneGateExperiment), nullptr);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

int ws_linear_acc(const float* A, int64_t M, int K, const float* W, int ldw, float* C, hipStream_t s) {
    if (tuning(kTuneGateVariant) == 0) {
        GateBfArgs a = {};
        a.e_in = A; a.e_out = C; a.E = M; a.B1h = C; a.ldn = K; a.W3 = W; a.ldw = ldw;
        return gate_bf_launch(K, 2, false, a, s);
    }
    return K == 128 ? launch_ws_acc<4, 1>(A, C, M, W, ldw, s) : launch_ws_acc<2, 2>(A, C, M, W, ldw, s);
}

template <int NB>
static int launch_gate(const float* e_in, float* e_out, int64_t E, const float* B1h, const float* B2h, int ldn,
                       const int32_t* ss, const int32_t* sd, const float* W3, int ldw, int norm, const float* scale,
                       const float* shift, hipStream_t s, int norm_width = 32 * NB) {
    const int64_t tiles = (E + kTileM - 1) / kTileM;
    GN_REQUIRE(tiles < (1ll << 31), "edge_gate: too many tiles");
    if (norm == 2) {
        hipLaunchKernelGGL((k_edge_gate<NB, 2>), dim3((unsigned)tiles), dim3(kGemmThreads), 0, s, e_in, e_out, E, B1h, B2h, ldn,
                           ss, sd, W3, ldw, scale, shift, (int)tiles, norm_width);
    } else if (norm == GNNOME_NORM_AFFINE) {
        hipLaunchKernelGGL((k_edge_gate<NB, GNNOME_NORM_AFFINE>), dim3((unsigned)tiles), dim3(kGemmThreads), 0, s, e_in,
                           e_out, E, B1h, B2h, ldn, ss, sd, W3, ldw, scale, shift, (int)tiles, norm_width);
    } else {
        hipLaunchKernelGGL((k_edge_gate<NB, GNNOME_NORM_LAYER>), dim3((unsigned)tiles), dim3(kGemmThreads), 0, s, e_in,
                           e_out, E, B1h, B2h, ldn, ss, sd, W3, ldw, scale, shift, (int)tiles, norm_width);
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
    // norm_kind = GNNOME_NORM_LAYER_OVER(w): LayerNorm whose statistics run over the first w channels (a zero-padded narrower model)
    const int norm_width = (norm_kind >> 8) ? (norm_kind >> 8) : hidden;
    norm_kind &= 0xFF;
    GN_REQUIRE(norm_kind == GNNOME_NORM_AFFINE || norm_kind == GNNOME_NORM_LAYER, "edge_gate: bad norm_kind %d", norm_kind);
    GN_REQUIRE(norm_width >= 1 && norm_width <= hidden && (norm_width == hidden || norm_kind == GNNOME_NORM_LAYER), "edge_gate: bad norm width %d", norm_width);
    GN_REQUIRE(ld_node >= hidden && ldw >= hidden && ldw % 4 == 0, "edge_gate: bad strides");
    GN_REQUIRE(((uintptr_t)e_in % 16 == 0) && ((uintptr_t)W3 % 16 == 0), "edge_gate: e_in and W3 must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    // variant 0 = shipped default (bf16x6 edge-tile kernel where it applies, else the tile kernel); 1 = tile kernel;
    // 5 / 6 = exact-fp32 wave-specialised kernel
    const int variant = tuning(kTuneGateVariant);
    const bool rows16 = ld_node % 4 == 0 && ((uintptr_t)B1h % 16 == 0) && ((uintptr_t)B2h % 16 == 0);
    const bool persistent_ok = norm_kind == GNNOME_NORM_AFFINE && (hidden == 64 || hidden == 128);
    if (persistent_ok && variant != 1) {
        if ((variant == 0 || variant == 7 || variant == 8) && rows16) {   // the shipped default: bf16x6 edge-tile kernel (edge_gate_bf.hip); 7: its plane form
            GateBfArgs a = {};
            a.e_in = e_in; a.e_out = e_out; a.E = num_edges; a.B1h = B1h; a.B2h = B2h; a.ldn = ld_node; a.srt_src = srt_src;
            a.srt_dst = srt_dst; a.W3 = W3; a.ldw = ldw; a.scale = norm_scale; a.shift = norm_shift;
            return gate_bf_launch(hidden, 0, false, a, s);
        }
        if ((variant == 5 || variant == 6) && rows16) {   // exact-fp32 MFMA, counter (5) / barrier (6) hand-over
            if (hidden == 128)
                return launch_gate_ws<4, 1>(e_in, e_out, num_edges, B1h, B2h, ld_node, srt_src, srt_dst, W3, ldw, norm_scale, norm_shift, s);
            return launch_gate_ws<2, 2>(e_in, e_out, num_edges, B1h, B2h, ld_node, srt_src, srt_dst, W3, ldw, norm_scale, norm_shift, s);
        }
    }
    // H = 256 with separate input and output buffers: the streaming kernel (its column chunks run in different workgroups,
    // so it cannot update e in place; in place - or LayerNorm - goes to the 128-edge tile kernel below)
    if (hidden == 256 && norm_kind == GNNOME_NORM_AFFINE && (variant == 0 || variant == 9) && e_in != e_out && ld_node % 4 == 0 &&
        ((uintptr_t)e_out % 16 == 0)) {
        if (variant == 0 && rows16) {   // the shipped default: wave-specialised plane form, W3 in registers (edge_gate_pl256.hip)
            GateBfArgs a = {};
            a.e_in = e_in; a.e_out = e_out; a.E = num_edges; a.B1h = B1h; a.B2h = B2h; a.ldn = ld_node; a.srt_src = srt_src;
            a.srt_dst = srt_dst; a.W3 = W3; a.ldw = ldw; a.scale = norm_scale; a.shift = norm_shift;
            return gate_pl256_launch(0, a, s);
        }
        return gate_stream_launch(e_in, e_out, num_edges, B1h, B2h, ld_node, srt_src, srt_dst, W3, ldw, norm_scale, norm_shift, s);   // variant 9: round 2's streaming kernel
    }
    switch (hidden) {
        case 64: return launch_gate<2>(e_in, e_out, num_edges, B1h, B2h, ld_node, srt_src, srt_dst, W3, ldw, norm_kind, norm_scale, norm_shift, s, norm_width);
        case 128: return launch_gate<4>(e_in, e_out, num_edges, B1h, B2h, ld_node, srt_src, srt_dst, W3, ldw, norm_kind, norm_scale, norm_shift, s, norm_width);
        case 256: return launch_gate<8>(e_in, e_out, num_edges, B1h, B2h, ld_node, srt_src, srt_dst, W3, ldw, norm_kind, norm_scale, norm_shift, s, norm_width);
        default: set_error("edge_gate: hidden=%d not in {64,128,256}", hidden); return GNNOME_EINVAL;
    }
}

extern "C" int gnnome_edge_gate_raw_f32(const float* e_in, float* x_out, int64_t num_edges, int hidden, const float* B1h,
                                        const float* B2h, int ld_node, const int32_t* srt_src, const int32_t* srt_dst,
                                        const float* W3, int ldw, void* stream) {
    using namespace gnnome;
    GN_REQUIRE(num_edges >= 0, "edge_gate_raw: negative edge count");
    if (num_edges == 0) return GNNOME_OK;
    GN_REQUIRE(e_in && x_out && x_out != e_in && B1h && B2h && srt_src && srt_dst && W3, "edge_gate_raw: bad pointers");
    GN_REQUIRE(ld_node >= hidden && ldw >= hidden && ldw % 4 == 0, "edge_gate_raw: bad strides");
    GN_REQUIRE(((uintptr_t)e_in % 16 == 0) && ((uintptr_t)W3 % 16 == 0), "edge_gate_raw: e_in and W3 must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (hidden == 256 && tuning(kTuneGateVariant) == 0 && ld_node % 4 == 0 && ((uintptr_t)B1h % 16 == 0) && ((uintptr_t)B2h % 16 == 0) &&
        ((uintptr_t)x_out % 16 == 0)) {   // the plane form's raw mode (no statistics)
        GateBfArgs a = {};
        a.e_in = e_in; a.e_out = x_out; a.E = num_edges; a.B1h = B1h; a.B2h = B2h; a.ldn = ld_node; a.srt_src = srt_src;
        a.srt_dst = srt_dst; a.W3 = W3; a.ldw = ldw;
        return gate_pl256_launch(1, a, s);
    }
    switch (hidden) {
        case 64: return launch_gate<2>(e_in, x_out, num_edges, B1h, B2h, ld_node, srt_src, srt_dst, W3, ldw, 2, nullptr, nullptr, s);
        case 128: return launch_gate<4>(e_in, x_out, num_edges, B1h, B2h, ld_node, srt_src, srt_dst, W3, ldw, 2, nullptr, nullptr, s);
        case 256: return launch_gate<8>(e_in, x_out, num_edges, B1h, B2h, ld_node, srt_src, srt_dst, W3, ldw, 2, nullptr, nullptr, s);
        default: set_error("edge_gate_raw: hidden=%d not in {64,128,256}", hidden); return GNNOME_EINVAL;
    }
}

extern "C" int gnnome_edge_gate_raw_stats_rows(int hidden, int* rows_host) {
    using namespace gnnome;
    GN_REQUIRE(rows_host && (hidden == 64 || hidden == 128 || hidden == 256), "edge_gate_raw_stats_rows: hidden=%d not in {64,128,256}", hidden);
    if (hidden == 256) {   // the plane form at H = 256: one row per load / store wave
        *rows_host = gate_pl256_stats_rows();
        return GNNOME_OK;
    }
    // bf16x6 kernel (variant 0): one row per load/store wave (8 per workgroup); exact-fp32 kernel: one per row block
    *rows_host = (tuning(kTuneGateVariant) == 0 || tuning(kTuneGateVariant) == 8) ? kNumCUs * 8 : kNumCUs * (hidden == 128 ? 1 : 2);
    return GNNOME_OK;
}

static int raw_stats_impl(const float* e_in, void* x_out, int64_t num_edges, int hidden, const float* B1h,
                          const float* B2h, int ld_node, const int32_t* srt_src, const int32_t* srt_dst,
                          const float* W3, int ldw, const float* center, float* stats_partial, void* stream, bool x16) {
    using namespace gnnome;
    GN_REQUIRE(num_edges > 0, "edge_gate_raw_stats: needs at least one edge");
    GN_REQUIRE(e_in && x_out != (const void*)e_in && B1h && B2h && srt_src && srt_dst && W3 && center && stats_partial,
               "edge_gate_raw_stats: bad pointers");
    // x_out == NULL: the statistics alone (the first pass of the two-pass training forward; hidden = 128, the plane form)
    GN_REQUIRE(x_out || ((hidden == 128 || (hidden == 256 && tuning(kTuneArith) == 0 && !x16)) && tuning(kTuneGateVariant) == 0),
               "edge_gate_raw_stats: x_out may be NULL at hidden = 128 and, as fp16x3 with fp32 storage, 256 (default kernels)");
    GN_REQUIRE(hidden == 64 || hidden == 128 || hidden == 256, "edge_gate_raw_stats: hidden=%d not in {64,128,256}", hidden);
    GN_REQUIRE(ld_node >= hidden && ld_node % 4 == 0 && ldw >= hidden && ldw % 4 == 0, "edge_gate_raw_stats: bad strides");
    GN_REQUIRE(((uintptr_t)e_in % 16 == 0) && ((uintptr_t)W3 % 16 == 0) && ((uintptr_t)B1h % 16 == 0) && ((uintptr_t)B2h % 16 == 0),
               "edge_gate_raw_stats: 16-byte alignment required");
    hipStream_t s = (hipStream_t)stream;
    if (hidden == 256) {
        GateBfArgs a = {};
        a.e_in = e_in; a.e_out = (float*)x_out; a.E = num_edges; a.B1h = B1h; a.B2h = B2h; a.ldn = ld_node; a.srt_src = srt_src;
        a.srt_dst = srt_dst; a.W3 = W3; a.ldw = ldw; a.scale = center; a.stats = stats_partial;
        return gate_pl256_launch(1, a, s, x16);
    }
    if (tuning(kTuneGateVariant) == 0 || tuning(kTuneGateVariant) == 8 || x16) {
        GateBfArgs a = {};
        a.e_in = e_in; a.e_out = (float*)x_out; a.E = num_edges; a.B1h = B1h; a.B2h = B2h; a.ldn = ld_node; a.srt_src = srt_src;
        a.srt_dst = srt_dst; a.W3 = W3; a.ldw = ldw; a.scale = center; a.stats = stats_partial;
        return gate_bf_launch(hidden, 1, false, a, s, x16, x_out == nullptr ? 1 : 0);
    }
    if (hidden == 128)
        return launch_ws_raw_stats<4, 1>(e_in, (float*)x_out, num_edges, B1h, B2h, ld_node, srt_src, srt_dst, W3, ldw, center, stats_partial, s);
    return launch_ws_raw_stats<2, 2>(e_in, (float*)x_out, num_edges, B1h, B2h, ld_node, srt_src, srt_dst, W3, ldw, center, stats_partial, s);
}

extern "C" int gnnome_edge_gate_raw_stats_f32(const float* e_in, float* x_out, int64_t num_edges, int hidden, const float* B1h,
                                              const float* B2h, int ld_node, const int32_t* srt_src, const int32_t* srt_dst,
                                              const float* W3, int ldw, const float* center, float* stats_partial, void* stream) {
    return raw_stats_impl(e_in, x_out, num_edges, hidden, B1h, B2h, ld_node, srt_src, srt_dst, W3, ldw, center, stats_partial, stream, false);
}

extern "C" int gnnome_edge_gate_raw_stats_x16(const float* e_in, uint16_t* x_out, int64_t num_edges, int hidden, const float* B1h,
                                              const float* B2h, int ld_node, const int32_t* srt_src, const int32_t* srt_dst,
                                              const float* W3, int ldw, const float* center, float* stats_partial, void* stream) {
    return raw_stats_impl(e_in, x_out, num_edges, hidden, B1h, B2h, ld_node, srt_src, srt_dst, W3, ldw, center, stats_partial, stream, true);
}

// The second pass of the two-pass training forward (hidden = 128): e_out = relu(xe * scale + shift) + e_in with the pre-normalisation rows
// xe = e_in W3^T + B1h[src] + B2h[dst] ALSO written to x_out (fp32, or bf16 with x16: e_out is then computed from the rounded rows).
static int gate_bn_impl(const float* e_in, float* e_out, void* x_out, int64_t num_edges, int hidden, const float* B1h, const float* B2h, int ld_node,
                        const int32_t* srt_src, const int32_t* srt_dst, const float* W3, int ldw, const float* scale, const float* shift,
                        void* stream, bool x16) {
    using namespace gnnome;
    GN_REQUIRE(num_edges >= 0, "edge_gate_bn: negative edge count");
    if (num_edges == 0) return GNNOME_OK;
    GN_REQUIRE(e_in && e_out && x_out && x_out != (const void*)e_in && x_out != (void*)e_out && B1h && B2h && srt_src && srt_dst && W3 && scale && shift,
               "edge_gate_bn: bad pointers");
    GN_REQUIRE((hidden == 128 || (hidden == 256 && tuning(kTuneArith) == 0 && !x16)) && tuning(kTuneGateVariant) == 0,
               "edge_gate_bn: built at hidden = 128 and, as fp16x3 with fp32 storage, 256 (default kernels), got %d", hidden);
    GN_REQUIRE(ld_node >= hidden && ld_node % 4 == 0 && ldw >= hidden && ldw % 4 == 0, "edge_gate_bn: bad strides");
    GN_REQUIRE(((uintptr_t)e_in % 16 == 0) && ((uintptr_t)e_out % 16 == 0) && ((uintptr_t)x_out % 16 == 0) && ((uintptr_t)W3 % 16 == 0) &&
                   ((uintptr_t)B1h % 16 == 0) && ((uintptr_t)B2h % 16 == 0), "edge_gate_bn: 16-byte alignment required");
    GateBfArgs a = {};
    a.e_in = e_in; a.e_out = e_out; a.E = num_edges; a.B1h = B1h; a.B2h = B2h; a.ldn = ld_node; a.srt_src = srt_src; a.srt_dst = srt_dst;
    a.W3 = W3; a.ldw = ldw; a.scale = scale; a.shift = shift; a.bnb.a_out = (float*)x_out;
    if (hidden == 256) return gate_pl256_launch(0, a, (hipStream_t)stream, false);   // (edge_tile_f16.hip: the gate writes xe when bnb.a_out is set)
    return gate_bf_launch(hidden, 0, false, a, (hipStream_t)stream, x16, 1);
}

extern "C" int gnnome_edge_gate_bn_f32(const float* e_in, float* e_out, float* x_out, int64_t num_edges, int hidden, const float* B1h,
                                       const float* B2h, int ld_node, const int32_t* srt_src, const int32_t* srt_dst, const float* W3, int ldw,
                                       const float* scale, const float* shift, void* stream) {
    return gate_bn_impl(e_in, e_out, x_out, num_edges, hidden, B1h, B2h, ld_node, srt_src, srt_dst, W3, ldw, scale, shift, stream, false);
}

extern "C" int gnnome_edge_gate_bn_x16(const float* e_in, float* e_out, uint16_t* x_out, int64_t num_edges, int hidden, const float* B1h,
                                       const float* B2h, int ld_node, const int32_t* srt_src, const int32_t* srt_dst, const float* W3, int ldw,
                                       const float* scale, const float* shift, void* stream) {
    return gate_bn_impl(e_in, e_out, x_out, num_edges, hidden, B1h, B2h, ld_node, srt_src, srt_dst, W3, ldw, scale, shift, stream, true);
}

extern "C" int gnnome_edge_gate_encode_f32(const float* e_raw, const int32_t* srt_eid, const float* encW1, const float* encb1,
                                           const float* encW2, const float* encb2, float* e_out, int64_t num_edges, int hidden,
                                           const float* B1h, const float* B2h, int ld_node, const int32_t* srt_src,
                                           const int32_t* srt_dst, const float* W3, int ldw, const float* norm_scale,
                                           const float* norm_shift, void* stream) {
    using namespace gnnome;
    GN_REQUIRE(num_edges >= 0, "edge_gate_encode: negative edge count");
    if (num_edges == 0) return GNNOME_OK;
    GN_REQUIRE(e_raw && srt_eid && encW1 && encb1 && encW2 && encb2 && e_out && B1h && B2h && srt_src && srt_dst && W3 &&
                   norm_scale && norm_shift, "edge_gate_encode: null pointer");
    GN_REQUIRE(hidden == 64 || hidden == 128 || (hidden == 256 && tuning(kTuneArith) == 0 && tuning(kTuneGateVariant) == 0),
               "edge_gate_encode: hidden=%d not in {64,128,256 (default kernels)}", hidden);
    GN_REQUIRE(ld_node % 4 == 0 && ldw % 4 == 0 && ((uintptr_t)B1h % 16 == 0) && ((uintptr_t)B2h % 16 == 0) && ((uintptr_t)W3 % 16 == 0),
               "edge_gate_encode: alignment");
    const GateEnc enc = {e_raw, srt_eid, encW1, encb1, encW2, encb2};
    hipStream_t s = (hipStream_t)stream;
    if (hidden == 256) {
        GN_REQUIRE((uintptr_t)e_out % 16 == 0, "edge_gate_encode: e_out must be 16-byte aligned");
        GateBfArgs a = {};
        a.e_out = e_out; a.E = num_edges; a.B1h = B1h; a.B2h = B2h; a.ldn = ld_node; a.srt_src = srt_src; a.srt_dst = srt_dst;
        a.W3 = W3; a.ldw = ldw; a.scale = norm_scale; a.shift = norm_shift; a.enc = enc;
        return gate_enc256_launch(a, s);
    }
    if (tuning(kTuneGateVariant) == 0 || tuning(kTuneGateVariant) == 7 || tuning(kTuneGateVariant) == 8) {
        GateBfArgs a = {};
        a.e_out = e_out; a.E = num_edges; a.B1h = B1h; a.B2h = B2h; a.ldn = ld_node; a.srt_src = srt_src; a.srt_dst = srt_dst;
        a.W3 = W3; a.ldw = ldw; a.scale = norm_scale; a.shift = norm_shift; a.enc = enc;
        return gate_bf_launch(hidden, 0, true, a, s);
    }
    if (hidden == 128)
        return launch_gate_ws<4, 1>(nullptr, e_out, num_edges, B1h, B2h, ld_node, srt_src, srt_dst, W3, ldw, norm_scale, norm_shift, s, &enc);
    return launch_gate_ws<2, 2>(nullptr, e_out, num_edges, B1h, B2h, ld_node, srt_src, srt_dst, W3, ldw, norm_scale, norm_shift, s, &enc);
}
