// Edge-tile kernel, second generation: the [32 x H] x [H x H] product of every tile runs on the bf16 matrix cores as
// an fp32-FAITHFUL three-way split, which turns the kernel from MFMA-bound (exact-fp32 MFMA, 157 TF peak) into
// HBM-bound.
//
//   x = x1 + x2 + x3 EXACTLY, each xi a bf16: x1 = the top 16 bits of x, x2 = the top 16 bits of x - x1, x3 = the rest
//   (24 mantissa bits = 3 x 8; truncation, so the remainders are exact and x3 is exactly representable).
//   a*b = sum of the six products a1b1, a1b2, a2b1, a2b2, a1b3, a3b1 + (a2b3 + a3b2 + a3b3 <= 3 * 2^-24 |ab|):
//   every product of two bf16 is exact in fp32, the accumulation is the matrix core's fp32 adder, and what is dropped
//   is of the size of ONE fp32 rounding.  Measured against an fp64 evaluation of the whole model the result is as
//   close as the exact-fp32 path or closer ("bf16x6").
//   Cost: 6 v_mfma_f32_32x32x16_bf16 (32 cycles each) per K = 16, against 8 v_mfma_f32_32x32x2_f32 (64 cycles each):
//   192 vs 512 cycles, and the tile's matrix work drops from 4096 to 1536 cycles per wave.
//
// Structure (see edge_gate.hip for the first generation, kept as variants 5/6):
//   * 4 compute waves (W3 split into 3 x bf16 in 96 VGPRs per lane, loaded and split once per workgroup),
//     4 load groups x 2 waves, a 4-slot LDS ring of fp32 tiles; slots are handed over through LDS counters, no barriers;
//   * the load waves fetch the e rows and the B1h[src] / B2h[dst] rows and store e and G = B1h[src] + B2h[dst] as fp32;
//   * a compute wave is a pure GEMM engine: it reads its fp32 A fragment (8 consecutive k of one row), splits it in
//     registers (~44 VALU operations per K = 16, dealt out under the previous step's MFMAs), runs the 6 MFMAs, and at
//     the end of the tile stores its accumulators - which started from the G tile - back over it: the slot then
//     holds x = e W3^T + G;
//   * the load waves are also the store waves: once the compute waves are done with a slot, the group that filled it
//     reads x and e back as whole rows (ds_read_b128), applies the epilogue (bn + relu + residual, or the raw forms) and
//     writes e' with 16-byte stores - every global access of the kernel is a full-row, 16-byte-per-lane one - then
//     refills the slot with the tile it fetched in the meantime.
#include "common.h"

namespace gnnome {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int CB, int RB>
struct GateBF {
    static_assert(CB * RB == 4, "four compute waves");
    static constexpr int H = 32 * CB, TM = 32 * RB, LDK = H + 4, RING = 4, LWAVES = 2;
    static constexpr int NT = 64 * (4 + RING * LWAVES);              // 768 threads
    static constexpr int NP = TM * (H / 4) / (64 * LWAVES);          // float4 pieces per load lane per tile
    static constexpr int kSlotFloats = TM * LDK;
    static constexpr int kLdsFloats = RING * 2 * kSlotFloats;        // e tiles + G tiles, fp32
};

__device__ __forceinline__ unsigned lds_addr_bf(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ void flag_wait_bf(unsigned addr, unsigned want, int nap) {
    unsigned v, spins = 0;
    for (;;) {
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
        if (__builtin_amdgcn_readfirstlane(v) >= want) break;
        if (++spins > (1u << 26)) __builtin_trap();  // a lost hand-over must end the launch, not hang the queue
        if (nap == 0) {
            __builtin_amdgcn_s_sleep(1);
        } else if (nap == 1) {
            __builtin_amdgcn_s_sleep(4);
        } else if (nap == 2) {
            __builtin_amdgcn_s_sleep(16);
        } else {
            __builtin_amdgcn_s_sleep(64);
        }
    }
}
__device__ __forceinline__ void flag_bump_bf(unsigned addr, int lane) {
    if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(1u) : "memory");
}

// exact three-way bf16 split of eight floats (one lane's share of a 32x32x16 MFMA operand: 8 consecutive k)
__device__ __forceinline__ void split3(const f32x4 lo4, const f32x4 hi4, uint4& p1, uint4& p2, uint4& p3) {
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = j < 4 ? lo4[j] : hi4[j - 4];
        h[j] = __float_as_uint(x) & 0xFFFF0000u;
        const float r = x - __uint_as_float(h[j]);      // exact
        m[j] = __float_as_uint(r) & 0xFFFF0000u;
        l[j] = __float_as_uint(r - __uint_as_float(m[j]));   // exact, <= 8 significant bits: a bf16 (low half zero)
    }
    // pack pairs: element 2j in the low half, 2j+1 in the high half (v_perm_b32: bytes 3,2 of each source)
    p1 = make_uint4(__builtin_amdgcn_perm(h[1], h[0], 0x07060302u), __builtin_amdgcn_perm(h[3], h[2], 0x07060302u),
                    __builtin_amdgcn_perm(h[5], h[4], 0x07060302u), __builtin_amdgcn_perm(h[7], h[6], 0x07060302u));
    p2 = make_uint4(__builtin_amdgcn_perm(m[1], m[0], 0x07060302u), __builtin_amdgcn_perm(m[3], m[2], 0x07060302u),
                    __builtin_amdgcn_perm(m[5], m[4], 0x07060302u), __builtin_amdgcn_perm(m[7], m[6], 0x07060302u));
    p3 = make_uint4(__builtin_amdgcn_perm(l[1], l[0], 0x07060302u), __builtin_amdgcn_perm(l[3], l[2], 0x07060302u),
                    __builtin_amdgcn_perm(l[5], l[4], 0x07060302u), __builtin_amdgcn_perm(l[7], l[6], 0x07060302u));
}

__device__ __forceinline__ bf16x8 as_bf(const uint4 v) { return __builtin_bit_cast(bf16x8, v); }

// MODE 0: y = relu((acc + G) * scale + shift) + e      (the gate, gated_gcn_full.py:97,104-110)
// MODE 1: y = acc + G, G = B1h[src] + B2h[dst]          (raw gate of the training step) + shifted column sums (scale = centre)
// MODE 2: y = acc + G, G = the old rows of C (in B1h)   (C += A W^T: the backward's d e_in = d e' + dxe W3)
// MODE 3: MODE 2 with A = BatchNorm-backward(old rows of C, rows at e_in) computed by the load waves and written to bnb.a_out
// X16 (modes 1 and 3): the xe rows (mode 1: the output; mode 3: the rows at e_in) and the dxe rows (mode 3: bnb.a_out) are bf16 in
// HBM - see common.h; mode 1's statistics are those of the ROUNDED values, the ones every later kernel reads.
template <int CB, int RB, int MODE, bool ENC, bool X16 = false>
__global__ __launch_bounds__(768) void k_edge_gate_bf(GateBfArgs a) {
    using P = GateBF<CB, RB>;
    constexpr int H = P::H, TM = P::TM, NP = P::NP, LDK = P::LDK, RING = P::RING, KS = H / 16, SLOT = P::kSlotFloats;
    constexpr int kEncFloats = ENC ? 16 * H + H + 48 : (MODE == 1 ? 8 * 64 * 8 : 0);   // MODE 1: the store waves' running column sums
    __shared__ __attribute__((aligned(16))) float lds[P::kLdsFloats + kEncFloats];
    __shared__ unsigned flags[2 * RING];   // full[RING], done[RING]
    float* Aring = lds;                    // [RING][TM][LDK]  e tiles (fp32)
    float* Gring = lds + RING * SLOT;      // [RING][TM][LDK]  G tiles
    float* w2t = lds + P::kLdsFloats;      // ENC only
    float* b2s = w2t + 16 * H;
    float* w1s = b2s + H;
    float* b1s = w1s + 32;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned full0 = lds_addr_bf(&flags[0]), done0 = lds_addr_bf(&flags[RING]);
    // interleaved tile order: in round r the whole chip works on one contiguous window of gridDim.x tiles, each XCD
    // (blocks b % 8) on a contiguous sub-window (see edge_gate.hip)
    const int per_xcd = gridDim.x / kXcds;
    const int first = (int)(blockIdx.x % kXcds) * per_xcd + (int)(blockIdx.x / kXcds);
    const int stride = (int)gridDim.x;
    const int n = first < a.num_tiles ? (a.num_tiles - first + stride - 1) / stride : 0;
    if (n <= 0) return;
    auto tile_of = [&](int r) { return first + r * stride; };
    auto tile_valid = [&](int r) { return (int)min((int64_t)TM, a.E - (int64_t)tile_of(r) * TM); };
    if (ENC) {
        for (int i = tid; i < 16 * H; i += P::NT) w2t[i] = a.enc.W2[(i % H) * 16 + (i / H)];
        for (int i = tid; i < H; i += P::NT) b2s[i] = a.enc.b2[i];
        if (tid < 32) w1s[tid] = a.enc.W1[tid];
        if (tid < 16) b1s[tid] = a.enc.b1[tid];
    }
    if (tid < 2 * RING) flags[tid] = 0;
    __syncthreads();

    if (wave < 4) {
        // ------------------------------------------------------------------ compute wave
        const int rb = wave % RB, cb = wave / RB, cl = lane & 31, half = lane >> 5;
        const int col = 32 * cb + cl;
        // B operand: W[col][16q + 8 half .. + 7], split once per workgroup
        uint4 w1[KS], w2[KS], w3[KS];
#pragma unroll
        for (int q = 0; q < KS; ++q) {
            const float* wp = a.W3 + (int64_t)col * a.ldw + 16 * q + 8 * half;
            split3(*reinterpret_cast<const f32x4*>(wp), *reinterpret_cast<const f32x4*>(wp + 4), w1[q], w2[q], w3[q]);
        }
        const int lrow = 32 * rb + 4 * half;   // accumulator element r sits in tile row lrow + crow(r)
        const int lane_lds = lrow * LDK + col;
        auto crow = [](int r) { return (r & 3) + 8 * (r >> 2); };

        long long t_wait = 0, t_pro = 0, t_loop = 0, t_x = 0, t0 = 0, t1 = 0;
        const long long c_begin = a.prof ? (long long)__builtin_readcyclecounter() : 0;
        const long long r_begin = a.prof ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
        for (int i = 0; i < n; ++i) {
            const int slot = i % RING;
            if (a.prof) t0 = __builtin_readcyclecounter();
            flag_wait_bf(full0 + 4 * slot, 2u * ((unsigned)(i / RING) + 1u), 0);
            if (a.prof) { t1 = __builtin_readcyclecounter(); t_wait += t1 - t0; t0 = t1; }
            const float* As = Aring + slot * SLOT;
            const float* ap = As + (32 * rb + cl) * LDK + 8 * half;   // + 16 q
            // the accumulator starts from the G tile (so the write-back below is a plain store of x = G + e W3^T)
            float* Gp = Gring + slot * SLOT + lane_lds;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = Gp[crow(r) * LDK];
            // Software pipeline, one basic block: while the 6 MFMAs of step q run, the fragment of step q+1 is split
            // (44 VALU operations, dealt out 8 per MFMA by the sched_group_barriers) and the one of step q+2 is read.
            uint4 a1, a2, a3;
            split3(*reinterpret_cast<const f32x4*>(ap), *reinterpret_cast<const f32x4*>(ap + 4), a1, a2, a3);
            f32x4 x0 = *reinterpret_cast<const f32x4*>(ap + (KS > 1 ? 16 : 0)), x1 = *reinterpret_cast<const f32x4*>(ap + (KS > 1 ? 20 : 4));
            if (a.prof) { asm volatile("" ::"v"(a1.x), "v"(x0[0])); t1 = __builtin_readcyclecounter(); t_pro += t1 - t0; t0 = t1; }
#pragma unroll
            for (int q = 0; q < KS; ++q) {
                const int q2 = q + 2 < KS ? q + 2 : KS - 1;
                const f32x4 n0 = *reinterpret_cast<const f32x4*>(ap + 16 * q2), n1 = *reinterpret_cast<const f32x4*>(ap + 16 * q2 + 4);
                uint4 b1 = a1, b2 = a2, b3 = a3;
                if (q + 1 < KS) split3(x0, x1, b1, b2, b3);
                // smallest terms first
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a3), as_bf(w1[q]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a1), as_bf(w3[q]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a2), as_bf(w2[q]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a2), as_bf(w1[q]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a1), as_bf(w2[q]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a1), as_bf(w1[q]), acc, 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // the two LDS reads of step q+2 first
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);   // eight VALU operations of the split under it
                }
                a1 = b1;
                a2 = b2;
                a3 = b3;
                x0 = n0;
                x1 = n1;
                __builtin_amdgcn_sched_barrier(0);
            }
            if (a.prof) { asm volatile("" ::"v"(acc[0])); t1 = __builtin_readcyclecounter(); t_loop += t1 - t0; t0 = t1; }
            // x = acc + G, in place in the G tile (rows past the end of the edge list are never stored)
#pragma unroll
            for (int r = 0; r < 16; ++r) Gp[crow(r) * LDK] = acc[r];
            flag_bump_bf(done0 + 4 * slot, lane);
            if (a.prof) { t1 = __builtin_readcyclecounter(); t_x += t1 - t0; }
        }
        if (a.prof && wave == 0 && lane == 0) {
            long long* o = a.prof + (int64_t)blockIdx.x * 8;
            o[0] = t_wait; o[1] = t_pro; o[2] = t_loop; o[3] = t_x; o[4] = n;
            o[5] = (long long)__builtin_readcyclecounter() - c_begin;           // shader cycles, loop start to end
            o[6] = (long long)__builtin_amdgcn_s_memrealtime() - r_begin;       // the same span in 100 MHz ticks
        }
    } else {
        // ------------------------------------------------------------------ load wave
        const int group = (wave - 4) / P::LWAVES;
        const int gl = ((wave - 4) % P::LWAVES) * 64 + lane;  // lane index inside the group, 0..127
        constexpr int RSTEP = 64 * P::LWAVES / (H / 4);
        const int r0 = gl / (H / 4), c4 = gl % (H / 4);
        f32x4 av[NP], g1[NP], g2[NP];
        float raw0[NP], raw1[NP];
        if (a.abl & 5) {
#pragma unroll
            for (int p = 0; p < NP; ++p) av[p] = g1[p] = g2[p] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const unsigned off_row = (unsigned)(r0 * H + 4 * c4);   // float offset of piece 0 inside a tile of rows
        // The fetch of a tile comes in two parts.  EARLY (issued while the compute waves still work on this group's
        // previous tile): the sorted indices and the e rows - nothing depends on them, 48 registers.  LATE (after the
        // previous tile's epilogue has given its registers back): the B1h[src] / B2h[dst] gathers, whose addresses the
        // early part has meanwhile delivered - so the index -> gather dependency costs no second round trip.
        int si[NP], di[NP], ei[NP];
        auto issue_early = [&](int r) {
            const int64_t row0 = (int64_t)tile_of(r) * TM;
            const int valid = tile_valid(r);
#pragma unroll
            for (int p = 0; p < NP; ++p) {   // rows past the end of the list read the last valid row (never stored)
                const int64_t row = row0 + min(r0 + p * RSTEP, valid - 1);
                if (MODE < 2) {
                    si[p] = a.srt_src[row];
                    di[p] = a.srt_dst[row];
                }
                if (ENC) ei[p] = a.enc.srt_eid[row];
            }
            if (!ENC && !(a.abl & 4)) {
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const int64_t row = row0 + min(r0 + p * RSTEP, valid - 1);
                    if (X16 && MODE == 3) {
                        const uint2 pk = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(a.e_in) + row * H + 4 * c4);
                        av[p] = unpack_bf16x4(pk);
                    } else {
                        av[p] = *reinterpret_cast<const f32x4*>(a.e_in + row * H + 4 * c4);
                    }
                }
            }
        };
        auto issue_late = [&](int r) {
            const int64_t row0 = (int64_t)tile_of(r) * TM;
            const int valid = tile_valid(r);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                if (ENC) {
                    raw0[p] = a.enc.e_raw[2 * (int64_t)ei[p]];
                    raw1[p] = a.enc.e_raw[2 * (int64_t)ei[p] + 1];
                }
                if (a.abl & 1) {
                } else if (MODE >= 2) {
                    const int64_t row = row0 + min(r0 + p * RSTEP, valid - 1);
                    g1[p] = *reinterpret_cast<const f32x4*>(a.B1h + row * a.ldn + 4 * c4);   // the old rows of C
                } else {
                    g1[p] = *reinterpret_cast<const f32x4*>(a.B1h + (int64_t)si[p] * a.ldn + 4 * c4);
                    g2[p] = *reinterpret_cast<const f32x4*>(a.B2h + (int64_t)di[p] * a.ldn + 4 * c4);
                }
            }
        };
        // ENC: e0[p,:] = W2e relu(W1e e_raw + b1e) + b2e for this lane's pieces (models/full_graph.py:27)
        auto encode_pending = [&]() {
            // in the reference's order (torch nn.Linear on the CPU = k-ascending fma chain from zero, then + bias)
#pragma unroll
            for (int p = 0; p < NP; ++p) av[p] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int j = 0; j < 16; ++j) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(w2t + j * H + 4 * c4);
                const float wa = w1s[2 * j], wb = w1s[2 * j + 1], bj = b1s[j];
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const float t = fmaxf(__builtin_fmaf(raw1[p], wb, raw0[p] * wa) + bj, 0.f);
#pragma unroll
                    for (int i = 0; i < 4; ++i) av[p][i] = __builtin_fmaf(t, w[i], av[p][i]);
                }
            }
            {
                const f32x4 b2v = *reinterpret_cast<const f32x4*>(b2s + 4 * c4);
#pragma unroll
                for (int p = 0; p < NP; ++p) av[p] += b2v;
            }
        };
        if (group < n) {
            issue_early(group);
            issue_late(group);
            if (ENC) encode_pending();
        }
        // this lane's four columns of the epilogue
        // MODE 1: this lane's running sums of its four columns live in LDS between tiles (the fetch phase has no
        // registers to spare for them)
        float* my_sums = w2t + ((wave - 4) * 64 + lane) * 8;
        if (MODE == 1) {
            *reinterpret_cast<f32x4*>(my_sums) = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(my_sums + 4) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        float* As = Aring + group * SLOT;
        float* Gs = Gring + group * SLOT;
        for (int r = group; r < n; r += RING) {
            // the slot is free: this group stored its previous tile itself (below)
            if (MODE == 3) {
                // A = BatchNorm backward of (dy = g1, x = av) for this lane's four columns; written out as dxe on the way
                const f32x4 ka = *reinterpret_cast<const f32x4*>(a.bnb.a + 4 * c4), k1 = *reinterpret_cast<const f32x4*>(a.bnb.c1 + 4 * c4);
                const f32x4 k2 = *reinterpret_cast<const f32x4*>(a.bnb.c2 + 4 * c4), km = *reinterpret_cast<const f32x4*>(a.bnb.mean + 4 * c4);
                const f32x4 kr = *reinterpret_cast<const f32x4*>(a.bnb.rstd + 4 * c4), ks = *reinterpret_cast<const f32x4*>(a.bnb.scale + 4 * c4);
                const f32x4 kh = *reinterpret_cast<const f32x4*>(a.bnb.shift + 4 * c4);
                const int valid3 = tile_valid(r);
                const int64_t once3 = a.bnb.n_once - (int64_t)tile_of(r) * TM;   // rows of this tile that get the mean terms
                float* aout = a.bnb.a_out + (int64_t)tile_of(r) * TM * H;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    f32x4 t;
                    const float on = r0 + p * RSTEP < once3 ? 1.f : 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float gm = (av[p][j] * ks[j] + kh[j] > 0.f) ? g1[p][j] : 0.f;
                        t[j] = ka[j] * (gm - on * (k1[j] + (av[p][j] - km[j]) * kr[j] * k2[j]));
                    }
                    av[p] = t;
                    if (r0 + p * RSTEP < valid3) {
                        if (X16)
                            *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.bnb.a_out) + (int64_t)tile_of(r) * TM * H +
                                                      (off_row + (unsigned)(p * RSTEP * H))) = pack_bf16x4(t);
                        else
                            *reinterpret_cast<f32x4*>(aout + (off_row + (unsigned)(p * RSTEP * H))) = t;
                    }
                }
            }
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                *reinterpret_cast<f32x4*>(As + (r0 + p * RSTEP) * LDK + 4 * c4) = av[p];
                *reinterpret_cast<f32x4*>(Gs + (r0 + p * RSTEP) * LDK + 4 * c4) = MODE >= 2 ? g1[p] : g1[p] + g2[p];
            }
            flag_bump_bf(full0 + 4 * group, lane);
            if (r + RING < n) issue_early(r + RING);
            // epilogue + store of tile r once the four compute waves have added their products into the G tile
            flag_wait_bf(done0 + 4 * group, 4u * ((unsigned)(r / RING) + 1u), a.xp & 3);
            const int valid = tile_valid(r);
            // (re-read per tile, L1-resident: eight registers fewer across the fetch phase)
            f32x4 sc4 = {0.f, 0.f, 0.f, 0.f}, sh4 = {0.f, 0.f, 0.f, 0.f};
            if (MODE < 2) sc4 = *reinterpret_cast<const f32x4*>(a.scale + 4 * c4);   // MODE 1: the columns' centres
            if (MODE == 0) sh4 = *reinterpret_cast<const f32x4*>(a.shift + 4 * c4);
            float* out = a.e_out + (int64_t)tile_of(r) * TM * H;   // uniform; the lane's part is off_row + p * const
            f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
            if (MODE == 1) {
                s1 = *reinterpret_cast<const f32x4*>(my_sums);
                s2 = *reinterpret_cast<const f32x4*>(my_sums + 4);
            }
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int row = r0 + p * RSTEP;
                const f32x4 x = *reinterpret_cast<const f32x4*>(Gs + row * LDK + 4 * c4);
                f32x4 y;
                if (MODE == 0) {
                    const f32x4 e = *reinterpret_cast<const f32x4*>(As + row * LDK + 4 * c4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[j] = fmaxf(x[j] * sc4[j] + sh4[j], 0.f) + e[j];
                } else {
                    y = x;
                }
                uint2 pk = {0u, 0u};
                if (X16 && MODE == 1) {
                    pk = pack_bf16x4(y);
                    y = unpack_bf16x4(pk);
                }
                if (row < valid) {
                    if (MODE == 1) {
                        const f32x4 d = y - sc4;
                        s1 += d;
                        s2 += d * d;
                    }
                    if (X16 && MODE == 1)
                        *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.e_out) + (int64_t)tile_of(r) * TM * H +
                                                  (off_row + (unsigned)(p * RSTEP * H))) = pk;
                    else if (!(a.abl & 2))
                        *reinterpret_cast<f32x4*>(out + (off_row + (unsigned)(p * RSTEP * H))) = y;
                }
                if (a.abl & 2) asm volatile("" ::"v"(y[0] + y[1] + y[2] + y[3]));
            }
            if (MODE == 1) {
                *reinterpret_cast<f32x4*>(my_sums) = s1;
                *reinterpret_cast<f32x4*>(my_sums + 4) = s2;
            }
            if (r + RING < n) {
                issue_late(r + RING);
                if (ENC) encode_pending();
            }
        }
        if (MODE == 1) {
            f32x4 s1 = *reinterpret_cast<const f32x4*>(my_sums), s2 = *reinterpret_cast<const f32x4*>(my_sums + 4);
            // lanes that share c4 hold different rows of the same four columns: fold them, then every load wave
            // leaves one row of partial sums (stats[(block * 2 RING + wave - 4)][2H])
#pragma unroll
            for (int o = H / 4; o < 64; o <<= 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s1[j] += __shfl_xor(s1[j], o);
                    s2[j] += __shfl_xor(s2[j], o);
                }
            }
            if (lane < H / 4) {
                float* dst = a.stats + ((int64_t)blockIdx.x * (RING * P::LWAVES) + (wave - 4)) * 2 * H;
                *reinterpret_cast<f32x4*>(dst + 4 * c4) = s1;
                *reinterpret_cast<f32x4*>(dst + H + 4 * c4) = s2;
            }
        }
    }
}

static long long* g_gate_prof = nullptr;
long long* gate_profile_buffer() { return g_gate_prof; }

// ---------------------------------------------------------------------------------------------------
// Third generation, H = 128, mode 0 (the inference gate): the A operand is split ONCE per tile, by the load waves.
// In k_edge_gate_bf the four compute waves each own 32 of the 128 output columns of the same 32 rows, so each of them reads
// the whole fp32 A tile and splits it - 352 VALU operations per tile per wave, four times over, interleaved with 48 MFMAs
// (2160 cycles in the loop against 1536 of MFMA).  Here the load waves, whose vector ALUs are idle most of the tile, store the
// tile as three bf16 PLANES ([row][k], 272-byte rows) and the compute loop is 3 ds_read_b128 + 6 MFMAs per K = 16 step, no VALU.
//   * LDS: a slot is the 26 KB of planes; once all four compute waves have read them (a second counter, `rd`) the slot is
//     reused for x = e W3^T (fp32 [32][132], 17 KB) - four slots in 104 KB, so the ring keeps its four load groups (the
//     round-1 attempt with planes AND a G tile per slot only had room for three and starved the compute waves);
//   * G = B1h[src] + B2h[dst] and the e rows (for the residual) never enter LDS: the lanes that fetched them keep them in
//     registers until their own epilogue (32 + 32 VGPRs), and the accumulators start from zero.
// ---------------------------------------------------------------------------------------------------
// fp16x3 arithmetic (round 4; the derivation and the error model are in edge_tile_f16.hip's header and tests/test_f16x3_model.py): an fp32
// operand as TWO fp16 planes, x1 = RN16(x) and x2 = RN16((x - x1) * 2048), three products instead of bf16x6's six, the two small ones in
// a second accumulator that is folded in with 2^-11 once per tile.
typedef _Float16 h2_pl __attribute__((ext_vector_type(2)));
typedef _Float16 h8_pl __attribute__((ext_vector_type(8)));
typedef float f32x2_pl __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split4_h(const f32x4 x, uint2& p1, uint2& p2) {
    h2_pl a[2], b[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const f32x2_pl v = {x[2 * j], x[2 * j + 1]};
        a[j] = __builtin_convertvector(v, h2_pl);
        const f32x2_pl big = v * 2048.f;
        const f32x2_pl r = {__builtin_fmaf((float)a[j][0], -2048.f, big[0]), __builtin_fmaf((float)a[j][1], -2048.f, big[1])};   // exact
        b[j] = __builtin_convertvector(r, h2_pl);
    }
    p1 = make_uint2(__builtin_bit_cast(unsigned, a[0]), __builtin_bit_cast(unsigned, a[1]));
    p2 = make_uint2(__builtin_bit_cast(unsigned, b[0]), __builtin_bit_cast(unsigned, b[1]));
}
__device__ __forceinline__ void split8_h(const f32x4 lo, const f32x4 hi, uint4& p1, uint4& p2) {
    uint2 a1, a2, b1, b2;
    split4_h(lo, a1, a2);
    split4_h(hi, b1, b2);
    p1 = make_uint4(a1.x, a1.y, b1.x, b1.y);
    p2 = make_uint4(a2.x, a2.y, b2.x, b2.y);
}
__device__ __forceinline__ h8_pl as_h8(const uint4 v) { return __builtin_bit_cast(h8_pl, v); }

__device__ __forceinline__ void split4_planes(const f32x4 x, uint2& p1, uint2& p2, uint2& p3) {
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h[j] = __float_as_uint(x[j]) & 0xFFFF0000u;
        const float r = x[j] - __uint_as_float(h[j]);        // exact
        m[j] = __float_as_uint(r) & 0xFFFF0000u;
        l[j] = __float_as_uint(r - __uint_as_float(m[j]));   // exact, a bf16
    }
    p1 = make_uint2(__builtin_amdgcn_perm(h[1], h[0], 0x07060302u), __builtin_amdgcn_perm(h[3], h[2], 0x07060302u));
    p2 = make_uint2(__builtin_amdgcn_perm(m[1], m[0], 0x07060302u), __builtin_amdgcn_perm(m[3], m[2], 0x07060302u));
    p3 = make_uint2(__builtin_amdgcn_perm(l[1], l[0], 0x07060302u), __builtin_amdgcn_perm(l[3], l[2], 0x07060302u));
}

// MODE 0: the inference gate.  MODE 1: raw gate + shifted column sums (training forward; a.scale = the centres).  MODE 3: the
// BatchNorm backward + data gradient of k_edge_gate_bf's mode 3 (A computed by the load waves from the old C rows and the rows at
// e_in, written to bnb.a_out; C += A W^T).  X16: xe / dxe stored as bf16 (see common.h).  Modes 1 and 3 hold no e rows for a residual.
// MODE 4 (round 3): C[M, 128 * a.num_cblocks] = A[M,128] W^T + bias - the node projection [N,H] -> [N,5H] on this kernel: a workgroup keeps
// ONE 128-column block of W in its compute waves' registers for the whole launch, the a.num_cblocks workgroups of an XCD that share a
// tile stream read an A row from HBM once (the others find it in that XCD's L2); no gathers, bias instead of the epilogue.  A has row
// stride a.ldn, C row stride a.ld_out, a.scale = bias (NULL: none).
// F16 (round 4, the default for the forward modes 0, 1 and 4; gnnome_set_tuning(10, 1) = bf16x6): two fp16 planes and three MFMAs per
// k step instead of three bf16 planes and six - the slot stays 26 KB (the x tile needs 17 KB of it either way).
// EXTRA (round 4, the training forward in two passes instead of three - train.py): MODE 1 + EXTRA = the statistics alone, nothing stored;
// MODE 0 + EXTRA = the gate with the pre-normalisation rows xe = x + G ALSO written out (to a.bnb.a_out; X16: as bf16, and e' is then
// computed from the rounded values, like gnnome_bn_relu_res_x16 reading them back).
template <bool ENC, int MODE = 0, bool X16 = false, bool F16 = false, int EXTRA = 0>
__global__ __launch_bounds__(768) void k_edge_gate_pl(GateBfArgs a) {
    static_assert(EXTRA == 0 || ((MODE == 0 || MODE == 1) && !ENC), "EXTRA belongs to modes 0 and 1");
    static_assert(MODE >= 0 && MODE <= 4, "modes of the plane form");   // 2: C += A W^T (the residual GEMM; mode 3 without the BatchNorm step)
    static_assert(!ENC || MODE == 0, "the folded encoder belongs to the inference gate");
    static_assert(!F16 || MODE == 0 || MODE == 1 || MODE == 4, "fp16x3 is built for the forward modes (gradients need a scale)");
    constexpr int H = 128, TM = 32, RING = 4, KS = H / 16, LDK = H + 4, PLD = 2 * H + 16, PLANE = TM * PLD, SLOTB = 3 * PLANE;
    constexpr int NP = 8, RSTEP = 4, NT = 768;
    static_assert(TM * LDK * 4 <= SLOTB, "the x tile reuses the planes' slot");
    constexpr int kEncFloats = ENC ? 16 * H + H + 48 : 4;
    __shared__ __attribute__((aligned(16))) unsigned char ring[RING * SLOTB];
    __shared__ __attribute__((aligned(16))) float encw[kEncFloats];
    __shared__ unsigned flags[4 * RING];   // full[RING], rd[RING], done[RING], drained[RING]
    __shared__ __attribute__((aligned(16))) float norm_lds[7 * H];   // per-channel constants of the mode, read per tile from here (no global load in the loop)
    float* w2t = encw;                     // ENC only
    float* b2s = w2t + 16 * H;
    float* w1s = b2s + H;
    float* b1s = w1s + 32;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned full0 = lds_addr_bf(&flags[0]), rd0 = lds_addr_bf(&flags[RING]), done0 = lds_addr_bf(&flags[2 * RING]),
                   drained0 = lds_addr_bf(&flags[3 * RING]);
    const int per_xcd = gridDim.x / kXcds;
    int first = (int)(blockIdx.x % kXcds) * per_xcd + (int)(blockIdx.x / kXcds);
    int stride = (int)gridDim.x;
    int cbk = 0;   // MODE 4: this workgroup's 128-column block of the output
    if (MODE == 4) {
        const int idx = blockIdx.x / kXcds, streams = per_xcd / a.num_cblocks;
        if (idx >= streams * a.num_cblocks) return;   // (32 workgroups per XCD, 5 column blocks: 6 tile streams, 2 idle workgroups)
        cbk = idx % a.num_cblocks;
        first = (int)(blockIdx.x % kXcds) * streams + idx / a.num_cblocks;
        stride = kXcds * streams;
    }
    const int n = first < a.num_tiles ? (a.num_tiles - first + stride - 1) / stride : 0;
    if (n <= 0) return;
    const int lda = MODE == 4 ? a.ldn : H, ldo = MODE == 4 ? a.ld_out : H;
    auto tile_of = [&](int r) { return first + r * stride; };
    auto tile_valid = [&](int r) { return (int)min((int64_t)TM, a.E - (int64_t)tile_of(r) * TM); };
    if (ENC) {
        for (int i = tid; i < 16 * H; i += NT) w2t[i] = a.enc.W2[(i % H) * 16 + (i / H)];
        for (int i = tid; i < H; i += NT) b2s[i] = a.enc.b2[i];
        if (tid < 32) w1s[tid] = a.enc.W1[tid];
        if (tid < 16) b1s[tid] = a.enc.b1[tid];
    }
    if (tid < 4 * RING) flags[tid] = 0;
    for (int i = tid; i < 7 * H; i += NT) {
        const int q = i / H, c = i % H;
        if (MODE == 4 && q < 1) norm_lds[i] = a.scale ? a.scale[H * cbk + c] : 0.f;   // the bias of this column block
        if (MODE == 0 && q < 2) norm_lds[i] = q == 0 ? a.scale[c] : a.shift[c];
        if (MODE == 1 && q < 1) norm_lds[i] = a.scale[c];   // the columns' centres
        if (MODE == 3) {
            const float* src = q == 0 ? a.bnb.a : q == 1 ? a.bnb.c1 : q == 2 ? a.bnb.c2 : q == 3 ? a.bnb.mean : q == 4 ? a.bnb.rstd : q == 5 ? a.bnb.scale : a.bnb.shift;
            norm_lds[i] = src[c];
        }
    }
    __syncthreads();

    if (wave < 4) {
        // ------------------------------------------------------------------ compute wave: 32 rows x 32 columns
        const int cl = lane & 31, half = lane >> 5, col = 32 * wave + cl;
        uint4 w1[KS], w2[KS], w3[F16 ? 1 : KS];
#pragma unroll
        for (int q = 0; q < KS; ++q) {
            const float* wp = a.W3 + (int64_t)(H * cbk + col) * a.ldw + 16 * q + 8 * half;
            if (F16)
                split8_h(*reinterpret_cast<const f32x4*>(wp), *reinterpret_cast<const f32x4*>(wp + 4), w1[q], w2[q]);
            else
                split3(*reinterpret_cast<const f32x4*>(wp), *reinterpret_cast<const f32x4*>(wp + 4), w1[q], w2[q], w3[q]);
        }
        auto crow = [](int r) { return (r & 3) + 8 * (r >> 2); };
        const int lane_x = 4 * half * LDK + col;   // accumulator element r sits in tile row 4 half + crow(r)
        long long t_wait = 0, t_rd = 0, t_loop = 0, t_x = 0, t0 = 0, t1 = 0;
        const long long c_begin = a.prof ? (long long)__builtin_readcyclecounter() : 0;
        const long long r_begin = a.prof ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
        for (int i = 0; i < n; ++i) {
            const int slot = i % RING;
            const unsigned use = (unsigned)(i / RING) + 1u;
            if (a.prof) t0 = __builtin_readcyclecounter();
            flag_wait_bf(full0 + 4 * slot, 2u * use, 0);
            if (a.prof) { t1 = __builtin_readcyclecounter(); t_wait += t1 - t0; t0 = t1; }
            const unsigned char* ap = ring + slot * SLOTB + cl * PLD + 16 * half;   // + 32 q, + PLANE * plane
            f32x16 acc, accC;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f, accC[r] = 0.f;
            // Plane fragments PD k steps ahead of their MFMAs.  F16 (round 4): three steps - hipcc had sunk the one-step-ahead reads to two MFMAs
            // (~70 cycles) before their use, an LDS read under this kernel's traffic takes longer than that, and the matrix pipe idled at every wait
            // (1205 cycles per tile for 24 MFMAs of 32); the scheduling barriers keep the reads where they are written.
            constexpr int PD = F16 ? 3 : 1;
            uint4 f1[KS], f2[KS], f3[F16 ? 1 : KS];
#pragma unroll
            for (int q = 0; q < PD; ++q) {
                f1[q] = *reinterpret_cast<const uint4*>(ap + 32 * q);
                f2[q] = *reinterpret_cast<const uint4*>(ap + 32 * q + PLANE);
                if (!F16) f3[q] = *reinterpret_cast<const uint4*>(ap + 32 * q + 2 * PLANE);
            }
#pragma unroll
            for (int q = 0; q < KS; ++q) {
                if (q + PD < KS) {
                    f1[q + PD] = *reinterpret_cast<const uint4*>(ap + 32 * (q + PD));
                    f2[q + PD] = *reinterpret_cast<const uint4*>(ap + 32 * (q + PD) + PLANE);
                    if (!F16) f3[q + PD] = *reinterpret_cast<const uint4*>(ap + 32 * (q + PD) + 2 * PLANE);
                }
                if (F16) __builtin_amdgcn_sched_barrier(0);
                const uint4 c1 = f1[q], c2 = f2[q], c3 = F16 ? c1 : f3[F16 ? 0 : q];
                if (F16) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(c1), as_h8(w1[q]), acc, 0, 0, 0);
                    accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(c1), as_h8(w2[q]), accC, 0, 0, 0);
                    accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(c2), as_h8(w1[q]), accC, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    // smallest terms first
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(c3), as_bf(w1[q]), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(c1), as_bf(w3[F16 ? 0 : q]), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(c2), as_bf(w2[q]), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(c2), as_bf(w1[q]), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(c1), as_bf(w2[q]), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(c1), as_bf(w1[q]), acc, 0, 0, 0);
                }
            }
            if (F16) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] += accC[r] * (1.0f / 2048.f);
            }
            if (a.prof) { asm volatile("" ::"v"(acc[0])); t1 = __builtin_readcyclecounter(); t_loop += t1 - t0; t0 = t1; }
            // every compute wave has read the planes -> the slot becomes the x tile
            flag_bump_bf(rd0 + 4 * slot, lane);
            flag_wait_bf(rd0 + 4 * slot, 4u * use, 0);
            if (a.prof) { t1 = __builtin_readcyclecounter(); t_rd += t1 - t0; t0 = t1; }
            float* X = reinterpret_cast<float*>(ring + slot * SLOTB) + lane_x;
#pragma unroll
            for (int r = 0; r < 16; ++r) X[crow(r) * LDK] = acc[r];
            flag_bump_bf(done0 + 4 * slot, lane);
            if (a.prof) { t1 = __builtin_readcyclecounter(); t_x += t1 - t0; }
        }
        if (a.prof && wave == 0 && lane == 0) {   // same record as k_edge_gate_bf; slot 1 = waiting for the other compute waves
            long long* o = a.prof + (int64_t)blockIdx.x * 8;
            o[0] = t_wait; o[1] = t_rd; o[2] = t_loop; o[3] = t_x; o[4] = n;
            o[5] = (long long)__builtin_readcyclecounter() - c_begin;
            o[6] = (long long)__builtin_amdgcn_s_memrealtime() - r_begin;
        }
    } else {
        // ------------------------------------------------------------------ load / store wave
        const int group = (wave - 4) / 2;
        const int gl = ((wave - 4) % 2) * 64 + lane;   // lane index inside the group, 0..127
        const int r0 = gl / (H / 4), c4 = gl % (H / 4);
        f32x4 av[NP], g1[NP], g2[NP], ek[NP], gk[NP];
        float raw0[NP], raw1[NP];
        int si[NP], di[NP], ei[NP];
        const unsigned off_row = (unsigned)(r0 * H + 4 * c4);
        auto issue_early = [&](int r) {
            const int64_t row0 = (int64_t)tile_of(r) * TM;
            const int valid = tile_valid(r);
#pragma unroll
            for (int p = 0; p < NP; ++p) {   // rows past the end of the list read the last valid row (never stored)
                const int64_t row = row0 + min(r0 + p * RSTEP, valid - 1);
                if (MODE < 2) {
                    si[p] = a.srt_src[row];
                    di[p] = a.srt_dst[row];
                }
                if (ENC) ei[p] = a.enc.srt_eid[row];
            }
        };
        // (the e rows are NOT fetched with the indices: 32 registers fewer are held across the wait for the compute waves - this
        //  role also keeps e and G of the tile in flight; see the epilogue)
        auto fetch_e = [&](int r) {
            const int64_t row0 = (int64_t)tile_of(r) * TM;
            const int valid = tile_valid(r);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int64_t row = row0 + min(r0 + p * RSTEP, valid - 1);
                av[p] = load4_as<(X16 && MODE == 3)>(a.e_in, row * lda + 4 * c4);   // mode 3: the xe rows; mode 4: the A rows
                if (MODE == 2 || MODE == 3) g1[p] = *reinterpret_cast<const f32x4*>(a.B1h + row * a.ldn + 4 * c4);   // the old rows of C
            }
        };
        auto issue_late = [&](int) {
            if (MODE >= 2) return;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                if (ENC) {
                    raw0[p] = a.enc.e_raw[2 * (int64_t)ei[p]];
                    raw1[p] = a.enc.e_raw[2 * (int64_t)ei[p] + 1];
                }
                g1[p] = *reinterpret_cast<const f32x4*>(a.B1h + (int64_t)si[p] * a.ldn + 4 * c4);
                g2[p] = *reinterpret_cast<const f32x4*>(a.B2h + (int64_t)di[p] * a.ldn + 4 * c4);
            }
        };
        auto encode_pending = [&]() {   // in the reference's order (see k_edge_gate_bf)
#pragma unroll
            for (int p = 0; p < NP; ++p) av[p] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int j = 0; j < 16; ++j) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(w2t + j * H + 4 * c4);
                const float wa = w1s[2 * j], wb = w1s[2 * j + 1], bj = b1s[j];
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const float t = fmaxf(__builtin_fmaf(raw1[p], wb, raw0[p] * wa) + bj, 0.f);
#pragma unroll
                    for (int i = 0; i < 4; ++i) av[p][i] = __builtin_fmaf(t, w[i], av[p][i]);
                }
            }
            const f32x4 b2v = *reinterpret_cast<const f32x4*>(b2s + 4 * c4);
#pragma unroll
            for (int p = 0; p < NP; ++p) av[p] += b2v;
        };
        if (group < n) {
            issue_early(group);
            if (!ENC) fetch_e(group);
            issue_late(group);
            if (ENC) encode_pending();
        }
        unsigned char* S = ring + group * SLOTB;
        const float* Xs = reinterpret_cast<const float*>(S);
        long long t_top = 0, t_split = 0, t_done = 0, t_epi = 0, t0 = 0, t1 = 0;
        f32x4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = st1;   // MODE 1: this lane's running shifted sums of its four columns
        float amax3 = 0.f;                             // MODE 3: max |dxe| over this lane's elements
        for (int r = group; r < n; r += RING) {
            const unsigned use = (unsigned)(r / RING) + 1u;
            if (a.prof) { t0 = __builtin_readcyclecounter(); asm volatile("" ::"v"(av[0][0]), "v"(g1[NP - 1][0])); t1 = __builtin_readcyclecounter(); t_top += t1 - t0; t0 = t1; }
            // The slot is free once BOTH waves of the group have read the previous x tile out of it: the planes of a row do not
            // lie where its x row lay, so one wave's plane stores would land on x rows the other wave has yet to read.
            flag_wait_bf(drained0 + 4 * group, 2u * (use - 1u), 0);
            if (MODE == 3) {
                // A = BatchNorm backward of (dy = the old C rows, x = the xe rows) for this lane's four columns, written out as dxe
                const f32x4 ka = *reinterpret_cast<const f32x4*>(norm_lds + 4 * c4), k1 = *reinterpret_cast<const f32x4*>(norm_lds + H + 4 * c4);
                const f32x4 k2 = *reinterpret_cast<const f32x4*>(norm_lds + 2 * H + 4 * c4), km = *reinterpret_cast<const f32x4*>(norm_lds + 3 * H + 4 * c4);
                const f32x4 kr = *reinterpret_cast<const f32x4*>(norm_lds + 4 * H + 4 * c4), ks = *reinterpret_cast<const f32x4*>(norm_lds + 5 * H + 4 * c4);
                const f32x4 kh = *reinterpret_cast<const f32x4*>(norm_lds + 6 * H + 4 * c4);
                const int valid3 = tile_valid(r);
                const int64_t once3 = a.bnb.n_once - (int64_t)tile_of(r) * TM;   // rows of this tile that get the mean terms
                const int64_t base3 = (int64_t)tile_of(r) * TM * H;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    f32x4 t;
                    const float on = r0 + p * RSTEP < once3 ? 1.f : 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float gm = (av[p][j] * ks[j] + kh[j] > 0.f) ? g1[p][j] : 0.f;
                        t[j] = ka[j] * (gm - on * (k1[j] + (av[p][j] - km[j]) * kr[j] * k2[j]));
                    }
                    av[p] = t;
                    if (r0 + p * RSTEP < valid3) {
                        store4_as<X16>(a.bnb.a_out, base3 + (off_row + (unsigned)(p * RSTEP * H)), t);
                        amax3 = fmaxf(fmaxf(amax3, fmaxf(fabsf(t[0]), fabsf(t[1]))), fmaxf(fabsf(t[2]), fabsf(t[3])));
                    }
                }
            }
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                uint2 p1, p2, p3;
                unsigned char* d = S + (r0 + p * RSTEP) * PLD + 8 * c4;
                if (F16) {
                    split4_h(av[p], p1, p2);
                } else {
                    split4_planes(av[p], p1, p2, p3);
                    *reinterpret_cast<uint2*>(d + 2 * PLANE) = p3;
                }
                *reinterpret_cast<uint2*>(d) = p1;
                *reinterpret_cast<uint2*>(d + PLANE) = p2;
                if (MODE == 0) ek[p] = av[p];
                if (MODE != 4) gk[p] = MODE >= 2 ? g1[p] : g1[p] + g2[p];
                // G is summed HERE, not where it is used: sunk into the epilogue, the sum would drag the wait for the gathers
                // behind that epilogue's own stores (one in-order counter for loads and stores) and stall on their completion
                if (MODE != 4) asm volatile("" : "+v"(gk[p]));
            }
            flag_bump_bf(full0 + 4 * group, lane);
            if (a.prof) { t1 = __builtin_readcyclecounter(); t_split += t1 - t0; t0 = t1; }
            if (r + RING < n) issue_early(r + RING);
            // MODE 4 (round 4): the A rows of this group's NEXT tile are requested here, ahead of the wait for the compute waves and of the epilogue's
            // stores - the loads then precede those stores in the in-order memory counter and have ~5000 cycles to arrive (phase counters before:
            // 2040 cycles per own tile waiting for rows requested after the epilogue); the other modes hold too many registers across the epilogue
            constexpr bool kFetchAhead = MODE == 4;
            if (kFetchAhead && r + RING < n && !(a.xp & 32)) fetch_e(r + RING);
            flag_wait_bf(done0 + 4 * group, 4u * use, a.xp & 3);
            if (a.prof) { t1 = __builtin_readcyclecounter(); t_done += t1 - t0; t0 = t1; }
            const int valid = tile_valid(r);
            // MODE 0: scale, shift; MODE 1: sc4 = the centres; MODE 3: unused
            const f32x4 sc4 = *reinterpret_cast<const f32x4*>(norm_lds + 4 * c4), sh4 = *reinterpret_cast<const f32x4*>(norm_lds + H + 4 * c4);
            float* out = a.e_out + (int64_t)tile_of(r) * TM * H;
            const int64_t obase = (int64_t)tile_of(r) * TM * H;
            float* out4 = a.e_out + (int64_t)tile_of(r) * TM * ldo + H * cbk + 4 * c4;   // MODE 4: row stride ldo, this column block
            // four x pieces are read together, BEFORE the row-validity branches: one LDS round trip (~400 cycles with the compute
            // waves reading planes flat out) per four pieces instead of one per piece
#pragma unroll
            for (int pb = 0; pb < NP; pb += 4) {
                f32x4 x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) x[u] = *reinterpret_cast<const f32x4*>(Xs + (r0 + (pb + u) * RSTEP) * LDK + 4 * c4);
                asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int p = pb + u, row = r0 + p * RSTEP;
                    f32x4 y;
                    if (MODE == 0) {
                        f32x4 xg = x[u] + gk[p];
                        if (EXTRA) {
                            if (X16) xg = unpack_bf16x4(pack_bf16x4(xg));
                            if (row < valid) store4_as<X16>(a.bnb.a_out, obase + (off_row + (unsigned)(p * RSTEP * H)), xg);
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float t = xg[j] * sc4[j] + sh4[j];
                            // F16: an operand beyond fp16's range reaches here as inf / NaN and must not leave the relu as 0 (t - t is 0 iff t is finite)
                            y[j] = (!F16 || t - t == 0.f) ? fmaxf(t, 0.f) + ek[p][j] : __builtin_nanf("");
                        }
                        if (row < valid) *reinterpret_cast<f32x4*>(out + (off_row + (unsigned)(p * RSTEP * H))) = y;
                    } else if (MODE == 4) {
                        y = x[u] + sc4;   // + bias
                        if (row < valid) *reinterpret_cast<f32x4*>(out4 + (int64_t)row * ldo) = y;
                    } else {
                        y = x[u] + gk[p];
                        if (MODE == 1 && X16) y = unpack_bf16x4(pack_bf16x4(y));   // the statistics are those of the stored values
                        if (row < valid) {
                            if (MODE == 1) {
                                const f32x4 dlt = y - sc4;
                                st1 += dlt;
                                st2 += dlt * dlt;
                                if (!EXTRA) store4_as<X16>(a.e_out, obase + (off_row + (unsigned)(p * RSTEP * H)), y);
                            } else {
                                *reinterpret_cast<f32x4*>(out + (off_row + (unsigned)(p * RSTEP * H))) = y;
                            }
                        }
                    }
                }
            }
            flag_bump_bf(drained0 + 4 * group, lane);
            if (a.prof) { t1 = __builtin_readcyclecounter(); t_epi += t1 - t0; }
            if (r + RING < n) {   // (woven into the epilogue piece by piece, the e-row fetch made the epilogue 1200 cycles longer and
                                  //  arrived no earlier: loads and stores share one in-order counter)
                if (!ENC && !(kFetchAhead && !(a.xp & 32))) fetch_e(r + RING);
                issue_late(r + RING);
                if (ENC) encode_pending();
            }
        }
        if (MODE == 3 && a.bnb.amax_bits != nullptr) {
            // max |dxe| for the consumer that scales dxe into fp16's range (gnnome_wgrad_scaled_f32): one atomicMax per load wave on the bits
            // of a non-negative float - a maximum does not depend on the order it is formed in.  (A NaN row fails every comparison here and
            // is caught by its own products downstream.)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) amax3 = fmaxf(amax3, __shfl_xor(amax3, o));
            if (lane == 0) atomicMax(a.bnb.amax_bits, __float_as_uint(amax3));
        }
        if (MODE == 1) {
            // lanes l and l + 32 hold different rows of the same four columns: fold them, then every load wave leaves one row of
            // partial sums (stats[(block * 2 RING + wave - 4)][2H], the layout of k_edge_gate_bf)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                st1[j] += __shfl_xor(st1[j], 32);
                st2[j] += __shfl_xor(st2[j], 32);
            }
            if (lane < H / 4) {
                float* dst = a.stats + ((int64_t)blockIdx.x * (RING * 2) + (wave - 4)) * 2 * H;
                *reinterpret_cast<f32x4*>(dst + 4 * c4) = st1;
                *reinterpret_cast<f32x4*>(dst + H + 4 * c4) = st2;
            }
        }
        if (a.prof && wave == 4 && lane == 0) {   // the first load wave's phases, after the 256 compute-wave records
            long long* o = a.prof + (int64_t)(256 + blockIdx.x) * 8;
            o[0] = t_top; o[1] = t_split; o[2] = t_done; o[3] = t_epi; o[4] = (n + RING - 1) / RING;
        }
    }
}

template <bool ENC, int MODE = 0, bool X16 = false, bool F16 = false, int EXTRA = 0>
static int launch_pl_arith(const GateBfArgs& args, hipStream_t s);
template <bool ENC, int MODE = 0, bool X16 = false, int EXTRA = 0>
static int launch_pl(const GateBfArgs& args, hipStream_t s) {
    // forward modes: fp16x3 unless gnnome_set_tuning(10, 1) asks for bf16x6 (the folded-encoder form keeps bf16x6; mode 1 with bf16 storage follows
    // the fp32-storage kernel, so that what it stores is that kernel's result rounded)
    constexpr bool kForward = !ENC && (MODE == 1 || (MODE == 0 && (EXTRA || !X16)) || (MODE == 4 && !X16));
    if (kForward && tuning(kTuneArith) == 0) return launch_pl_arith<ENC, kForward ? MODE : 0, kForward ? X16 : false, kForward, kForward ? EXTRA : 0>(args, s);
    return launch_pl_arith<ENC, MODE, X16, false, EXTRA>(args, s);
}
template <bool ENC, int MODE, bool X16, bool F16, int EXTRA>
static int launch_pl_arith(const GateBfArgs& args, hipStream_t s) {
    GateBfArgs a = args;
    const int64_t tiles = (a.E + 31) / 32;
    GN_REQUIRE(tiles < (1ll << 31), "edge_gate: too many tiles");
    a.num_tiles = (int)tiles;
    a.xp = tuning(kTuneGateExperiment);
    a.prof = g_gate_prof;
    if (MODE == 1) GN_HIP(hipMemsetAsync(a.stats, 0, sizeof(float) * kNumCUs * 8 * 2 * 128, s));   // idle waves leave zeros
    GN_REQUIRE(MODE != 4 || (a.num_cblocks >= 1 && a.num_cblocks <= persistent_grid() / kXcds && a.ldn >= 128 && a.ldn % 4 == 0 && a.ld_out % 4 == 0),
               "linear (K = 128): %d column blocks / strides %d, %d", a.num_cblocks, a.ldn, a.ld_out);
    hipLaunchKernelGGL((k_edge_gate_pl<ENC, MODE, X16, F16, EXTRA>), dim3(persistent_grid()), dim3(768), 0, s, a);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}


// ---------------------------------------------------------------------------------------------------
// Layer 0 at H = 128 with the edge encoder folded ALGEBRAICALLY.  e0 = t W2^T + b2 with t = relu(W1 e_raw + b1) of width 16,
// so B_3(e0) = t (W3 W2)^T + W3 b2: both the gate's GEMM and the residual e0 are K = 16 products of the same [32 x 16] tile t.
// k_edge_gate_pl<ENC> computed e0 [32 x 128] in the load waves (~900 VALU operations per lane and tile) and then ran the K = 128
// GEMM on it; here the load waves produce t (one row and four hidden units per lane), the compute waves run 6 + 6 MFMAs per tile
// (x = t W23^T and e0 = t W2^T into two accumulators) and the kernel is bound by its 512 MB of output.  W23 = W3 W2 and
// b23 = W3 b2 come from k_fold_encoder (fp32, one wave per element, fixed summation tree) on the same stream.
// Three load groups: a slot holds the three [32 x 16] bf16 planes of t and the two fp32 tiles x and e0 (38 KB).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fold_encoder(const float* __restrict__ W3, int ldw, const float* __restrict__ W2,
                                                      const float* __restrict__ b2, int H, float* __restrict__ W23, float* __restrict__ b23) {
    // one wave per output element: H*16 products W23[col][j] = sum_k W3[col][k] W2[k][j], then H bias terms sum_k W3[col][k] b2[k];
    // lane l takes k = l, l + 64, ...; the 64 lane sums are folded by a butterfly (a fixed tree: the same bits on every launch)
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= H * 17) return;
    const bool bias = i >= H * 16;
    const int col = bias ? i - H * 16 : i / 16, j = i % 16;
    float s = 0.f;
    for (int k = lane; k < H; k += 64) s = __builtin_fmaf(W3[(int64_t)col * ldw + k], bias ? b2[k] : W2[k * 16 + j], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) (bias ? b23[col] : W23[i]) = s;
}

__global__ __launch_bounds__(640) void k_edge_gate_enc16(GateBfArgs a) {
    constexpr int H = 128, TM = 32, RING = 3, LDK = H + 4, TPLD = 48, TPLANE = TM * TPLD, TILEF = TM * LDK;
    constexpr int SLOTB = 3 * TPLANE + 2 * TILEF * 4, NP = 8, RSTEP = 4;
    __shared__ __attribute__((aligned(16))) unsigned char ring[RING * SLOTB];
    __shared__ __attribute__((aligned(16))) float consts[4 * H];   // scale | shift | b23 | b2
    __shared__ float w1s[32], b1s[16];
    __shared__ unsigned flags[2 * RING];   // full[RING], done[RING]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned full0 = lds_addr_bf(&flags[0]), done0 = lds_addr_bf(&flags[RING]);
    const int per_xcd = gridDim.x / kXcds;
    const int first = (int)(blockIdx.x % kXcds) * per_xcd + (int)(blockIdx.x / kXcds);
    const int stride = (int)gridDim.x;
    const int n = first < a.num_tiles ? (a.num_tiles - first + stride - 1) / stride : 0;
    if (n <= 0) return;
    auto tile_of = [&](int r) { return first + r * stride; };
    auto tile_valid = [&](int r) { return (int)min((int64_t)TM, a.E - (int64_t)tile_of(r) * TM); };
    if (tid < 4 * H) {
        const int q = tid / H, c = tid % H;
        consts[tid] = q == 0 ? a.scale[c] : q == 1 ? a.shift[c] : q == 2 ? a.enc.b23[c] : a.enc.b2[c];
    }
    if (tid >= 4 * H && tid < 4 * H + 32) w1s[tid - 4 * H] = a.enc.W1[tid - 4 * H];
    if (tid >= 4 * H + 32 && tid < 4 * H + 48) b1s[tid - 4 * H - 32] = a.enc.b1[tid - 4 * H - 32];
    if (tid < 2 * RING) flags[tid] = 0;
    __syncthreads();

    if (wave < 4) {
        // ------------------------------------------------------------------ compute wave: 32 rows x 32 columns, K = 16
        const int cl = lane & 31, half = lane >> 5, col = 32 * wave + cl;
        uint4 u1, u2, u3, v1, v2, v3;
        {
            const float* p = a.enc.W23 + col * 16 + 8 * half;
            split3(*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4), u1, u2, u3);
            const float* q = a.enc.W2 + col * 16 + 8 * half;
            split3(*reinterpret_cast<const f32x4*>(q), *reinterpret_cast<const f32x4*>(q + 4), v1, v2, v3);
        }
        auto crow = [](int r) { return (r & 3) + 8 * (r >> 2); };
        const int lane_x = 4 * half * LDK + col;
        const float bxc = consts[2 * H + col], bec = consts[3 * H + col];
        for (int i = 0; i < n; ++i) {
            const int slot = i % RING;
            const unsigned use = (unsigned)(i / RING) + 1u;
            flag_wait_bf(full0 + 4 * slot, 2u * use, 0);
            const unsigned char* tp = ring + slot * SLOTB + cl * TPLD + 16 * half;
            const uint4 c1 = *reinterpret_cast<const uint4*>(tp), c2 = *reinterpret_cast<const uint4*>(tp + TPLANE),
                        c3 = *reinterpret_cast<const uint4*>(tp + 2 * TPLANE);
            f32x16 ax, ae;
#pragma unroll
            for (int r = 0; r < 16; ++r) ax[r] = ae[r] = 0.f;
            // smallest terms first; the two products alternate
            ax = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(c3), as_bf(u1), ax, 0, 0, 0);
            ae = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(c3), as_bf(v1), ae, 0, 0, 0);
            ax = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(c1), as_bf(u3), ax, 0, 0, 0);
            ae = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(c1), as_bf(v3), ae, 0, 0, 0);
            ax = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(c2), as_bf(u2), ax, 0, 0, 0);
            ae = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(c2), as_bf(v2), ae, 0, 0, 0);
            ax = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(c2), as_bf(u1), ax, 0, 0, 0);
            ae = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(c2), as_bf(v1), ae, 0, 0, 0);
            ax = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(c1), as_bf(u2), ax, 0, 0, 0);
            ae = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(c1), as_bf(v2), ae, 0, 0, 0);
            ax = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(c1), as_bf(u1), ax, 0, 0, 0);
            ae = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(c1), as_bf(v1), ae, 0, 0, 0);
            float* X = reinterpret_cast<float*>(ring + slot * SLOTB + 3 * TPLANE) + lane_x;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                X[crow(r) * LDK] = ax[r] + bxc;          // (+ b23 / + b2 here, one column per lane, rather than as two more float4
                X[TILEF + crow(r) * LDK] = ae[r] + bec;  //  constants in the store waves: the same two additions, the same bits)
            }
            flag_bump_bf(done0 + 4 * slot, lane);
        }
    } else {
        // ------------------------------------------------------------------ load / store wave
        const int group = (wave - 4) / 2;
        const int gl = ((wave - 4) % 2) * 64 + lane;   // 0..127
        const int r0 = gl / (H / 4), c4 = gl % (H / 4);   // epilogue role: rows r0 + 4 p, columns 4 c4 .. + 3
        const int trow = gl >> 2, jq = gl & 3;             // encoder role: row trow, hidden units 4 jq .. + 3
        f32x4 g1[NP], g2[NP], gk[NP];
        int si[NP], di[NP], eid = 0;
        float raw0 = 0.f, raw1 = 0.f;
        const unsigned off_row = (unsigned)(r0 * H + 4 * c4);
        auto issue_early = [&](int r) {
            const int64_t row0 = (int64_t)tile_of(r) * TM;
            const int valid = tile_valid(r);
#pragma unroll
            for (int p = 0; p < NP; ++p) {   // rows past the end of the list read the last valid row (never stored)
                const int64_t row = row0 + min(r0 + p * RSTEP, valid - 1);
                si[p] = a.srt_src[row];
                di[p] = a.srt_dst[row];
            }
            eid = a.enc.srt_eid[row0 + min(trow, valid - 1)];
        };
        auto issue_late = [&]() {
            raw0 = a.enc.e_raw[2 * (int64_t)eid];
            raw1 = a.enc.e_raw[2 * (int64_t)eid + 1];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                g1[p] = *reinterpret_cast<const f32x4*>(a.B1h + (int64_t)si[p] * a.ldn + 4 * c4);
                g2[p] = *reinterpret_cast<const f32x4*>(a.B2h + (int64_t)di[p] * a.ldn + 4 * c4);
            }
        };
        // Order of a group's requests (round 4): the gathers of its NEXT tile go out before the epilogue of the current one, from indices that were
        // fetched a tile earlier still - g1 / g2 are free once G is summed, the loads precede the epilogue's stores in the in-order memory counter,
        // and they have the wait for the compute waves + the epilogue to arrive.  Before, indices -> gathers -> use were strung out behind each
        // epilogue and a group spent most of its cycle waiting for them: with three groups and half the gate's memory traffic this launch was bound
        // by that chain (0.19 ms against 0.09 ms of store time), not by memory.
        if (group < n) {
            issue_early(group);
            issue_late();
            if (group + RING < n) issue_early(group + RING);
        }
        unsigned char* S = ring + group * SLOTB;
        const float* Xs = reinterpret_cast<const float*>(S + 3 * TPLANE);
        for (int r = group; r < n; r += RING) {
            const unsigned use = (unsigned)(r / RING) + 1u;
            {   // t = relu(W1 e_raw + b1) in the reference's order (models/full_graph.py:27), four hidden units of one row
                f32x4 t;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int hu = 4 * jq + j;
                    t[j] = fmaxf(__builtin_fmaf(raw1, w1s[2 * hu + 1], raw0 * w1s[2 * hu]) + b1s[hu], 0.f);
                }
                uint2 p1, p2, p3;
                split4_planes(t, p1, p2, p3);
                unsigned char* d = S + trow * TPLD + 8 * jq;
                *reinterpret_cast<uint2*>(d) = p1;
                *reinterpret_cast<uint2*>(d + TPLANE) = p2;
                *reinterpret_cast<uint2*>(d + 2 * TPLANE) = p3;
            }
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                gk[p] = g1[p] + g2[p];
                asm volatile("" : "+v"(gk[p]));   // summed here, not in the epilogue (see k_edge_gate_pl)
            }
            flag_bump_bf(full0 + 4 * group, lane);
            if (r + RING < n) issue_late();                       // tile r + RING: its indices arrived during the previous iteration
            if (r + 2 * RING < n) issue_early(r + 2 * RING);
            flag_wait_bf(done0 + 4 * group, 4u * use, a.xp & 3);
            const int valid = tile_valid(r);
            const f32x4 sc4 = *reinterpret_cast<const f32x4*>(consts + 4 * c4), sh4 = *reinterpret_cast<const f32x4*>(consts + H + 4 * c4);
            float* out = a.e_out + (int64_t)tile_of(r) * TM * H;
            // (two pieces per LDS round trip, not four: the next tile's 64 gather registers are live across this epilogue now)
#pragma unroll
            for (int pb = 0; pb < NP; pb += 2) {
                f32x4 x[2], e0[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    x[u] = *reinterpret_cast<const f32x4*>(Xs + (r0 + (pb + u) * RSTEP) * LDK + 4 * c4);
                    e0[u] = *reinterpret_cast<const f32x4*>(Xs + TILEF + (r0 + (pb + u) * RSTEP) * LDK + 4 * c4);
                }
                asm volatile("" : "+v"(x[0]), "+v"(x[1]));
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int p = pb + u, row = r0 + p * RSTEP;
                    f32x4 y;
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[j] = fmaxf((x[u][j] + gk[p][j]) * sc4[j] + sh4[j], 0.f) + e0[u][j];
                    if (row < valid) *reinterpret_cast<f32x4*>(out + (off_row + (unsigned)(p * RSTEP * H))) = y;
                }
            }
        }
    }
}

struct EncFoldScratch {
    float* w23 = nullptr;   // [128*16 + 128]
};
static EncFoldScratch g_enc_fold[16];

static int launch_enc16(const GateBfArgs& args, hipStream_t s) {
    GateBfArgs a = args;
    const int64_t tiles = (a.E + 31) / 32;
    GN_REQUIRE(tiles < (1ll << 31), "edge_gate: too many tiles");
    a.num_tiles = (int)tiles;
    a.xp = tuning(kTuneGateExperiment);
    int dev = 0;
    GN_HIP(hipGetDevice(&dev));
    GN_REQUIRE(dev >= 0 && dev < 16, "edge_gate_encode: device index %d", dev);
    // W3 W2 and W3 b2, recomputed at every call (the weights may have changed) into one small per-device buffer: calls on
    // different streams of one device must not overlap (as for the aggregation's hub scratch)
    if (!g_enc_fold[dev].w23) GN_HIP(hipMalloc(&g_enc_fold[dev].w23, sizeof(float) * (128 * 16 + 128)));
    float* w23 = g_enc_fold[dev].w23;
    hipLaunchKernelGGL(k_fold_encoder, dim3((128 * 17 + 3) / 4), dim3(256), 0, s, a.W3, a.ldw, a.enc.W2, a.enc.b2, 128, w23, w23 + 128 * 16);
    GN_LAUNCH_CHECK();
    a.enc.W23 = w23;
    a.enc.b23 = w23 + 128 * 16;
    hipLaunchKernelGGL(k_edge_gate_enc16, dim3(persistent_grid()), dim3(640), 0, s, a);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

// H = 256 (round 4): the same fold on the fp16x3 edge-tile kernel (edge_tile_f16.hip, mode 5)
static EncFoldScratch g_enc_fold256[16];
int gate_enc256_launch(const GateBfArgs& args, hipStream_t s) {
    GateBfArgs a = args;
    const int64_t tiles = (a.E + 31) / 32;
    GN_REQUIRE(tiles < (1ll << 31), "edge_gate: too many tiles");
    a.num_tiles = (int)tiles;
    a.prof = g_gate_prof;
    int dev = 0;
    GN_HIP(hipGetDevice(&dev));
    GN_REQUIRE(dev >= 0 && dev < 16, "edge_gate_encode: device index %d", dev);
    if (!g_enc_fold256[dev].w23) GN_HIP(hipMalloc(&g_enc_fold256[dev].w23, sizeof(float) * (256 * 16 + 256)));
    float* w23 = g_enc_fold256[dev].w23;
    hipLaunchKernelGGL(k_fold_encoder, dim3((256 * 17 + 3) / 4), dim3(256), 0, s, a.W3, a.ldw, a.enc.W2, a.enc.b2, 256, w23, w23 + 256 * 16);
    GN_LAUNCH_CHECK();
    a.enc.W23 = w23;
    a.enc.b23 = w23 + 256 * 16;
    int g = persistent_grid();
    g -= g % 16;   // pairs of workgroups per XCD (as grid_pl256)
    return gate_f16_launch(5, a, g < 16 ? 16 : g, s);
}

template <int CB, int RB, int MODE, bool ENC, bool X16 = false>
static int launch_bf(const GateBfArgs& args, hipStream_t s) {
    using P = GateBF<CB, RB>;
    GateBfArgs a = args;
    const int64_t tiles = (a.E + P::TM - 1) / P::TM;
    GN_REQUIRE(tiles < (1ll << 31), "edge_gate: too many tiles");
    a.num_tiles = (int)tiles;
    a.abl = tuning(kTuneGateAblation);
    a.xp = tuning(kTuneGateExperiment);
    a.prof = g_gate_prof;
    if (MODE == 1) GN_HIP(hipMemsetAsync(a.stats, 0, sizeof(float) * kNumCUs * P::RING * P::LWAVES * 2 * P::H, s));   // idle waves leave zeros
    hipLaunchKernelGGL((k_edge_gate_bf<CB, RB, MODE, ENC, X16>), dim3(persistent_grid()), dim3(P::NT), 0, s, a);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

int gate_bf_launch(int hidden, int mode, bool enc, const GateBfArgs& a, hipStream_t s, bool x16, int extra) {
    const bool planes = hidden == 128 && tuning(kTuneGateVariant) != 8;   // the plane form (k_edge_gate_pl), modes 0, 1 and 3
    if (extra) {   // the two-pass training forward: statistics only (mode 1) / gate + xe out (mode 0); H = 128 plane form only
        GN_REQUIRE(hidden == 128 && !enc && (mode == 0 || mode == 1), "edge-tile kernel: the two-pass training forward is built at hidden = 128");
        if (mode == 1) return x16 ? launch_pl<false, 1, true, 1>(a, s) : launch_pl<false, 1, false, 1>(a, s);
        return x16 ? launch_pl<false, 0, true, 1>(a, s) : launch_pl<false, 0, false, 1>(a, s);
    }
    if (x16) {   // bf16 storage of xe / dxe: modes 1 and 3 only
        if (mode == 1 && planes) return launch_pl<false, 1, true>(a, s);
        if (mode == 3 && planes) return launch_pl<false, 3, true>(a, s);
        if (mode == 1) return hidden == 128 ? launch_bf<4, 1, 1, false, true>(a, s) : launch_bf<2, 2, 1, false, true>(a, s);
        if (mode == 3) return hidden == 128 ? launch_bf<4, 1, 3, false, true>(a, s) : launch_bf<2, 2, 3, false, true>(a, s);
        set_error("edge-tile kernel: bf16 storage exists for modes 1 and 3 only");
        return GNNOME_EINVAL;
    }
    if (hidden == 128) {
        // mode 0: the plane form (k_edge_gate_pl) is the default, with the folded encoder too (its 24 extra registers spill
        // there, and it still measures 0.025 ms ahead); variant 8 forces the second-generation kernel
        if (mode == 0 && enc && tuning(kTuneGateVariant) == 0) return launch_enc16(a, s);   // the encoder folded algebraically: K = 16
        if (mode == 0 && tuning(kTuneGateVariant) != 8) return enc ? launch_pl<true>(a, s) : launch_pl<false>(a, s);
        if (mode == 1 && planes) return launch_pl<false, 1>(a, s);
        if (mode == 3 && planes) return launch_pl<false, 3>(a, s);
        if (mode == 4) return launch_pl<false, 4>(a, s);
        if (mode == 2 && planes) return launch_pl<false, 2>(a, s);
        if (mode == 0) return enc ? launch_bf<4, 1, 0, true>(a, s) : launch_bf<4, 1, 0, false>(a, s);
        if (mode == 3) return launch_bf<4, 1, 3, false>(a, s);
        return mode == 1 ? launch_bf<4, 1, 1, false>(a, s) : launch_bf<4, 1, 2, false>(a, s);
    }
    if (mode == 0) return enc ? launch_bf<2, 2, 0, true>(a, s) : launch_bf<2, 2, 0, false>(a, s);
    if (mode == 3) return launch_bf<2, 2, 3, false>(a, s);
    return mode == 1 ? launch_bf<2, 2, 1, false>(a, s) : launch_bf<2, 2, 2, false>(a, s);
}

}  // namespace gnnome

extern "C" int gnnome_debug_gate_profile(void* counters) {
    gnnome::g_gate_prof = (long long*)counters;
    return GNNOME_OK;
}

// C[M,H] += BatchNormBackward(C, X) W^T and dxe = BatchNormBackward(C, X) written out: gnnome_bn_bwd_apply_f32 followed by
// gnnome_linear_acc_f32 in ONE pass over the [E,H] tensors (the A tile never comes from HBM: the load waves compute it).
static int bn_bwd_dgrad_impl(float* C, const void* X, int64_t rows, int64_t rows_once, int hidden, const float* scale, const float* shift,
                             const float* a, const float* c1, const float* c2, const float* mean, const float* rstd,
                             const float* W, int ldw, void* dxe, void* stream, bool x16, unsigned* amax_bits = nullptr) {
    using namespace gnnome;
    GN_REQUIRE(rows >= 0 && (hidden == 64 || hidden == 128), "bn_bwd_dgrad: hidden=%d not in {64,128}", hidden);
    GN_REQUIRE(rows_once >= 0 && rows_once <= rows, "bn_bwd_dgrad: rows_once=%lld outside [0, rows]", (long long)rows_once);
    if (rows == 0) return GNNOME_OK;
    GN_REQUIRE(C && X && scale && shift && a && c1 && c2 && mean && rstd && W && dxe && dxe != (void*)C && ldw >= hidden && ldw % 4 == 0,
               "bn_bwd_dgrad: bad arguments");
    GN_REQUIRE(((uintptr_t)C % 16 == 0) && ((uintptr_t)X % 16 == 0) && ((uintptr_t)dxe % 16 == 0) && ((uintptr_t)W % 16 == 0),
               "bn_bwd_dgrad: tensors must be 16-byte aligned");
    GateBfArgs g = {};
    g.e_in = (const float*)X; g.e_out = C; g.E = rows; g.B1h = C; g.ldn = hidden; g.W3 = W; g.ldw = ldw;
    g.bnb = GateBnBwd{a, c1, c2, mean, rstd, scale, shift, (float*)dxe, rows_once};
    if (amax_bits != nullptr) {
        GN_REQUIRE(hidden == 128 && tuning(kTuneGateVariant) != 8, "bn_bwd_dgrad_amax: hidden = 128 on the plane-form kernel only");
        GN_HIP(hipMemsetAsync(amax_bits, 0, sizeof(unsigned), (hipStream_t)stream));
        g.bnb.amax_bits = amax_bits;
    }
    return gate_bf_launch(hidden, 3, false, g, (hipStream_t)stream, x16);
}

extern "C" int gnnome_bn_bwd_dgrad_f32(float* C, const float* X, int64_t rows, int64_t rows_once, int hidden, const float* scale, const float* shift,
                                       const float* a, const float* c1, const float* c2, const float* mean, const float* rstd,
                                       const float* W, int ldw, float* dxe, void* stream) {
    return bn_bwd_dgrad_impl(C, X, rows, rows_once, hidden, scale, shift, a, c1, c2, mean, rstd, W, ldw, dxe, stream, false);
}

// ... and max |dxe| left at amax_bits (the bits of a non-negative float) for gnnome_wgrad_scaled_f32
extern "C" int gnnome_bn_bwd_dgrad_amax_f32(float* C, const float* X, int64_t rows, int64_t rows_once, int hidden, const float* scale,
                                            const float* shift, const float* a, const float* c1, const float* c2, const float* mean,
                                            const float* rstd, const float* W, int ldw, float* dxe, unsigned* amax_bits, void* stream) {
    GN_REQUIRE(amax_bits != nullptr, "bn_bwd_dgrad_amax: null amax_bits");
    return bn_bwd_dgrad_impl(C, X, rows, rows_once, hidden, scale, shift, a, c1, c2, mean, rstd, W, ldw, dxe, stream, false, amax_bits);
}

extern "C" int gnnome_bn_bwd_dgrad_x16(float* C, const uint16_t* X, int64_t rows, int64_t rows_once, int hidden, const float* scale, const float* shift,
                                       const float* a, const float* c1, const float* c2, const float* mean, const float* rstd,
                                       const float* W, int ldw, uint16_t* dxe, void* stream) {
    return bn_bwd_dgrad_impl(C, X, rows, rows_once, hidden, scale, shift, a, c1, c2, mean, rstd, W, ldw, dxe, stream, true);
}

// The same pass at hidden = 256, OUT OF PLACE: two workgroups (column halves) read whole rows of C while each writes its half, so the
// updated rows go to C_out != C_in (edge_gate_pl256.hip, mode 3).
static int bn_bwd_dgrad_out_impl(const float* C_in, float* C_out, const float* X, int64_t rows, int64_t rows_once, int hidden,
                                 const float* scale, const float* shift, const float* a, const float* c1, const float* c2,
                                 const float* mean, const float* rstd, const float* W, int ldw, float* dxe, void* stream, bool x16,
                                 unsigned* amax_bits = nullptr);
// ... and max |dxe| left at amax_bits (round 6: gnnome_wgrad_scaled_f32 then runs B_3's [256, 256] weight gradient as fp16x3)
extern "C" int gnnome_bn_bwd_dgrad_out_amax_f32(const float* C_in, float* C_out, const float* X, int64_t rows, int64_t rows_once, int hidden,
                                                const float* scale, const float* shift, const float* a, const float* c1, const float* c2,
                                                const float* mean, const float* rstd, const float* W, int ldw, float* dxe, unsigned* amax_bits,
                                                void* stream) {
    GN_REQUIRE(amax_bits != nullptr, "bn_bwd_dgrad_out_amax: null amax_bits");
    return bn_bwd_dgrad_out_impl(C_in, C_out, X, rows, rows_once, hidden, scale, shift, a, c1, c2, mean, rstd, W, ldw, dxe, stream, false, amax_bits);
}
extern "C" int gnnome_bn_bwd_dgrad_out_f32(const float* C_in, float* C_out, const float* X, int64_t rows, int64_t rows_once, int hidden,
                                           const float* scale, const float* shift, const float* a, const float* c1, const float* c2,
                                           const float* mean, const float* rstd, const float* W, int ldw, float* dxe, void* stream) {
    return bn_bwd_dgrad_out_impl(C_in, C_out, X, rows, rows_once, hidden, scale, shift, a, c1, c2, mean, rstd, W, ldw, dxe, stream, false);
}
// X (the xe rows) and dxe as bf16 (round 4: bf16 activation storage at hidden = 256)
extern "C" int gnnome_bn_bwd_dgrad_out_x16(const float* C_in, float* C_out, const uint16_t* X, int64_t rows, int64_t rows_once, int hidden,
                                           const float* scale, const float* shift, const float* a, const float* c1, const float* c2,
                                           const float* mean, const float* rstd, const float* W, int ldw, uint16_t* dxe, void* stream) {
    return bn_bwd_dgrad_out_impl(C_in, C_out, (const float*)X, rows, rows_once, hidden, scale, shift, a, c1, c2, mean, rstd, W, ldw, (float*)dxe, stream, true);
}
static int bn_bwd_dgrad_out_impl(const float* C_in, float* C_out, const float* X, int64_t rows, int64_t rows_once, int hidden,
                                 const float* scale, const float* shift, const float* a, const float* c1, const float* c2,
                                 const float* mean, const float* rstd, const float* W, int ldw, float* dxe, void* stream, bool x16,
                                 unsigned* amax_bits) {
    using namespace gnnome;
    GN_REQUIRE(rows >= 0 && hidden == 256, "bn_bwd_dgrad_out: hidden=%d (256 only; 64 / 128 update C in place: gnnome_bn_bwd_dgrad_f32)", hidden);
    if (amax_bits != nullptr) GN_HIP(hipMemsetAsync(amax_bits, 0, sizeof(unsigned), (hipStream_t)stream));   // (before the rows == 0 return: an empty product has maximum 0)
    GN_REQUIRE(rows_once >= 0 && rows_once <= rows, "bn_bwd_dgrad_out: rows_once=%lld outside [0, rows]", (long long)rows_once);
    if (rows == 0) return GNNOME_OK;
    GN_REQUIRE(C_in && C_out && C_out != C_in && X && scale && shift && a && c1 && c2 && mean && rstd && W && dxe && dxe != C_out && dxe != C_in &&
                   ldw >= hidden && ldw % 4 == 0, "bn_bwd_dgrad_out: bad arguments");
    GN_REQUIRE(((uintptr_t)C_in % 16 == 0) && ((uintptr_t)C_out % 16 == 0) && ((uintptr_t)X % 16 == 0) && ((uintptr_t)dxe % 16 == 0) &&
                   ((uintptr_t)W % 16 == 0), "bn_bwd_dgrad_out: tensors must be 16-byte aligned");
    (void)x16;
    GateBfArgs g = {};
    g.e_in = X; g.e_out = C_out; g.E = rows; g.B1h = C_in; g.ldn = hidden; g.W3 = W; g.ldw = ldw;
    g.bnb = GateBnBwd{a, c1, c2, mean, rstd, scale, shift, dxe, rows_once};
    g.bnb.amax_bits = amax_bits;
    return gate_pl256_launch(3, g, (hipStream_t)stream, x16);
}
