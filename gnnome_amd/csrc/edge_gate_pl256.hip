// Edge-tile kernel for H = 256 in the wave-specialised PLANE form (round 3; k_edge_gate_stream stays as variant 9).
//
// k_edge_gate_stream keeps one 64-column chunk of W3 as bf16 planes in LDS (101 KB) and lets every wave fetch, split and
// multiply its own rows: four workgroups read and split every e row, and a wave's costs - gathers, operand split, MFMAs,
// residual, stores - add up serially at two waves per SIMD (2.36 ms per launch at 2.5M edges against
// ~1.3 ms of matrix-core / HBM time).  This kernel is k_edge_gate_pl's structure at K = 256:
//   * W3 lives in REGISTERS: a compute wave holds the three bf16 planes of its 32 output columns for all 256 k
//     (16 steps x 3 planes x 4 VGPRs = 192 of the 256 registers a wave has at two waves per SIMD); four compute waves = 128
//     columns, so TWO workgroups (column halves, same XCD, same tile sequence) cover a row - the e rows are read and split
//     twice, not four times;
//   * four load / store waves in two groups: a group fetches a tile's 32 full e rows, splits them ONCE into three bf16 planes
//     in LDS (50 KB per slot, two slots), and - while the compute waves run 96 MFMAs per tile each, pure matrix work - fetches
//     the next tile; it then applies the epilogue (G = B1h[src] + B2h[dst], normalise, relu, residual) to the x tile the
//     compute waves left in a separate 17 KB LDS buffer, with 16-byte row pieces;
//   * hand-over through two LDS counters per slot (full: planes ready, 2 bumps; done: x ready, 4 bumps); the x tile has its
//     own buffer, so the compute waves never wait for one another and no third or fourth counter is needed (cf. k_edge_gate_pl).
// Per tile and compute wave: 16 x 6 MFMAs = 3072 matrix-pipe cycles (3560 measured at the 1.9 GHz the chip holds under this load).
// MEASURED (tools/gate_phase_profile.py --hidden 256, profiles/r03_gate256_phases.txt): 5700 cycles per tile, the compute waves
// busy 63 % of the time; the load / store waves are the long pole - per tile a wave handles (every second one) 3100 cycles of
// split + plane stores (352 VALU operations and 48 ds_write_b64 per lane; one VALU-issuing wave per SIMD retires an instruction
// every ~5 cycles), 3600 of epilogue (residual rebuilt from the planes, G, normalise, eight 16-byte stores) and 3100 issuing the
// next requests.  What was learned on the way (each measured on the box):
//   * with ALL global traffic compiled out the kernel still took 1.57 ms: the load waves' own instruction streams bound it, not HBM;
//   * the hipcc waitcnt pass turns a wait that sits between CONDITIONAL stores into s_waitcnt vmcnt(0) - the wave then sits out
//     the HBM write latency of its own stores (500 cycles per epilogue piece): every operand that came through the vector-memory
//     queue is awaited before the first store, and whole tiles store unconditionally;
//   * a register copy of a value whose load was just issued (the destination-row reuse below, first version) serialises the
//     gathers - vmcnt(1) after every pair, 4800 cycles of "issue" - so the reuse is resolved in the epilogue, a period later;
//   * loads issued BEFORE the wait for the compute waves make the epilogue's stores queue behind HBM misses (3760 cycles);
//   * the split of tile r + 2 goes BEFORE the epilogue of tile r (chain done -> full 3100 cycles instead of 8400);
//   * B2h[dst] is fetched once per run of equal destinations (a lane's pieces are consecutive, destination-sorted rows) and the
//     residual is rebuilt from the planes ((x1 + x2) + x3 is exact) instead of being read again: 96 -> ~69 KB per tile-half
//     through the CU's vector-memory queue;
//   * no effect: s_setprio for the load waves, the two workgroups of a pair walking their tiles in opposite order.
// 2.36 ms (k_edge_gate_stream) -> 1.98 ms per launch at the 2.5M-edge shard; the matrix-pipe floor of this form is 1.25 ms.
// e_out must not alias e_in (the two column halves of a row are written by different workgroups while both read whole rows).
#include "common.h"

#include <type_traits>

namespace gnnome {
namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
constexpr bool kResidualFromPlanes = true;   // the epilogue's residual rebuilt from the LDS planes (true; 5724 cycles per tile) or fetched again (false; 5949)

__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p; }
__device__ __forceinline__ void flag_wait(unsigned addr, unsigned want) {
    unsigned v, spins = 0;
    for (;;) {
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
        if (__builtin_amdgcn_readfirstlane(v) >= want) break;
        if (++spins > (1u << 26)) __builtin_trap();   // a lost hand-over must end the launch, not hang the queue
        __builtin_amdgcn_s_sleep(1);
    }
}
__device__ __forceinline__ void flag_bump(unsigned addr, int lane) {
    if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(1u) : "memory");
}
// exact three-way bf16 split (see edge_gate_bf.hip): eight floats -> one MFMA operand per plane
__device__ __forceinline__ void split8(const f32x4 lo4, const f32x4 hi4, uint4& p1, uint4& p2, uint4& p3) {
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = j < 4 ? lo4[j] : hi4[j - 4];
        h[j] = __float_as_uint(x) & 0xFFFF0000u;
        const float r = x - __uint_as_float(h[j]);
        m[j] = __float_as_uint(r) & 0xFFFF0000u;
        l[j] = __float_as_uint(r - __uint_as_float(m[j]));
    }
    p1 = make_uint4(__builtin_amdgcn_perm(h[1], h[0], 0x07060302u), __builtin_amdgcn_perm(h[3], h[2], 0x07060302u),
                    __builtin_amdgcn_perm(h[5], h[4], 0x07060302u), __builtin_amdgcn_perm(h[7], h[6], 0x07060302u));
    p2 = make_uint4(__builtin_amdgcn_perm(m[1], m[0], 0x07060302u), __builtin_amdgcn_perm(m[3], m[2], 0x07060302u),
                    __builtin_amdgcn_perm(m[5], m[4], 0x07060302u), __builtin_amdgcn_perm(m[7], m[6], 0x07060302u));
    p3 = make_uint4(__builtin_amdgcn_perm(l[1], l[0], 0x07060302u), __builtin_amdgcn_perm(l[3], l[2], 0x07060302u),
                    __builtin_amdgcn_perm(l[5], l[4], 0x07060302u), __builtin_amdgcn_perm(l[7], l[6], 0x07060302u));
}
__device__ __forceinline__ void split4(const f32x4 x, uint2& p1, uint2& p2, uint2& p3) {
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h[j] = __float_as_uint(x[j]) & 0xFFFF0000u;
        const float r = x[j] - __uint_as_float(h[j]);
        m[j] = __float_as_uint(r) & 0xFFFF0000u;
        l[j] = __float_as_uint(r - __uint_as_float(m[j]));
    }
    p1 = make_uint2(__builtin_amdgcn_perm(h[1], h[0], 0x07060302u), __builtin_amdgcn_perm(h[3], h[2], 0x07060302u));
    p2 = make_uint2(__builtin_amdgcn_perm(m[1], m[0], 0x07060302u), __builtin_amdgcn_perm(m[3], m[2], 0x07060302u));
    p3 = make_uint2(__builtin_amdgcn_perm(l[1], l[0], 0x07060302u), __builtin_amdgcn_perm(l[3], l[2], 0x07060302u));
}
__device__ __forceinline__ bf16x8_t as_bf8(const uint4 v) { return __builtin_bit_cast(bf16x8_t, v); }

// fp16x3 planes (edge_tile_f16.hip's header): x1 = RN16(x), x2 = RN16((x - x1) 2048); and, for the one-accumulator form of mode 3, x1 2048
typedef _Float16 pl_h2 __attribute__((ext_vector_type(2)));
typedef _Float16 pl_h8 __attribute__((ext_vector_type(8)));
typedef float pl_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pl_h8 as_h8(const uint4 v) { return __builtin_bit_cast(pl_h8, v); }
__device__ __forceinline__ void split2h(const pl_f2 v, unsigned& p1, unsigned& p2, unsigned& p1x) {
    const pl_h2 a = __builtin_convertvector(v, pl_h2);
    const pl_f2 af = {(float)a[0], (float)a[1]};
    const pl_f2 r = {__builtin_fmaf(af[0], -2048.f, v[0] * 2048.f), __builtin_fmaf(af[1], -2048.f, v[1] * 2048.f)};   // exact
    p1 = __builtin_bit_cast(unsigned, a);
    p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, pl_h2));
    p1x = __builtin_bit_cast(unsigned, __builtin_convertvector(af * 2048.f, pl_h2));   // exact: |x| < 32 (the caller's scale), subnormal a1 included
}
__device__ __forceinline__ void split4h3(const f32x4 x, uint2& p1, uint2& p2, uint2& p1x) {
    split2h(pl_f2{x[0], x[1]}, p1.x, p2.x, p1x.x);
    split2h(pl_f2{x[2], x[3]}, p1.y, p2.y, p1x.y);
}
__device__ __forceinline__ void split8h(const f32x4 lo4, const f32x4 hi4, uint4& p1, uint4& p2) {
    unsigned unused;
    split2h(pl_f2{lo4[0], lo4[1]}, p1.x, p2.x, unused);
    split2h(pl_f2{lo4[2], lo4[3]}, p1.y, p2.y, unused);
    split2h(pl_f2{hi4[0], hi4[1]}, p1.z, p2.z, unused);
    split2h(pl_f2{hi4[2], hi4[3]}, p1.w, p2.w, unused);
}

// MODE 0: e' = relu((e W3^T + B1h[src] + B2h[dst]) * scale + shift) + e    (gated_gcn_full.py:97,104-110)
// MODE 1: xe = e W3^T + B1h[src] + B2h[dst] and its shifted column sums (training forward; a.scale = the centres, a.stats out)
// MODE 2: C += A W^T (A = e_in, C = e_out = the rows at B1h; the backward's d e_in = d e' + dxe W3)
// MODE 3: MODE 2 with A = BatchNorm-backward(old C rows, xe rows at e_in) computed by the load waves and written to bnb.a_out
// MODE 4: C[M, 128 * num_cblocks] = A[M,256] W^T + bias (a.e_in = A with row stride a.ldn, a.e_out = C with row stride a.ld_out, a.scale = bias):
//         the node projection [N,256] -> [N,1280] and the scorer's node halves at this width
// X16 (MODE 3, round 4): the xe rows are read and the dxe rows written as bf16 (the product C += dxe W^T uses the unrounded dxe - common.h)
// F16 (MODE 3, round 6): the product dxe W^T as fp16x3 in ONE accumulator.  dxe is a gradient computed right here, so its
// scale cannot come from a maximum known beforehand: every load wave scales ITS sixteen rows of a tile by the power of two that brings their
// largest |element| into [8, 16) and multiplies the rows of x by the inverse on the way out (rows of A scale rows of C).  Planes of A: a1,
// a2 2^11 and a1 2^11 (exact: a1 < 16), of W: w1 and w2 2^11, so that 2^11 a w = (a1 2^11) w1 + a1 (w2 2^11) + (a2 2^11) w1 are three MFMAs
// into the same accumulator - 48 per tile instead of bf16x6's 96, W in 128 registers instead of 192.  Same error bound as every fp16x3
// product; W (weights) must lie in fp16's range.
template <int MODE, int PROBE = 0, bool X16 = false, bool F16 = false>   // PROBE (measurement only, wrong results): 1 = no MFMAs, 2 = no plane reads either
__global__ __launch_bounds__(512) void k_edge_gate_pl256(GateBfArgs a) {
    static_assert(!X16 || MODE == 3, "bf16 storage in this kernel: mode 3 only");
    static_assert(!F16 || MODE == 3, "the one-accumulator fp16x3 form is built for mode 3");
    constexpr int H = 256, HC = 128, TM = 32, KS = H / 16, PLD = 2 * H + 16, PLANE = TM * PLD, SLOTB = 3 * PLANE, LDK = HC + 4, XT = TM * LDK;
    constexpr int NPF = 16, NPE = 8;   // pieces per lane: fetch mapping (whole rows), epilogue mapping (this workgroup's column half)
    __shared__ __attribute__((aligned(16))) unsigned char ring[2 * SLOTB];
    __shared__ __attribute__((aligned(16))) float xt[2 * XT];
    __shared__ __attribute__((aligned(16))) float norm_lds[(MODE == 3 ? 7 : 2) * (MODE == 3 ? H : HC)];
    __shared__ unsigned flags[6];   // full[2], done[2], drained[2]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned full0 = lds_addr(&flags[0]), done0 = lds_addr(&flags[2]), drained0 = lds_addr(&flags[4]);
    // pairs of workgroups on one XCD (blocks b and b + 8 share b % 8) take the two column halves of the same tiles; in round r
    // the chip works on one contiguous window of tiles, each XCD on a contiguous part of it
    const int per_xcd = gridDim.x / kXcds, xcd = blockIdx.x % kXcds, idx = blockIdx.x / kXcds;
    // MODE 4 (C = A W^T + bias with Nout = 128 * a.num_cblocks columns): a workgroup keeps ONE 128-column block of W for the whole
    // launch; the a.num_cblocks workgroups of an XCD that share (idx / num_cblocks) walk the same tiles, so an A row comes from
    // HBM once per XCD and from its L2 for the other column blocks
    const int ncb = MODE == 4 ? a.num_cblocks : 2, streams = per_xcd / ncb;
    if (idx >= streams * ncb) return;   // (32 workgroups per XCD, e.g. 10 column blocks: 3 streams, 2 idle workgroups)
    const int hh = idx % ncb, first = xcd * streams + idx / ncb, stride = kXcds * streams;
    const int n = first < a.num_tiles ? (a.num_tiles - first + stride - 1) / stride : 0;
    if (n <= 0) return;
    const int lda = MODE == 4 ? a.ldn : H, ldo = MODE == 4 ? a.ld_out : H;   // row strides of the A rows / of the output
    // The two workgroups of a pair walk their common tiles in opposite order within every two: workgroup 0 takes t0, t1, t2, t3, ...,
    // workgroup 1 takes t1, t0, t3, t2, ...  Each e row is then requested from HBM by ONE of the two and found in the XCD's L2 a tile
    // later by the other.  In step, both miss together: the second request merges into the first in L2 but still holds its L1
    // miss slots for the whole HBM latency, and a CU's ~32 KB of misses in flight is what bounds this kernel's fetch.
    auto tile_of = [&](int r) {
        const int rr = (MODE != 4 && hh == 1 && (r ^ 1) < n) ? (r ^ 1) : r;
        return first + rr * stride;
    };
    auto tile_valid = [&](int r) { return (int)min((int64_t)TM, a.E - (int64_t)tile_of(r) * TM); };
    const int colh = HC * hh;   // first global column of this workgroup's half
    if (tid < 6) flags[tid] = 0;
    if (MODE == 3) {
        for (int i = tid; i < 7 * H; i += 512) {
            const int q = i / H, c = i % H;
            const float* src = q == 0 ? a.bnb.a : q == 1 ? a.bnb.c1 : q == 2 ? a.bnb.c2 : q == 3 ? a.bnb.mean : q == 4 ? a.bnb.rstd : q == 5 ? a.bnb.scale : a.bnb.shift;
            norm_lds[i] = src[c];
        }
    } else if (MODE == 4) {
        for (int i = tid; i < HC; i += 512) norm_lds[i] = a.scale ? a.scale[colh + i] : 0.f;   // the bias of this column block
    } else if (MODE < 2) {
        for (int i = tid; i < 2 * HC; i += 512) {
            const int q = i / HC, c = i % HC;
            norm_lds[i] = q == 0 ? (a.scale ? a.scale[colh + c] : 0.f) : (MODE == 0 ? a.shift[colh + c] : 0.f);   // MODE 1: the centres (NULL: none)
        }
    }
    __syncthreads();

    if (wave < 4) {
        // ------------------------------------------------------------------ compute wave: 32 rows x 32 columns, all of K in registers
        const int cl = lane & 31, half = lane >> 5, col = colh + 32 * wave + cl;
        // k numbering of the matrix-core steps = k_edge_gate_stream's (so that the two kernels give the same bits): step 4 b + q of the
        // lower / upper half wave takes k in [64 b + 32 half + 8 q, + 8)
        uint4 w1[KS], w2[KS], w3[KS];
#pragma unroll
        for (int q = 0; q < KS; ++q) {
            const float* wp = a.W3 + (int64_t)col * a.ldw + 64 * (q >> 2) + 32 * half + 8 * (q & 3);
            if (F16) split8h(*reinterpret_cast<const f32x4*>(wp), *reinterpret_cast<const f32x4*>(wp + 4), w1[q], w2[q]);
            else split8(*reinterpret_cast<const f32x4*>(wp), *reinterpret_cast<const f32x4*>(wp + 4), w1[q], w2[q], w3[q]);
        }
        auto crow = [](int r) { return (r & 3) + 8 * (r >> 2); };
        const int lane_x = 4 * half * LDK + 32 * wave + cl;   // accumulator element r sits in tile row 4 half + crow(r)
        long long t_wait = 0, t_loop = 0, t_x = 0, t0 = 0, t1 = 0;
        const long long c_begin = a.prof ? (long long)__builtin_readcyclecounter() : 0;
        const long long r_begin = a.prof ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
        for (int i = 0; i < n; ++i) {
            const int slot = i & 1;
            const unsigned use = (unsigned)(i >> 1) + 1u;
            if (a.prof) t0 = __builtin_readcyclecounter();
            flag_wait(full0 + 4 * slot, 2u * use);
            if (a.prof) { t1 = __builtin_readcyclecounter(); t_wait += t1 - t0; t0 = t1; }
            const unsigned char* ap = ring + slot * SLOTB + cl * PLD + 64 * half;   // + 128 b + 16 q (step 4 b + q), + PLANE * plane
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            uint4 c1 = *reinterpret_cast<const uint4*>(ap), c2 = *reinterpret_cast<const uint4*>(ap + PLANE),
                  c3 = *reinterpret_cast<const uint4*>(ap + 2 * PLANE);
#pragma unroll
            for (int q = 0; q < KS; ++q) {
                const int qn = q + 1 < KS ? q + 1 : q, on = 128 * (qn >> 2) + 16 * (qn & 3);
                const uint4 n1 = *reinterpret_cast<const uint4*>(ap + on), n2 = *reinterpret_cast<const uint4*>(ap + on + PLANE),
                            n3 = *reinterpret_cast<const uint4*>(ap + on + 2 * PLANE);
                if (PROBE >= 1) {
                    acc[q & 15] += __uint_as_float(c1.x ^ c2.y ^ c3.z ^ w1[q].x ^ w2[q].y ^ w3[q].z);
                    c1 = n1, c2 = n2, c3 = n3;
                    continue;
                }
                if (F16) {   // planes of A: c1 = a1, c2 = a2 2^11, c3 = a1 2^11; smallest terms first
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(c2), as_h8(w1[q]), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(c1), as_h8(w2[q]), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(c3), as_h8(w1[q]), acc, 0, 0, 0);
                    c1 = n1;
                    c2 = n2;
                    c3 = n3;
                    continue;
                }
                // smallest terms first
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(c3), as_bf8(w1[q]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(c1), as_bf8(w3[q]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(c2), as_bf8(w2[q]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(c2), as_bf8(w1[q]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(c1), as_bf8(w2[q]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(c1), as_bf8(w1[q]), acc, 0, 0, 0);
                c1 = n1;
                c2 = n2;
                c3 = n3;
            }
            if (a.prof) { asm volatile("" ::"v"(acc[0])); t1 = __builtin_readcyclecounter(); t_loop += t1 - t0; t0 = t1; }
            // the x buffer of this slot must have been read out by its group's epilogue of the tile before last (the group publishes
            // the next tile's planes BEFORE that epilogue, so this is a real wait - normally long satisfied)
            flag_wait(drained0 + 4 * slot, 2u * (use - 1u));
            float* X = xt + slot * XT + lane_x;
#pragma unroll
            for (int r = 0; r < 16; ++r) X[crow(r) * LDK] = acc[r];
            flag_bump(done0 + 4 * slot, lane);
            if (a.prof) { t1 = __builtin_readcyclecounter(); t_x += t1 - t0; }
        }
        if (a.prof && wave == 0 && lane == 0) {   // the record layout of k_edge_gate_pl (tools/gate_phase_profile.py)
            long long* o = a.prof + (int64_t)blockIdx.x * 8;
            o[0] = t_wait; o[1] = 0; o[2] = t_loop; o[3] = t_x; o[4] = n;
            o[5] = (long long)__builtin_readcyclecounter() - c_begin;
            o[6] = (long long)__builtin_amdgcn_s_memrealtime() - r_begin;
        }
    } else {
        // ------------------------------------------------------------------ load / store wave
        const int group = (wave - 4) >> 1;
        // (tried: s_setprio 2 for these - the younger - waves: no change, 6330 cycles per tile either way; they are not losing issue slots
        //  to the compute waves, they are waiting on the CU's vector-memory queue - see the header)
        const int gl = ((wave - 4) & 1) * 64 + lane;        // lane index inside the group, 0..127
        const int c4f = gl & 63, r0f = gl >> 6;             // fetch mapping: whole rows, rows 16 r0f + p: a wave splits AND post-processes
                                                            // the same sixteen rows, so it may read tile r's planes while its sibling already writes tile r + 2's
        const int c4e = gl & 31, r0e = gl >> 5;             // epilogue mapping: this half's 128 columns, rows 8 r0e + p (CONSECUTIVE rows
                                                            // per lane: the rows are destination-sorted, so B2h[dst] repeats from piece to piece)
        auto erow = [&](int p) { return 8 * r0e + p; };
        f32x4 av[NPF], ek[NPE], g1[NPE], g2[NPE];
        f32x4 dyv[MODE == 3 ? NPF : 1];   // MODE 3: the old C rows (dy), whole rows like av
        auto fetch_rows = [&](int r) {   // the A operand rows (MODE 3: the xe rows and the old C rows) - whole rows
            const int64_t row0 = (int64_t)tile_of(r) * TM;
            const int valid = tile_valid(r);
#pragma unroll
            for (int p = 0; p < NPF; ++p) {   // rows past the end of the list read the last valid row (never stored)
                const int64_t row = row0 + min(16 * r0f + p, valid - 1);
                av[p] = load4_as<X16>(a.e_in, row * lda + 4 * c4f);
                if (MODE == 3) dyv[p] = *reinterpret_cast<const f32x4*>(a.B1h + row * H + 4 * c4f);
            }
        };
        unsigned g2_fresh = 0;        // (wave-uniform) bit p: B2h[dst] was fetched for piece p of the tile whose epilogue comes next
        int si_all = 0, di_all = 0;   // lane l: the endpoints of tile row l % 32 (two loads per wave and tile instead of sixteen)
        auto fetch_index = [&](int r) {
            if (MODE >= 2) return;
            const int64_t row = (int64_t)tile_of(r) * TM + min(lane & 31, tile_valid(r) - 1);
            si_all = a.srt_src[row];
            di_all = a.srt_dst[row];
        };
        auto fetch_side = [&](int r) {   // this half's pieces: gathers / old C rows (the residual comes back out of the planes, see below)
            if (MODE == 4) return;
            const int64_t row0 = (int64_t)tile_of(r) * TM;
            const int valid = tile_valid(r);
            int dprev = -1;
            g2_fresh = 0;
#pragma unroll
            for (int p = 0; p < NPE; ++p) {
                const int64_t row = row0 + min(erow(p), valid - 1);
                if (MODE == 0 && !kResidualFromPlanes) ek[p] = *reinterpret_cast<const f32x4*>(a.e_in + row * H + colh + 4 * c4e);
                if (MODE < 2) {
                    const int lr = min(erow(p), valid - 1);
                    const int sp = __shfl(si_all, lr), dp = __shfl(di_all, lr);
                    g1[p] = *reinterpret_cast<const f32x4*>(a.B1h + (int64_t)sp * a.ldn + colh + 4 * c4e);
                    // B2h[dst]: a run of equal destinations is ~10 rows long, a lane's pieces are consecutive rows - the row is fetched again
                    // only when some lane of the wave needs a new one (a wave-uniform branch: ~2.3 of 8 loads survive on assembly graphs)
                    // (the skipped pieces are filled in from their predecessors in the epilogue, when everything has arrived: a register
                    //  copy HERE would wait for the load just issued and serialise the gathers - measured: 4800 cycles of issue)
                    const bool fresh = p == 0 || dp != dprev;
                    if (__builtin_amdgcn_ballot_w64(fresh) != 0) {
                        g2[p] = *reinterpret_cast<const f32x4*>(a.B2h + (int64_t)dp * a.ldn + colh + 4 * c4e);
                        g2_fresh |= 1u << p;
                    }
                    dprev = dp;
                } else {
                    g1[p] = *reinterpret_cast<const f32x4*>(a.B1h + row * H + colh + 4 * c4e);   // the old rows of C
                }
            }
        };
        unsigned char* S = ring + group * SLOTB;
        const float* Xs = xt + group * XT;
        f32x4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = st1;   // MODE 1: this lane's running shifted sums of its four columns
        long long t_split = 0, t_done = 0, t_epi = 0, t_issue = 0, t0 = 0, t1 = 0;
        float amax3 = 0.f;   // MODE 3: max |dxe| over the elements this lane writes out (round 6: for the fp16x3 weight gradient of B_3)
        float un_pub = 1.f;  // F16: 2^-11 / (the scale of this wave's sixteen rows of the tile it published last)
        // split the rows in av (tile r) into the group's planes slot and publish them; MODE 3 first turns them into A = BatchNorm
        // backward of (dy = the old C rows, x = the xe rows) and writes this workgroup's column half of it out as dxe
        auto split_and_publish = [&](int r) {
            float mrow = 0.f;   // F16: max |.| over this lane's pieces of the wave's sixteen rows
            if (MODE == 3) {
                const f32x4 ka = *reinterpret_cast<const f32x4*>(norm_lds + 4 * c4f), k1 = *reinterpret_cast<const f32x4*>(norm_lds + H + 4 * c4f);
                const f32x4 k2 = *reinterpret_cast<const f32x4*>(norm_lds + 2 * H + 4 * c4f), km = *reinterpret_cast<const f32x4*>(norm_lds + 3 * H + 4 * c4f);
                const f32x4 kr = *reinterpret_cast<const f32x4*>(norm_lds + 4 * H + 4 * c4f), ks = *reinterpret_cast<const f32x4*>(norm_lds + 5 * H + 4 * c4f);
                const f32x4 kh = *reinterpret_cast<const f32x4*>(norm_lds + 6 * H + 4 * c4f);
                const int valid3 = tile_valid(r);
                const int64_t once3 = a.bnb.n_once - (int64_t)tile_of(r) * TM;   // rows of this tile that get the mean terms
                const int64_t aoff = (int64_t)tile_of(r) * TM * H + 4 * c4f;
                const bool mine = (c4f >> 5) == hh;
#pragma unroll
                for (int p = 0; p < NPF; ++p) {
                    const int row = 16 * r0f + p;
                    const float on = row < once3 ? 1.f : 0.f;
                    f32x4 t;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float gm = (av[p][j] * ks[j] + kh[j] > 0.f) ? dyv[p][j] : 0.f;
                        t[j] = ka[j] * (gm - on * (k1[j] + (av[p][j] - km[j]) * kr[j] * k2[j]));
                    }
                    av[p] = t;
                    if (F16) mrow = fmaxf(fmaxf(mrow, fmaxf(fabsf(t[0]), fabsf(t[1]))), fmaxf(fabsf(t[2]), fabsf(t[3])));
                    if (mine && row < valid3) {
                        // (the offset is made opaque so that the sixteen row addresses are formed here, one at a time: hoisted out of the loop they were
                        // spilled, and every reload came with an s_waitcnt vmcnt(0) in front of its store - the stores of a tile went out one by one)
                        int roff = row * H;
                        asm volatile("" : "+v"(roff));
                        store4_as<X16>(a.bnb.a_out, aoff + (int64_t)roff, t);
                        amax3 = fmaxf(fmaxf(amax3, fmaxf(fabsf(t[0]), fabsf(t[1]))), fmaxf(fabsf(t[2]), fabsf(t[3])));
                    }
                }
            }
            float row_scale = 1.f;
            if (F16) {
                float m = mrow;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
                const unsigned bits = __float_as_uint(m);
                const int ex = (int)((bits >> 23) & 0xFFu) - 127;
                int k = 0;
                if (bits != 0u && ex < 128) k = max(-100, min(100, 3 - ex));   // (all zero: any scale; inf / NaN: scale 1 and NaN rows, as it should be)
                row_scale = __uint_as_float((unsigned)(k + 127) << 23);
                un_pub = __uint_as_float((unsigned)(127 - k - 11) << 23);
            }
#pragma unroll
            for (int p = 0; p < NPF; ++p) {
                uint2 p1, p2, p3;
                if (F16) split4h3(av[p] * row_scale, p1, p2, p3);
                else split4(av[p], p1, p2, p3);
                unsigned char* d = S + (16 * r0f + p) * PLD + 8 * c4f;
                *reinterpret_cast<uint2*>(d) = p1;
                *reinterpret_cast<uint2*>(d + PLANE) = p2;
                *reinterpret_cast<uint2*>(d + 2 * PLANE) = p3;
            }
            flag_bump(full0 + 4 * group, lane);
        };
        // Software pipeline over this group's tiles r, r + 2, ...: when the compute waves are done with tile r, the group FIRST splits
        // and publishes tile r + 2 (its rows were requested a whole period ago) and only THEN runs tile r's epilogue, so the compute
        // waves' next-but-one tile is ready ~2600 cycles after `done`, not after epilogue + fetch latency + split (measured: 8400).
        // With two groups the chain done(r) -> full(r + 2) has one tile's matrix time (~3600 cycles) to hide in.
        if (group < n) {
            fetch_index(group);
            fetch_rows(group);
            fetch_side(group);
            split_and_publish(group);
            if (group + 2 < n) {
                fetch_index(group + 2);   // (after fetch_side(group): it consumed the previous indices as addresses)
                fetch_rows(group + 2);
            }
        }
        for (int r = group; r < n; r += 2) {
            const unsigned use = (unsigned)(r >> 1) + 1u;
            if (a.prof) t0 = __builtin_readcyclecounter();
            flag_wait(done0 + 4 * group, 4u * use);   // x(r) is ready; the planes slot is free (all four compute waves have read it)
            if (a.prof) { t1 = __builtin_readcyclecounter(); t_done += t1 - t0; t0 = t1; }
            if (MODE == 0 && kResidualFromPlanes) {
                // the residual e[row, this half] is NOT fetched a second time: the planes of tile r are still in the slot, and the
                // three-way split is exact - x = (x1 + x2) + x3 bit for bit (x1 + x2 = x with its low mantissa bits cleared) - so
                // 24 ds_read_b64 + 20 VALU operations per piece replace 16 KB per tile through the vector-memory queue
#pragma unroll
                for (int p = 0; p < NPE; ++p) {
                    const unsigned char* q = S + erow(p) * PLD + 2 * (colh + 4 * c4e);
                    const uint2 u1 = *reinterpret_cast<const uint2*>(q), u2 = *reinterpret_cast<const uint2*>(q + PLANE),
                                u3 = *reinterpret_cast<const uint2*>(q + 2 * PLANE);
                    auto lo = [](unsigned v) { return __uint_as_float(v << 16); };
                    auto hi = [](unsigned v) { return __uint_as_float(v & 0xFFFF0000u); };
                    ek[p] = f32x4{(lo(u1.x) + lo(u2.x)) + lo(u3.x), (hi(u1.x) + hi(u2.x)) + hi(u3.x), (lo(u1.y) + lo(u2.y)) + lo(u3.y),
                                  (hi(u1.y) + hi(u2.y)) + hi(u3.y)};
                }
            }
            const float un_use = un_pub;   // F16: tile r's factor (this wave split its sixteen rows of tile r - the rows its epilogue lanes own)
            if (r + 2 < n) split_and_publish(r + 2);
            if (a.prof) { t1 = __builtin_readcyclecounter(); t_split += t1 - t0; t0 = t1; }
            const int valid = tile_valid(r);
            const f32x4 sc4 = (MODE < 2 || MODE == 4) ? *reinterpret_cast<const f32x4*>(norm_lds + 4 * c4e) : f32x4{0.f, 0.f, 0.f, 0.f};   // MODE 4: the bias
            const f32x4 sh4 = MODE == 0 ? *reinterpret_cast<const f32x4*>(norm_lds + HC + 4 * c4e) : f32x4{0.f, 0.f, 0.f, 0.f};
            float* out = a.e_out + (int64_t)tile_of(r) * TM * ldo + colh + 4 * c4e;
            // Every operand that came through the vector-memory queue is awaited HERE, before the first store: loads and stores share
            // one in-order counter, and a wait that the compiler places between (conditional) stores becomes s_waitcnt vmcnt(0) - the
            // wave then sits out the HBM write latency of its own stores (measured: 500 cycles per piece; ISA: vmcnt(0) before piece 7)
            if (MODE != 4) {
                if (MODE < 2) {
#pragma unroll
                    for (int p = 0; p < NPE; ++p) asm volatile("" : "+v"(g2[p]));
                    const unsigned fresh_r = __builtin_amdgcn_readfirstlane(g2_fresh);
#pragma unroll
                    for (int p = 1; p < NPE; ++p)
                        if (!((fresh_r >> p) & 1u)) g2[p] = g2[p - 1];   // same destination as the row above: the same B2h row
                }
#pragma unroll
                for (int p = 0; p < NPE; ++p) {
                    if (MODE < 2) g1[p] += g2[p];   // G = B1h[src] + B2h[dst]
                    asm volatile("" : "+v"(g1[p]));
                    if (MODE == 0) asm volatile("" : "+v"(ek[p]));
                }
            }
            // whole tiles (all but the last of an edge list) store unconditionally: no branch per piece, the store count is static
            auto pieces = [&](auto full_tile) {
                constexpr bool FULL = decltype(full_tile)::value;
#pragma unroll
                for (int pb = 0; pb < NPE; pb += 4) {
                    f32x4 x[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) x[u] = *reinterpret_cast<const f32x4*>(Xs + erow(pb + u) * LDK + 4 * c4e);
                    asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int p = pb + u, row = erow(p);
                        f32x4 y;
                        if (MODE == 0) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) y[j] = fmaxf((x[u][j] + g1[p][j]) * sc4[j] + sh4[j], 0.f) + ek[p][j];
                        } else if (MODE == 4) {
                            y = x[u] + sc4;
                        } else if (F16) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) y[j] = __builtin_fmaf(x[u][j], un_use, g1[p][j]);
                        } else {
                            y = x[u] + g1[p];
                        }
                        if (FULL || row < valid) {
                            if (MODE == 1) {
                                const f32x4 dlt = y - sc4;
                                st1 += dlt;
                                st2 += dlt * dlt;
                            }
                            *reinterpret_cast<f32x4*>(out + (int64_t)row * ldo) = y;
                        }
                    }
                }
            };
            if (valid == TM)
                pieces(std::true_type{});
            else
                pieces(std::false_type{});
            flag_bump(drained0 + 4 * group, lane);   // x(r) has been read: the compute waves may write x(r + 2) over it
            if (a.prof) { t1 = __builtin_readcyclecounter(); t_epi += t1 - t0; t0 = t1; }
            // requests for the coming tiles, issued behind the epilogue's stores (one in-order vector-memory queue per CU): tile r + 2's
            // gathers / residual (consumed by its epilogue, a period from now) and tile r + 4's rows (consumed by its split, a period from now)
            if (r + 2 < n) fetch_side(r + 2);
            if (r + 4 < n) {
                fetch_index(r + 4);
                fetch_rows(r + 4);
            }
            if (a.prof) { t1 = __builtin_readcyclecounter(); t_issue += t1 - t0; }
        }
        if (MODE == 3 && a.bnb.amax_bits != nullptr) wave_amax_to(a.bnb.amax_bits, amax3);   // (one atomicMax per wave, only when larger)
        if (a.prof && wave == 4 && lane == 0) {   // the first load wave's phases, after the 256 compute-wave records
            long long* o = a.prof + (int64_t)(256 + blockIdx.x) * 8;
            o[0] = 0; o[1] = t_split; o[2] = t_done; o[3] = t_epi; o[4] = (n + 1) / 2; o[5] = t_issue;
        }
        if (MODE == 1 && a.stats != nullptr) {
            // lanes l and l + 32 hold different rows of the same four columns: fold them, then every load wave leaves one row of partial
            // sums for its 128 columns
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                st1[j] += __shfl_xor(st1[j], 32);
                st2[j] += __shfl_xor(st2[j], 32);
            }
            if (lane < 32) {   // one row per load wave in each of two [rows][H] matrices: sums and sums of squares of (x - centre), this half's columns, the rest zero
                const int64_t srow = (int64_t)blockIdx.x * 4 + (wave - 4), srows = (int64_t)gridDim.x * 4;
                *reinterpret_cast<f32x4*>(a.stats + srow * H + colh + 4 * c4e) = st1;              // [0][row][H]: sums
                *reinterpret_cast<f32x4*>(a.stats + (srows + srow) * H + colh + 4 * c4e) = st2;    // [1][row][H]: sums of squares
            }
        }
    }
}

int grid_pl256() {
    int g = persistent_grid();
    g -= g % 16;   // pairs of workgroups per XCD
    return g < 16 ? 16 : g;
}

template <int MODE, int PROBE = 0, bool X16 = false, bool F16 = false>
int launch_pl256(const GateBfArgs& args, hipStream_t s) {
    GateBfArgs a = args;
    const int64_t tiles = (a.E + 31) / 32;
    GN_REQUIRE(tiles < (1ll << 31), "edge_gate: too many tiles");
    GN_REQUIRE(a.e_out != a.e_in, "edge_gate (H = 256): the output must not alias the input rows");
    GN_REQUIRE(MODE != 4 || (a.num_cblocks >= 1 && a.num_cblocks <= grid_pl256() / kXcds && a.ldn >= 256 && a.ldn % 4 == 0 && a.ld_out % 4 == 0),
               "linear (K = 256): %d column blocks / strides %d, %d", a.num_cblocks, a.ldn, a.ld_out);
    GN_REQUIRE(MODE != 3 || a.e_out != a.B1h, "bn_bwd_dgrad (H = 256): C_out must not alias C_in (two workgroups read whole rows of it)");
    a.num_tiles = (int)tiles;
    a.prof = gate_profile_buffer();
    if (MODE == 1 && a.stats != nullptr) GN_HIP(hipMemsetAsync(a.stats, 0, sizeof(float) * (size_t)grid_pl256() * 4 * 2 * 256, s));   // idle waves / the other half
    if (PROBE == 0 && (MODE == 0 || MODE == 1 || MODE == 4) && tuning(kTuneArith) == 0)   // the shipped default: fp16x3 + LDS-DMA (edge_tile_f16.hip)
        return gate_f16_launch(MODE, a, grid_pl256(), s);
    hipLaunchKernelGGL((k_edge_gate_pl256<MODE, PROBE, X16, F16>), dim3(grid_pl256()), dim3(512), 0, s, a);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

}  // namespace

// mode 0: the gate; mode 1: the raw gate (+ shifted column sums into a.stats[2][gate_pl256_stats_rows()][256] when given, a.scale = centres);
// mode 2: C += A W^T (a.e_in = A, a.e_out = a.B1h = C); mode 3: C_out = C_in + BatchNormBackward(C_in, X) W^T with dxe written out
// (a.e_in = X, a.B1h = C_in, a.e_out = C_out != C_in, a.bnb)
int gate_pl256_stats_rows() { return grid_pl256() * 4; }
int gate_pl256_launch(int mode, const GateBfArgs& a, hipStream_t s, bool x16) {
    if (x16) {   // bf16 storage of xe / dxe at H = 256 (round 4)
        if (mode == 3) return tuning(kTuneArith) == 0 && tuning(kTuneGateExperiment) != 80 ? launch_pl256<3, 0, true, true>(a, s) : launch_pl256<3, 0, true>(a, s);
        GN_REQUIRE(mode == 1 && tuning(kTuneArith) == 0, "edge-tile kernel (H = 256): bf16 storage exists for modes 1 (fp16x3 kernel) and 3");
        GateBfArgs b = a;
        b.num_tiles = (int)((a.E + 31) / 32);
        b.prof = gate_profile_buffer();
        if (b.stats != nullptr) GN_HIP(hipMemsetAsync(b.stats, 0, sizeof(float) * (size_t)grid_pl256() * 4 * 2 * 256, s));
        return gate_f16_launch(1, b, grid_pl256(), s, true);
    }
    if (mode == 0 && tuning(kTuneGateAblation) == 8) return launch_pl256<0, 1>(a, s);   // measurement only: the kernel without its MFMAs
    if (mode == 0) return launch_pl256<0>(a, s);
    if (mode == 1) return launch_pl256<1>(a, s);
    if (mode == 2) return launch_pl256<2>(a, s);
    if (mode == 3)   // fp16x3 in one accumulator (round 6); gnnome_set_tuning(10, 1) or (4, 80): bf16x6
        return tuning(kTuneArith) == 0 && tuning(kTuneGateExperiment) != 80 ? launch_pl256<3, 0, false, true>(a, s) : launch_pl256<3>(a, s);
    if (mode == 4) return launch_pl256<4>(a, s);
    set_error("edge-tile kernel (H = 256): mode %d is not built", mode);
    return GNNOME_EINVAL;
}

}  // namespace gnnome
