// Node projection, round 6: P[N, Nout] = h[N, K] W^T + bias as ONE fp16x3 GEMM whose operands never meet in LDS in fp32.
//
// What it stands in for: the five nn.Linear calls on node rows, gated_gcn_full.py:91-96 (A_1, A_2, A_3, B_1, B_2 over the row-concatenated
// weights, Nout = 5H) and the node halves of predictor.W1 (score_predictor.py:13-14, Nout = 2 hs).  Arithmetic: fp16x3 exactly as
// edge_tile_f16.hip's header derives it (two fp16 planes per fp32 operand, three of the four plane products, the two small ones in a
// second fp32 accumulator folded in once per output; |x| < 65504 or the row leaves as NaN - loud, never silently wrong).
//
// Why a new kernel: the plane-form edge-tile kernel in mode 4 (edge_gate_bf.hip, K = 128: 87 us at N = 1e5 for
// 307 MB = 0.44 of HBM) and edge_tile_f16.hip<4> (K = 256: 0.66 ms at N = 253k for 1.55 GB = 0.29) keep a 128-column block of W in the
// compute waves' registers and bring h tiles in through a ring, 32 rows at a time, with load, compute and store waves handing every
// tile over through LDS counters: five (ten) workgroups per XCD re-read and re-split every h row, and a tile's latency is the sum of
// three hand-overs.  Here:
//   * h is the STATIONARY operand: a wave loads its own 32 rows straight into MFMA fragment order (two 16-byte loads per k step),
//     splits them into the two fp16 planes in registers ONCE (K = 128: 64 registers, K = 256: 128) and keeps them for every output column;
//   * W is split into planes ONCE PER WEIGHT MATRIX by k_weight_planes (gnnome_weight_planes_f16) into fragment order
//     [column block of 32][k chunk of 128][k step][plane][lane][8 halves]; a GRANULE of 32 KB (K = 256: one column block, K = 128: two) is
//     thirty-two 1 KB pieces that global_load_lds_dwordx4 copies verbatim from L2 into one of two LDS slots - no registers, no VALU, no bank
//     conflicts on the way out (one ds_read_b128 per plane and k step) - while the other slot is being multiplied; ONE barrier per granule;
//   * the MFMA takes W as its A operand and h as its B operand, so a lane ends up with FOUR CONSECUTIVE output columns of one node row
//     per accumulator quad; a 4 x 4 transpose among the four lanes of a quad (DPP) and a permuted column order inside the block (frag_col)
//     turn that into 64 contiguous bytes per quad and store: full 128-byte lines leave from the accumulators, no LDS transpose, no store waves
//     (32-byte pieces, the untransposed layout, measured 83 against 66 us for the stores alone at N = 1e5);
//   * PERSISTENT, BALANCED workgroups: two per CU, each takes an equal share of the (row tile, granule) units as one contiguous run - it may
//     start and end inside a row tile, whose h rows the neighbour loads again.  (One workgroup per row tile: 782 tiles on 512 slots are two
//     rounds of 38 us each at N = 1e5, the second half empty.)
// Bound: HBM write of 4 Nout bytes per row; the matrix cores need 3 x 2 K Nout flop per row = 0.4 (K = 128) to 1.0 (K = 256) of that time
// at the clocks the chip holds under MFMA load.  Every byte of W a workgroup multiplies by comes through the CU's vector memory path
// (K / 128 bytes per output byte at 128 rows per workgroup), which is what keeps K = 256 above its HBM time.
//
// vmcnt accounting.  gfx950 has ONE vector-memory counter for loads, stores and LDS-DMA, decremented in issue order.  In iteration n a wave
// issues its 8 pieces of granule n + 1 (between the MFMAs) and then 4 row-piece stores per column block, ALWAYS 4 (rows past M are clamped on
// the way in and store the clamped row's own bits again): what is younger than granule n's DMA when it is needed is exactly the stores of
// iteration n - 1, so `s_waitcnt vmcnt(4 * blocks per granule)` waits for the granule and for nothing issued after it.
#include <cstdlib>

#include "common.h"

#pragma clang diagnostic ignored "-Winline-asm"

namespace gnnome {
namespace {

typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr float kLoScale = 2048.f, kLoInv = 1.0f / 2048.f;

__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p; }

// 1 KB of LDS-DMA: lane l's 16 bytes at src + voff (voff = 16 l) land at lds + 16 l.  No "memory" clobber: between the two barriers that bracket
// it nothing reads the slot it fills, and with the clobber hipcc may not move the NEXT k steps' ds_reads above it - it then waits out one LDS
// latency per k step (measured: 224 cycles per step against 96 of MFMA).  asm volatile keeps it ordered with the barriers and the waits.
__device__ __forceinline__ void dma_piece(const void* src, unsigned voff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(lds) : "m0");
}

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// 16 bytes from p + OFF into a register that the compiler must not touch before one of the waits below has named it
template <int OFF>
__device__ __forceinline__ void load16_at(f32x4& dst, const float* p) {
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(p), "n"(OFF) : "memory");
}
__device__ __forceinline__ void load4_raw(float& dst, const float* p) { asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(p) : "memory"); }
// s_waitcnt vmcnt(0) that the eight named registers' readers are ordered behind
__device__ __forceinline__ void landed8(f32x4* r, bool wait) {
    if (wait)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])::"memory");
    else
        asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])::"memory");
}

// lane l <- lane l ^ 1 / l ^ 2 (DPP quad_perm [1,0,3,2] / [2,3,0,1])
__device__ __forceinline__ float quad_xor1(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ float quad_xor2(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true)); }
// E[q] (four floats each) of the four lanes of a quad, transposed: afterwards lane a's E[r] is what lane r's E[a] was.  Per exchanged pair
// two selects whose one source is the partner lane's register (the partner of an upper lane is a lower lane and sends its Y, and vice versa).
__device__ __forceinline__ void quad_transpose(f32x4& E0, f32x4& E1, f32x4& E2, f32x4& E3, int lane) {
    const bool hi2 = lane & 2, hi1 = lane & 1;
#define GN_SWAP(X, Y, COND, XOR)                       \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) {    \
        const float x = X[t], y = Y[t];                \
        const float py = XOR(y), px = XOR(x);          \
        X[t] = COND ? py : x;                          \
        Y[t] = COND ? y : px;                          \
    }
    GN_SWAP(E0, E2, hi2, quad_xor2) GN_SWAP(E1, E3, hi2, quad_xor2)
    GN_SWAP(E0, E1, hi1, quad_xor1) GN_SWAP(E2, E3, hi1, quad_xor1)
#undef GN_SWAP
}

// the two fp16 planes of eight floats: p1 = RN16(x), p2 = RN16((x - p1) * 2048)   (x - p1 is exact in fp32)
__device__ __forceinline__ void split8(const f32x4 lo, const f32x4 hi, h8_t& p1, h8_t& p2) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x2 v = j < 2 ? f32x2{lo[2 * j], lo[2 * j + 1]} : f32x2{hi[2 * j - 4], hi[2 * j - 3]};
        const h2_t a = __builtin_convertvector(v, h2_t);
        const f32x2 big = v * kLoScale;
        const f32x2 r = {__builtin_fmaf((float)a[0], -kLoScale, big[0]), __builtin_fmaf((float)a[1], -kLoScale, big[1])};
        const h2_t b = __builtin_convertvector(r, h2_t);
        p1[2 * j] = a[0];
        p1[2 * j + 1] = a[1];
        p2[2 * j] = b[0];
        p2[2 * j + 1] = b[1];
    }
}

}  // namespace (the helpers above stay private to this file; the kernels carry plain gnnome:: names for the profilers' tables)

// Which output column of its 32-column block the MFMA's row index i = 8 a + 4 h + t stands for: 16 h + 4 a + t.  With W as the A operand, lane
// (j, h) of the result holds i = 8 q + 4 h + t in accumulator 4 q + t; after the 4 x 4 transpose among the four lanes of a quad (epilogue) lane
// (4 g + a, h) holds i = 8 a + 4 h + t of node row 4 g + r in slot r - and with this numbering that is columns 16 h + 4 a .. + 3: the four lanes
// of a quad write 64 contiguous bytes of one row, the two wave halves the two halves of a 128-byte line.
__host__ __device__ constexpr int frag_col(int i) { return 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3); }

// W[Nout, K] (row stride ldw) -> planes in fragment order: [cb = col / 32][s = k / 16][plane][lane][8], where lane = i + 32 * ((k % 16) / 8) holds
// k % 8 = 0 .. 7 of column 32 cb + frag_col(i) (for K = 128 / 256 that is [cb][kc = k / 128][(k % 128) / 16], the 16 KB chunks of the header).  One thread
// per (cb, s, lane).
__global__ __launch_bounds__(256) void k_weight_planes(const float* __restrict__ W, int ldw, int Nout, int K, uint4* __restrict__ planes) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int lane = t & 63, ksteps = K / 16, s = (t >> 6) % ksteps, cb = (t >> 6) / ksteps;
    if (cb * 32 >= Nout) return;
    const int col = 32 * cb + frag_col(lane & 31), k = 16 * s + 8 * (lane >> 5);
    const float* w = W + (int64_t)col * ldw + k;
    h8_t p1, p2;
    split8(*reinterpret_cast<const f32x4*>(w), *reinterpret_cast<const f32x4*>(w + 4), p1, p2);
    uint4* dst = planes + ((int64_t)cb * ksteps + s) * 128 + lane;
    dst[0] = __builtin_bit_cast(uint4, p1);
    dst[64] = __builtin_bit_cast(uint4, p2);
}

constexpr int kMaxNout = 1536;
constexpr int kGranuleBytes = 32768;   // (K = 64: 16 KB - two column blocks, as at K = 128)

// K in {64, 128, 256}; 4 waves; two granule slots.  PROBE (measurement only; gnnome_set_tuning(2, 20) + gnnome_set_tuning(1, mask)): 1 no stores, 2 no MFMAs,
// 4 no DMA inside the loop, 8 no quad transpose (all four: wrong results), 16 (right results) no prefetch of the next row tile; 32 (right results): wave 0's cycle counters to gnnome_debug_gate_profile's
// buffer ([workgroups][8] int64)
template <int K, int PROBE = 0>
__global__ __launch_bounds__(256, 2) void k_node_project(const float* __restrict__ A, int64_t M, int lda, const unsigned char* __restrict__ planes,
                                                          const float* __restrict__ bias, int Nout, float* __restrict__ C, int ldc, int64_t total_units,
                                                          long long* prof, int xp) {
    static_assert(K == 64 || K == 128 || K == 256, "K is 64, 128 or 256");
    constexpr int KS = K / 16, GB = K == 64 ? kGranuleBytes / 2 : kGranuleBytes, SPG = GB / 2048, BPG = SPG / KS;   // k steps per column block; granule
    // bytes; k steps per granule (2 KB each: two planes x 64 lanes x 16 bytes); column blocks per granule (2, 2, 1)
    constexpr int NW = 4, NP = GB / 1024 / NW, NT = 64 * NW, TM = 32 * NW;   // waves; DMA pieces per wave and granule; threads; rows per tile
    constexpr int NSTORE = (PROBE & 1) ? 0 : 4 * BPG;   // vector-memory operations a wave issues per iteration after its DMA pieces
    __shared__ __attribute__((aligned(1024))) unsigned char ring[2 * kGranuleBytes];
    __shared__ __attribute__((aligned(16))) float bias_lds[kMaxNout];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    long long t_start = 0, t_wait = 0, t_load = 0, r_start = 0;   // PROBE 32
    if (PROBE & 32) { t_start = __builtin_readcyclecounter(); r_start = __builtin_amdgcn_s_memrealtime(); }
    // this workgroup's run of units: unit u = (row tile u / gpt, granule u % gpt); an XCD's workgroups take neighbouring runs
    const int gpt = Nout / (32 * BPG);
    // A CU's two workgroups (blocks p and p + gridDim / 2) share one contiguous run of units, cut at 55 %: the block dispatched first is the older
    // one and wins the arbitration for every pipe the two share - cut in halves, the first ones finish at 48 us and the second ones at 62 (N = 1e5).
    // xp (experiment knob, gnnome_set_tuning key 4): another percentage (50: plain equal runs).  Measured level: s_setprio 1 for the second workgroup,
    // throughout or in every second iteration; one 8-wave workgroup per CU (256 rows per tile: its waves meet at the barrier, 1400 cycles per granule).
    const int share = (gridDim.x & 1) == 0 ? (xp > 0 && xp < 100 ? xp : 55) : 50;
    const int half = gridDim.x / 2;
    const bool second = (gridDim.x & 1) == 0 && (int)blockIdx.x >= half;
    int64_t u0, u1;
    if (share == 50) {
        const int64_t w = xcd_remap(blockIdx.x, gridDim.x);
        u0 = w * total_units / gridDim.x;
        u1 = (w + 1) * total_units / gridDim.x;
    } else {   // pair p = the two workgroups of one CU (blocks p and p + half): a contiguous run of units, cut at `share` percent
        const int64_t pr = xcd_remap(blockIdx.x % half, half);
        const int64_t p0 = pr * total_units / half, p1 = (pr + 1) * total_units / half, cut = p0 + (p1 - p0) * share / 100;
        u0 = second ? cut : p0;
        u1 = second ? p1 : cut;
    }
    const int count = (int)(u1 - u0);
    if (count <= 0) return;
    const unsigned ring0 = lds_addr(ring), voff = 16u * lane;
    const unsigned char* mine = planes + (NP * wave) * 1024;   // this wave's pieces of every granule

    int g = (int)(u0 % gpt);          // granule of the current unit
    int64_t tile = u0 / gpt;
#pragma unroll
    for (int p = 0; p < NP; ++p) dma_piece(mine + (int64_t)g * GB + p * 1024, voff, ring0 + (NP * wave + p) * 1024);
    {
        float bv[kMaxNout / NT];
#pragma unroll
        for (int i = 0; i < kMaxNout / NT; ++i) {
            bv[i] = 0.f;
            if (bias != nullptr && NT * i < Nout) load4_raw(bv[i], bias + min(tid + NT * i, Nout - 1));   // uniform condition
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < kMaxNout / NT; ++i) {
            asm volatile("" : "+v"(bv[i]));
            if (tid + NT * i < Nout) bias_lds[tid + NT * i] = bv[i];
        }
    }

    h8_t a1[KS], a2[KS];
    float* crow[4];
    constexpr bool PREFETCH_ROWS = K <= 128 && !(PROBE & 16);
    f32x4 raw[2 * KS];
    bool ahead = false;   // raw[] holds (requests for) the rows of the tile that starts with the next iteration
    auto request_rows = [&](f32x4 (&dst)[2 * KS], int64_t t) {
        int64_t row = t * TM + 32 * wave + (lane & 31);
        if (row >= M) row = M - 1;   // clamped: the lane computes and stores row M - 1's own bits once more
        const float* arow = A + row * lda + 8 * (lane >> 5);
#define GN_LD(S)                                                         \
    if constexpr (S < KS) {                                              \
        load16_at<64 * S>(dst[2 * (S < KS ? S : 0)], arow);              \
        load16_at<64 * S + 16>(dst[2 * (S < KS ? S : 0) + 1], arow);     \
    }
        GN_LD(0) GN_LD(1) GN_LD(2) GN_LD(3) GN_LD(4) GN_LD(5) GN_LD(6) GN_LD(7)
        GN_LD(8) GN_LD(9) GN_LD(10) GN_LD(11) GN_LD(12) GN_LD(13) GN_LD(14) GN_LD(15)
#undef GN_LD
        static_assert(KS <= 16, "offsets up to 1008 bytes");
    };
    bool fresh = true;   // the unit starts a row tile (for this workgroup): its h rows have to come in
    for (int n = 0; n < count; ++n) {
        long long t_in = 0;
        if (PROBE & 32) t_in = __builtin_readcyclecounter();
        if (fresh) {
            // this wave's 32 rows of h in fragment order (lane (j, h): row j, floats 16 s + 8 h .. + 7 of k step s), through inline assembly:
            // hipcc moves plain loads from read-only arguments across asm statements (it put them BEHIND a wait, two latencies in a row)
            const int64_t row0 = tile * TM + 32 * wave;
            if (!ahead) {
                request_rows(raw, tile);
                landed8(raw, true);   // everything issued so far has landed: these rows, this wave's pieces of granule n, older stores
            } else {
                wait_vm<NSTORE>();    // requested an iteration ago, BEFORE the pieces of granule n: what landed those landed these
                landed8(raw, false);
            }
#pragma unroll
            for (int q = 1; q < KS / 4; ++q) landed8(raw + 8 * q, false);
#pragma unroll
            for (int s = 0; s < KS; ++s) split8(raw[2 * s], raw[2 * s + 1], a1[s], a2[s]);
            ahead = false;
            // where this lane's four 16-byte pieces of a column block go (see frag_col): rows 4 g + r of the wave's 32, columns 16 h + 4 a .. + 3
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int64_t rr = row0 + (lane & 28) + r;
                if (rr >= M) rr = M - 1;   // the lanes that computed row M - 1 again store it again
                crow[r] = C + rr * ldc + 16 * (lane >> 5) + 4 * (lane & 3);
            }
            if (PROBE & 32) t_load += __builtin_readcyclecounter() - t_in;
        } else {
            wait_vm<NSTORE>();   // granule n is in (this wave's pieces); the stores of iteration n - 1 may still be on their way
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave's pieces of granule n are in; every wave is through with granule n - 1
        if ((PROBE & 32) && !fresh) t_wait += __builtin_readcyclecounter() - t_in;

        // the next unit, and its granule on the way into the other slot
        int g_next = g + 1;
        const bool wrap = g_next == gpt;
        if (wrap) g_next = 0;
        // K = 128 (64 registers to spare): the NEXT row tile's rows are requested one iteration ahead - first thing in the iteration, so that in the
        // counter's order they are older than the next granule's pieces and cost no wait of their own.  (A wait for loads is a wait for every older
        // store as well: the unprefetched form stands 7500 cycles per tile, 19 % of a workgroup's life at N = 1e5.)
        if (PREFETCH_ROWS && wrap && n + 1 < count) {
            request_rows(raw, tile + 1);
            ahead = true;
        }
        // (the last iteration fetches a granule nobody reads, into the slot nobody reads any more: unconditional requests keep the k steps one
        // basic block, and the accounting uniform)
        const unsigned next_slot = ring0 + ((n + 1) & 1) * GB + (NP * wave) * 1024;
        const unsigned char* next_src = mine + (int64_t)g_next * GB;
        const unsigned char* slot = ring + (n & 1) * GB + 16 * lane;
        // W fragments two k steps ahead of the MFMAs that take them (hipcc by itself requests a step's pair only after the previous step's MFMAs)
        constexpr int AHEAD = 2;
        h8_t wq1[16], wq2[16];
#pragma unroll
        for (int st = 0; st < AHEAD; ++st) {
            wq1[st] = *reinterpret_cast<const h8_t*>(slot + (2 * st) * 1024);
            wq2[st] = *reinterpret_cast<const h8_t*>(slot + (2 * st + 1) * 1024);
        }
#pragma unroll
        for (int b = 0; b < BPG; ++b) {
            f32x16 accM, accC;
#pragma unroll
            for (int r = 0; r < 16; ++r) accM[r] = accC[r] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int step = b * KS + s;   // 0 .. SPG - 1 inside the granule
                // this wave's pieces of the next granule, all of them BEFORE this iteration's first stores (the accounting above): spread over
                // the k steps of the granule's first column block
                constexpr int EVERY = KS / NP;
                if (!(PROBE & 4) && step % EVERY == 0 && step / EVERY < NP) dma_piece(next_src + (step / EVERY) * 1024, voff, next_slot + (step / EVERY) * 1024);
                if (step + AHEAD < SPG) {
                    wq1[step + AHEAD] = *reinterpret_cast<const h8_t*>(slot + (2 * (step + AHEAD)) * 1024);
                    wq2[step + AHEAD] = *reinterpret_cast<const h8_t*>(slot + (2 * (step + AHEAD) + 1) * 1024);
                }
                const h8_t w1 = wq1[step], w2 = wq2[step];
                if (!(PROBE & 2)) {
                    accM = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, a1[s], accM, 0, 0, 0);
                    accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, a2[s], accC, 0, 0, 0);
                    accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2, a1[s], accC, 0, 0, 0);
                } else {
                    accM[s % 16] += (float)w1[0] + (float)w2[0] + (float)a1[s][0] + (float)a2[s][1];
                }
                __builtin_amdgcn_sched_barrier(0);   // (keeps the requests above where they are: the scheduler sinks them to their first use otherwise)
            }
            // lane (j, h): accumulator 4 q + t = node row j, MFMA row 8 q + 4 h + t; transposed within the quad it becomes four row pieces of 16 bytes
            const int cb = g * BPG + b;
            f32x4 E[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int t = 0; t < 4; ++t) E[q][t] = __builtin_fmaf(accC[4 * q + t], kLoInv, accM[4 * q + t]);
            if (!(PROBE & 8)) quad_transpose(E[0], E[1], E[2], E[3], lane);
            const f32x4 bq = *reinterpret_cast<const f32x4*>(bias_lds + 32 * cb + 16 * (lane >> 5) + 4 * (lane & 3));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const f32x4 y = E[r] + bq;
                if (!(PROBE & 1)) *reinterpret_cast<f32x4*>(crow[r] + 32 * cb) = y;
                else if (y[0] == 123.456f) crow[r][0] = y[1];
            }
        }
        g = g_next;
        fresh = wrap;
        if (wrap) ++tile;
    }
    if ((PROBE & 32) && prof != nullptr && tid == 0) {
        const long long t_end = __builtin_readcyclecounter();
        long long* p = prof + 8ll * blockIdx.x;
        p[0] = t_load;                     // row tiles coming in: requests, landing (everything older as well), split
        p[1] = t_end - t_start;            // the workgroup's life
        p[2] = t_wait;                     // waiting for a granule + the barrier (iterations that did not load rows)
        p[3] = r_start;                    // 100 MHz clock at the start
        p[4] = __builtin_amdgcn_s_memrealtime();
        p[5] = count;
    }
    wait_vm<0>();   // nothing of this workgroup's is on its way into LDS when the slots are handed on
}

template <int K, int PROBE>
static int launch_project(const float* A, int64_t M, int lda, const void* planes, const float* bias, int Nout, float* C, int ldc, hipStream_t s) {
    constexpr int BPG = K == 256 ? 1 : 2;
    const int64_t tiles = (M + 127) / 128, units = tiles * (Nout / (32 * BPG));
    const int64_t slots = 2 * persistent_grid();   // workgroups that are resident together: two per CU
    const int grid = (int)(units < slots ? units : slots);
    hipLaunchKernelGGL((k_node_project<K, PROBE>), dim3(grid), dim3(256), 0, s, A, M, lda, (const unsigned char*)planes, bias, Nout, C, ldc, units,
                       (PROBE & 32) ? gate_profile_buffer() : nullptr, tuning(kTuneGateExperiment));
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

bool project_supported(int K, int Nout) { return (K == 64 || K == 128 || K == 256) && Nout > 0 && Nout % (K == 256 ? 32 : 64) == 0 && Nout <= kMaxNout; }

int weight_planes_launch(const float* W, int ldw, int Nout, int K, void* planes, hipStream_t s) {
    const int threads = Nout / 32 * (K / 16) * 64;
    hipLaunchKernelGGL(k_weight_planes, dim3((threads + 255) / 256), dim3(256), 0, s, W, ldw, Nout, K, reinterpret_cast<uint4*>(planes));
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

int project_launch(const float* A, int64_t M, int K, int lda, const void* planes, const float* bias, int Nout, float* C, int ldc, hipStream_t s) {
    const int probe = (K != 64 && tuning(kTuneLinearVariant) == 20) ? (tuning(kTuneGateAblation) & 63) : 0;
#define GN_PROBE(P) \
    if (probe == P) return K == 128 ? launch_project<128, P>(A, M, lda, planes, bias, Nout, C, ldc, s) : launch_project<256, P>(A, M, lda, planes, bias, Nout, C, ldc, s);
    GN_PROBE(1) GN_PROBE(2) GN_PROBE(3) GN_PROBE(4) GN_PROBE(7) GN_PROBE(8) GN_PROBE(16) GN_PROBE(32) GN_PROBE(33) GN_PROBE(34) GN_PROBE(35) GN_PROBE(39)
#undef GN_PROBE
    if (K == 64) return launch_project<64, 0>(A, M, lda, planes, bias, Nout, C, ldc, s);   // (round 6, late: the shipped checkpoint's width; no probes)
    return K == 128 ? launch_project<128, 0>(A, M, lda, planes, bias, Nout, C, ldc, s) : launch_project<256, 0>(A, M, lda, planes, bias, Nout, C, ldc, s);
}

}  // namespace gnnome

// Planes of a weight matrix for gnnome_linear_planes_f32: W[Nout, K] fp32 (row stride ldw, 16-byte aligned rows) -> Nout * K * 4 bytes at
// `planes` (256-byte aligned).  Once per weight matrix (gnnome_amd.engine.Prepared keeps them beside Wcat); 2-3 us.
extern "C" int gnnome_weight_planes_f16(const float* W, int ldw, int Nout, int K, void* planes, void* stream) {
    using namespace gnnome;
    GN_REQUIRE(W && planes, "weight_planes: null pointer");
    GN_REQUIRE(project_supported(K, Nout), "weight_planes: K=%d must be 64, 128 or 256 and Nout=%d a multiple of 32 (K <= 128: of 64) up to 1536", K, Nout);
    GN_REQUIRE(ldw >= K && ldw % 4 == 0 && (uintptr_t)W % 16 == 0 && (uintptr_t)planes % 256 == 0, "weight_planes: bad stride / alignment");
    return weight_planes_launch(W, ldw, Nout, K, planes, (hipStream_t)stream);
}

// C[M, Nout] = A[M, K] W^T + bias with W given as gnnome_weight_planes_f16's planes (gated_gcn_full.py:91-96 as one GEMM; bias may be NULL).
// A, C row-strided (lda, ldc multiples of 4, 16-byte aligned bases), C must not alias A.  A row's bits depend on that row and W alone.
extern "C" int gnnome_linear_planes_f32(const float* A, int64_t M, int K, int lda, const void* planes, const float* bias, int Nout, float* C,
                                        int ldc, void* stream) {
    using namespace gnnome;
    GN_REQUIRE(M >= 0, "linear_planes: bad row count %lld", (long long)M);
    if (M == 0) return GNNOME_OK;
    GN_REQUIRE(A && planes && C, "linear_planes: null pointer");
    GN_REQUIRE(project_supported(K, Nout), "linear_planes: K=%d must be 64, 128 or 256 and Nout=%d a multiple of 32 (K <= 128: of 64) up to 1536", K, Nout);
    GN_REQUIRE(lda >= K && ldc >= Nout && lda % 4 == 0 && ldc % 4 == 0 && (uintptr_t)A % 16 == 0 && (uintptr_t)C % 16 == 0 &&
                   (uintptr_t)planes % 256 == 0,
               "linear_planes: bad stride / alignment");
    GN_REQUIRE((const void*)A != (const void*)C, "linear_planes: C must not alias A");
    return project_launch(A, M, K, lda, planes, bias, Nout, C, ldc, (hipStream_t)stream);
}

// The one rule that sends a product to the kernel above (include/gnnome_hip.h): shapes, the caller's planes, the tuning switches, the A/B environment switch.
extern "C" int gnnome_linear_planes_route(int64_t M, int K, int Nout, int given_planes) {
    using namespace gnnome;
    static const bool enabled = [] {
        const char* v = getenv("GNNOME_PLANES_LINEAR");
        return !(v != nullptr && v[0] == '0' && v[1] == '\0');
    }();
    const int variant = tuning(kTuneLinearVariant);
    if (!enabled || M <= 0 || tuning(kTuneArith) != 0 || !(variant == 0 || variant == 9 || variant == 20) || !project_supported(K, Nout)) return 0;
    return (given_planes || (K >= 128 && Nout % 128 == 0 && (K == 256 || Nout >= 256))) ? 1 : 0;   // (K = 64: only on the caller's planes)
}
