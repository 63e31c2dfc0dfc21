// Edge-tile kernel for H = 256, round 4: fp16x3 arithmetic on the f16 matrix cores, A tiles brought in by LDS-DMA.
//
// What it replaces: k_edge_gate_pl256 (edge_gate_pl256.hip) in modes 0, 1 and 4 - the gate of gated_gcn_full.py:97,104-110, its raw
// form for the training forward, and the K = 256 node projection (gated_gcn_full.py:91-96) - which stays in the tree as the bf16x6
// form (gnnome_set_tuning(10, 1)) and keeps modes 2 and 3 (the backward's products).
//
// ARITHMETIC.  bf16x6 writes an fp32 operand as three bf16 planes (8 + 8 + 8 significant bits) and needs six of the nine plane
// products.  fp16 carries 11 significant bits, so TWO planes hold 22 and THREE products suffice:
//     x = x1 + x2 / 2048 + rx,   x1 = RN16(x),   x2 = RN16((x - x1) * 2048),   |rx| <= 2^-22 |x|      (x - x1 is exact in fp32)
//     x w  =  x1 w1  +  (x1 w2 + x2 w1) / 2048  +  [ x2 w2 / 2^22 + rx w + x rw ]                       dropped: <= 3 * 2^-22 |x w|
// The second planes are stored SCALED by 2^11 (so they are fp16 normals whenever the first plane is - nothing here relies on fp16
// denormals) and their two products go to a second fp32 accumulator that is folded in once per tile: x1 w1 takes 16 roundings per
// output instead of bf16x6's 96.  Measured against an fp64 product (K = 256, random operands; tests/test_f16x3_model.py restates the
// arithmetic in numpy): max error / sum |x||w| = 8e-8 for this form, 2.3e-7 for bf16x6, 3.3e-7 for an fp32 BLAS product - i.e. no
// further from the exact product than the reference's own fp32 GEMM, at HALF the matrix-core work of bf16x6.  Range: |x| < 65504
// (fp16's; beyond it the plane is inf and the output row NaN - loud, not silent); below 2^-14 the relative precision decays
// gracefully (absolute error <= 2^-36 per operand).
//
// STRUCTURE.  With half the MFMAs the plane form's long pole - the load waves that split every tile into planes (352 VALU operations
// + 48 ds_write_b64 per lane and tile, 5700 cycles per tile against 3580 of MFMA) - would be all that is left.  So the load waves are gone:
//   * the four COMPUTE waves bring the raw fp32 rows of a tile in by LDS-DMA (global_load_lds_dwordx4: one instruction = one 1 KB row,
//     no registers, no VALU; 8 rows per wave and tile, issued between the MFMAs three tiles ahead into a ring of four 33 KB slots);
//   * a wave turns ITS OWN eight rows into the two fp16 planes IN PLACE, one tile ahead and also between the MFMAs: a row's 256 floats
//     (1 KB) become 256 + 256 halves in the same kilobyte - one ds_read_b128, 12 VALU operations and two ds_write_b64 per lane and row,
//     no other wave involved, so the split is done once per workgroup (a first version let every compute wave split the whole tile on the
//     fly from the fp32 slot: 24 VALU operations per k step and wave, and a lone wave issues one instruction every four cycles - 3340
//     cycles per tile in the loop against 1536 of MFMA, measured);
//     (moving this conversion to the epilogue waves made THEM the long pole: 1.68 against 1.34 ms per launch, measured; the raw rows through
//     registers - global_load_dwordx4 during tile j, ds_write_b128 during tile j + 1 - instead of LDS-DMA: 1.437 against 1.369 ms, the DMA is not what binds);
//   * the MFMA loop then reads a fragment per plane and step (two ds_read_b128) and issues three MFMAs; W3's two planes for the wave's
//     32 output columns live in 128 registers;
//   * the four EPILOGUE waves never touch a tile: they gather G = B1h[src] + B2h[dst] and the residual two tiles ahead into three
//     register sets, read x from a 17 KB LDS buffer the compute waves leave it in, and store 16-byte row pieces;
//   * hand-over: full[slot] (a compute wave has turned its rows of the next tile into planes AND has issued its last read of the
//     current one; the DMA itself is awaited by a counted s_waitcnt vmcnt(8) - it is issued through inline assembly, so the compiler's
//     own wait insertion neither sees nor drains it), done (x is in the buffer), drained (x has been read).  Passing full[j] implies
//     every compute wave is through with tile j - 1, so refilling that tile's slot needs no fourth counter.
// As in the plane form two workgroups on one XCD take the two 128-column halves of the same tiles (mode 4: a.num_cblocks workgroups
// take the column blocks of the output); rows past the end of the list are clamped on the way in and not stored.  A row with an
// operand beyond fp16's range leaves the matrix cores as inf / NaN; the epilogue keeps it that way through the relu (fmaxf would
// turn NaN into 0): the output row is NaN, never silently wrong.
#include "common.h"

#include <type_traits>

#pragma clang diagnostic ignored "-Winline-asm"

namespace gnnome {
namespace {

typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr float kLoScale = 2048.f, kLoInv = 1.0f / 2048.f;

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
// one row piece of LDS-DMA: the wave's 64 lanes x 16 bytes land at lds .. lds + 1023 (lane-linear); the source is row + voff (voff = 16 lane)
__device__ __forceinline__ void dma_row(const float* row, unsigned voff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(row), "s"(lds) : "memory", "m0");
}

// the two fp16 planes of eight floats (see the header): p1 = RN16(x), p2 = RN16((x - p1) * 2048)
__device__ __forceinline__ void split8h(const f32x4 lo, const f32x4 hi, h8_t& p1, h8_t& p2) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x2 v = j < 2 ? f32x2{lo[2 * j], lo[2 * j + 1]} : f32x2{hi[2 * j - 4], hi[2 * j - 3]};
        const h2_t a = __builtin_convertvector(v, h2_t);
        const f32x2 big = v * kLoScale;
        const f32x2 r = {__builtin_fmaf((float)a[0], -kLoScale, big[0]), __builtin_fmaf((float)a[1], -kLoScale, big[1])};   // exact
        const h2_t b = __builtin_convertvector(r, h2_t);
        p1[2 * j] = a[0];
        p1[2 * j + 1] = a[1];
        p2[2 * j] = b[0];
        p2[2 * j + 1] = b[1];
    }
}

// four floats -> the two planes' four halves each
__device__ __forceinline__ void split4h(const f32x4 x, uint2& p1, uint2& p2) {
    h2_t a[2], b[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const f32x2 v = {x[2 * j], x[2 * j + 1]};
        a[j] = __builtin_convertvector(v, h2_t);
        const f32x2 big = v * kLoScale;
        const f32x2 r = {__builtin_fmaf((float)a[j][0], -kLoScale, big[0]), __builtin_fmaf((float)a[j][1], -kLoScale, big[1])};   // exact
        b[j] = __builtin_convertvector(r, h2_t);
    }
    p1 = make_uint2(__builtin_bit_cast(unsigned, a[0]), __builtin_bit_cast(unsigned, a[1]));
    p2 = make_uint2(__builtin_bit_cast(unsigned, b[0]), __builtin_bit_cast(unsigned, b[1]));
}
__device__ __forceinline__ h8_t as_h8(const uint4 v) { return __builtin_bit_cast(h8_t, v); }

// MODE 0: e' = relu((e W3^T + B1h[src] + B2h[dst]) * scale + shift) + e    (gated_gcn_full.py:97,104-110)
// MODE 1: xe = e W3^T + B1h[src] + B2h[dst] and its shifted column sums (training forward; a.scale = the centres, a.stats out)
// MODE 4: C[M, 128 * num_cblocks] = A[M,256] W^T + bias (a.e_in = A with row stride a.ldn, a.e_out = C with row stride a.ld_out, a.scale = bias)
// MODE 5: MODE 0 for layer 0 with the edge encoder folded ALGEBRAICALLY (as k_edge_gate_enc16 does at H = 128; full_graph.py:27 + gated_gcn_full.py:97,104-110):
//         e0 = t W2^T + b2 with t = relu(W1 e_raw + b1) only 16 wide, so e0 W3^T = t (W3 W2)^T + W3 b2 - the gate's product and the residual are both K = 16
//         products of one [32 x 16] tile t that every compute wave builds in registers from the raw edge features; no e tile, no DMA, nothing read from HBM
//         but two floats per edge and the gathers (a.enc: e_raw, srt_eid, W1, b1, W2, b2 and W23 = W3 W2, b23 = W3 b2 from k_fold_encoder)
// PROBE (measurement only, wrong results; gnnome_set_tuning(1, 100 + mask)): 1 no DMA inside the loop, 2 no plane conversion inside the loop,
// 4 no MFMAs, 8 no gathers / residual loads, 16 no stores, 32 (correct results) B2h[dst] fetched for every piece
// X16 (MODE 1): xe stored as bf16 (rounded to nearest even; the statistics are those of the rounded values - common.h)
template <int MODE, int PROBE = 0, bool X16 = false>
__global__ __launch_bounds__(512) void k_edge_tile_f16(GateBfArgs a) {
    static_assert(!X16 || MODE == 1, "bf16 storage belongs to the raw gate");
    constexpr bool ENC = MODE == 5, GATE = MODE == 0 || MODE == 5;
    constexpr int H = 256, HC = 128, TM = 32, KS = H / 16, NS = 4;
    constexpr int RSB = 4 * H + 16, SLOTB = TM * RSB;   // a raw row in LDS: 1024 bytes + 16 (consecutive rows start 4 banks apart)
    constexpr int LDK = HC + 4, XT = TM * LDK;
    constexpr int NP = 4;   // epilogue pieces per lane and tile: a wave owns 8 rows x 128 columns
    __shared__ __attribute__((aligned(16))) unsigned char ring[ENC ? 16 : NS * SLOTB];
    __shared__ __attribute__((aligned(16))) float xt[ENC ? 2 * XT : XT];   // ENC: the x tile and the e0 tile
    __shared__ __attribute__((aligned(16))) float norm_lds[2 * HC];
    __shared__ unsigned flags[2 * NS + 2];   // full[NS], done, drained, landed[NS] (EPI_CONV)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned full0 = lds_addr(&flags[0]), done0 = lds_addr(&flags[NS]), drained0 = lds_addr(&flags[NS + 1]), landed0 = lds_addr(&flags[NS + 2]);
    // MODE 4 (no gathers, no residual: the epilogue waves only add a bias and store): THEY turn the landed rows into planes, the compute waves
    // keep DMA + MFMA.  (In the gate the same split made the epilogue waves the long pole: 1.68 against 1.34 ms per launch.  Letting the epilogue waves
    // issue the DMA of their rows as well - they have no other loads in this mode - measured level: 0.66 against 0.63-0.66 ms per projection.)
    constexpr bool EPI_CONV = MODE == 4;
    // the work distribution of k_edge_gate_pl256: pairs of workgroups on one XCD (blocks b and b + 8 share b % 8) take the two column
    // halves of the same tiles; mode 4: the a.num_cblocks workgroups of an XCD that share idx / num_cblocks walk the same tiles
    const int per_xcd = gridDim.x / kXcds, xcd = blockIdx.x % kXcds, idx = blockIdx.x / kXcds;
    const int ncb = MODE == 4 ? a.num_cblocks : 2, streams = per_xcd / ncb;
    if (idx >= streams * ncb) return;
    const int hh = idx % ncb, first = xcd * streams + idx / ncb, stride = kXcds * streams;
    const int n = first < a.num_tiles ? (a.num_tiles - first + stride - 1) / stride : 0;
    if (n <= 0) return;
    const int lda = MODE == 4 ? a.ldn : H, ldo = MODE == 4 ? a.ld_out : H;
    // ordinals past the end repeat the last tile (their loads are issued so that every wait counts the same instructions; nothing reads them)
    auto tile_of = [&](int r) {
        r = min(r, n - 1);
        // Round 5: both workgroups of a pair walk the tiles in the SAME order (their requests for a row meet in the L2's miss queue), and the
        // gate's e' rows leave through nontemporal stores so that they do not push the e rows the pair shares out of the L2: HBM read per
        // launch 5489 -> 3588 MB at the 2.5M-edge shard (e rows: 2560 MB), 1.515 -> 1.46 ms (profiles/r05_gate256_pair_sharing.txt).
        // PROBE 64: round 4's order (the second workgroup swaps every two tiles, as k_edge_gate_pl256 does); PROBE 128: plain stores.
        const int rr = (MODE != 4 && (PROBE & 64) && hh == 1 && (r ^ 1) < n) ? (r ^ 1) : r;
        return first + rr * stride;
    };
    auto tile_valid = [&](int r) { return (int)min((int64_t)TM, a.E - (int64_t)tile_of(r) * TM); };
    const int colh = HC * hh;
    if (tid < 2 * NS + 2) flags[tid] = 0;
    if (MODE == 4) {
        for (int i = tid; i < HC; i += 512) norm_lds[i] = a.scale ? a.scale[colh + i] : 0.f;
    } else {
        for (int i = tid; i < 2 * HC; i += 512) {
            const int q = i / HC, c = i % HC;
            // ENC: x lacks W3 b2 (= b23): (x + b23 + G) scale + shift = (x + G) scale + (shift + b23 scale)
            norm_lds[i] = q == 0 ? (a.scale ? a.scale[colh + c] : 0.f)
                                 : (MODE == 0 ? a.shift[colh + c] : (ENC ? __builtin_fmaf(a.enc.b23[colh + c], a.scale[colh + c], a.shift[colh + c]) : 0.f));
        }
    }
    __syncthreads();

    if (ENC && wave < 4) {
        // ------------------------------------------------------------------ compute wave, folded encoder: t in registers, two K = 16 products
        const int cl = lane & 31, half = lane >> 5, col = colh + 32 * wave + cl;
        h8_t wxa, wxb, wea, web;   // this lane's B fragments (column col, k = 8 half .. + 7) of W23 and of W2
        {
            const float* p23 = a.enc.W23 + col * 16 + 8 * half;
            const float* p2 = a.enc.W2 + col * 16 + 8 * half;
            split8h(f32x4{p23[0], p23[1], p23[2], p23[3]}, f32x4{p23[4], p23[5], p23[6], p23[7]}, wxa, wxb);
            split8h(f32x4{p2[0], p2[1], p2[2], p2[3]}, f32x4{p2[4], p2[5], p2[6], p2[7]}, wea, web);
        }
        const float b2c = a.enc.b2[col];
        float w1a[8], w1b[8], b1v[8];   // hidden units 8 half .. + 7 of the edge encoder's first layer (in_features = 2)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            w1a[k] = a.enc.W1[2 * (8 * half + k)], w1b[k] = a.enc.W1[2 * (8 * half + k) + 1], b1v[k] = a.enc.b1[8 * half + k];
        }
        auto crow = [](int r) { return (r & 3) + 8 * (r >> 2); };
        float* X = xt + 4 * half * LDK + 32 * wave + cl;
        // the raw features of tile row cl: two dependent loads (edge id, then the two floats) - the edge id is fetched TWO tiles ahead, the features one
        // (a tile lasts ~1 us here: with both one tile ahead the chain of two memory round trips was exposed in every tile)
        auto eid_of = [&](int r) { return a.enc.srt_eid[(int64_t)tile_of(r) * TM + min(cl, tile_valid(r) - 1)]; };
        float r0 = 0.f, r1 = 0.f;
        {
            const int64_t e0i = eid_of(0);
            r0 = a.enc.e_raw[2 * e0i], r1 = a.enc.e_raw[2 * e0i + 1];
        }
        int eid_next = eid_of(1);
#pragma unroll 1
        for (int j = 0; j < n; ++j) {
            const int eid_after = eid_of(j + 2);   // (ordinals past the end repeat the last tile)
            const float n0 = a.enc.e_raw[2 * (int64_t)eid_next], n1 = a.enc.e_raw[2 * (int64_t)eid_next + 1];
            f32x4 tl, th;   // t = relu(W1 e_raw + b1) in the reference's order (k_edge_gate_enc16)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                tl[k] = fmaxf(__builtin_fmaf(r1, w1b[k], r0 * w1a[k]) + b1v[k], 0.f);
                th[k] = fmaxf(__builtin_fmaf(r1, w1b[4 + k], r0 * w1a[4 + k]) + b1v[4 + k], 0.f);
            }
            h8_t ta, tb;
            split8h(tl, th, ta, tb);
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
            f32x16 xm = __builtin_amdgcn_mfma_f32_32x32x16_f16(ta, wxa, z, 0, 0, 0);
            f32x16 xc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ta, wxb, z, 0, 0, 0);
            xc = __builtin_amdgcn_mfma_f32_32x32x16_f16(tb, wxa, xc, 0, 0, 0);
            f32x16 em = __builtin_amdgcn_mfma_f32_32x32x16_f16(ta, wea, z, 0, 0, 0);
            f32x16 ec = __builtin_amdgcn_mfma_f32_32x32x16_f16(ta, web, z, 0, 0, 0);
            ec = __builtin_amdgcn_mfma_f32_32x32x16_f16(tb, wea, ec, 0, 0, 0);
            flag_wait(drained0, 4u * (unsigned)j);   // x(j - 1) and e0(j - 1) have been read by all four epilogue waves
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                X[crow(r) * LDK] = xm[r] + xc[r] * kLoInv;
                X[XT + crow(r) * LDK] = (em[r] + ec[r] * kLoInv) + b2c;
            }
            flag_bump(done0, lane);
            r0 = n0, r1 = n1;
            eid_next = eid_after;
        }
    } else if (!ENC && wave < 4) {
        // ------------------------------------------------------------------ compute wave: DMA in, planes in place, 32 rows x 32 columns of MFMA
        const int cl = lane & 31, half = lane >> 5, col = colh + 32 * wave + cl;
        h8_t w1[KS], w2[KS];
#pragma unroll
        for (int q = 0; q < KS; ++q) {   // step q: k in [16 q + 8 half, + 8)
            const float* wp = a.W3 + (int64_t)col * a.ldw + 16 * q + 8 * half;
            split8h(*reinterpret_cast<const f32x4*>(wp), *reinterpret_cast<const f32x4*>(wp + 4), w1[q], w2[q]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the W loads: from here on this wave's only vector-memory traffic is its DMA)
        const unsigned ring0 = lds_addr(ring), voff = 16u * (unsigned)lane;
        // this wave's rows 8 wave .. 8 wave + 7 of tile ordinal r: the source row of piece 0 (rows past the end: the last valid row), the
        // stride to the next piece (0 once clamped), the slot address of piece 0
        const float* d_row = nullptr;
        int d_left = 0;
        unsigned d_lds = 0;
        auto dma_begin = [&](int r) {
            const int valid = tile_valid(r), trow = min(8 * wave, valid - 1);
            d_row = a.e_in + ((int64_t)tile_of(r) * TM + trow) * lda;
            d_left = valid - 1 - trow;   // rows that follow trow inside the tile
            d_lds = ring0 + (unsigned)((r % NS) * SLOTB + 8 * wave * RSB);
        };
        auto dma_piece = [&]() {
            dma_row(d_row, voff, d_lds);
            if (d_left > 0) d_row += lda, --d_left;
            d_lds += RSB;
        };
        // a row of this wave in place: 256 floats -> 256 + 256 halves in the same kilobyte (all 64 lanes read before any of them writes:
        // LDS operations of one wave execute in order)
        auto row_read = [&](int r, int p) { return *reinterpret_cast<const f32x4*>(ring + (r % NS) * SLOTB + (8 * wave + p) * RSB + 16 * lane); };
        auto row_write = [&](int r, int p, const f32x4 x) {
            uint2 p1, p2;
            split4h(x, p1, p2);
            unsigned char* d = ring + (r % NS) * SLOTB + (8 * wave + p) * RSB + 8 * lane;
            *reinterpret_cast<uint2*>(d) = p1;
            *reinterpret_cast<uint2*>(d + 2 * H) = p2;
        };
#pragma unroll 1
        for (int r = 0; r < 3; ++r) {
            dma_begin(r);
#pragma unroll
            for (int p = 0; p < 8; ++p) dma_piece();
        }
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        if (EPI_CONV) {
            flag_bump(landed0, lane);   // my rows of tile 0 are in the slot
        } else {
            f32x4 t[8];
#pragma unroll
            for (int p = 0; p < 8; ++p) t[p] = row_read(0, p);
#pragma unroll
            for (int p = 0; p < 8; ++p) row_write(0, p, t[p]);
            flag_bump(full0, lane);
        }
        auto crow = [](int r) { return (r & 3) + 8 * (r >> 2); };
        float* X = xt + 4 * half * LDK + 32 * wave + cl;   // accumulator element r sits in tile row 4 half + crow(r)
        long long t_wait = 0, t_loop = 0, t_x = 0, t0 = 0, t1 = 0;
        const long long c_begin = a.prof ? (long long)__builtin_readcyclecounter() : 0;
        const long long r_begin = a.prof ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
#pragma unroll 1
        for (int j = 0; j < n; ++j) {
            const int slot = j % NS;
            if (a.prof) t0 = __builtin_readcyclecounter();
            if (!(PROBE & 1)) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // this wave's rows of tile j + 1 have landed (tile j + 2's may be in flight)
            if (EPI_CONV) flag_bump(landed0 + 4 * ((j + 1) % NS), lane);
            flag_wait(full0 + 4 * slot, 4u * ((unsigned)(j / NS) + 1u));   // tile j's planes are complete; (!EPI_CONV:) every compute wave is through with tile j - 1
            if (EPI_CONV) flag_wait(done0, 4u * (unsigned)j);              // every compute wave is through with tile j - 1: its slot may be refilled
            if (a.prof) { t1 = __builtin_readcyclecounter(); t_wait += t1 - t0; t0 = t1; }
            const unsigned char* ap = ring + slot * SLOTB + cl * RSB + 16 * half;   // + 32 q: the lane's eight halves of step q; + 2 H: the second plane
            dma_begin(j + 3);   // tile j + 3 goes into the slot tile j - 1 has left
            f32x16 accM, accC;
#pragma unroll
            for (int r = 0; r < 16; ++r) accM[r] = 0.f, accC[r] = 0.f;
            // Plane fragments PD k steps ahead of their MFMAs, and the scheduling barriers that keep them there: hipcc sinks LDS reads to the
            // instruction before their use, and a read issued one or two MFMAs ahead had every k step wait for its LDS round trip (2963 cycles
            // per tile in this loop for 48 MFMAs of 32 - the matrix pipe idle half of the time).
            constexpr int PD = 3;
            uint4 f1[KS], f2[KS];
#pragma unroll
            for (int q = 0; q < PD; ++q) {
                f1[q] = *reinterpret_cast<const uint4*>(ap + 32 * q);
                f2[q] = *reinterpret_cast<const uint4*>(ap + 32 * q + 2 * H);
            }
            f32x4 raw = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < KS; ++q) {
                if (q + PD < KS) {
                    f1[q + PD] = *reinterpret_cast<const uint4*>(ap + 32 * (q + PD));
                    f2[q + PD] = *reinterpret_cast<const uint4*>(ap + 32 * (q + PD) + 2 * H);
                }
                if ((q & 1) == 0) {
                    if (!(PROBE & 1)) dma_piece();
                    if (!(PROBE & 2) && !EPI_CONV) raw = row_read(j + 1, q >> 1);        // tile j + 1, this wave's row q / 2 ...
                }
                __builtin_amdgcn_sched_barrier(0);
                if ((q & 1) != 0) {
                    if (!(PROBE & 2) && !EPI_CONV) row_write(j + 1, q >> 1, raw);        // ... becomes planes a step later (under this step's MFMAs)
                }
                const uint4 c1 = f1[q], c2 = f2[q];
                if (PROBE & 4) {
                    accM[q] += __uint_as_float(c1.x ^ c2.y ^ __builtin_bit_cast(uint4, w1[q]).x ^ __builtin_bit_cast(uint4, w2[q]).y);
                } else {
                    accM = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(c1), w1[q], accM, 0, 0, 0);
                    accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(c1), w2[q], accC, 0, 0, 0);
                    accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(c2), w1[q], accC, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!EPI_CONV) flag_bump(full0 + 4 * ((j + 1) % NS), lane);   // my rows of tile j + 1 are planes, my reads of tile j are issued
            if (a.prof) { asm volatile("" ::"v"(accM[0])); t1 = __builtin_readcyclecounter(); t_loop += t1 - t0; t0 = t1; }
            flag_wait(drained0, 4u * (unsigned)j);   // x(j - 1) has been read by all four epilogue waves
#pragma unroll
            for (int r = 0; r < 16; ++r) X[crow(r) * LDK] = accM[r] + accC[r] * kLoInv;
            flag_bump(done0, lane);
            if (a.prof) { t1 = __builtin_readcyclecounter(); t_x += t1 - t0; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the surplus tiles' rows: the LDS must not be handed back with DMA writes in flight)
        if (a.prof && wave == 0 && lane == 0) {   // the record layout of k_edge_gate_pl (tools/gate_phase_profile.py)
            long long* o = a.prof + (int64_t)blockIdx.x * 8;
            o[0] = t_wait; o[1] = 0; o[2] = t_loop; o[3] = t_x; o[4] = n;
            o[5] = (long long)__builtin_readcyclecounter() - c_begin;
            o[6] = (long long)__builtin_amdgcn_s_memrealtime() - r_begin;
        }
    } else {
        // ------------------------------------------------------------------ epilogue wave: rows 8 ew .. 8 ew + 7 of every tile
        const int ew = wave - 4;
        const int c4 = lane & 31, hl = 4 * (lane >> 5), rl = 8 * ew + hl;   // piece p: tile row rl + p, columns colh + 4 c4 .. + 3
        f32x4 g1[3][NP], g2[3][NP], ek[3][NP];
        unsigned g2_fresh[3] = {0u, 0u, 0u};   // (wave-uniform) bit p: B2h[dst] was fetched for piece p of the set
        constexpr bool FRESH_G2 = !(PROBE & 32);
        int si[3], di[3];   // lane l: the endpoints of tile row 8 ew + l % 8 (one load per array, wave and tile)
        auto fetch_index = [&](auto set, int r) {
            constexpr int S = decltype(set)::value;
            if (MODE == 4) return;
            const int64_t row = (int64_t)tile_of(r) * TM + min(8 * ew + (lane & 7), tile_valid(r) - 1);
            si[S] = a.srt_src[row];
            di[S] = a.srt_dst[row];
        };
        auto fetch_side = [&](auto set, int r) {
            constexpr int S = decltype(set)::value;
            if (MODE == 4) return;
            if (PROBE & 8) {
#pragma unroll
                for (int p = 0; p < NP; ++p) g1[S][p] = g2[S][p] = ek[S][p] = f32x4{0.f, 0.f, 0.f, 0.f};
                return;
            }
            const int64_t row0 = (int64_t)tile_of(r) * TM;
            const int valid = tile_valid(r);
            int dprev = -1;
            g2_fresh[S] = 0;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int sp = __shfl(si[S], hl + p), dp = __shfl(di[S], hl + p);
                g1[S][p] = *reinterpret_cast<const f32x4*>(a.B1h + (int64_t)sp * a.ldn + colh + 4 * c4);
                // B2h[dst]: the rows are destination-sorted and a lane's pieces are consecutive rows - the row is fetched again only when some
                // lane of the wave needs a new one (a wave-uniform branch; the skipped pieces are filled in from their predecessors when consumed)
                const bool fresh = p == 0 || dp != dprev;
                if (!FRESH_G2 || __builtin_amdgcn_ballot_w64(fresh) != 0) {
                    g2[S][p] = *reinterpret_cast<const f32x4*>(a.B2h + (int64_t)dp * a.ldn + colh + 4 * c4);
                    g2_fresh[S] |= 1u << p;
                }
                dprev = dp;
                if (MODE == 0) ek[S][p] = *reinterpret_cast<const f32x4*>(a.e_in + (row0 + min(rl + p, valid - 1)) * H + colh + 4 * c4);   // (ENC: e0 comes out of LDS with x)
            }
        };
        // EPI_CONV: rows 8 ew .. 8 ew + 7 of tile ordinal r become planes in place (all 64 lanes read a row before any of them writes it)
        auto to_planes = [&](int r) {
            flag_wait(landed0 + 4 * (r % NS), 4u * ((unsigned)(r / NS) + 1u));
            unsigned char* rows = ring + (r % NS) * SLOTB + 8 * ew * RSB;
            f32x4 t[8];
#pragma unroll
            for (int p = 0; p < 8; ++p) t[p] = *reinterpret_cast<const f32x4*>(rows + p * RSB + 16 * lane);
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                uint2 p1, p2;
                split4h(t[p], p1, p2);
                *reinterpret_cast<uint2*>(rows + p * RSB + 8 * lane) = p1;
                *reinterpret_cast<uint2*>(rows + p * RSB + 8 * lane + 2 * H) = p2;
            }
            flag_bump(full0 + 4 * (r % NS), lane);
        };
        f32x4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = st1;   // MODE 1: this lane's running shifted sums of its four columns
        long long t_done = 0, t_epi = 0, t_issue = 0, t0 = 0, t1 = 0;
        const f32x4 sc4 = *reinterpret_cast<const f32x4*>(norm_lds + 4 * c4);   // MODE 0: scale; 1: centres; 4: bias
        const f32x4 sh4 = GATE ? *reinterpret_cast<const f32x4*>(norm_lds + HC + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
        const float* Xs = xt + rl * LDK + 4 * c4;
        auto tile = [&](auto set, int i) {
            constexpr int S = decltype(set)::value, S2 = (S + 2) % 3;
            if (a.prof) t0 = __builtin_readcyclecounter();
            if (EPI_CONV && i + 1 < n) to_planes(i + 1);   // (while the compute waves are busy with tile i)
            flag_wait(done0, 4u * ((unsigned)i + 1u));
            if (a.prof) { t1 = __builtin_readcyclecounter(); t_done += t1 - t0; t0 = t1; }
            f32x4 x[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) x[p] = *reinterpret_cast<const f32x4*>(Xs + p * LDK);
            if (ENC) {
#pragma unroll
                for (int p = 0; p < NP; ++p) ek[S][p] = *reinterpret_cast<const f32x4*>(Xs + XT + p * LDK);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ek[S][0]), "+v"(ek[S][1]), "+v"(ek[S][2]), "+v"(ek[S][3]));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
            flag_bump(drained0, lane);
            const int valid = tile_valid(i);
            if (MODE != 4 && FRESH_G2) {
                const unsigned fr = __builtin_amdgcn_readfirstlane(g2_fresh[S]);
#pragma unroll
                for (int p = 1; p < NP; ++p)
                    if (!((fr >> p) & 1u)) g2[S][p] = g2[S][p - 1];   // same destination as the row above: the same B2h row
            }
            float* out = a.e_out + ((int64_t)tile_of(i) * TM + rl) * ldo + colh + 4 * c4;
            // round 5, the two-pass training forward at this width: the gate (mode 0) ALSO writes the pre-normalisation rows xe = x + G to
            // a.bnb.a_out when that is set; the raw gate (mode 1) with a.e_out = NULL leaves its statistics alone (both wave-uniform tests)
            float* xe_out = (GATE && a.bnb.a_out != nullptr) ? a.bnb.a_out + ((int64_t)tile_of(i) * TM + rl) * ldo + colh + 4 * c4 : nullptr;
            const bool stats_only = MODE == 1 && a.e_out == nullptr;
            auto pieces = [&](auto full_tile) {
                constexpr bool FULL = decltype(full_tile)::value;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    f32x4 y, xg = {0.f, 0.f, 0.f, 0.f};
                    if (GATE) {
                        const f32x4 g = g1[S][p] + g2[S][p];
                        xg = x[p] + g;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float t = xg[k] * sc4[k] + sh4[k];
                            // (t - t is 0 for a finite t and NaN otherwise: an operand beyond fp16's range must not come out of the relu as 0)
                            y[k] = (t - t == 0.f) ? fmaxf(t, 0.f) + ek[S][p][k] : __builtin_nanf("");
                        }
                    } else if (MODE == 4) {
                        y = x[p] + sc4;
                    } else {
                        y = x[p] + (g1[S][p] + g2[S][p]);
                    }
                    uint2 pk = {0u, 0u};
                    if (X16) {
                        pk = pack_bf16x4(y);
                        y = unpack_bf16x4(pk);
                    }
                    if (FULL || rl + p < valid) {
                        if (MODE == 1) {
                            const f32x4 dlt = y - sc4;
                            st1 += dlt;
                            st2 += dlt * dlt;
                        }
                        if (GATE && xe_out != nullptr) __builtin_nontemporal_store(xg, reinterpret_cast<f32x4*>(xe_out + (int64_t)p * ldo));
                        if (PROBE & 16)
                            asm volatile("" ::"v"(y));
                        else if (stats_only)
                            ;
                        else if (X16)
                            *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.e_out) + ((int64_t)tile_of(i) * TM + rl + p) * ldo + colh + 4 * c4) = pk;
                        else if (GATE && !(PROBE & 128))   // nontemporal: the e' rows are not read again by this launch - the L2 is for the e rows the pair shares
                            __builtin_nontemporal_store(y, reinterpret_cast<f32x4*>(out + (int64_t)p * ldo));
                        else
                            *reinterpret_cast<f32x4*>(out + (int64_t)p * ldo) = y;
                    }
                }
            };
            if (valid == TM)
                pieces(std::true_type{});
            else
                pieces(std::false_type{});
            if (a.prof) { t1 = __builtin_readcyclecounter(); t_epi += t1 - t0; t0 = t1; }
            // behind the stores: tile i + 3's endpoints FIRST, then tile i + 2's gathers through the endpoints fetched an iteration ago -
            // the memory pipeline returns in order, so waiting for those endpoints then means waiting for last iteration's stores only,
            // not for last iteration's gathers (the other order serialised one memory latency per tile: 2000 cycles of "issue", measured)
            fetch_index(set, i + 3);
            fetch_side(std::integral_constant<int, S2>{}, i + 2);
            if (a.prof) { t1 = __builtin_readcyclecounter(); t_issue += t1 - t0; }
        };
        if (EPI_CONV) to_planes(0);
        fetch_index(std::integral_constant<int, 0>{}, 0);
        fetch_index(std::integral_constant<int, 1>{}, 1);
        fetch_side(std::integral_constant<int, 0>{}, 0);
        fetch_index(std::integral_constant<int, 2>{}, 2);
        fetch_side(std::integral_constant<int, 1>{}, 1);
#pragma unroll 1
        for (int i = 0; i < n; i += 3) {
            tile(std::integral_constant<int, 0>{}, i);
            if (i + 1 < n) tile(std::integral_constant<int, 1>{}, i + 1);
            if (i + 2 < n) tile(std::integral_constant<int, 2>{}, i + 2);
        }
        if (a.prof && wave == 4 && lane == 0) {   // the first epilogue wave's phases, after the 256 compute-wave records
            long long* o = a.prof + (int64_t)(256 + blockIdx.x) * 8;
            o[0] = 0; o[1] = 0; o[2] = t_done; o[3] = t_epi; o[4] = n; o[5] = t_issue;
        }
        if (MODE == 1 && a.stats != nullptr) {
            // lanes l and l + 32 hold different rows of the same four columns: fold them, then every epilogue wave leaves one row of
            // partial sums for its 128 columns (the layout of k_edge_gate_pl256: [2][grid * 4][H])
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                st1[k] += __shfl_xor(st1[k], 32);
                st2[k] += __shfl_xor(st2[k], 32);
            }
            if (lane < 32) {
                const int64_t srow = (int64_t)blockIdx.x * 4 + ew, srows = (int64_t)gridDim.x * 4;
                *reinterpret_cast<f32x4*>(a.stats + srow * H + colh + 4 * c4) = st1;
                *reinterpret_cast<f32x4*>(a.stats + (srows + srow) * H + colh + 4 * c4) = st2;
            }
        }
    }
}

template <int MODE, int PROBE = 0, bool X16 = false>
int launch_f16(const GateBfArgs& args, int grid, hipStream_t s) {
    GateBfArgs a = args;
    hipLaunchKernelGGL((k_edge_tile_f16<MODE, PROBE, X16>), dim3(grid), dim3(512), 0, s, a);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

}  // namespace

// called by gate_pl256_launch (edge_gate_pl256.hip) with the arguments checked and num_tiles / prof filled in
int gate_f16_launch(int mode, const GateBfArgs& a, int grid, hipStream_t s, bool x16) {
    if (x16) {
        GN_REQUIRE(mode == 1, "edge-tile kernel (H = 256, fp16x3): bf16 storage exists for the raw gate (mode 1) only");
        return launch_f16<1, 0, true>(a, grid, s);
    }
    if (mode == 0 && tuning(kTuneGateAblation) >= 100) {   // measurement only (tools/gate_time.py --ablations 101,102,...)
        switch (tuning(kTuneGateAblation) - 100) {
            case 1: return launch_f16<0, 1>(a, grid, s);
            case 2: return launch_f16<0, 2>(a, grid, s);
            case 3: return launch_f16<0, 3>(a, grid, s);
            case 4: return launch_f16<0, 4>(a, grid, s);
            case 7: return launch_f16<0, 7>(a, grid, s);
            case 8: return launch_f16<0, 8>(a, grid, s);
            case 16: return launch_f16<0, 16>(a, grid, s);
            case 24: return launch_f16<0, 24>(a, grid, s);
            case 31: return launch_f16<0, 31>(a, grid, s);
            case 32: return launch_f16<0, 32>(a, grid, s);   // B2h[dst] fetched for every piece (the form before the destination-run skip)
            case 64: return launch_f16<0, 64>(a, grid, s);     // (correct results) round 4's tile order: the second workgroup of a pair swaps every two tiles
            case 128: return launch_f16<0, 128>(a, grid, s);   // (correct results) plain e' stores
            case 192: return launch_f16<0, 192>(a, grid, s);   // both = round 4's form
            default: break;
        }
    }
    if (mode == 0) return launch_f16<0>(a, grid, s);
    if (mode == 1) return launch_f16<1>(a, grid, s);
    if (mode == 4) return launch_f16<4>(a, grid, s);
    if (mode == 5 && tuning(kTuneGateAblation) == 308) return launch_f16<5, 8>(a, grid, s);    // measurement only: no gathers
    if (mode == 5 && tuning(kTuneGateAblation) == 316) return launch_f16<5, 16>(a, grid, s);   // measurement only: no stores
    if (mode == 5) return launch_f16<5>(a, grid, s);
    set_error("edge-tile kernel (H = 256, fp16x3): mode %d is not built", mode);
    return GNNOME_EINVAL;
}

}  // namespace gnnome
