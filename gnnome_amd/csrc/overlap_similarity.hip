// gnnome_overlap_edit_distance: the edit distances behind `overlap_similarity`, on the device.
//
// Reference lines replaced: graph_parser.py:101-117 (calculate_similarities) - for every edge (src, dst) with overlap length ol
//     edit_distance = edlib.align(read_src[-ol:], read_dst[:ol])['editDistance']      (edlib defaults: global / NW alignment)
//     overlap_similarity = 1 - edit_distance / ol
// with read_seqs[2r] = read r and read_seqs[2r+1] = its reverse complement (graph_parser.py:365).  edlib is a third-party aligner
// that is not in the reference tree; what it returns here is the plain Levenshtein distance of the two strings, which is what
// this kernel computes exactly (any exact algorithm gives the same integer).
//
// Algorithm: Myers' bit-vector dynamic programme in Hyyro's block form (the formulation edlib itself uses), 32 query rows per
// word, laid out for a wavefront:
//   * one wave per overlap; lane L owns the B consecutive 32-row blocks [L*B, (L+1)*B) of the query (B = 1..32 chosen from
//     the query length: up to 65 536 rows), their vertical deltas Pv / Mv in registers;
//   * the lanes run SKEWED by one column: at step t lane L processes target column t - L, so the horizontal delta that
//     leaves lane L-1's last block for a column is exactly what lane L needs one step later - it travels with the column's
//     symbol in ONE cross-lane move per step (2 bits of hout + the symbol), lane 0 taking hin = +1 (row 0 of the NW matrix)
//     and the next target symbol from a 64-symbol buffer the wave loads with one coalesced read every 64 steps;
//   * match masks Peq[symbol][block] live in LDS as [symbol][k][lane] words: the lanes of a step sit at different columns, i.e.
//     use different symbols, and still hit 64 different banks;
//   * every lane tracks D at the bottom of its own rows; the lane that owns the query's last row subtracts the vertical
//     deltas of the padding rows below it (pad rows match nothing, so they never influence the rows above).
// Work per overlap ~ m * n / 32 block steps of ~20 integer VALU operations: no floating point, no matrix cores, no HBM
// traffic to speak of (the two strings are read once) - this path is bound by the integer VALU rate.
// Reads are addressed in place: node 2r reads read r forwards, node 2r+1 reads it backwards through the complement half of
// the symbol table - no reverse-complemented copy of the reads exists.
//
// Round 4: an Ukkonen BAND in front of it (k_overlap_banded).  The full matrix costs m * n cells; overlaps between accurate
// reads differ in a few dozen positions, and an alignment of cost d never leaves the diagonals [min(0, n-m) - d, max(0, n-m) + d].
// The banded pass runs the same block recurrence over a window of 8 blocks (256 query rows) that slides down one block every
// 32 target columns - ONE THREAD per overlap, window state in registers, no LDS besides the symbol table:
//   * cells outside the window count as upper bounds (a block enters with Pv = all +1, the first block of a column takes
//     hin = +1), so the banded value is >= the true distance, and it IS the true distance whenever it is <= the band's k
//     (then the optimal path lies inside the band and every cell on it was computed from true predecessors);
//   * match masks are not tabulated: a block keeps its 32 query symbols as NP bit planes (NP = 2 / 3 / 4 for alphabets of
//     <= 4 / 8 / 16 symbols) and Eq = AND_k (plane_k XOR ~column_bit_k) costs 2 NP - 1 integer operations per block step;
//   * an overlap whose banded value exceeds k (k = 96 for m = n; smaller when the lengths differ, none when they differ by
//     more than ~190), or whose alphabet has more than 16 symbols, is left marked for the full-matrix wave kernels below.
// ~8 block steps per column instead of m / 32: a 7.7 kb HiFi-grade overlap costs 1/30 of the cells.
#include "common.h"

#include <algorithm>

namespace gnnome {
namespace {

constexpr int kMaxSyms = 32;

__device__ __forceinline__ int myers_block(uint32_t& Pv, uint32_t& Mv, uint32_t Eq, int hin) {
    const uint32_t hneg = (uint32_t)hin >> 31;   // 1 iff hin == -1
    const uint32_t Xv = Eq | Mv;
    Eq |= hneg;
    const uint32_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
    uint32_t Ph = Mv | ~(Xh | Pv);
    uint32_t Mh = Pv & Xh;
    const int hout = (int)(Ph >> 31) - (int)(Mh >> 31);
    Ph = (Ph << 1) | (uint32_t)(hin > 0);
    Mh = (Mh << 1) | hneg;
    Pv = Mh | ~(Xv | Ph);
    Mv = Ph & Xv;
    return hout;
}

// blocks per lane for a query of m rows: the smallest class that covers it with 64 lanes (0: longer than the largest class)
__device__ __forceinline__ int class_of(int m) {
    const int need = (m + 2047) / 2048;   // blocks per lane at 64 lanes x 32 rows
    return need <= 4 ? need : need <= 6 ? 6 : need <= 8 ? 8 : need <= 12 ? 12 : need <= 16 ? 16 : need <= 24 ? 24 : need <= 32 ? 32 : 0;
}

constexpr int kNeedFull = -3;   // dist_out value of an overlap that still wants the full-matrix kernel (never leaves this file)
constexpr int kBandW = 8;       // blocks in the sliding window

// m, n of an edge exactly as the full kernel defines them (read_src[-ol:] is the whole read when ol exceeds its length)
__device__ __forceinline__ void overlap_shape(const int64_t* __restrict__ read_off, int u, int v, int L, int& m, int& n, int& ulen, int& vlen) {
    ulen = (int)(read_off[(u >> 1) + 1] - read_off[u >> 1]);
    vlen = (int)(read_off[(v >> 1) + 1] - read_off[v >> 1]);
    m = max(min(L, ulen), 0);
    n = max(min(L, vlen), 0);
}

// every edge: the trivial cases are answered here, the rest is marked for the banded / full kernels
__global__ void k_overlap_prepare(const int64_t* __restrict__ read_off, int64_t num_reads, const int32_t* __restrict__ src,
                                  const int32_t* __restrict__ dst, const int32_t* __restrict__ ol, int64_t E, int32_t* __restrict__ dist_out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= E) return;
    // an endpoint outside [0, 2 num_reads) would index read_off out of bounds: the edge is reported (-1) and no later kernel touches it
    // (they take kNeedFull entries only, and test that before they read an offset) - ADVICE r3
    if (src[i] < 0 || dst[i] < 0 || (src[i] >> 1) >= num_reads || (dst[i] >> 1) >= num_reads) {
        dist_out[i] = -1;
        return;
    }
    int m, n, ulen, vlen;
    overlap_shape(read_off, src[i], dst[i], ol[i], m, n, ulen, vlen);
    dist_out[i] = (m == 0 || n == 0) ? max(m, n) : kNeedFull;   // an empty side: the distance is the other side's length
}

__global__ void k_overlap_finish(int64_t E, int32_t* __restrict__ dist_out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < E && dist_out[i] == kNeedFull) dist_out[i] = -1;   // no kernel could take it (too long / alphabet too large for LDS): reported, not guessed
}

template <int NP>
__global__ __launch_bounds__(64) void k_overlap_banded(const uint8_t* __restrict__ reads, const int64_t* __restrict__ read_off,
                                                       const uint8_t* __restrict__ symtab, const int nsym, const int32_t* __restrict__ src,
                                                       const int32_t* __restrict__ dst, const int32_t* __restrict__ ol, int64_t E,
                                                       int* __restrict__ ticket, int32_t* __restrict__ dist_out) {
    __shared__ uint8_t st[512];
    const int lane = threadIdx.x;
    // entries >= num_symbols are read as num_symbols - 1, as the header says and the full-matrix kernel does: a raw entry > 15 would spill
    // into the neighbouring 4-bit nibbles of the packed target symbols
    for (int i = lane; i < 512; i += 64) st[i] = (uint8_t)min((int)symtab[i], nsym - 1);
    __syncthreads();
    for (;;) {
        int first = 0;
        if (lane == 0) first = atomicAdd(ticket, 64);
        first = __builtin_amdgcn_readfirstlane(first);
        if (first >= E) break;
        const int64_t ed = (int64_t)first + lane;
        if (ed >= E || dist_out[ed] != kNeedFull) continue;
        const int u = src[ed], v = dst[ed];
        int m, n, ulen, vlen;
        overlap_shape(read_off, u, v, ol[ed], m, n, ulen, vlen);
        // the band: diagonals c - r in [lo, hi] = [min(0, n-m) - k, max(0, n-m) + k]; in chunk q (columns 32 q .. 32 q + 31) the
        // window holds the blocks q - U .. q - U + 7, which covers the band when U >= ceil(hi / 32) and 7 - U >= floor((31 - lo) / 32)
        const int delta = n - m, a = max(delta, 0), bneg = -min(delta, 0);
        int k = -1, U = 0;
        for (int kk = 96; kk >= 16; kk -= 16) {
            const int uu = (a + kk + 31) / 32, dn = (31 + kk + bneg) / 32;
            if (uu + dn <= kBandW - 1) { k = kk, U = uu; break; }
        }
        if (k < 0) continue;   // the lengths differ by too much for this window: the full kernel
        const int64_t uo = read_off[u >> 1], vo = read_off[v >> 1];
        const bool urc = u & 1, vrc = v & 1;
        const uint8_t* stu = st + (urc ? 256 : 0);
        const uint8_t* stv = st + (vrc ? 256 : 0);
        const int bmax = (m - 1) / 32;
        uint32_t Pv[kBandW], Mv[kBandW], pl[NP][kBandW];
        // the NP bit planes of query block b (rows past the end of the query keep plane bits 0: whatever they match, rows above
        // them do not depend on it, and their vertical deltas are taken off the score at the end)
        auto load_block = [&](int b, uint32_t (&out)[NP]) {
#pragma unroll
            for (int j = 0; j < NP; ++j) out[j] = 0u;
            const int base = 32 * b;
            uint8_t raw[32];
#pragma unroll
            for (int bit = 0; bit < 32; ++bit) {
                const int p = ulen - m + min(base + bit, m - 1);
                raw[bit] = urc ? reads[uo + (ulen - 1 - p)] : reads[uo + p];
            }
#pragma unroll
            for (int bit = 0; bit < 32; ++bit) {
                const uint32_t sy = base + bit < m ? stu[raw[bit]] : 0u;
#pragma unroll
                for (int j = 0; j < NP; ++j) out[j] |= ((sy >> j) & 1u) << bit;
            }
        };
        int score = 0;
#pragma unroll
        for (int i = 0; i < kBandW; ++i) {
            Pv[i] = 0xFFFFFFFFu, Mv[i] = 0u;
            const int b = i - U;
            uint32_t t[NP];
#pragma unroll
            for (int j = 0; j < NP; ++j) t[j] = 0u;
            if (b >= 0 && b <= bmax) {
                load_block(b, t);
                score += 32;   // column 0 of the NW matrix: D[i][0] = i
            }
#pragma unroll
            for (int j = 0; j < NP; ++j) pl[j][i] = t[j];
        }
        const int chunks = (n + 31) / 32;
        for (int q = 0; q < chunks; ++q) {
            const int fb = q - U;   // block at window index 0
            if (q > 0) {
#pragma unroll
                for (int i = 0; i + 1 < kBandW; ++i) {
                    Pv[i] = Pv[i + 1], Mv[i] = Mv[i + 1];
#pragma unroll
                    for (int j = 0; j < NP; ++j) pl[j][i] = pl[j][i + 1];
                }
                const int b = fb + kBandW - 1;
                uint32_t t[NP];
#pragma unroll
                for (int j = 0; j < NP; ++j) t[j] = 0u;
                Pv[kBandW - 1] = 0xFFFFFFFFu, Mv[kBandW - 1] = 0u;
                if (b >= 0 && b <= bmax) {
                    load_block(b, t);
                    score += 32;   // the block enters one column late as an upper bound: every row + 1 on the row above
                }
#pragma unroll
                for (int j = 0; j < NP; ++j) pl[j][kBandW - 1] = t[j];
            }
            const int il = min(kBandW - 1, bmax - fb);   // window index of the lowest block that exists: its bottom row carries `score`
            // this chunk's 32 target symbols, 4 bits each
            uint32_t tsym[4] = {0u, 0u, 0u, 0u};
            {
                uint8_t raw[32];
#pragma unroll
                for (int cc = 0; cc < 32; ++cc) {
                    const int j = min(32 * q + cc, n - 1);
                    raw[cc] = vrc ? reads[vo + (vlen - 1 - j)] : reads[vo + j];
                }
#pragma unroll
                for (int cc = 0; cc < 32; ++cc) tsym[cc >> 3] |= (uint32_t)stv[raw[cc]] << (4 * (cc & 7));
            }
            const int cols = min(32, n - 32 * q);
            for (int cc = 0; cc < cols; ++cc) {
                const uint32_t sy = (tsym[cc >> 3] >> (4 * (cc & 7))) & 15u;
                uint32_t inv[NP];   // ~(bit j of the column's symbol, spread over the word): plane ^ inv has a 1 where the plane bit EQUALS it
#pragma unroll
                for (int j = 0; j < NP; ++j) inv[j] = ((sy >> j) & 1u) - 1u;
                int h = 1;   // the first block of the column: row 0 of the NW matrix (+1 per column), or an upper bound below it
#pragma unroll
                for (int i = 0; i < kBandW; ++i) {
                    if (fb + i >= 0 && i <= il) {
                        uint32_t eq = pl[0][i] ^ inv[0];
#pragma unroll
                        for (int j = 1; j < NP; ++j) eq &= pl[j][i] ^ inv[j];
                        h = myers_block(Pv[i], Mv[i], eq, h);
                        if (i == il) score += h;
                    }
                }
            }
        }
        // D[m][n]: take the vertical deltas of the rows below the query's last one off the bottom of block bmax
        const int ib = bmax - (chunks - 1 - U);
        int result = kNeedFull;
        if (ib >= 0 && ib < kBandW) {
            uint32_t pv = 0u, mv = 0u;
#pragma unroll
            for (int i = 0; i < kBandW; ++i)
                if (i == ib) pv = Pv[i], mv = Mv[i];
            const int used = m - 32 * bmax;   // rows of the last block that belong to the query (1..32)
            const uint32_t mask = used >= 32 ? 0u : (0xFFFFFFFFu << used);
            const int d = score - (__popc(pv & mask) - __popc(mv & mask));
            if (d <= k) result = d;   // inside the band: exact.  Otherwise only an upper bound: the full kernel decides
        }
        dist_out[ed] = result;
        if (result != kNeedFull) atomicAdd(ticket + 1, 1);   // workspace int 11: overlaps the band settled (statistics only)
    }
}

// One workgroup = one wave.  Every class's launch walks the whole edge list through its own ticket counter and takes the
// edges of its class (a skipped edge costs three integer loads); no host-side sort, no host sync.
template <int B>
__global__ __launch_bounds__(64) void k_overlap_edit_distance(const uint8_t* __restrict__ reads, const int64_t* __restrict__ read_off,
                                                              const uint8_t* __restrict__ symtab, int nsym, const int32_t* __restrict__ src,
                                                              const int32_t* __restrict__ dst, const int32_t* __restrict__ ol, int64_t E,
                                                              int* __restrict__ ticket, int32_t* __restrict__ dist_out) {
    extern __shared__ uint32_t lds[];
    uint32_t* peq = lds;                                            // [nsym][B][64]
    uint8_t* st = reinterpret_cast<uint8_t*>(lds + nsym * B * 64);  // [512]: symbol of byte b, symbol of complement(b)
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) st[i] = (uint8_t)min((int)symtab[i], nsym - 1);   // (an entry >= nsym would index past the mask table)
    __syncthreads();
    for (;;) {
        int first = 0;
        if (lane == 0) first = atomicAdd(ticket, 64);
        first = __builtin_amdgcn_readfirstlane(first);
        if (first >= E) break;
        // 64 candidate edges per ticket: each lane classifies one, the wave then runs the ones of this class in turn
        const int64_t cand = (int64_t)first + lane;
        int mine_m = 0, mine_n = 0;
        bool take = false;
        if (cand < E && dist_out[cand] == kNeedFull) {   // (the mark first: an edge with an endpoint out of range never reaches read_off)
            const int L = ol[cand], u = src[cand], v = dst[cand];
            const int ulen = (int)(read_off[(u >> 1) + 1] - read_off[u >> 1]);
            const int vlen = (int)(read_off[(v >> 1) + 1] - read_off[v >> 1]);
            mine_m = max(min(L, ulen), 0);   // read_src[-ol:] is the whole read when ol exceeds its length
            mine_n = max(min(L, vlen), 0);
            // (empty sides were answered by k_overlap_prepare, overlaps inside the band by k_overlap_banded; a query longer than
            //  65 536 rows has no class and keeps its mark, which k_overlap_finish turns into -1: reported, not guessed)
            take = mine_m > 0 && mine_n > 0 && class_of(mine_m) == B && dist_out[cand] == kNeedFull;
        }
        unsigned long long todo = __ballot(take);
        while (todo) {
            const int who = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int64_t ed = (int64_t)first + who;
            const int m = __shfl(mine_m, who), n = __shfl(mine_n, who);
            const int u = src[ed], v = dst[ed];
            const int64_t uo = read_off[u >> 1], vo = read_off[v >> 1];
            const int ulen = (int)(read_off[(u >> 1) + 1] - uo), vlen = (int)(read_off[(v >> 1) + 1] - vo);
            const bool urc = u & 1, vrc = v & 1;
            const uint8_t* stu = st + (urc ? 256 : 0);
            const uint8_t* stv = st + (vrc ? 256 : 0);
            // ---- match masks of this lane's rows.  query[r] = read_src[ulen - m + r]
            for (int i = 0; i < nsym * B; ++i) peq[i * 64 + lane] = 0;
            const int row0 = lane * B * 32;
            for (int k = 0; k < B; ++k) {
                const int base = row0 + k * 32;
                if (base >= m) break;
                const int cnt = min(32, m - base);
                for (int bit = 0; bit < cnt; ++bit) {
                    const int p = ulen - m + base + bit;
                    const uint8_t byte = urc ? reads[uo + (ulen - 1 - p)] : reads[uo + p];
                    peq[(stu[byte] * B + k) * 64 + lane] |= 1u << bit;
                }
            }
            // ---- the skewed wavefront over the target columns
            uint32_t Pv[B], Mv[B];
#pragma unroll
            for (int k = 0; k < B; ++k) Pv[k] = 0xFFFFFFFFu, Mv[k] = 0u;
            int score = (lane + 1) * B * 32;   // D[bottom row of this lane][column 0]
            const int nl = (m + B * 32 - 1) / (B * 32);
            const int steps = n + nl - 1;
            auto target_sym = [&](int j) -> uint32_t {
                if (j >= n) return 0u;
                return stv[vrc ? reads[vo + (vlen - 1 - j)] : reads[vo + j]];
            };
            uint32_t tbuf = target_sym(lane), tnext = target_sym(64 + lane);
            uint32_t out_prev = 0;
            for (int t = 0; t < steps; ++t) {
                if ((t & 63) == 0 && t > 0) {
                    tbuf = tnext;
                    tnext = target_sym(t + 64 + lane);
                }
                const uint32_t fresh = (uint32_t)__builtin_amdgcn_readlane((int)tbuf, t & 63);
                uint32_t in = (uint32_t)__shfl_up((int)out_prev, 1);
                if (lane == 0) in = 2u | (fresh << 2);   // row 0 of the NW matrix grows by one per column: hin = +1
                const int c = t - lane;
                uint32_t out = in;
                if (c >= 0 && c < n && lane < nl) {
                    int h = (int)(in & 3u) - 1;
                    const uint32_t s = in >> 2;
                    const uint32_t* pe = peq + (s * B) * 64 + lane;
#pragma unroll
                    for (int k = 0; k < B; ++k) h = myers_block(Pv[k], Mv[k], pe[k * 64], h);
                    score += h;
                    out = (uint32_t)(h + 1) | (s << 2);
                }
                out_prev = out;
            }
            // ---- D[m][n]: the owner of the last query row removes the padding rows below it
            if (lane == (m - 1) / (B * 32)) {
                int excess = 0;
#pragma unroll
                for (int k = 0; k < B; ++k) {
                    const int base = row0 + k * 32;
                    if (base + 32 > m) {
                        const uint32_t mask = base >= m ? 0xFFFFFFFFu : (0xFFFFFFFFu << (m - base));
                        excess += __popc(Pv[k] & mask) - __popc(Mv[k] & mask);
                    }
                }
                dist_out[ed] = score - excess;
            }
        }
    }
}

template <int B>
int launch_class(const uint8_t* reads, const int64_t* read_off, const uint8_t* symtab, int nsym, const int32_t* src, const int32_t* dst,
                 const int32_t* ol, int64_t E, int* ticket, int32_t* dist, int grid, hipStream_t s) {
    const size_t lds = (size_t)nsym * B * 64 * 4 + 512;
    GN_REQUIRE(lds <= 160 * 1024, "overlap_edit_distance: %d symbols x %d blocks per lane need %zu bytes of LDS", nsym, B, lds);
    if (lds > 64 * 1024)
        GN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_overlap_edit_distance<B>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_overlap_edit_distance<B>), dim3((unsigned)grid), dim3(64), lds, s, reads, read_off, symtab, nsym, src, dst, ol, E, ticket, dist);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

__global__ void k_similarity(const int32_t* __restrict__ dist, const int32_t* __restrict__ ol, int64_t E, float* __restrict__ sim) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= E) return;
    // graph_parser.py:108-113: 1 - edit_distance / ol_length in Python floats (doubles), 0.5 for a zero-length overlap
    sim[i] = ol[i] > 0 ? (float)(1.0 - (double)dist[i] / (double)ol[i]) : 0.5f;
}

}  // namespace
}  // namespace gnnome

extern "C" int gnnome_overlap_workspace_bytes(size_t* bytes_host) {
    using namespace gnnome;
    GN_REQUIRE(bytes_host, "overlap_workspace_bytes: null pointer");
    *bytes_host = 16 * sizeof(int);
    return GNNOME_OK;
}

extern "C" int gnnome_overlap_edit_distance(const uint8_t* reads, const int64_t* read_off, int64_t num_reads, const uint8_t* symtab,
                                            int num_symbols, const int32_t* src, const int32_t* dst, const int32_t* overlap_length,
                                            int64_t num_edges, int32_t* dist_out, float* similarity_out, void* workspace,
                                            size_t workspace_bytes, void* stream) {
    using namespace gnnome;
    GN_REQUIRE(num_edges >= 0 && num_reads >= 0, "overlap_edit_distance: negative size");
    if (num_edges == 0) return GNNOME_OK;
    GN_REQUIRE(reads && read_off && symtab && src && dst && overlap_length && dist_out && workspace, "overlap_edit_distance: null pointer");
    GN_REQUIRE(num_symbols >= 1 && num_symbols <= kMaxSyms, "overlap_edit_distance: %d symbols (1..%d supported)", num_symbols, kMaxSyms);
    GN_REQUIRE(workspace_bytes >= 16 * sizeof(int), "overlap_edit_distance: workspace too small");
    // (every wave draws one more 64-edge ticket after the list is exhausted: the int counter must have room for that)
    GN_REQUIRE(num_edges < (1ll << 31) - 64 - (int64_t)persistent_grid() * 32 * 64, "overlap_edit_distance: too many edges");
    hipStream_t s = (hipStream_t)stream;
    int* tickets = reinterpret_cast<int*>(workspace);
    GN_HIP(hipMemsetAsync(tickets, 0, 16 * sizeof(int), s));
    const unsigned eb = (unsigned)((num_edges + 255) / 256);
    hipLaunchKernelGGL(k_overlap_prepare, dim3(eb), dim3(256), 0, s, read_off, num_reads, src, dst, overlap_length, num_edges, dist_out);
    GN_LAUNCH_CHECK();
    if (tuning(kTuneOverlapBand) != 1 && num_symbols <= 16) {   // (key 9 = 1: full-matrix kernels only, for A/B runs and cross-checks)
        const int bgrid = (int)std::min<int64_t>((num_edges + 63) / 64, (int64_t)persistent_grid() * 32);
        if (num_symbols <= 4)
            hipLaunchKernelGGL((k_overlap_banded<2>), dim3((unsigned)bgrid), dim3(64), 0, s, reads, read_off, symtab, num_symbols, src, dst, overlap_length, num_edges, tickets + 10, dist_out);
        else if (num_symbols <= 8)
            hipLaunchKernelGGL((k_overlap_banded<3>), dim3((unsigned)bgrid), dim3(64), 0, s, reads, read_off, symtab, num_symbols, src, dst, overlap_length, num_edges, tickets + 10, dist_out);
        else
            hipLaunchKernelGGL((k_overlap_banded<4>), dim3((unsigned)bgrid), dim3(64), 0, s, reads, read_off, symtab, num_symbols, src, dst, overlap_length, num_edges, tickets + 10, dist_out);
        GN_LAUNCH_CHECK();
    }
    // persistent waves: enough to fill every SIMD several times over (the kernel is VALU-bound, 8 waves per SIMD hide the
    // cross-lane and LDS latencies of one another), never more than there are 64-edge tickets
    const int grid = (int)std::min<int64_t>((num_edges + 63) / 64, (int64_t)persistent_grid() * 16);
    int rc = GNNOME_OK;
#define GN_CLASS(IDX, B_)                                                                                                          \
    if (rc == GNNOME_OK && (size_t)num_symbols * B_ * 64 * 4 + 512 <= 160 * 1024)                                                  \
        rc = launch_class<B_>(reads, read_off, symtab, num_symbols, src, dst, overlap_length, num_edges, tickets + IDX, dist_out, grid, s)
    // (a class whose masks do not fit LDS for this alphabet is not launched: its edges keep the caller's fill value, -1)
    GN_CLASS(9, 32);   // the longest first: they set the tail
    GN_CLASS(8, 24);
    GN_CLASS(7, 16);
    GN_CLASS(6, 12);
    GN_CLASS(5, 8);
    GN_CLASS(4, 6);
    GN_CLASS(3, 4);
    GN_CLASS(2, 3);
    GN_CLASS(1, 2);
    GN_CLASS(0, 1);
#undef GN_CLASS
    if (rc != GNNOME_OK) return rc;
    hipLaunchKernelGGL(k_overlap_finish, dim3(eb), dim3(256), 0, s, num_edges, dist_out);
    GN_LAUNCH_CHECK();
    if (similarity_out) {
        hipLaunchKernelGGL(k_similarity, dim3((unsigned)((num_edges + 255) / 256)), dim3(256), 0, s, dist_out, overlap_length, num_edges, similarity_out);
        GN_LAUNCH_CHECK();
    }
    return GNNOME_OK;
}
