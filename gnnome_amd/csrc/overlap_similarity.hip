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
    for (int i = lane; i < 128; i += 64) reinterpret_cast<uint32_t*>(st)[i] = reinterpret_cast<const uint32_t*>(symtab)[i];
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
        if (cand < E) {
            const int L = ol[cand], u = src[cand], v = dst[cand];
            const int ulen = (int)(read_off[(u >> 1) + 1] - read_off[u >> 1]);
            const int vlen = (int)(read_off[(v >> 1) + 1] - read_off[v >> 1]);
            mine_m = max(min(L, ulen), 0);   // read_src[-ol:] is the whole read when ol exceeds its length
            mine_n = max(min(L, vlen), 0);
            if (mine_m == 0 || mine_n == 0) {
                if (B == 1) dist_out[cand] = max(mine_m, mine_n);   // an empty side: the distance is the other side's length
            } else {
                const int cls = class_of(mine_m);
                if (cls == 0 && B == 32) dist_out[cand] = -1;       // longer than 65 536 rows: reported, not guessed
                take = cls == B;
            }
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
    GN_REQUIRE(num_edges < (1ll << 31) - 64, "overlap_edit_distance: too many edges");
    hipStream_t s = (hipStream_t)stream;
    int* tickets = reinterpret_cast<int*>(workspace);
    GN_HIP(hipMemsetAsync(tickets, 0, 16 * sizeof(int), s));
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
    if (similarity_out) {
        hipLaunchKernelGGL(k_similarity, dim3((unsigned)((num_edges + 255) / 256)), dim3(256), 0, s, dist_out, overlap_length, num_edges, similarity_out);
        GN_LAUNCH_CHECK();
    }
    return GNNOME_OK;
}
