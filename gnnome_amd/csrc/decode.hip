// Greedy decode of the edge scores into contig walks: what inference.py:99-164 does with Python
// dicts and sets - one greedy walk forwards from every sampled edge's head and one backwards (forwards on the
// reverse-complement strand) from its tail - as ONE launch with one wavefront per candidate edge.
//
//   reference                                            here
//   greedy_forwards        inference.py:70-114           walk<>() from dst
//   greedy_backwards_rc    inference.py:117-158          walk<>() from src ^ 1, afterwards (it must not enter what the
//                                                        forward walk visited: run_greedy_both_ways, :161-165)
//   get_contig_length      inference.py:29-36            prefix lengths summed along the way (mate edges for the backward half)
//   best walk, transitive nodes  inference.py:304-323    gnnome_mark_walk_visited
//
// A walk is a chain of dependent memory reads (successor list -> visited flags + log-probabilities of the successors
// -> next node); what a GPU can add is width, not speed per step: the 64 lanes of a wave test and rank a node's
// successors at once, and all candidates (hyperparameters.py:47, num_decoding_paths = 100) walk concurrently.  State
// per candidate: one bit per node (its own visited set, inference.py:76 / :123), kept in HBM - 100 candidates x 10M
// nodes = 125 MB - and the walk itself, written as it grows.  Node pairs (2r, 2r+1) are a read and its reverse
// complement (graph_parser.py:174-181), so "x ^ 1" is the mate, as in the reference.
//
// Ties.  Log-probabilities saturate at exactly 0 for scores above ~17, so exact ties for the maximum are real, and which
// of the tied successors the reference takes is whatever torch.topk(k=1) returns: on the CPU that is
// std::nth_element(begin, begin, end, greater) of libstdc++ over the (value, index) pairs of the UNVISITED successors in
// list order (edge-id order, graph_parser.py:31-37) - an introselect whose answer among equals depends on the whole
// sequence (first maximum for up to three candidates, e.g. the fourth of [0, 0, -.2, 0, 0, -.7, 0]); from 64 candidates
// up torch switches to std::partial_sort, which keeps the first maximum.  A unique maximum is found by a wave reduction;
// on a tie among fewer than 64 candidates lane 0 replays that introselect over the candidates in LDS (nth0_libstdcxx;
// checked against torch.topk on 40k random tied lists), so the walks are the reference's walks on ties as well
// (tests/test_decode.py, golden case "ties").
#include "common.h"

namespace gnnome {

struct WalkGraph {
    const int32_t* succ_ptr;    // [N+1]  successors of node u: slots succ_ptr[u] .. succ_ptr[u+1], in edge-id order
    const int32_t* succ_nbr;    // [E]
    const int32_t* succ_eid;    // [E]    edge id of the slot
    const float* logp;          // [E]    log(sigmoid(score)) by edge id (inference.py:184)
    const int32_t* prefix_len;  // [E]    by edge id
    const int32_t* read_len;    // [N]
    const uint8_t* visited;     // [N]    nodes consumed by earlier contigs (inference.py:173, :334)
    int64_t num_nodes;
};

constexpr int kTieCap = 64;   // torch.topk(k = 1) takes std::nth_element below 64 candidates, std::partial_sort (= first maximum) from 64 up

// libstdc++'s std::nth_element(first, first, last, comp) with comp(a, b) = a.value > b.value on (value, position) pairs:
// __introselect (bits/stl_algo.h) with __move_median_to_first, __unguarded_partition, __heap_select, __insertion_sort,
// for nth = first.  Afterwards element 0 is what torch.topk(k = 1) returns on the CPU.  One lane, sequential, on LDS.
__device__ void nth0_libstdcxx(float* v, int* p, int n) {
    auto swp = [&](int a, int b) {
        const float tv = v[a];
        const int tp = p[a];
        v[a] = v[b], p[a] = p[b];
        v[b] = tv, p[b] = tp;
    };
    int first = 0, last = n;
    int depth = 0;
    for (int m = n; m > 1; m >>= 1) ++depth;
    depth *= 2;
    while (last - first > 3) {
        if (depth == 0) {   // __heap_select(first, first + 1, last) + iter_swap(first, nth): a heap of one element
            for (int i = first + 1; i < last; ++i)
                if (v[i] > v[first]) swp(i, first);
            return;
        }
        --depth;
        const int mid = first + (last - first) / 2, a = first + 1, b = mid, c = last - 1;
        if (v[a] > v[b]) {              // __move_median_to_first(first, a, b, c)
            if (v[b] > v[c]) swp(first, b);
            else if (v[a] > v[c]) swp(first, c);
            else swp(first, a);
        } else if (v[a] > v[c]) swp(first, a);
        else if (v[b] > v[c]) swp(first, c);
        else swp(first, b);
        int lo = first + 1, hi = last;  // __unguarded_partition(first + 1, last, pivot = first)
        for (;;) {
            while (v[lo] > v[first]) ++lo;
            --hi;
            while (v[first] > v[hi]) --hi;
            if (!(lo < hi)) break;
            swp(lo, hi);
            ++lo;
        }
        last = lo;                      // nth = the first element: always the left part
    }
    for (int i = first + 1; i < last; ++i) {   // __insertion_sort(first, last)
        const float tv = v[i];
        const int tp = p[i];
        if (tv > v[first]) {
            for (int j = i; j > first; --j) v[j] = v[j - 1], p[j] = p[j - 1];
            v[first] = tv, p[first] = tp;
        } else {
            int j = i;
            while (tv > v[j - 1]) {
                v[j] = v[j - 1], p[j] = p[j - 1];
                --j;
            }
            v[j] = tv, p[j] = tp;
        }
    }
}

__device__ __forceinline__ bool bit_test(const uint32_t* bits, int node) {
    // written by this wave's lane 0 with device-scope atomics: read past the (non-coherent) vector L1
    return (__hip_atomic_load(bits + (node >> 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (node & 31)) & 1u;
}

// One greedy walk by one wave (inference.py:70-114; the backward walk :117-158 is the same loop started at src ^ 1).
// Returns the number of nodes written to `out`; `sum` accumulates the chosen log-probabilities in fp32 in walk order
// like the reference's `sumLogProb +=`, `plen` the prefix lengths of the contig's edges: of the edges walked (forward),
// or of their mates (nb ^ 1 -> cur ^ 1) when MATES (the backward half is reversed and complemented afterwards, :157).
template <bool MATES>
__device__ int greedy_walk(const WalkGraph& g, uint32_t* bits, int start, int s, int d, int32_t* out, int cap, float& sum,
                           int64_t& plen, int& status, int& last, float* tie_v, int* tie_p) {
    const int lane = threadIdx.x & 63;
    int current = start, len = 0;
    last = start;
    for (;;) {
        if (len >= cap) {
            status |= 1;
            break;
        }
        if (lane == 0) {
            out[len] = current;
            const uint32_t a = atomicOr(bits + (current >> 5), 1u << (current & 31));
            const uint32_t b = atomicOr(bits + ((current ^ 1) >> 5), 1u << ((current ^ 1) & 31));
            asm volatile("" ::"v"(a), "v"(b));   // returned atomics: complete before the reads below
        }
        ++len;
        last = current;
        const int row = g.succ_ptr[current], deg = g.succ_ptr[current + 1] - row;
        if (deg == 0) break;
        auto excluded = [&](int n) {
            return g.visited[n] != 0 || n == s || n == (s ^ 1) || n == d || n == (d ^ 1) || bit_test(bits, n);
        };
        int next, eid;
        if (deg == 1) {          // :83-90 - a single successor is taken without ranking
            next = g.succ_nbr[row];
            if (excluded(next)) break;
            eid = g.succ_eid[row];
            sum += g.logp[eid];
        } else {                 // :91-111 - the unvisited successor with the highest log-probability (torch.topk, k = 1)
            float best = -INFINITY;
            int best_pos = 0x7fffffff, n_cand = 0, n_best = 0;
            for (int base = 0; base < deg; base += 64) {
                const int i = base + lane;
                const bool ok = i < deg && !excluded(g.succ_nbr[row + i]);
                const float mine = ok ? g.logp[g.succ_eid[row + i]] : -INFINITY;
                // the unvisited successors, in list order, into LDS (only read if the maximum turns out to be tied)
                const unsigned long long mask = __ballot(ok);
                const int slot = n_cand + __popcll(mask & ((1ull << lane) - 1ull));
                if (ok && slot < kTieCap) {
                    tie_v[slot] = mine;
                    tie_p[slot] = i;
                }
                n_cand += __popcll(mask);
                float v = mine;
                int pos = ok ? i : 0x7fffffff;
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) {
                    const float ov = __shfl_xor(v, m);
                    const int op = __shfl_xor(pos, m);
                    if (op != 0x7fffffff && (pos == 0x7fffffff || ov > v || (ov == v && op < pos))) {
                        v = ov;
                        pos = op;
                    }
                }
                if (pos != 0x7fffffff) {
                    const int ties_here = __popcll(__ballot(ok && mine == v));
                    if (best_pos == 0x7fffffff || v > best) {
                        best = v;
                        best_pos = pos;
                        n_best = ties_here;
                    } else if (v == best) {
                        n_best += ties_here;
                    }
                }
            }
            if (best_pos == 0x7fffffff) break;   // :93-94 every successor is visited
            if (n_best > 1 && n_cand < kTieCap) {
                __syncthreads();                 // (one wave per block) the LDS writes above are in place
                if (lane == 0) nth0_libstdcxx(tie_v, tie_p, n_cand);
                __syncthreads();
                best_pos = tie_p[0];
                __syncthreads();                 // before the next step refills the arrays
            }   // 64 candidates or more: torch's k * 64 <= n branch, std::partial_sort, keeps the FIRST maximum = best_pos
            next = g.succ_nbr[row + best_pos];
            eid = g.succ_eid[row + best_pos];
            sum += best;
        }
        if (!MATES) {
            plen += g.prefix_len[eid];
        } else {
            // the contig will contain (next ^ 1) -> (current ^ 1): find that edge among the successors of next ^ 1
            const int u = next ^ 1, w = current ^ 1;
            const int r2 = g.succ_ptr[u], d2 = g.succ_ptr[u + 1] - r2;
            int found = -1;
            for (int base = 0; base < d2; base += 64) {
                const int i = base + lane;
                int hit = (i < d2 && g.succ_nbr[r2 + i] == w) ? i : -1;
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) hit = max(hit, __shfl_xor(hit, m));   // the edges dict keeps the LAST id of a pair (:77-80)
                found = max(found, hit);
            }
            if (found >= 0) {
                plen += g.prefix_len[g.succ_eid[r2 + found]];
            } else {
                status |= 2;   // no mate edge: the reference's DGL lookup would raise here
            }
        }
        current = next;
    }
    return len;
}

// One wave per candidate start edge (src, dst, eid): inference.py:161-165 (run_greedy_both_ways) + :29-36.
__global__ __launch_bounds__(64) void k_greedy_walks(WalkGraph g, const int32_t* __restrict__ cand_src, const int32_t* __restrict__ cand_dst,
                                                     const int32_t* __restrict__ cand_eid, int num_cand, uint32_t* bitmaps,
                                                     int64_t words_per_cand, int32_t* walks_f, int32_t* walks_b, int64_t cap,
                                                     int32_t* len_f, int32_t* len_b, float* sum_f, float* sum_b,
                                                     int64_t* contig_len, int32_t* status_out) {
    const int c = blockIdx.x;
    if (c >= num_cand) return;
    const int s = cand_src[c], d = cand_dst[c];
    uint32_t* bits = bitmaps + (int64_t)c * words_per_cand;
    float sf = 0.f, sb = 0.f;
    int64_t plen = g.prefix_len[cand_eid[c]];   // the sampled edge itself joins the two halves
    int status = 0;
    __shared__ float tie_v[kTieCap];
    __shared__ int tie_p[kTieCap];
    int last = d, last_b = 0;
    const int lf = greedy_walk<false>(g, bits, d, s, d, walks_f + (int64_t)c * cap, (int)cap, sf, plen, status, last, tie_v, tie_p);
    const int lb = greedy_walk<true>(g, bits, s ^ 1, s, d, walks_b + (int64_t)c * cap, (int)cap, sb, plen, status, last_b, tie_v, tie_p);
    if ((threadIdx.x & 63) == 0) {
        len_f[c] = lf;
        len_b[c] = lb;
        sum_f[c] = sf;
        sum_b[c] = sb;
        contig_len[c] = plen + g.read_len[last];
        status_out[c] = status;
    }
}

// After a walk has been chosen (inference.py:304-334): every node of the walk and its mate, plus the nodes the walk
// jumped over - for consecutive (ss, dd): succs[ss] & preds[dd], and their mates (:313-318) - become visited.
// One wave per consecutive pair; walk[] is the contig (walk_b reversed and complemented, then walk_f).
__global__ __launch_bounds__(64) void k_mark_walk_visited(const int32_t* __restrict__ succ_ptr, const int32_t* __restrict__ succ_nbr,
                                                          const int32_t* __restrict__ walk, int64_t len, uint8_t* visited) {
    const int64_t i = blockIdx.x;
    const int lane = threadIdx.x & 63;
    if (i >= len) return;
    const int ss = walk[i];
    if (lane == 0) {
        visited[ss] = 1;
        visited[ss ^ 1] = 1;
    }
    if (i + 1 >= len) return;
    const int dd = walk[i + 1];
    const int row = succ_ptr[ss], deg = succ_ptr[ss + 1] - row;
    for (int j = 0; j < deg; ++j) {   // t in succs[ss]: is dd among the successors of t (t in preds[dd])?
        const int t = succ_nbr[row + j];
        const int r2 = succ_ptr[t], d2 = succ_ptr[t + 1] - r2;
        bool hit = false;
        for (int base = 0; base < d2; base += 64) hit |= (base + lane < d2) && succ_nbr[r2 + base + lane] == dd;
        if (__any(hit) && lane == 0) {
            visited[t] = 1;
            visited[t ^ 1] = 1;
        }
    }
}

}  // namespace gnnome

extern "C" int gnnome_greedy_walks_workspace_bytes(int64_t num_nodes, int num_candidates, size_t* bytes_host) {
    using namespace gnnome;
    GN_REQUIRE(num_nodes >= 0 && num_candidates >= 0 && bytes_host, "greedy_walks_workspace_bytes: bad arguments");
    *bytes_host = (size_t)num_candidates * (size_t)((num_nodes + 31) / 32) * sizeof(uint32_t) + 16;
    return GNNOME_OK;
}

extern "C" int gnnome_greedy_walks(const int32_t* succ_ptr, const int32_t* succ_nbr, const int32_t* succ_eid, const float* logp,
                                   const int32_t* prefix_len, const int32_t* read_len, const uint8_t* visited, int64_t num_nodes,
                                   const int32_t* cand_src, const int32_t* cand_dst, const int32_t* cand_eid, int num_candidates,
                                   int32_t* walks_f, int32_t* walks_b, int64_t capacity, int32_t* len_f, int32_t* len_b, float* sum_f,
                                   float* sum_b, int64_t* contig_len, int32_t* status, void* workspace, size_t workspace_bytes,
                                   void* stream) {
    using namespace gnnome;
    GN_REQUIRE(num_nodes >= 0 && num_candidates >= 0 && capacity >= 1, "greedy_walks: bad sizes");
    if (num_candidates == 0) return GNNOME_OK;
    GN_REQUIRE(num_nodes % 2 == 0, "greedy_walks: nodes come in (read, reverse complement) pairs, N must be even");
    GN_REQUIRE(succ_ptr && succ_nbr && succ_eid && logp && prefix_len && read_len && visited && cand_src && cand_dst && cand_eid && walks_f &&
                   walks_b && len_f && len_b && sum_f && sum_b && contig_len && status && workspace,
               "greedy_walks: null pointer");
    const int64_t words = (num_nodes + 31) / 32;
    GN_REQUIRE(workspace_bytes >= (size_t)num_candidates * (size_t)words * sizeof(uint32_t), "greedy_walks: workspace too small");
    GN_REQUIRE(capacity < (1ll << 31), "greedy_walks: capacity must fit an int32");
    hipStream_t s = (hipStream_t)stream;
    GN_HIP(hipMemsetAsync(workspace, 0, (size_t)num_candidates * (size_t)words * sizeof(uint32_t), s));
    WalkGraph g{succ_ptr, succ_nbr, succ_eid, logp, prefix_len, read_len, visited, num_nodes};
    hipLaunchKernelGGL(k_greedy_walks, dim3((unsigned)num_candidates), dim3(64), 0, s, g, cand_src, cand_dst, cand_eid, num_candidates,
                       (uint32_t*)workspace, words, walks_f, walks_b, capacity, len_f, len_b, sum_f, sum_b, contig_len, status);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

extern "C" int gnnome_mark_walk_visited(const int32_t* succ_ptr, const int32_t* succ_nbr, const int32_t* walk, int64_t walk_len,
                                        uint8_t* visited, void* stream) {
    using namespace gnnome;
    GN_REQUIRE(walk_len >= 0 && walk_len < (1ll << 31), "mark_walk_visited: bad length");
    if (walk_len == 0) return GNNOME_OK;
    GN_REQUIRE(succ_ptr && succ_nbr && walk && visited, "mark_walk_visited: null pointer");
    hipLaunchKernelGGL(k_mark_walk_visited, dim3((unsigned)walk_len), dim3(64), 0, (hipStream_t)stream, succ_ptr, succ_nbr, walk, walk_len,
                       visited);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}
