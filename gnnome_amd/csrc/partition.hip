// Multilevel k-way clustering for mini-batch training (train.py:333-346: dgl.metis_partition(g, k, extra_cached_hops) = METIS 5.1.0's
// k-way partitioner behind DGL 0.8.1, neither of them in this image).  The two inner loops of the published scheme
// (Karypis & Kumar, "A Fast and High Quality Multilevel Scheme for Partitioning Irregular Graphs", SIAM J. Sci. Comput. 20(1), 1998, section 3:
// heavy-edge matching; "Multilevel k-way Partitioning Scheme for Irregular Graphs", JPDC 48(1), 1998, section 4: greedy k-way refinement)
// as deterministic, atomic-free kernels over an undirected weighted CSR; the levels, the contraction (sort + segmented sums) and the
// move selection under the balance constraint are sequenced by gnnome_amd/partition.py.
#include <algorithm>
#include <vector>

#include "common.h"

namespace gnnome {

// Heavy-edge matching, one proposal round.  Every unmatched vertex v names its heaviest unmatched neighbour u with
// vwgt[v] + vwgt[u] <= max_vwgt (ties: the lighter vertex, then the smaller id); the host pairs v and u when they named each other
// (a handshake: no atomics, the result is a function of the graph alone).  proposal[v] = -1: nobody to name.
__global__ __launch_bounds__(256) void k_hem_propose(const int32_t* __restrict__ ptr, const int32_t* __restrict__ adj, const int32_t* __restrict__ wgt,
                                                     const int32_t* __restrict__ vwgt, const int32_t* __restrict__ match, int64_t n, int max_vwgt,
                                                     int32_t* __restrict__ proposal) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    int best = -1, best_w = -1, best_vw = 0;
    if (match[v] < 0) {
        const int vw = vwgt[v];
        for (int k = ptr[v]; k < ptr[v + 1]; ++k) {
            const int u = adj[k];
            if (u == v || match[u] >= 0) continue;
            const int uw = vwgt[u];
            if (vw + uw > max_vwgt) continue;
            const int w = wgt[k];
            if (w > best_w || (w == best_w && (uw < best_vw || (uw == best_vw && u < best)))) best = u, best_w = w, best_vw = uw;
        }
    }
    proposal[v] = best;
}

// Greedy k-way refinement, the gains of one pass: for every vertex the part among its neighbours' parts (other than its own) it is
// connected to most heavily, and gain = that connectivity - the connectivity to its own part (the cut shrinks by `gain` if v alone moves).
// best_part[v] = -1: an interior vertex.  Ties: the smaller part id.  A vertex adjacent to more than kSlots distinct parts (assembly graphs:
// degree ~10-20; coarse vertices and repeat hubs can exceed it) is evaluated EXACTLY by the storage-free form below - for every neighbouring
// part the row is scanned once more (round 6; rounds 4-5 dropped the parts beyond the first kSlots).
constexpr int kSlots = 24;
__global__ __launch_bounds__(256) void k_kway_gains(const int32_t* __restrict__ ptr, const int32_t* __restrict__ adj, const int32_t* __restrict__ wgt,
                                                    const int32_t* __restrict__ label, int64_t n, int32_t* __restrict__ best_part,
                                                    int32_t* __restrict__ gain) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const int own = label[v];
    int parts[kSlots], conn[kSlots], used = 0, internal = 0;
    bool overflow = false;
    for (int k = ptr[v]; k < ptr[v + 1]; ++k) {
        const int u = adj[k];
        if (u == v) continue;
        const int p = label[u], w = wgt[k];
        if (p == own) { internal += w; continue; }
        int s = 0;
        while (s < used && parts[s] != p) ++s;
        if (s == used) {
            if (used == kSlots) { overflow = true; break; }
            parts[used] = p, conn[used] = 0, ++used;
        }
        conn[s] += w;
    }
    int bp = -1, bc = -1;
    if (!overflow) {
        for (int s = 0; s < used; ++s)
            if (conn[s] > bc || (conn[s] == bc && parts[s] < bp)) bp = parts[s], bc = conn[s];
    } else {
        internal = 0;
        for (int k = ptr[v]; k < ptr[v + 1]; ++k) {
            const int u = adj[k];
            if (u == v) continue;
            const int p = label[u];
            if (p == own) { internal += wgt[k]; continue; }
            bool seen = false;   // this part's connectivity is summed at the FIRST neighbour that carries it
            for (int j = ptr[v]; j < k && !seen; ++j) seen = adj[j] != v && label[adj[j]] == p;
            if (seen) continue;
            int c = 0;
            for (int j = k; j < ptr[v + 1]; ++j)
                if (adj[j] != v && label[adj[j]] == p) c += wgt[j];
            if (c > bc || (c == bc && p < bp)) bp = p, bc = c;
        }
    }
    best_part[v] = bp;
    gain[v] = bp < 0 ? 0 : bc - internal;
}

}  // namespace gnnome

// The coarsest graph's initial partition on the HOST (a few thousand to a few ten thousand vertices), as METIS does it and as
// gnnome_amd/partition.py::_greedy_growing_py states it: k - 1 regions grown one after the other from the free vertex with the smallest id, always
// taking the free vertex most heavily connected to the region (ties: the one that became a neighbour of the region first), until the region holds
// its share of the weight; the rest is the last part.  Same labels as the Python form, bit for bit (tests/test_partition.py); it replaces a
// Python heap loop that took 0.3 s per call at k = 500.
extern "C" int gnnome_greedy_growing_host(const int32_t* ptr_host, const int32_t* adj_host, const int32_t* wgt_host, const int32_t* vwgt_host,
                                          int64_t num_vertices, int num_parts, int32_t* label_host) {
    using namespace gnnome;
    GN_REQUIRE(num_vertices >= 0 && num_parts >= 1, "greedy_growing: bad sizes");
    if (num_vertices == 0) return GNNOME_OK;
    GN_REQUIRE(ptr_host && vwgt_host && label_host, "greedy_growing: null pointer");
    const int64_t n = num_vertices;
    struct Item {
        long long conn;
        long long seen;
        int v;
    };
    auto worse = [](const Item& a, const Item& b) {   // max-heap on (conn, earlier first seen, smaller id): heapq's (-conn, first_seen, u) order
        if (a.conn != b.conn) return a.conn < b.conn;
        if (a.seen != b.seen) return a.seen > b.seen;
        return a.v > b.v;
    };
    long long total = 0, assigned = 0;
    for (int64_t v = 0; v < n; ++v) label_host[v] = -1, total += vwgt_host[v];
    std::vector<long long> conn(n, 0), first_seen(n, -1);
    std::vector<int> touched;
    int64_t next_free = 0;
    for (int p = 0; p < num_parts - 1; ++p) {
        const double target = (double)(total - assigned) / (double)(num_parts - p);
        std::vector<Item> heap;
        long long size = 0, tick = 0;
        for (int u : touched) conn[u] = 0, first_seen[u] = -1;
        touched.clear();
        while ((double)size < target) {
            while (!heap.empty() && (label_host[heap.front().v] >= 0 || heap.front().conn != conn[heap.front().v])) {
                std::pop_heap(heap.begin(), heap.end(), worse);
                heap.pop_back();
            }
            int v;
            if (!heap.empty()) {
                v = heap.front().v;
                std::pop_heap(heap.begin(), heap.end(), worse);
                heap.pop_back();
            } else {
                while (next_free < n && label_host[next_free] >= 0) ++next_free;
                if (next_free == n) break;
                v = (int)next_free;
            }
            label_host[v] = p;
            size += vwgt_host[v];
            for (int q = ptr_host[v]; q < ptr_host[v + 1]; ++q) {
                const int u = adj_host[q];
                if (label_host[u] >= 0) continue;
                if (first_seen[u] < 0) first_seen[u] = tick++, touched.push_back(u);
                conn[u] += wgt_host[q];
                heap.push_back({conn[u], first_seen[u], u});
                std::push_heap(heap.begin(), heap.end(), worse);
            }
        }
        assigned += size;
    }
    for (int64_t v = 0; v < n; ++v)
        if (label_host[v] < 0) label_host[v] = num_parts - 1;
    return GNNOME_OK;
}

extern "C" int gnnome_hem_propose(const int32_t* ptr, const int32_t* adj, const int32_t* wgt, const int32_t* vwgt, const int32_t* match,
                                  int64_t num_vertices, int max_vwgt, int32_t* proposal, void* stream) {
    using namespace gnnome;
    GN_REQUIRE(num_vertices >= 0 && num_vertices < (1ll << 31), "hem_propose: vertex count out of range");
    if (num_vertices == 0) return GNNOME_OK;
    GN_REQUIRE(ptr && vwgt && match && proposal, "hem_propose: null pointer");
    hipLaunchKernelGGL(k_hem_propose, dim3((unsigned)((num_vertices + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ptr, adj, wgt, vwgt, match,
                       num_vertices, max_vwgt, proposal);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

extern "C" int gnnome_kway_gains(const int32_t* ptr, const int32_t* adj, const int32_t* wgt, const int32_t* label, int64_t num_vertices,
                                 int32_t* best_part, int32_t* gain, void* stream) {
    using namespace gnnome;
    GN_REQUIRE(num_vertices >= 0 && num_vertices < (1ll << 31), "kway_gains: vertex count out of range");
    if (num_vertices == 0) return GNNOME_OK;
    GN_REQUIRE(ptr && label && best_part && gain, "kway_gains: null pointer");
    hipLaunchKernelGGL(k_kway_gains, dim3((unsigned)((num_vertices + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ptr, adj, wgt, label,
                       num_vertices, best_part, gain);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}
