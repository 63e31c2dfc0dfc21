// Multilevel k-way clustering for mini-batch training (train.py:333-346: dgl.metis_partition(g, k, extra_cached_hops) = METIS 5.1.0's
// k-way partitioner behind DGL 0.8.1, neither of them in this image).  The two inner loops of the published scheme
// (Karypis & Kumar, "A Fast and High Quality Multilevel Scheme for Partitioning Irregular Graphs", SIAM J. Sci. Comput. 20(1), 1998, section 3:
// heavy-edge matching; "Multilevel k-way Partitioning Scheme for Irregular Graphs", JPDC 48(1), 1998, section 4: greedy k-way refinement)
// as deterministic, atomic-free kernels over an undirected weighted CSR; the levels, the contraction (sort + segmented sums) and the
// move selection under the balance constraint are sequenced by gnnome_amd/partition.py.
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
// best_part[v] = -1: an interior vertex.  A vertex adjacent to more than kSlots distinct parts keeps the first kSlots it meets
// (assembly graphs: degree ~10-20).  Ties: the smaller part id.
constexpr int kSlots = 24;
__global__ __launch_bounds__(256) void k_kway_gains(const int32_t* __restrict__ ptr, const int32_t* __restrict__ adj, const int32_t* __restrict__ wgt,
                                                    const int32_t* __restrict__ label, int64_t n, int32_t* __restrict__ best_part,
                                                    int32_t* __restrict__ gain) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const int own = label[v];
    int parts[kSlots], conn[kSlots], used = 0, internal = 0;
    for (int k = ptr[v]; k < ptr[v + 1]; ++k) {
        const int u = adj[k];
        if (u == v) continue;
        const int p = label[u], w = wgt[k];
        if (p == own) { internal += w; continue; }
        int s = 0;
        while (s < used && parts[s] != p) ++s;
        if (s == used) {
            if (used == kSlots) continue;
            parts[used] = p, conn[used] = 0, ++used;
        }
        conn[s] += w;
    }
    int bp = -1, bc = -1;
    for (int s = 0; s < used; ++s)
        if (conn[s] > bc || (conn[s] == bc && parts[s] < bp)) bp = parts[s], bc = conn[s];
    best_part[v] = bp;
    gain[v] = bp < 0 ? 0 : bc - internal;
}

}  // namespace gnnome

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
