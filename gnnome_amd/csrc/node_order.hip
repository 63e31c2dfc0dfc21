// Locality-restoring node order: the two device kernels behind gnnome_amd/node_order.py.
//
// Why: the destination-range partition (gnnome_amd/dist.py) and the L2 reuse of the aggregation both assume that node ids follow
// the layout - read r overlaps reads near r.  The reference numbers nodes in S-line order of the GFA (graph_parser.py:174-181),
// which need not be layout order: with shuffled ids 7/8 of the edges are cut at 8 ranks and every gather misses L2.  An overlap
// graph is locally transitive - if a overlaps b and both overlap c the three sit together on the genome - while repeat-induced
// edges join reads that share no neighbour.  So: (1) keep the edges of the undirected read graph whose endpoints have a common
// neighbour (k_adjacency_support), (2) breadth-first levels over the kept edges from a far end of every component
// (k_bfs_levels), (3) number the reads by (level, old id).  A plain Cuthill-McKee order over ALL edges does not survive 1 % of
// long-range edges (cut 40 % at 8 ranks on the synthetic banded graph); over the supported edges it restores the layout (cut
// 1.1 % against 1.0 % for the generator's own order) - tools/node_order_quality.py.
#include "common.h"

namespace gnnome {
namespace {

// supported[p] = 1 iff the endpoints of adjacency entry p (row a, column adj[p]) have a common neighbour.  Sorted rows: one merge.
__global__ __launch_bounds__(256) void k_adjacency_support(const int32_t* __restrict__ ptr, const int32_t* __restrict__ adj,
                                                           const int32_t* __restrict__ row_of, int64_t nnz, uint8_t* __restrict__ supported) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= nnz) return;
    const int a = row_of[p], b = adj[p];
    int i = ptr[a], j = ptr[b];
    const int ie = ptr[a + 1], je = ptr[b + 1];
    uint8_t hit = 0;
    while (i < ie && j < je) {
        const int x = adj[i], y = adj[j];
        if (x == y) { hit = 1; break; }
        i += x < y;
        j += y < x;
    }
    supported[p] = hit;
}

constexpr int kBfsThreads = 1024;

__device__ __forceinline__ int load_agent(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Breadth-first levels of every component of a CSR graph, ONE workgroup for the whole graph: the frontier of an overlap graph is a
// few dozen reads wide and tens of thousands of levels deep, so what matters is the cost of a level (one pass of 16 waves over
// the frontier, one workgroup barrier), not the width of the machine.  A wave takes a frontier node, its lanes the neighbours.
//   level_key[v]  out: a counter that grows by one per level and never restarts between components - sorting by it lays the
//                 components out one after another, each in level order; -1 for nodes that were never reached (degree 0)
//   seeds         component starts, in order (NULL: the smallest unvisited id with a neighbour starts the next component)
//   far_node[c]   out: the smallest id in the LAST level of component c - the start of the second pass
// Levels and far nodes depend on the graph alone (which wave claims a node first does not change its level).
__global__ __launch_bounds__(kBfsThreads) void k_bfs_levels(const int32_t* __restrict__ ptr, const int32_t* __restrict__ adj, int num_nodes,
                                                            const int32_t* __restrict__ seeds, const int32_t* __restrict__ num_seeds,
                                                            int32_t* __restrict__ level_key, int32_t* __restrict__ frontier_a,
                                                            int32_t* __restrict__ frontier_b, int32_t* __restrict__ far_node,
                                                            int32_t* __restrict__ num_components) {
    __shared__ int cur_size, nxt_size, nxt_min, pick, cursor, comps, key, seed_at, stop;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int kWaves = kBfsThreads / 64;
    if (tid == 0) cur_size = 0, nxt_size = 0, nxt_min = 0x7FFFFFFF, cursor = 0, comps = 0, key = 0, seed_at = 0, stop = 0, pick = 0x7FFFFFFF;
    __syncthreads();
    const int nseeds = seeds ? *num_seeds : 0;
    int32_t* cur = frontier_a;
    int32_t* nxt = frontier_b;
    int cur_min = 0;
    for (;;) {
        if (cur_size == 0) {
            // ---- start the next component
            if (seeds) {
                if (tid == 0) {
                    if (seed_at >= nseeds) stop = 1;
                    else pick = seeds[seed_at++];
                }
                __syncthreads();
            } else {
                // the smallest unvisited node with at least one neighbour, from `cursor` on (the cursor only moves forward)
                for (;;) {
                    const int base = cursor;
                    __syncthreads();
                    if (base >= num_nodes) {
                        if (tid == 0) stop = 1;
                        break;
                    }
                    const int v = base + tid;
                    if (v < num_nodes && load_agent(level_key + v) < 0 && ptr[v + 1] > ptr[v]) atomicMin(&pick, v);
                    __syncthreads();
                    if (pick != 0x7FFFFFFF) {
                        if (tid == 0) cursor = pick + 1;
                        break;
                    }
                    if (tid == 0) cursor = base + kBfsThreads;
                    __syncthreads();
                }
                __syncthreads();
            }
            if (stop) break;
            const int s = pick;
            __syncthreads();
            if (tid == 0) {
                level_key[s] = key;
                cur[0] = s;
                cur_size = 1;
                pick = 0x7FFFFFFF;
                __threadfence();
            }
            cur_min = s;
            __syncthreads();
        }
        // ---- one level
        const int size = cur_size, next_key = key + 1;
        for (int f = wave; f < size; f += kWaves) {
            const int u = load_agent(cur + f);
            const int b = ptr[u], e = ptr[u + 1];
            for (int j = b + lane; j < e; j += 64) {
                const int v = adj[j];
                if (atomicCAS(level_key + v, -1, next_key) == -1) {
                    const int pos = atomicAdd(&nxt_size, 1);
                    __hip_atomic_store(nxt + pos, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    atomicMin(&nxt_min, v);
                }
            }
        }
        __threadfence();
        __syncthreads();
        const int found = nxt_size, found_min = nxt_min;
        __syncthreads();
        if (tid == 0) {
            if (found == 0) far_node[comps++] = cur_min;   // that was the component's last level
            cur_size = found;
            nxt_size = 0;
            nxt_min = 0x7FFFFFFF;
            key = next_key;
        }
        cur_min = found_min;
        int32_t* t = cur;
        cur = nxt;
        nxt = t;
        __syncthreads();
    }
    if (tid == 0) *num_components = comps;
}

}  // namespace
}  // namespace gnnome

// ---- C ABI (include/gnnome_hip.h, "Node order") ------------------------------------------------------------------------------
extern "C" int gnnome_adjacency_support(const int32_t* ptr, const int32_t* adj, const int32_t* row_of, int64_t num_rows, int64_t nnz,
                                        uint8_t* supported, void* stream) {
    using namespace gnnome;
    GN_REQUIRE(num_rows >= 0 && nnz >= 0, "adjacency_support: negative size");
    if (nnz == 0) return GNNOME_OK;
    GN_REQUIRE(ptr && adj && row_of && supported, "adjacency_support: null pointer");
    const int64_t blocks = (nnz + 255) / 256;
    GN_REQUIRE(blocks < (1ll << 31), "adjacency_support: too many entries");
    hipLaunchKernelGGL(k_adjacency_support, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, ptr, adj, row_of, nnz, supported);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

extern "C" int gnnome_bfs_levels(const int32_t* ptr, const int32_t* adj, int64_t num_nodes, const int32_t* seeds, const int32_t* num_seeds,
                                 int32_t* level_key, int32_t* frontier_workspace, int32_t* far_node, int32_t* num_components, void* stream) {
    using namespace gnnome;
    GN_REQUIRE(num_nodes >= 0 && num_nodes < (1ll << 31) - 2048, "bfs_levels: node count out of range");
    GN_REQUIRE((seeds == nullptr) == (num_seeds == nullptr), "bfs_levels: seeds and num_seeds go together");
    GN_REQUIRE(num_components, "bfs_levels: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (num_nodes == 0) {
        GN_HIP(hipMemsetAsync(num_components, 0, sizeof(int32_t), s));
        return GNNOME_OK;
    }
    GN_REQUIRE(ptr && adj && level_key && frontier_workspace && far_node, "bfs_levels: null pointer");
    GN_HIP(hipMemsetAsync(level_key, 0xFF, sizeof(int32_t) * (size_t)num_nodes, s));   // -1: not reached
    hipLaunchKernelGGL(k_bfs_levels, dim3(1), dim3(kBfsThreads), 0, s, ptr, adj, (int)num_nodes, seeds, num_seeds, level_key, frontier_workspace,
                       frontier_workspace + num_nodes, far_node, num_components);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}
