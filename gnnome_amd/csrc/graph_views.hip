// gnnome_build_graph_views: one edge list -> in-edge (by dst) and out-edge (by src) orderings.
//
// DGL builds these lazily (CSR/CSC of the DGLGraph) the first time g.update_all runs on g and on
// dgl.reverse(g) (gated_gcn_full.py:99,112-113,125-126).  Here they are explicit, int32, and built
// once per graph; because both views index the SAME destination-sorted edge storage, reversing the
// graph (train.py:165) costs nothing.
//
// Sorting is rocPRIM's LSD radix sort (stable), so ties keep edge-id order and every downstream
// reduction order is a pure function of the input edge list.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"

namespace gnnome {

__global__ void k_iota(int32_t* out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (int32_t)i;
}

__global__ void k_take(const int32_t* __restrict__ table, const int32_t* __restrict__ idx, int32_t* __restrict__ out,
                       int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = table[idx[i]];
}

// ptr[i] = number of sorted keys < i, for i in [0, n_nodes]
__global__ void k_lower_bound(const int32_t* __restrict__ keys, int64_t n_keys, int32_t* __restrict__ ptr,
                              int64_t n_nodes) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n_nodes; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t lo = 0, hi = n_keys;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (keys[mid] < (int32_t)i) lo = mid + 1; else hi = mid;
        }
        ptr[i] = (int32_t)lo;
    }
}

static unsigned key_bits(int64_t n) {
    unsigned b = 1;
    while (b < 31 && (1ll << b) < n) ++b;
    return b;
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static int sort_temp_bytes(int64_t E, unsigned bits, size_t* out) {
    size_t bytes = 0;
    GN_HIP(rocprim::radix_sort_pairs(nullptr, bytes, (const int32_t*)nullptr, (int32_t*)nullptr, (const int32_t*)nullptr,
                                     (int32_t*)nullptr, (size_t)E, 0u, bits, (hipStream_t)0));
    *out = bytes;
    return GNNOME_OK;
}

static unsigned grid_for(int64_t n) {
    int64_t b = (n + 255) / 256;
    if (b < 1) b = 1;
    if (b > kNumCUs * 8) b = kNumCUs * 8;
    return (unsigned)b;
}

}  // namespace gnnome

extern "C" int gnnome_graph_views_workspace_bytes(int64_t num_nodes, int64_t num_edges, size_t* bytes_host) {
    using namespace gnnome;
    GN_REQUIRE(bytes_host != nullptr, "graph_views: null output");
    GN_REQUIRE(num_nodes >= 0 && num_edges >= 0 && num_nodes < (1ll << 31) && num_edges < (1ll << 31),
               "graph_views: N=%lld E=%lld out of int32 range", (long long)num_nodes, (long long)num_edges);
    size_t sort_bytes = 0;
    if (num_edges > 0) {
        const int rc = sort_temp_bytes(num_edges, key_bits(num_nodes), &sort_bytes);
        if (rc != GNNOME_OK) return rc;
    }
    *bytes_host = 2 * align256((size_t)num_edges * sizeof(int32_t)) + align256(sort_bytes) + 256;
    return GNNOME_OK;
}

extern "C" int gnnome_build_graph_views(const int32_t* src, const int32_t* dst, int64_t num_nodes, int64_t num_edges,
                                        int32_t* in_ptr, int32_t* srt_src, int32_t* srt_dst, int32_t* srt_eid,
                                        int32_t* out_ptr, int32_t* out_pos, int32_t* out_dst, void* workspace,
                                        size_t workspace_bytes, void* stream) {
    using namespace gnnome;
    GN_REQUIRE(num_nodes >= 0 && num_edges >= 0 && num_nodes < (1ll << 31) && num_edges < (1ll << 31),
               "graph_views: N=%lld E=%lld out of int32 range", (long long)num_nodes, (long long)num_edges);
    GN_REQUIRE(in_ptr && out_ptr, "graph_views: null pointer");
    hub_cache_invalidate();   // the aggregation's per-device hub list belongs to the graph whose views were built before
    hipStream_t s = (hipStream_t)stream;
    const int64_t N = num_nodes, E = num_edges;
    if (E == 0) {
        GN_HIP(hipMemsetAsync(in_ptr, 0, (size_t)(N + 1) * sizeof(int32_t), s));
        GN_HIP(hipMemsetAsync(out_ptr, 0, (size_t)(N + 1) * sizeof(int32_t), s));
        return GNNOME_OK;
    }
    GN_REQUIRE(src && dst && srt_src && srt_dst && srt_eid && out_pos && out_dst && workspace, "graph_views: null pointer");
    const unsigned bits = key_bits(N);
    size_t sort_bytes = 0;
    {
        const int rc = sort_temp_bytes(E, bits, &sort_bytes);
        if (rc != GNNOME_OK) return rc;
    }
    const size_t ebytes = align256((size_t)E * sizeof(int32_t));
    if (workspace_bytes < 2 * ebytes + sort_bytes) {
        set_error("graph_views: workspace %zu < %zu bytes", workspace_bytes, 2 * ebytes + sort_bytes);
        return GNNOME_EWORKSPACE;
    }
    char* ws = (char*)workspace;
    int32_t* iota = (int32_t*)ws;
    int32_t* keys_tmp = (int32_t*)(ws + ebytes);
    void* sort_tmp = ws + 2 * ebytes;

    hipLaunchKernelGGL(k_iota, dim3(grid_for(E)), dim3(256), 0, s, iota, E);
    GN_LAUNCH_CHECK();
    // in-edge view: stable sort of (dst, edge id)
    GN_HIP(rocprim::radix_sort_pairs(sort_tmp, sort_bytes, dst, srt_dst, (const int32_t*)iota, srt_eid, (size_t)E, 0u, bits, s));
    hipLaunchKernelGGL(k_take, dim3(grid_for(E)), dim3(256), 0, s, src, (const int32_t*)srt_eid, srt_src, E);
    GN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_lower_bound, dim3(grid_for(N + 1)), dim3(256), 0, s, (const int32_t*)srt_dst, E, in_ptr, N);
    GN_LAUNCH_CHECK();
    // out-edge view: stable sort of (src of sorted position, sorted position)
    GN_HIP(rocprim::radix_sort_pairs(sort_tmp, sort_bytes, (const int32_t*)srt_src, keys_tmp, (const int32_t*)iota, out_pos,
                                     (size_t)E, 0u, bits, s));
    hipLaunchKernelGGL(k_lower_bound, dim3(grid_for(N + 1)), dim3(256), 0, s, (const int32_t*)keys_tmp, E, out_ptr, N);
    GN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_take, dim3(grid_for(E)), dim3(256), 0, s, (const int32_t*)srt_dst, (const int32_t*)out_pos, out_dst, E);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}
