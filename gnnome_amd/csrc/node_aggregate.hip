// gnnome_node_aggregate_f32: both gated aggregations of SymGatedGCN plus the node update, one wave
// per destination node, no atomics.
//
//   s_p   = sigmoid(e[p,:])
//   fwd_i = sum_{p in in(i)}  s_p * A2h[src_p,:] / (sum_{p in in(i)}  s_p + 1e-6)
//   bwd_i = sum_{p in out(i)} s_p * A3h[dst_p,:] / (sum_{p in out(i)} s_p + 1e-6)
//   h'_i  = relu(norm_h(A1h_i + fwd_i + bwd_i)) + h_i
//
// Reference lines replaced: gated_gcn_full.py:111 / :124 (sigmoid), :112-113 / :125-126 (DGL gspmm
// u_mul_e+sum and copy_e+sum on g and on dgl.reverse(g)), :114 / :127 (divide), :129-137 (sum,
// bn_h, relu, residual).  In-edges of a node are a contiguous run of sorted positions (CSR by dst),
// so the forward reduce streams e rows; out-edges are reached through out_pos and re-read the same
// e rows as whole H*4-byte lines.  Bound: HBM - 2 reads of e[E,H] per layer (8*H bytes per edge).
//
// Lane mapping: a row of H floats is covered by H/4 lanes holding a float4 each; a wave64 therefore
// walks G = 64/(H/4) edges at once (4 at H=64, 2 at H=128, 1 at H=256), lane group g taking every
// G-th item of the node's list, and the groups are combined with __shfl_xor at the end.  The
// summation order is a function of the graph alone, never of scheduling: results are bit-reproducible.
//
// Skew ("Skew"): one wave per node is the right shape for assembly graphs (degree ~10), but a repeat-induced hub
// with 10^5 incident edges would keep ONE wave busy for milliseconds while the chip idles.  Nodes whose in + out list
// exceeds kHubThreshold items are therefore found on the device at every call (k_find_hubs: N threads, microseconds),
// their lists are cut into kHubChunks fixed chunks that a first launch reduces to per-chunk partial sums, one wave per
// chunk (k_hub_partials), and the node's own wave then adds the partials IN CHUNK ORDER instead of walking the list: the
// result is still a function of the graph alone (bit-reproducible), only a different - fixed - association than the
// single-wave sum.  Up to kHubCap hubs per call take this path; any further ones fall back to the single wave.
#include "common.h"

#include <algorithm>
#include <map>
#include <mutex>
#include <utility>

namespace gnnome {

constexpr int kAggThreads = 256;
constexpr int kHubThreshold = 4096;   // items (in-edges + out-edges) above which a node's list is split
constexpr int kHubChunks = 128;       // chunks per hub = waves working on it
constexpr int kHubCap = 64;           // hubs per call that get the split path
constexpr int kHubMaxH = 256;
constexpr int kRecItems = 20;         // neighbours per direction that fit a node's 64-word record (4 + 3 * 20)

struct HubScratch {
    const void *key_in = nullptr, *key_out = nullptr;   // the CSR arrays the hub list was last built from
    int64_t key_n = -1;
    int* count = nullptr;       // [1]
    int* nodes = nullptr;       // [kHubCap]
    float* partials = nullptr;  // [kHubCap][kHubChunks][4][H]
    // The hub count of the graph the list was last built for, read back WITHOUT a host sync: an asynchronous copy into pinned memory
    // behind k_find_hubs, polled with hipEventQuery on later calls.  Once it is known to be zero - every assembly graph without a
    // repeat-induced hub - the two hub launches (8 layers x 2 x ~4.7 us: 9 % of an E. coli-sized forward) are skipped.
    int* host_count = nullptr;  // pinned
    hipEvent_t ready = nullptr;
    bool pending = false;
    int known = -1;             // -1: not known yet
};

// Scratch of the hub path, one per (device, stream), allocated on first use (never inside a stream capture: the first call of any
// process is an eager one - CapturedForward and the benchmarks warm up before they record).  Round 4 (VERDICT r3 12.i / ADVICE r3): the
// scratch used to be one buffer per DEVICE behind a "do not run two aggregations of graphs with hubs on two streams at once" contract;
// it is now keyed by the stream as well, so launches on different streams never share it, and every read or write of the host-side
// state (key, known, pending) happens under the mutex, which launch_agg holds for its whole hub section.
static std::map<std::pair<int, hipStream_t>, HubScratch> g_hub_table;
static std::mutex g_hub_guard;

// gnnome_build_graph_views calls this: new CSR arrays may land at the addresses of freed ones (torch's caching allocator does
// that routinely), and the list - and the "this graph has no hubs" fact - of the previous graph must not outlive it.
void hub_cache_invalidate() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    std::lock_guard<std::mutex> lock(g_hub_guard);
    for (auto& kv : g_hub_table) {
        if (kv.first.first != dev) continue;
        kv.second.key_in = kv.second.key_out = nullptr;
        kv.second.key_n = -1;
        kv.second.known = -1;
        kv.second.pending = false;
    }
}

// (called with g_hub_guard held)
static HubScratch* hub_scratch(hipStream_t s, bool* capturing_unallocated, const void* in_ptr = nullptr, const void* out_ptr = nullptr,
                               int64_t n = -1) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    auto it = g_hub_table.find(std::make_pair(dev, s));
    if (it == g_hub_table.end() || it->second.partials == nullptr) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone) {
            // a capture runs on a stream of its own and must not allocate: it records against the scratch the eager warm-up on this device
            // left (the recording then shares it with that stream - replays of graphs WITH hubs must not overlap eager aggregations there)
            // ... preferably the one that knows THIS graph (round 5: taking the first one found - often the default stream's, keyed to another
            // graph - recorded k_find_hubs and the two hub launches of every aggregation into the training step's hipGraph: 25 launches that
            // find nothing on a graph the warm-up had already found hub-free)
            for (auto& kv : g_hub_table) {
                const HubScratch& h = kv.second;
                if (kv.first.first == dev && h.partials != nullptr && h.key_n == n &&
                    ((h.key_in == in_ptr && h.key_out == out_ptr) || (h.key_in == out_ptr && h.key_out == in_ptr)))
                    return &kv.second;
            }
            for (auto& kv : g_hub_table)
                if (kv.first.first == dev && kv.second.partials != nullptr) return &kv.second;
            *capturing_unallocated = true;
            return nullptr;
        }
    }
    HubScratch& h = g_hub_table[std::make_pair(dev, s)];
    if (h.partials == nullptr) {
        if (h.count == nullptr && hipMalloc(&h.count, sizeof(int)) != hipSuccess) return nullptr;
        if (h.nodes == nullptr && hipMalloc(&h.nodes, sizeof(int) * kHubCap) != hipSuccess) return nullptr;
        if (hipMalloc(&h.partials, sizeof(float) * (size_t)kHubCap * kHubChunks * 4 * kHubMaxH) != hipSuccess) return nullptr;
        if (hipHostMalloc(&h.host_count, sizeof(int), hipHostMallocDefault) != hipSuccess) h.host_count = nullptr;   // (optional: without it the count stays unknown)
        if (h.host_count && hipEventCreateWithFlags(&h.ready, hipEventDisableTiming) != hipSuccess) h.ready = nullptr;
    }
    return &h;
}

__global__ __launch_bounds__(256) void k_find_hubs(const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ out_ptr, int64_t n,
                                                   int* __restrict__ count, int* __restrict__ nodes) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int cnt = (in_ptr[i + 1] - in_ptr[i]) + (out_ptr[i + 1] - out_ptr[i]);
    if (cnt > kHubThreshold) {
        const int slot = atomicAdd(count, 1);
        if (slot < kHubCap) nodes[slot] = (int)i;
    }
}

template <int H>
__device__ __forceinline__ float group_sum(float v) {
    // all-reduce over the lane groups (lanes with equal lane % (H/4))
    constexpr int LPR = H / 4;
#pragma unroll
    for (int m = LPR; m < 64; m <<= 1) v += __shfl_xor(v, m);
    return v;
}

template <int H>
__device__ __forceinline__ float row_sum(float v) {
    // all-reduce over the H/4 lanes of one row
    constexpr int LPR = H / 4;
#pragma unroll
    for (int m = 1; m < LPR; m <<= 1) v += __shfl_xor(v, m);
    return v;
}

// The gated sums over items [lo, hi) of one node's work list (in-edges first, then out-edges), lane group g taking every
// G-th item of every 64-item batch; per-lane-group partial sums (combine with group_sum).
template <int H, int U = (H == 256 ? 8 : 4)>
__device__ __forceinline__ void accumulate_items(const float* __restrict__ e, const float* __restrict__ A2h, const float* __restrict__ A3h,
                                                 int ldn, const int32_t* __restrict__ srt_src, const int32_t* __restrict__ out_pos,
                                                 const int32_t* __restrict__ out_dst, int ib, int din, int ob, int lo, int hi, int lane,
                                                 int group, int c, f32x4& nf, f32x4& df, f32x4& nb, f32x4& db) {
    constexpr int LPR = H / 4, G = 64 / LPR;
    for (int base = lo; base < hi; base += 64) {
        // lane l owns item base + l
        const int j = base + lane;
        int my_p = 0, my_n = 0;
        if (j < din) {
            my_p = ib + j;
            my_n = srt_src[my_p];
        } else if (j < hi) {
            my_p = out_pos[ob + j - din];
            my_n = out_dst[ob + j - din];
        }
        const int m = min(64, hi - base);
        for (int j0 = 0; j0 < m; j0 += G * U) {
            f32x4 x[U], a[U];
            bool live[U], fwd[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int item = j0 + u * G + group;
                live[u] = item < m;
                const int it = live[u] ? item : 0;
                const int p = __shfl(my_p, it), nn = __shfl(my_n, it);
                fwd[u] = base + it < din;
                const float* tb = fwd[u] ? A2h : A3h;
                x[u] = *reinterpret_cast<const f32x4*>(e + (int64_t)p * H + c);
                a[u] = *reinterpret_cast<const f32x4*>(tb + (int64_t)nn * ldn + c);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (live[u]) {
                    f32x4 s;
#pragma unroll
                    for (int k = 0; k < 4; ++k) s[k] = sigmoidf_(x[u][k]);
                    if (fwd[u]) {
                        nf += s * a[u];
                        df += s;
                    } else {
                        nb += s * a[u];
                        db += s;
                    }
                }
            }
        }
    }
}

// The same sums with the same association (every item goes to the same lane group and is added in the same order: bit-identical
// results), restructured so that the e rows of the IN-edges - whose positions follow from the CSR pointer alone - are requested
// before the wave waits for its index loads: the in-items and the out-items of a 64-item batch run as two loops (the step that
// straddles the boundary is cut in two; the two directions have separate accumulators, so the order within each is unchanged),
// and inside the in-loop all row loads go out first, then the neighbour indices are awaited and the table rows gathered.
// NOBR: the accumulation of a dead item (past the end of its list: its loads were clamped to a live one) is masked arithmetically
// instead of being branched around - every live item still adds the same values in the same order (a dead one adds +0): same bits.
// recv (round 6 experiment, kNodeRecord): this lane's word of the node's 256-byte record (k_build_node_records) - when the node's two lists fit
// it (rec_small), the neighbour ids and out-edge positions come out of that ONE load by cross-lane reads instead of a second trip to memory.
template <int H, int U = (H == 256 ? 8 : 4), bool NOBR = false, bool NTOUT = false, bool TBLFIX = false>
__device__ __forceinline__ void accumulate_items_split(const float* __restrict__ e, const float* __restrict__ A2h, const float* __restrict__ A3h,
                                                       int ldn, const int32_t* __restrict__ srt_src, const int32_t* __restrict__ out_pos,
                                                       const int32_t* __restrict__ out_dst, int ib, int din, int ob, int lo, int hi, int lane,
                                                       int group, int c, f32x4& nf, f32x4& df, f32x4& nb, f32x4& db, int recv = 0,
                                                       bool rec_small = false) {
    constexpr int LPR = H / 4, G = 64 / LPR;
    for (int base = lo; base < hi; base += 64) {
        const int j = base + lane;   // lane l owns item base + l
        int my_p = 0, my_n = 0;
        if (rec_small) {   // (wave-uniform; one batch: hi <= 2 * kRecItems)
            const int jo = min(max(j - din, 0), kRecItems - 1);
            my_n = __shfl(recv, j < din ? 4 + min(j, kRecItems - 1) : 4 + 2 * kRecItems + jo);
            my_p = __shfl(recv, 4 + kRecItems + jo);
        } else if (j < din) {
            my_n = srt_src[ib + j];
        } else if (j < hi) {
            my_p = out_pos[ob + j - din];
            my_n = out_dst[ob + j - din];
        }
        const int m = min(64, hi - base);
        const int m_in = min(m, max(din - base, 0));   // items [0, m_in) of this batch are in-edges
        for (int j0 = 0; j0 < m_in; j0 += G * U) {
            f32x4 x[U], a[U];
            bool live[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int item = j0 + u * G + group;
                live[u] = item < m_in;
                x[u] = *reinterpret_cast<const f32x4*>(e + (int64_t)(ib + base + (live[u] ? item : 0)) * H + c);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int item = j0 + u * G + group;
                const int nn = TBLFIX ? (ib & 1023) : __shfl(my_n, live[u] ? item : 0);   // TBLFIX (measurement only, wrong results): one table row per node
                a[u] = *reinterpret_cast<const f32x4*>(A2h + (int64_t)nn * ldn + c);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (NOBR) {
                    f32x4 s;
#pragma unroll
                    for (int k = 0; k < 4; ++k) s[k] = live[u] ? sigmoidf_(x[u][k]) : 0.f;
                    nf += s * a[u];
                    df += s;
                } else if (live[u]) {
                    f32x4 s;
#pragma unroll
                    for (int k = 0; k < 4; ++k) s[k] = sigmoidf_(x[u][k]);
                    nf += s * a[u];
                    df += s;
                }
            }
        }
        for (int j0 = m_in / (G * U) * (G * U); j0 < m; j0 += G * U) {
            f32x4 x[U], a[U];
            bool live[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int item = j0 + u * G + group;
                live[u] = item >= m_in && item < m;
                const int it = live[u] ? item : m_in;   // (m_in < m here: a valid out-item)
                const int p = __shfl(my_p, it), nn = TBLFIX ? (ib & 1023) : __shfl(my_n, it);
                // NTOUT (variant 14): the out-edge pass is an e' row's LAST use in this launch - read past the L2's replacement order, so that the
                // rows the in-edge passes have just brought in (and whose out-edge use is still to come) stay
                x[u] = NTOUT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(e + (int64_t)p * H + c))
                             : *reinterpret_cast<const f32x4*>(e + (int64_t)p * H + c);
                a[u] = *reinterpret_cast<const f32x4*>(A3h + (int64_t)nn * ldn + c);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (NOBR) {
                    f32x4 s;
#pragma unroll
                    for (int k = 0; k < 4; ++k) s[k] = live[u] ? sigmoidf_(x[u][k]) : 0.f;
                    nb += s * a[u];
                    db += s;
                } else if (live[u]) {
                    f32x4 s;
#pragma unroll
                    for (int k = 0; k < 4; ++k) s[k] = sigmoidf_(x[u][k]);
                    nb += s * a[u];
                    db += s;
                }
            }
        }
    }
}

// First launch of the hub path: wave (blockIdx.x * 4 + wave) reduces chunk c of EVERY split hub to (sum s*A2h, sum s,
// sum s*A3h, sum s) partials; chunk boundaries are a function of the node's item count only.
template <int H>
__global__ __launch_bounds__(kAggThreads) void k_hub_partials(const float* __restrict__ e, const float* __restrict__ A2h,
                                                              const float* __restrict__ A3h, int ldn, const int32_t* __restrict__ in_ptr,
                                                              const int32_t* __restrict__ srt_src, const int32_t* __restrict__ out_ptr,
                                                              const int32_t* __restrict__ out_pos, const int32_t* __restrict__ out_dst,
                                                              int64_t n_out, const int* __restrict__ hub_count,
                                                              const int* __restrict__ hub_nodes, float* __restrict__ partials) {
    constexpr int LPR = H / 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ch = blockIdx.x * (kAggThreads / 64) + wave;
    const int group = lane / LPR, c = (lane % LPR) * 4;
    const int nh = min(*hub_count, kHubCap);
    for (int h = 0; h < nh; ++h) {
        const int64_t node = hub_nodes[h];
        if (node >= n_out) continue;
        const int ib = in_ptr[node], din = in_ptr[node + 1] - ib;
        const int ob = out_ptr[node], cnt = din + out_ptr[node + 1] - ob;
        const int per = ((cnt + kHubChunks - 1) / kHubChunks + 63) / 64 * 64;   // whole 64-item batches per chunk
        const int lo = min(cnt, ch * per), hi = min(cnt, lo + per);
        f32x4 nf = {0.f, 0.f, 0.f, 0.f}, df = nf, nb = nf, db = nf;
        accumulate_items<H>(e, A2h, A3h, ldn, srt_src, out_pos, out_dst, ib, din, ob, lo, hi, lane, group, c, nf, df, nb, db);
        f32x4 o0, o1, o2, o3;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            o0[k] = group_sum<H>(nf[k]);
            o1[k] = group_sum<H>(df[k]);
            o2[k] = group_sum<H>(nb[k]);
            o3[k] = group_sum<H>(db[k]);
        }
        if (group == 0) {
            float* pp = partials + ((int64_t)h * kHubChunks + ch) * 4 * H + c;
            *reinterpret_cast<f32x4*>(pp) = o0;
            *reinterpret_cast<f32x4*>(pp + H) = o1;
            *reinterpret_cast<f32x4*>(pp + 2 * H) = o2;
            *reinterpret_cast<f32x4*>(pp + 3 * H) = o3;
        }
    }
}

// One wave per node.  The node's in-edges and out-edges form ONE work list of (sorted position,
// neighbour row, direction) items; each lane first fetches the indices of one item (so the dependent
// index loads happen once per 64 items, not once per edge), then the wave walks the list with U items
// per lane group in flight: 2*U*G independent H*4-byte row loads per wave.
// MODE 0: the fused inference update.  MODE 1 (train forward): h_out = A1h + fwd + bwd (pre-normalisation) and the
// four node tables the backward needs (aux0..3 = fwd, 1/(den_f+eps), bwd, 1/(den_b+eps)).  MODE 2 (aggregation
// backward): aux0 = sum_in s*A2h[src], aux2 = sum_out s*A3h[dst], the raw gated sums with caller-chosen tables.
// HUBFIN = false: the regular launch, one wave per node; a node on the hub list (and still longer than the threshold) is
// left to the HUBFIN = true launch, which runs one wave per hub-list slot, adds the chunk partials in chunk order and
// shares everything after the sums (two launches instead of a branch: the partial-sum loop cost the regular kernel 22
// registers and a third of its occupancy).  U: items per lane group in flight, WPS: waves per SIMD asked of the register
// allocator (0 = unconstrained).
template <int H, int NORM, int MODE, bool HUBFIN = false, int U = (H == 256 ? 8 : 4), int WPS = 0, int SPLIT = 0, bool REC = false>
__global__ __launch_bounds__(kAggThreads, (WPS > 0 ? WPS : 1)) void k_node_aggregate(
    const float* __restrict__ e, int64_t n_out, const float* __restrict__ A1h, const float* __restrict__ A2h,
    const float* __restrict__ A3h, int ldn, const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ srt_src,
    const int32_t* __restrict__ out_ptr, const int32_t* __restrict__ out_pos, const int32_t* __restrict__ out_dst,
    const float* __restrict__ h_in, int ldh, float* __restrict__ h_out, const float* __restrict__ scale,
    const float* __restrict__ shift, int total_blocks, float* __restrict__ aux0, float* __restrict__ aux1,
    float* __restrict__ aux2, float* __restrict__ aux3, const int* __restrict__ hub_count, const int* __restrict__ hub_nodes,
    const float* __restrict__ hub_partials, int64_t node0, int norm_width, const int32_t* __restrict__ records = nullptr) {
    constexpr int LPR = H / 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t node;
    int hub_slot = -1;
    if (HUBFIN) {
        hub_slot = blockIdx.x * (kAggThreads / 64) + wave;
        if (hub_slot >= min(*hub_count, kHubCap)) return;
        node = hub_nodes[hub_slot];
    } else {
        node = node0 + (int64_t)xcd_remap(blockIdx.x, total_blocks) * (kAggThreads / 64) + wave;
    }
    if (node >= n_out) return;
    const int group = lane / LPR, c = (lane % LPR) * 4;

    int ib, din, ob, cnt, recv = 0;
    if (REC) {   // one 256-byte load: [ib, din, ob, dout, 20 in-neighbours, 20 out positions, 20 out-neighbours]
        recv = records[node * 64 + lane];
        ib = __builtin_amdgcn_readlane(recv, 0), din = __builtin_amdgcn_readlane(recv, 1);
        ob = __builtin_amdgcn_readlane(recv, 2), cnt = din + __builtin_amdgcn_readlane(recv, 3);
    } else {
        ib = in_ptr[node], din = in_ptr[node + 1] - ib;
        ob = out_ptr[node], cnt = din + out_ptr[node + 1] - ob;
    }
    const bool rec_small = REC && din <= kRecItems && cnt - din <= kRecItems;
    if (HUBFIN) {
        if (cnt <= kHubThreshold) return;   // a stale list entry: the regular launch has done this node
    } else if (cnt > kHubThreshold && hub_nodes != nullptr) {   // wave-uniform: is this node on the hub list?
        const int nh = min(*hub_count, kHubCap);
        bool listed = false;
        for (int base = 0; base < nh; base += 64) listed |= __ballot(base + lane < nh && hub_nodes[base + lane] == (int)node) != 0;
        if (listed) return;
    }
    f32x4 a1 = {0.f, 0.f, 0.f, 0.f};
    if (MODE != 2) a1 = *reinterpret_cast<const f32x4*>(A1h + node * ldn + c);

    f32x4 nf = {0.f, 0.f, 0.f, 0.f}, df = nf, nb = nf, db = nf;
    if (HUBFIN) {
        // add the chunk partials in chunk order (lane group 0 carries the sums; the others stay zero for group_sum)
        if (group == 0) {
            const float* pp = hub_partials + ((int64_t)hub_slot * kHubChunks) * 4 * H + c;
            for (int ch = 0; ch < kHubChunks; ++ch, pp += 4 * H) {
                nf += *reinterpret_cast<const f32x4*>(pp);
                df += *reinterpret_cast<const f32x4*>(pp + H);
                nb += *reinterpret_cast<const f32x4*>(pp + 2 * H);
                db += *reinterpret_cast<const f32x4*>(pp + 3 * H);
            }
        }
    } else {
        if (SPLIT == 2)   // MEASUREMENT ONLY (variant 7, wrong results): the out-edges alone - what the aggregation would cost if the in-edge half
                          // were done elsewhere (inside the gate's store waves); see DESIGN.md, round 3
            accumulate_items_split<H, U>(e, A2h, A3h, ldn, srt_src, out_pos, out_dst, ib, 0, ob, 0, cnt - din, lane, group, c, nf, df, nb, db);
        else if (SPLIT == 4)   // variant 14: nontemporal out-edge loads of e'
            accumulate_items_split<H, U, false, true>(e, A2h, A3h, ldn, srt_src, out_pos, out_dst, ib, din, ob, 0, cnt, lane, group, c, nf, df, nb, db);
        else if (SPLIT == 5)   // variant 15, MEASUREMENT ONLY (wrong results): every table row of a node replaced by one L1-resident row - the kernel without its gathers' L1 misses
            accumulate_items_split<H, U, false, false, true>(e, A2h, A3h, ldn, srt_src, out_pos, out_dst, ib, din, ob, 0, cnt, lane, group, c, nf, df, nb, db);
        else if (SPLIT == 3)   // variant 8: the split loop without branches around dead items
            accumulate_items_split<H, U, true>(e, A2h, A3h, ldn, srt_src, out_pos, out_dst, ib, din, ob, 0, cnt, lane, group, c, nf, df, nb, db);
        else if (SPLIT == 1)
            accumulate_items_split<H, U>(e, A2h, A3h, ldn, srt_src, out_pos, out_dst, ib, din, ob, 0, cnt, lane, group, c, nf, df, nb, db, recv, rec_small);
        else
            accumulate_items<H, U>(e, A2h, A3h, ldn, srt_src, out_pos, out_dst, ib, din, ob, 0, cnt, lane, group, c, nf, df, nb, db);
    }

    f32x4 v, t0, t1, t2, t3;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float num_f = group_sum<H>(nf[k]), den_f = group_sum<H>(df[k]);
        const float num_b = group_sum<H>(nb[k]), den_b = group_sum<H>(db[k]);
        v[k] = a1[k] + num_f / (den_f + kAggEps) + num_b / (den_b + kAggEps);
        if (MODE == 1) {
            t0[k] = num_f / (den_f + kAggEps);
            t1[k] = 1.0f / (den_f + kAggEps);
            t2[k] = num_b / (den_b + kAggEps);
            t3[k] = 1.0f / (den_b + kAggEps);
        } else if (MODE == 2) {
            t0[k] = num_f;
            t2[k] = num_b;
        }
    }
    if (MODE != 0) {
        if (group == 0) {
            *reinterpret_cast<f32x4*>(aux0 + node * H + c) = t0;
            *reinterpret_cast<f32x4*>(aux2 + node * H + c) = t2;
            if (MODE == 1) {
                *reinterpret_cast<f32x4*>(aux1 + node * H + c) = t1;
                *reinterpret_cast<f32x4*>(aux3 + node * H + c) = t3;
                *reinterpret_cast<f32x4*>(h_out + node * H + c) = v;
            }
        }
        return;
    }
    if (NORM == GNNOME_NORM_LAYER) {
        // (norm_width < H: a zero-padded narrower model - statistics over its own channels, the padded ones hold exact zeros)
        const float inv_w = 1.0f / (float)norm_width;
        const float mean = row_sum<H>(v[0] + v[1] + v[2] + v[3]) * inv_w;
        float s2 = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) s2 += (c + k < norm_width) ? (v[k] - mean) * (v[k] - mean) : 0.f;
        const float rstd = rsqrtf(row_sum<H>(s2) * inv_w + kNormEps);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (v[k] - mean) * rstd;
    }
    if (group == 0) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + c);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(h_in + node * ldh + c);
        f32x4 y;
#pragma unroll
        for (int k = 0; k < 4; ++k) y[k] = relu_keep_nan(v[k] * sc[k] + sh[k]) + hi[k];
        *reinterpret_cast<f32x4*>(h_out + node * H + c) = y;
    }
}

// Round 4 experiment (tuning key 7 = 9, H = 128, the fused inference update): TWO NODES PER WAVE.  The kernel above gives a wave one node and
// lets its two 32-lane groups take alternate items of that node's list: a list of 10 + 10 items costs four rounds of 8 row slots (62 % of
// them live) and a cross-group reduction.  Here each 32-lane group owns a node of its own (consecutive nodes: mates 2r / 2r + 1 on an
// assembly graph, with lists of equal length): three rounds of 4 slots per direction and node (83 % live), half as many waves, no
// cross-lane sums - every node's items are added in list order by one lane group, so the result is a function of the graph alone here
// too (another association than the kernel above: equal to fp32 rounding, not bit for bit).
// MEASURED (tools/agg_time.py 128 variants, tools/forward_ab.py 0,9,11,12,13 7; configs[1]): 0.2011 against 0.2051 ms per launch, forward 4.405 against
// 4.419 ms - level: the kernel's time does not depend on how its waves are cut (as with U and occupancy in rounds 2-3).  Not the default.
// PERSIST (variants 11 / 12): a workgroup walks a CONTIGUOUS chunk of node pairs in order instead of taking one group of four pairs, so that the
// table rows A2h[src] / A3h[dst] of one step - neighbours of neighbouring reads, largely the same rows - are still in that CU's L1 at the next;
// NT: the e rows (no reuse inside a CU) are loaded nontemporally so that they do not evict them.
// MEASURED NEGATIVE: 0.294 (4 workgroups per CU) / 0.303 (+ nontemporal) / 0.311 ms (8 per CU) against 0.205: a CU walking its own chunk gives
// up what the plain launch has - many CUs of an XCD streaming ADJACENT e rows at the same time (DRAM pages, shared L2 lines).
template <int NORM, int U = 4, bool PERSIST = false, bool NT = false>
__global__ __launch_bounds__(kAggThreads) void k_node_aggregate_pair(
    const float* __restrict__ e, int64_t n_out, const float* __restrict__ A1h, const float* __restrict__ A2h, const float* __restrict__ A3h,
    int ldn, const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ srt_src, const int32_t* __restrict__ out_ptr,
    const int32_t* __restrict__ out_pos, const int32_t* __restrict__ out_dst, const float* __restrict__ h_in, int ldh,
    float* __restrict__ h_out, const float* __restrict__ scale, const float* __restrict__ shift, int total_blocks,
    const int* __restrict__ hub_count, const int* __restrict__ hub_nodes, int64_t node0) {
    constexpr int H = 128;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, grp = lane >> 5, gl = lane & 31, c = 4 * gl;
    auto load_e = [&](int64_t row) {
        const f32x4* p = reinterpret_cast<const f32x4*>(e + row * H + c);
        return NT ? __builtin_nontemporal_load(p) : *p;
    };
    // total_blocks: PERSIST - the number of 4-pair steps of the whole range, dealt out in contiguous chunks to the gridDim.x workgroups
    const int steps_per = PERSIST ? (total_blocks + (int)gridDim.x - 1) / (int)gridDim.x : 1;
    const int chunk = PERSIST ? xcd_remap(blockIdx.x, gridDim.x) : 0;
  for (int it = 0; it < steps_per; ++it) {
    const int64_t step = PERSIST ? (int64_t)chunk * steps_per + it : (int64_t)xcd_remap(blockIdx.x, total_blocks);
    if (PERSIST && step >= total_blocks) break;
    const int64_t node = node0 + 2 * (step * (kAggThreads / 64) + wave) + grp;
    bool valid = node < n_out;
    int ib = 0, din = 0, ob = 0, cnt = 0;
    if (valid) {
        ib = in_ptr[node], din = in_ptr[node + 1] - ib;
        ob = out_ptr[node], cnt = din + out_ptr[node + 1] - ob;
        if (cnt > kHubThreshold && hub_nodes != nullptr) {   // a listed hub is left to the HUBFIN launch of the kernel above
            const int nh = min(*hub_count, kHubCap);
            for (int k = 0; k < nh; ++k) valid = valid && hub_nodes[k] != (int)node;
        }
    }
    if (!valid) cnt = din = 0;
    f32x4 a1 = {0.f, 0.f, 0.f, 0.f};
    if (valid) a1 = *reinterpret_cast<const f32x4*>(A1h + node * ldn + c);
    f32x4 nf = {0.f, 0.f, 0.f, 0.f}, df = nf, nb = nf, db = nf;
    const int cnt_w = max(__shfl(cnt, 0), __shfl(cnt, 32));
    for (int base = 0; base < cnt_w; base += 32) {
        const int j = base + gl;   // lane gl of a group owns item base + gl of ITS node
        int my_p = 0, my_n = 0;
        if (j < din) {
            my_n = srt_src[ib + j];
        } else if (j < cnt) {
            my_p = out_pos[ob + j - din];
            my_n = out_dst[ob + j - din];
        }
        const int m = max(min(32, cnt - base), 0), m_in = min(m, max(din - base, 0));
        const int m_in_w = max(__shfl(m_in, 0), __shfl(m_in, 32)), m_out_w = max(__shfl(m - m_in, 0), __shfl(m - m_in, 32));
        for (int j0 = 0; j0 < m_in_w; j0 += U) {   // in-edges: the rows follow from the CSR pointer alone - requested before the indices are awaited
            f32x4 x[U], a[U];
            bool live[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                live[u] = j0 + u < m_in;
                x[u] = load_e((int64_t)(ib + base + (live[u] ? j0 + u : 0)));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int nn = __shfl(my_n, 32 * grp + (live[u] ? j0 + u : 0));
                a[u] = *reinterpret_cast<const f32x4*>(A2h + (int64_t)nn * ldn + c);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                f32x4 sg;
#pragma unroll
                for (int k = 0; k < 4; ++k) sg[k] = live[u] ? sigmoidf_(x[u][k]) : 0.f;
                nf += sg * a[u];
                df += sg;
            }
        }
        for (int j0 = 0; j0 < m_out_w; j0 += U) {   // out-edges: item m_in + j0 + u of this batch
            f32x4 x[U], a[U];
            bool live[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int item = m_in + j0 + u;
                live[u] = item < m;
                const int src_lane = 32 * grp + (live[u] ? item : (m > 0 ? m - 1 : 0));
                const int pp = __shfl(my_p, src_lane), nn = __shfl(my_n, src_lane);
                x[u] = load_e((int64_t)pp);
                a[u] = *reinterpret_cast<const f32x4*>(A3h + (int64_t)nn * ldn + c);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                f32x4 sg;
#pragma unroll
                for (int k = 0; k < 4; ++k) sg[k] = live[u] ? sigmoidf_(x[u][k]) : 0.f;
                nb += sg * a[u];
                db += sg;
            }
        }
    }
    if (!valid) continue;
    f32x4 v;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = a1[k] + nf[k] / (df[k] + kAggEps) + nb[k] / (db[k] + kAggEps);
    if (NORM == GNNOME_NORM_LAYER) {
        float s1 = v[0] + v[1] + v[2] + v[3];
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) s1 += __shfl_xor(s1, o);
        const float mean = s1 * (1.0f / H);
        float s2 = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) s2 += (v[k] - mean) * (v[k] - mean);
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) s2 += __shfl_xor(s2, o);
        const float rstd = rsqrtf(s2 * (1.0f / H) + kNormEps);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (v[k] - mean) * rstd;
    }
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c), sh = *reinterpret_cast<const f32x4*>(shift + c);
    const f32x4 hi = *reinterpret_cast<const f32x4*>(h_in + node * ldh + c);
    f32x4 y;
#pragma unroll
    for (int k = 0; k < 4; ++k) y[k] = relu_keep_nan(v[k] * sc[k] + sh[k]) + hi[k];
    *reinterpret_cast<f32x4*>(h_out + node * H + c) = y;
  }
}

// A node's 256-byte record for the REC form of the aggregation (round 6 experiment): word 0 ib, 1 in-degree, 2 ob, 3 out-degree, then the first
// kRecItems of srt_src[ib ..], out_pos[ob ..], out_dst[ob ..].  One thread per word.
__global__ __launch_bounds__(256) void k_build_node_records(const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ srt_src,
                                                            const int32_t* __restrict__ out_ptr, const int32_t* __restrict__ out_pos,
                                                            const int32_t* __restrict__ out_dst, int64_t n, int32_t* __restrict__ rec) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t node = t >> 6;
    const int w = (int)(t & 63);
    if (node >= n) return;
    const int ib = in_ptr[node], din = in_ptr[node + 1] - ib, ob = out_ptr[node], dout = out_ptr[node + 1] - ob;
    int v = 0;
    if (w < 4) v = w == 0 ? ib : w == 1 ? din : w == 2 ? ob : dout;
    else if (w < 4 + kRecItems) v = (w - 4) < din ? srt_src[ib + w - 4] : 0;
    else if (w < 4 + 2 * kRecItems) v = (w - 4 - kRecItems) < dout ? out_pos[ob + w - 4 - kRecItems] : 0;
    else v = (w - 4 - 2 * kRecItems) < dout ? out_dst[ob + w - 4 - 2 * kRecItems] : 0;
    rec[t] = v;
}
static thread_local const int32_t* g_node_records = nullptr;   // gnnome_debug_node_records

template <int H>
static int launch_agg(const float* e, int64_t n_out, const float* A1h, const float* A2h, const float* A3h, int ldn,
                      const int32_t* in_ptr, const int32_t* ss, const int32_t* out_ptr, const int32_t* out_pos,
                      const int32_t* od, const float* h_in, int ldh, float* h_out, int norm, const float* scale,
                      const float* shift, hipStream_t s, int mode = 0, float* aux0 = nullptr, float* aux1 = nullptr,
                      float* aux2 = nullptr, float* aux3 = nullptr, int64_t node_begin = 0, int64_t node_end = -1, int norm_width = 0) {
    if (norm_width <= 0) norm_width = H;
    // [node_begin, node_end): the rows the regular launch covers (gnnome_node_aggregate_range_f32); the hub path always works
    // on the whole graph [0, n_out) and runs with the range that starts at node 0
    if (node_end < 0) node_end = n_out;
    const int64_t node0 = node_begin;
    const int64_t blocks = (node_end - node_begin + (kAggThreads / 64) - 1) / (kAggThreads / 64);
    GN_REQUIRE(blocks < (1ll << 31), "node_aggregate: too many nodes");
    GN_REQUIRE(node_begin >= 0 && node_begin < node_end && node_end <= n_out, "node_aggregate: bad node range [%lld, %lld) of %lld",
               (long long)node_begin, (long long)node_end, (long long)n_out);
    // the hub path (see the header comment): find the long lists, reduce them chunk-wise, let the node's wave add the chunks
    bool capturing_unallocated = false;
    std::lock_guard<std::mutex> hub_lock(g_hub_guard);   // the scratch's host-side state is read and written below, up to the launches
    HubScratch* hub = tuning(kTuneAggHubs) == 1 ? nullptr : hub_scratch(s, &capturing_unallocated, in_ptr, out_ptr, n_out);
    GN_REQUIRE(!capturing_unallocated, "node_aggregate: first call on this device inside a stream capture - run one eager call first "
                                       "(the hub scratch is allocated on first use)");
    // the hub list is a function of in-degree + out-degree: the same for a graph and its reversed views (in_ptr / out_ptr swapped)
    auto same_graph = [&](const HubScratch* h) {
        return h->key_n == n_out && ((h->key_in == in_ptr && h->key_out == out_ptr) || (h->key_in == out_ptr && h->key_out == in_ptr));
    };
    if (hub != nullptr && same_graph(hub)) {
        hipStreamCaptureStatus capq = hipStreamCaptureStatusNone;
        const bool capturing = hub->pending && (hipStreamIsCapturing(s, &capq) != hipSuccess || capq != hipStreamCaptureStatusNone);
        if (hub->pending && !capturing && hipEventQuery(hub->ready) == hipSuccess) {   // (an event query is not allowed inside a stream capture)
            hub->known = *hub->host_count;
            hub->pending = false;
        }
        if (hub->known == 0) hub = nullptr;   // this graph has no node above the threshold: the plain launch is the whole aggregation
    }
    const bool hub_pass = hub != nullptr && node_begin == 0;
    const int* hub_count = hub ? hub->count : nullptr;
    const int* hub_nodes = hub ? hub->nodes : nullptr;
    const float* hub_partials = hub ? hub->partials : nullptr;
    if (hub && !hub_pass)   // a later range of the same call sequence: the hub list must be the one its first range built
        GN_REQUIRE(same_graph(hub), "node_aggregate_range: the ranges of one aggregation must start with the range that begins at node 0");
    if (hub_pass) {
        // The list is rebuilt when the CSR arrays change (8 layers share one graph).  A stale list - another graph at the
        // same addresses - costs speed only: the kernels re-read every count, a listed node that is no hub takes the normal
        // path, an unlisted hub the single-wave path, and the partials are recomputed at every call.
        if (!same_graph(hub)) {
            GN_HIP(hipMemsetAsync(hub->count, 0, sizeof(int), s));
            hipLaunchKernelGGL(k_find_hubs, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, s, in_ptr, out_ptr, n_out, hub->count, hub->nodes);
            hub->key_in = in_ptr, hub->key_out = out_ptr, hub->key_n = n_out;
            hub->known = -1, hub->pending = false;
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            if (hub->ready != nullptr && hipStreamIsCapturing(s, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone) {   // (not inside a capture)
                if (hipMemcpyAsync(hub->host_count, hub->count, sizeof(int), hipMemcpyDeviceToHost, s) == hipSuccess &&
                    hipEventRecord(hub->ready, s) == hipSuccess)
                    hub->pending = true;
            }
        }
        hipLaunchKernelGGL((k_hub_partials<H>), dim3(kHubChunks / (kAggThreads / 64)), dim3(kAggThreads), 0, s, e, A2h, A3h, ldn, in_ptr, ss, out_ptr,
                           out_pos, od, n_out, hub->count, hub->nodes, hub->partials);
    }
    // (measurement knob: unused dynamic LDS caps the workgroups resident per CU, i.e. the window of nodes in flight)
    const size_t dyn = (size_t)tuning(kTuneAggLdsKiB) * 1024;
#define GN_AGG_LAUNCH_S(NORM_, MODE_, FIN_, U_, WPS_, GRID_, SPLIT_)                                                                           \
    hipLaunchKernelGGL((k_node_aggregate<H, NORM_, MODE_, FIN_, U_, WPS_, SPLIT_>), dim3((unsigned)(GRID_)), dim3(kAggThreads), (FIN_) ? 0 : dyn, s, e, \
                       (FIN_) ? n_out : node_end, A1h, A2h, A3h, ldn, in_ptr, ss, out_ptr, out_pos, od, h_in, ldh, h_out, scale, shift,   \
                       (int)blocks, aux0, aux1, aux2, aux3, hub_count, hub_nodes, hub_partials, (FIN_) ? (int64_t)0 : node0, norm_width)
#define GN_AGG_LAUNCH(NORM_, MODE_, FIN_, U_, WPS_, GRID_) GN_AGG_LAUNCH_S(NORM_, MODE_, FIN_, U_, WPS_, GRID_, 1)
    // The regular launches run the split item loop (accumulate_items_split: 0.2207 -> 0.2103 ms per launch at configs[1], whole forward
    // 4.89 -> 4.78 ms, the same bits, the same 782 MB fetched); variant 6 keeps the single loop for A/B.  (A single loop with a
    // wave-uniform fast path for steps of in-edges only - one round fewer, the overlap only when a node has >= G U in-edges - measured
    // the same as the split form: 4.875 against 4.881 ms.)
    // items per lane group in flight: 4 at H <= 128.  Measured at configs[1] (tools/agg_time.py <H> variants): 1 item 0.2105 ms,
    // 2 items 0.2115, 4 items 0.2197, 8 items 0.2533 - the launch time hardly depends on the loads in flight per wave (the
    // kernel sits at the HBM rate this access pattern sustains) - but with 2 items the 8 resident waves per SIMD widen the
    // window of nodes in flight and the out-edge pass finds fewer of its rows still in L2: 979 MB fetched per launch against
    // 764 MB.  4 % of launch time is not worth 28 % more HBM traffic.
    // H = 256 (one 1 KB row per load instruction): 2 since round 4 - measured inside the 2.5M-edge forward 1.036 (U = 8) / 0.980 (2) / 0.969 ms (1) per launch,
    // the same bits (one lane group: the items are added in list order whatever U)
    constexpr int UD = H == 256 ? 2 : 4;
    constexpr int kFinGrid = kHubCap / (kAggThreads / 64);
    if (mode == 1) {
        GN_AGG_LAUNCH(GNNOME_NORM_AFFINE, 1, false, UD, 0, blocks);
        if (hub_pass) GN_AGG_LAUNCH(GNNOME_NORM_AFFINE, 1, true, UD, 0, kFinGrid);
    } else if (mode == 2) {
        GN_AGG_LAUNCH(GNNOME_NORM_AFFINE, 2, false, UD, 0, blocks);
        if (hub_pass) GN_AGG_LAUNCH(GNNOME_NORM_AFFINE, 2, true, UD, 0, kFinGrid);
    } else if (norm == GNNOME_NORM_AFFINE) {
        switch (tuning(kTuneAggVariant)) {   // A/B of occupancy against items in flight (tools/agg_time.py)
            case 1: GN_AGG_LAUNCH(GNNOME_NORM_AFFINE, 0, false, 4, 8, blocks); break;
            case 2: GN_AGG_LAUNCH(GNNOME_NORM_AFFINE, 0, false, 2, 8, blocks); break;
            case 3: GN_AGG_LAUNCH(GNNOME_NORM_AFFINE, 0, false, 1, 0, blocks); break;
            case 4: GN_AGG_LAUNCH(GNNOME_NORM_AFFINE, 0, false, 8, 0, blocks); break;
            case 5: GN_AGG_LAUNCH(GNNOME_NORM_AFFINE, 0, false, 2, 0, blocks); break;
            case 6: GN_AGG_LAUNCH_S(GNNOME_NORM_AFFINE, 0, false, UD, 0, blocks, 0); break;   // the unsplit item loop (in-edge rows requested after the index wait)
            case 7: GN_AGG_LAUNCH_S(GNNOME_NORM_AFFINE, 0, false, UD, 0, blocks, 2); break;   // measurement only: out-edges alone
            case 8: GN_AGG_LAUNCH_S(GNNOME_NORM_AFFINE, 0, false, UD, 0, blocks, 3); break;   // no branches around dead items
            case 14: GN_AGG_LAUNCH_S(GNNOME_NORM_AFFINE, 0, false, UD, 0, blocks, 4); break;  // nontemporal out-edge loads of e' 
            case 15: GN_AGG_LAUNCH_S(GNNOME_NORM_AFFINE, 0, false, UD, 0, blocks, 5); break;  // measurement only: no table-row misses
            case 11:   // ... walked as contiguous chunks by persistent workgroups (4 per CU)
            case 12:   // ... with nontemporal e loads
            case 13:   // the same, 8 workgroups per CU
                if (H == 128) {
                    const int64_t pblocks = ((node_end - node_begin + 1) / 2 + (kAggThreads / 64) - 1) / (kAggThreads / 64);
                    const int per_cu = tuning(kTuneAggVariant) == 13 ? 8 : 4;
                    const unsigned pgrid = (unsigned)std::min<int64_t>(pblocks, (int64_t)persistent_grid() * per_cu);
                    if (tuning(kTuneAggVariant) == 11)
                        hipLaunchKernelGGL((k_node_aggregate_pair<GNNOME_NORM_AFFINE, 4, true, false>), dim3(pgrid), dim3(kAggThreads), dyn, s, e, node_end, A1h, A2h, A3h, ldn,
                                           in_ptr, ss, out_ptr, out_pos, od, h_in, ldh, h_out, scale, shift, (int)pblocks, hub_count, hub_nodes, node0);
                    else
                        hipLaunchKernelGGL((k_node_aggregate_pair<GNNOME_NORM_AFFINE, 4, true, true>), dim3(pgrid), dim3(kAggThreads), dyn, s, e, node_end, A1h, A2h, A3h, ldn,
                                           in_ptr, ss, out_ptr, out_pos, od, h_in, ldh, h_out, scale, shift, (int)pblocks, hub_count, hub_nodes, node0);
                } else {
                    GN_AGG_LAUNCH(GNNOME_NORM_AFFINE, 0, false, UD, 0, blocks);
                }
                break;
            case 9:   // two nodes per wave (k_node_aggregate_pair), H = 128
            case 10:
                if (H == 128) {
                    const int64_t pblocks = ((node_end - node_begin + 1) / 2 + (kAggThreads / 64) - 1) / (kAggThreads / 64);
                    if (tuning(kTuneAggVariant) == 9)
                        hipLaunchKernelGGL((k_node_aggregate_pair<GNNOME_NORM_AFFINE, 4>), dim3((unsigned)pblocks), dim3(kAggThreads), dyn, s, e, node_end, A1h, A2h, A3h, ldn,
                                           in_ptr, ss, out_ptr, out_pos, od, h_in, ldh, h_out, scale, shift, (int)pblocks, hub_count, hub_nodes, node0);
                    else
                        hipLaunchKernelGGL((k_node_aggregate_pair<GNNOME_NORM_AFFINE, 2>), dim3((unsigned)pblocks), dim3(kAggThreads), dyn, s, e, node_end, A1h, A2h, A3h, ldn,
                                           in_ptr, ss, out_ptr, out_pos, od, h_in, ldh, h_out, scale, shift, (int)pblocks, hub_count, hub_nodes, node0);
                } else {
                    GN_AGG_LAUNCH(GNNOME_NORM_AFFINE, 0, false, UD, 0, blocks);
                }
                break;
            default:
                if (g_node_records != nullptr && node_begin == 0 && node_end == n_out)
                    hipLaunchKernelGGL((k_node_aggregate<H, GNNOME_NORM_AFFINE, 0, false, UD, 0, 1, true>), dim3((unsigned)blocks), dim3(kAggThreads), dyn, s, e, node_end,
                                       A1h, A2h, A3h, ldn, in_ptr, ss, out_ptr, out_pos, od, h_in, ldh, h_out, scale, shift, (int)blocks, aux0, aux1, aux2, aux3,
                                       hub_count, hub_nodes, hub_partials, node0, norm_width, g_node_records);
                else
                    GN_AGG_LAUNCH(GNNOME_NORM_AFFINE, 0, false, UD, 0, blocks);
                break;
        }
        if (hub_pass) GN_AGG_LAUNCH(GNNOME_NORM_AFFINE, 0, true, UD, 0, kFinGrid);
    } else {
        GN_AGG_LAUNCH(GNNOME_NORM_LAYER, 0, false, UD, 0, blocks);
        if (hub_pass) GN_AGG_LAUNCH(GNNOME_NORM_LAYER, 0, true, UD, 0, kFinGrid);
    }
#undef GN_AGG_LAUNCH
#undef GN_AGG_LAUNCH_S
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}

}  // namespace gnnome

// Round 6 experiment: records[N][64] for the aggregation's REC form, and the switch that makes the following
// gnnome_node_aggregate_f32 calls of this thread (inference form, BatchNorm, whole node range) read them.  NULL = off.
extern "C" int gnnome_build_node_records(const int32_t* in_ptr, const int32_t* srt_src, const int32_t* out_ptr, const int32_t* out_pos,
                                         const int32_t* out_dst, int64_t num_nodes, int32_t* records, void* stream) {
    using namespace gnnome;
    GN_REQUIRE(num_nodes >= 0 && num_nodes < (1ll << 25), "build_node_records: node count out of range");
    if (num_nodes == 0) return GNNOME_OK;
    GN_REQUIRE(in_ptr && out_ptr && records && (uintptr_t)records % 256 == 0, "build_node_records: null or misaligned pointer");
    hipLaunchKernelGGL(k_build_node_records, dim3((unsigned)((num_nodes * 64 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in_ptr, srt_src, out_ptr,
                       out_pos, out_dst, num_nodes, records);
    GN_LAUNCH_CHECK();
    return GNNOME_OK;
}
extern "C" int gnnome_debug_node_records(const int32_t* records) {
    gnnome::g_node_records = records;
    return GNNOME_OK;
}

extern "C" int gnnome_node_aggregate_f32(const float* e, int hidden, int64_t num_nodes_out, const float* A1h,
                                         const float* A2h, const float* A3h, int ld_node, const int32_t* in_ptr,
                                         const int32_t* srt_src, const int32_t* out_ptr, const int32_t* out_pos,
                                         const int32_t* out_dst, const float* h_in, int ld_h, float* h_out,
                                         int norm_kind, const float* norm_scale, const float* norm_shift, void* stream) {
    using namespace gnnome;
    GN_REQUIRE(num_nodes_out >= 0, "node_aggregate: negative node count");
    if (num_nodes_out == 0) return GNNOME_OK;
    // e / srt_src / out_pos / out_dst may be NULL for a graph without edges (never dereferenced then)
    GN_REQUIRE(A1h && A2h && A3h && in_ptr && out_ptr && h_in && h_out && norm_scale && norm_shift,
               "node_aggregate: null pointer");
    const int norm_width = (norm_kind >> 8) ? (norm_kind >> 8) : hidden;   // GNNOME_NORM_LAYER_OVER(w), see gnnome_edge_gate_f32
    norm_kind &= 0xFF;
    GN_REQUIRE(norm_width >= 1 && norm_width <= hidden && (norm_width == hidden || norm_kind == GNNOME_NORM_LAYER), "node_aggregate: bad norm width %d", norm_width);
    GN_REQUIRE(norm_kind == GNNOME_NORM_AFFINE || norm_kind == GNNOME_NORM_LAYER, "node_aggregate: bad norm_kind %d",
               norm_kind);
    GN_REQUIRE(ld_node >= hidden && ld_node % 4 == 0 && ld_h >= hidden && ld_h % 4 == 0, "node_aggregate: bad strides");
    GN_REQUIRE(((uintptr_t)A1h % 16 == 0) && ((uintptr_t)A2h % 16 == 0) && ((uintptr_t)A3h % 16 == 0) &&
                   ((uintptr_t)h_in % 16 == 0) && ((uintptr_t)h_out % 16 == 0) && ((uintptr_t)e % 16 == 0),
               "node_aggregate: tensors must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    switch (hidden) {
        case 64: return launch_agg<64>(e, num_nodes_out, A1h, A2h, A3h, ld_node, in_ptr, srt_src, out_ptr, out_pos, out_dst, h_in, ld_h, h_out, norm_kind, norm_scale, norm_shift, s, 0, nullptr, nullptr, nullptr, nullptr, 0, -1, norm_width);
        case 128: return launch_agg<128>(e, num_nodes_out, A1h, A2h, A3h, ld_node, in_ptr, srt_src, out_ptr, out_pos, out_dst, h_in, ld_h, h_out, norm_kind, norm_scale, norm_shift, s, 0, nullptr, nullptr, nullptr, nullptr, 0, -1, norm_width);
        case 256: return launch_agg<256>(e, num_nodes_out, A1h, A2h, A3h, ld_node, in_ptr, srt_src, out_ptr, out_pos, out_dst, h_in, ld_h, h_out, norm_kind, norm_scale, norm_shift, s, 0, nullptr, nullptr, nullptr, nullptr, 0, -1, norm_width);
        default: set_error("node_aggregate: hidden=%d not in {64,128,256}", hidden); return GNNOME_EINVAL;
    }
}

extern "C" int gnnome_node_aggregate_range_f32(const float* e, int hidden, int64_t num_nodes_out, int64_t node_begin, int64_t node_end,
                                               const float* A1h, const float* A2h, const float* A3h, int ld_node, const int32_t* in_ptr,
                                               const int32_t* srt_src, const int32_t* out_ptr, const int32_t* out_pos,
                                               const int32_t* out_dst, const float* h_in, int ld_h, float* h_out, int norm_kind,
                                               const float* norm_scale, const float* norm_shift, void* stream) {
    using namespace gnnome;
    GN_REQUIRE(num_nodes_out > 0, "node_aggregate_range: empty graph");
    GN_REQUIRE(A1h && A2h && A3h && in_ptr && out_ptr && h_in && h_out && norm_scale && norm_shift, "node_aggregate_range: null pointer");
    const int norm_width = (norm_kind >> 8) ? (norm_kind >> 8) : hidden;   // GNNOME_NORM_LAYER_OVER(w), see gnnome_edge_gate_f32
    norm_kind &= 0xFF;
    GN_REQUIRE(norm_width >= 1 && norm_width <= hidden && (norm_width == hidden || norm_kind == GNNOME_NORM_LAYER), "node_aggregate_range: bad norm width %d", norm_width);
    GN_REQUIRE(norm_kind == GNNOME_NORM_AFFINE || norm_kind == GNNOME_NORM_LAYER, "node_aggregate_range: bad norm_kind %d", norm_kind);
    GN_REQUIRE(ld_node >= hidden && ld_node % 4 == 0 && ld_h >= hidden && ld_h % 4 == 0, "node_aggregate_range: bad strides");
    GN_REQUIRE(((uintptr_t)A1h % 16 == 0) && ((uintptr_t)A2h % 16 == 0) && ((uintptr_t)A3h % 16 == 0) &&
                   ((uintptr_t)h_in % 16 == 0) && ((uintptr_t)h_out % 16 == 0) && ((uintptr_t)e % 16 == 0),
               "node_aggregate_range: tensors must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    switch (hidden) {
        case 64: return launch_agg<64>(e, num_nodes_out, A1h, A2h, A3h, ld_node, in_ptr, srt_src, out_ptr, out_pos, out_dst, h_in, ld_h, h_out, norm_kind, norm_scale, norm_shift, s, 0, nullptr, nullptr, nullptr, nullptr, node_begin, node_end, norm_width);
        case 128: return launch_agg<128>(e, num_nodes_out, A1h, A2h, A3h, ld_node, in_ptr, srt_src, out_ptr, out_pos, out_dst, h_in, ld_h, h_out, norm_kind, norm_scale, norm_shift, s, 0, nullptr, nullptr, nullptr, nullptr, node_begin, node_end, norm_width);
        case 256: return launch_agg<256>(e, num_nodes_out, A1h, A2h, A3h, ld_node, in_ptr, srt_src, out_ptr, out_pos, out_dst, h_in, ld_h, h_out, norm_kind, norm_scale, norm_shift, s, 0, nullptr, nullptr, nullptr, nullptr, node_begin, node_end, norm_width);
        default: set_error("node_aggregate_range: hidden=%d not in {64,128,256}", hidden); return GNNOME_EINVAL;
    }
}

extern "C" int gnnome_node_aggregate_raw_f32(const float* e, int hidden, int64_t num_nodes, int mode, const float* A1h,
                                             const float* A2h, const float* A3h, int ld_node, const int32_t* in_ptr,
                                             const int32_t* srt_src, const int32_t* out_ptr, const int32_t* out_pos,
                                             const int32_t* out_dst, float* v_out, float* aux0, float* aux1, float* aux2,
                                             float* aux3, void* stream) {
    using namespace gnnome;
    GN_REQUIRE(num_nodes >= 0 && (mode == 1 || mode == 2), "node_aggregate_raw: mode must be 1 or 2");
    if (num_nodes == 0) return GNNOME_OK;
    GN_REQUIRE(A2h && A3h && in_ptr && out_ptr && aux0 && aux2 && ld_node >= hidden && ld_node % 4 == 0, "node_aggregate_raw: bad arguments");
    GN_REQUIRE(mode == 2 || (A1h && v_out && aux1 && aux3), "node_aggregate_raw: mode 1 needs A1h, v_out, aux1, aux3");
    hipStream_t s = (hipStream_t)stream;
    switch (hidden) {
        case 64: return launch_agg<64>(e, num_nodes, A1h, A2h, A3h, ld_node, in_ptr, srt_src, out_ptr, out_pos, out_dst, nullptr, hidden, v_out, 0, nullptr, nullptr, s, mode, aux0, aux1, aux2, aux3);
        case 128: return launch_agg<128>(e, num_nodes, A1h, A2h, A3h, ld_node, in_ptr, srt_src, out_ptr, out_pos, out_dst, nullptr, hidden, v_out, 0, nullptr, nullptr, s, mode, aux0, aux1, aux2, aux3);
        case 256: return launch_agg<256>(e, num_nodes, A1h, A2h, A3h, ld_node, in_ptr, srt_src, out_ptr, out_pos, out_dst, nullptr, hidden, v_out, 0, nullptr, nullptr, s, mode, aux0, aux1, aux2, aux3);
        default: set_error("node_aggregate_raw: hidden=%d not in {64,128,256}", hidden); return GNNOME_EINVAL;
    }
}
