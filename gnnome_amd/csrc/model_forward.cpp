// gnnome_model_forward_f32 / gnnome_model_forward_buffers_f32: models/full_graph.py:22-30 as one host call (round 6).
//
// A sequencer over the library's own per-kernel entries, in the order gnnome_amd/engine.py::run_stack calls them: it adds no kernel and no
// arithmetic, it removes ~30 interpreter round trips (host time per forward with an empty launch queue 0.15 against 0.48 ms; a forward that
// keeps the GPU busy was never waiting for them).  Every decision run_stack takes per layer is taken here the same way: the
// reference-order kernels per layer, the edge encoder folded into layer 0's gate, the two [E,H] buffers taking turns at H = 256, the
// projection on the weights' fp16x3 planes.  The buffers come as one workspace block or one by one (placement in HBM: DESIGN.md section 4).
#include "common.h"

namespace gnnome {
namespace {

constexpr size_t kAlign = 256;
size_t rounded(size_t bytes) { return (bytes + kAlign - 1) / kAlign * kAlign; }

struct Workspace {
    size_t h[2], P, e[2], PQ, total;
};
Workspace layout(int64_t n, int64_t e, int hidden, int hs) {
    Workspace w;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off += rounded(bytes);
        return at;
    };
    w.h[0] = take(sizeof(float) * n * hidden);
    w.h[1] = take(sizeof(float) * n * hidden);
    w.P = take(sizeof(float) * n * 5 * hidden);
    w.e[0] = take(sizeof(float) * e * hidden);
    w.e[1] = hidden == 256 ? take(sizeof(float) * e * hidden) : w.e[0];   // H = 256: the gate writes a second buffer, the two take turns
    w.PQ = take(sizeof(float) * n * 2 * hs);
    w.total = off;
    return w;
}

// gnnome_debug_forward_events: HIP events the NEXT forwards record around one layer's gate and aggregation launches (bench.py's live roofline)
struct ForwardEvents {
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    int layer = -1;
};
thread_local ForwardEvents g_events;
int mark(int which, int layer, void* stream) {
    if (layer != g_events.layer || g_events.ev[which] == nullptr) return GNNOME_OK;
    GN_HIP(hipEventRecord(g_events.ev[which], (hipStream_t)stream));
    return GNNOME_OK;
}

int project(const float* A, int64_t M, int K, const float* W, const void* planes, const float* bias, int Nout, float* C, void* stream) {
    if (planes != nullptr && gnnome_linear_planes_route(M, K, Nout, 1)) return gnnome_linear_planes_f32(A, M, K, K, planes, bias, Nout, C, Nout, stream);
    return gnnome_linear_f32(A, M, K, K, W, K, bias, Nout, C, Nout, stream);
}

}  // namespace
}  // namespace gnnome

extern "C" int gnnome_debug_forward_events(void* gate_start, void* gate_stop, void* aggregate_start, void* aggregate_stop, int layer) {
    using namespace gnnome;
    g_events.ev[0] = (hipEvent_t)gate_start;
    g_events.ev[1] = (hipEvent_t)gate_stop;
    g_events.ev[2] = (hipEvent_t)aggregate_start;
    g_events.ev[3] = (hipEvent_t)aggregate_stop;
    g_events.layer = layer;
    return GNNOME_OK;
}

extern "C" int gnnome_model_forward_workspace_bytes(int64_t num_nodes, int64_t num_edges, int hidden, int score_hidden, size_t* bytes_host) {
    using namespace gnnome;
    GN_REQUIRE(bytes_host && num_nodes >= 0 && num_edges >= 0 && hidden > 0 && score_hidden > 0, "model_forward_workspace_bytes: bad argument");
    *bytes_host = layout(num_nodes, num_edges, hidden, score_hidden).total;
    return GNNOME_OK;
}

#define GN_TRY(call)                       \
    do {                                   \
        const int rc__ = (call);           \
        if (rc__ != GNNOME_OK) return rc__; \
    } while (0)

namespace gnnome {
namespace {
int forward_on(const gnnome_model_params* m, const gnnome_views* g, const float* x, const float* e_raw, float* logits, float* const h[2], float* P,
               float* const eb[2], float* PQ, void* stream);
}  // namespace
}  // namespace gnnome

static int check_model(const gnnome_model_params* m, const gnnome_views* g) {
    using namespace gnnome;
    GN_REQUIRE(m && g, "model_forward: null parameter block");
    const int H = m->hidden, hs = m->score_hidden, L = m->num_layers;
    GN_REQUIRE(g->num_nodes >= 0 && g->num_edges >= 0 && L >= 0 && (L == 0 || m->layers_host != nullptr), "model_forward: bad sizes");
    GN_REQUIRE(H == 64 || H == 128 || H == 256, "model_forward: hidden=%d (the kernels are built for 64 / 128 / 256; pad narrower models)", H);
    GN_REQUIRE(hs == 32 || hs == 64 || hs == 128, "model_forward: score_hidden=%d (built for 32 / 64 / 128)", hs);
    return GNNOME_OK;
}

extern "C" int gnnome_model_forward_f32(const gnnome_model_params* m, const gnnome_views* g, const float* x, const float* e_raw, float* logits,
                                        void* workspace, size_t workspace_bytes, void* stream) {
    using namespace gnnome;
    GN_TRY(check_model(m, g));
    const int64_t N = g->num_nodes, E = g->num_edges;
    const Workspace ws = layout(N, E, m->hidden, m->score_hidden);
    if (workspace_bytes < ws.total) {
        set_error("model_forward: workspace of %zu bytes, %zu needed", workspace_bytes, ws.total);
        return GNNOME_EWORKSPACE;
    }
    GN_REQUIRE((N == 0 && E == 0) || (workspace != nullptr && (uintptr_t)workspace % kAlign == 0), "model_forward: workspace must be 256-byte aligned");
    GN_REQUIRE(E == 0 || logits != nullptr, "model_forward: null logits");
    char* base = static_cast<char*>(workspace);
    float* h[2] = {reinterpret_cast<float*>(base + ws.h[0]), reinterpret_cast<float*>(base + ws.h[1])};
    float* eb[2] = {reinterpret_cast<float*>(base + ws.e[0]), reinterpret_cast<float*>(base + ws.e[1])};
    return forward_on(m, g, x, e_raw, logits, h, reinterpret_cast<float*>(base + ws.P), eb, reinterpret_cast<float*>(base + ws.PQ), stream);
}

// The same forward on buffers the caller allocated ONE BY ONE (round 6).  Where the driver places a buffer in HBM is worth 3 % of configs[1]'s forward, and
// one block holding everything comes out on the slow side of that more often than five blocks of their own do (4.24 against 4.10 ms on the same box, the
// kernels being the same: NOTES round 6, profiles/r06_placement_*.txt) - gnnome_amd.ops.model_forward hands over torch allocations.
extern "C" int gnnome_model_forward_buffers_f32(const gnnome_model_params* m, const gnnome_views* g, const float* x, const float* e_raw, float* logits,
                                                const gnnome_forward_buffers* b, void* stream) {
    using namespace gnnome;
    GN_TRY(check_model(m, g));
    GN_REQUIRE(b != nullptr, "model_forward_buffers: null buffer block");
    const int64_t N = g->num_nodes, E = g->num_edges;
    GN_REQUIRE(E == 0 || logits != nullptr, "model_forward_buffers: null logits");
    GN_REQUIRE(N == 0 || (b->h[0] && b->h[1] && b->P && b->PQ && b->h[0] != b->h[1]), "model_forward_buffers: node buffers h[0], h[1], P, PQ");
    GN_REQUIRE(E == 0 || (b->e[0] && (m->hidden != 256 || (b->e[1] && b->e[1] != b->e[0]))), "model_forward_buffers: edge buffers e[0] (and e[1] at hidden = 256)");
    for (const void* p : {(const void*)b->h[0], (const void*)b->h[1], (const void*)b->P, (const void*)b->e[0], (const void*)b->e[1], (const void*)b->PQ})
        GN_REQUIRE((uintptr_t)p % 16 == 0, "model_forward_buffers: buffers must be 16-byte aligned");
    float* h[2] = {b->h[0], b->h[1]};
    float* eb[2] = {b->e[0], m->hidden == 256 ? b->e[1] : b->e[0]};
    return forward_on(m, g, x, e_raw, logits, h, b->P, eb, b->PQ, stream);
}

namespace gnnome {
namespace {
int forward_on(const gnnome_model_params* m, const gnnome_views* g, const float* x, const float* e_raw, float* logits, float* const h[2], float* P,
               float* const eb[2], float* PQ, void* stream) {
    const int64_t N = g->num_nodes, E = g->num_edges;
    const int H = m->hidden, hs = m->score_hidden, L = m->num_layers;

    // models/full_graph.py:26 - the node encoder, rows in the views' numbering
    GN_TRY(gnnome_encode_f32(x, N, m->node_features, g->node_gather, m->node_W1, m->node_b1, m->hidden_ne, m->node_W2, m->node_b2, H, h[0], stream));
    // :27 - the edge encoder: folded into layer 0's gate where that kernel exists (engine.encode_edges / ops.can_fuse_edge_encoder), else here,
    // already in destination-sorted order
    const bool wide_ok = tuning(kTuneArith) == 0 && tuning(kTuneGateVariant) == 0;
    const bool fold = L > 0 && m->edge_features == 2 && m->hidden_ne == 16 && (H == 64 || H == 128 || (H == 256 && wide_ok)) &&
                      m->layers_host[0].norm_kind == GNNOME_NORM_AFFINE && E > 0;
    float* e = nullptr;   // NULL until the first gate has produced it (fold)
    if (!fold) {
        e = eb[0];
        GN_TRY(gnnome_encode_f32(e_raw, E, m->edge_features, g->srt_eid, m->edge_W1, m->edge_b1, m->hidden_ne, m->edge_W2, m->edge_b2, H, e, stream));
    }
    int cur = 0;
    for (int i = 0; i < L; ++i) {
        const gnnome_layer_params& lw = m->layers_host[i];
        // gated_gcn_full.py:91-96 as one GEMM
        if (lw.reference_order) GN_TRY(gnnome_linear_ref_f32(h[cur], N, H, H, lw.Wcat, H, lw.bcat, 5 * H, P, 5 * H, stream));
        else GN_TRY(project(h[cur], N, H, lw.Wcat, lw.Wcat_planes, lw.bcat, 5 * H, P, stream));
        const float *A1 = P, *A2 = P + H, *A3 = P + 2 * H, *B1 = P + 3 * H, *B2 = P + 4 * H;
        if (g->transposed) {   // dgl.reverse(g): src <-> dst (GraphViews.reversed)
            const float* t = A2; A2 = A3; A3 = t;
            t = B1; B1 = B2; B2 = t;
        }
        // :97,104-110 (+117-122, identical)
        GN_TRY(mark(0, i, stream));
        if (lw.reference_order) {
            float* out = e ? e : eb[0];
            GN_TRY(gnnome_edge_gate_ref_f32(e, out, E, H, B1, B2, 5 * H, g->srt_src, g->srt_dst, lw.W3, H, lw.b3, lw.scale_e, lw.shift_e,
                                            e ? nullptr : e_raw, e ? nullptr : g->srt_eid, e ? nullptr : m->edge_W1, e ? nullptr : m->edge_b1,
                                            e ? nullptr : m->edge_W2, e ? nullptr : m->edge_b2, stream));
            e = out;
        } else if (e == nullptr) {
            GN_TRY(gnnome_edge_gate_encode_f32(e_raw, g->srt_eid, m->edge_W1, m->edge_b1, m->edge_W2, m->edge_b2, eb[0], E, H, B1, B2, 5 * H, g->srt_src,
                                               g->srt_dst, lw.W3, H, lw.scale_e, lw.shift_e, stream));
            e = eb[0];
        } else if (H == 256 && lw.norm_kind == GNNOME_NORM_AFFINE) {   // engine.gate_update: out of place, the two buffers take turns
            float* out = e == eb[0] ? eb[1] : eb[0];
            GN_TRY(gnnome_edge_gate_f32(e, out, E, H, B1, B2, 5 * H, g->srt_src, g->srt_dst, lw.W3, H, lw.norm_kind, lw.scale_e, lw.shift_e, stream));
            e = out;
        } else {
            GN_TRY(gnnome_edge_gate_f32(e, e, E, H, B1, B2, 5 * H, g->srt_src, g->srt_dst, lw.W3, H, lw.norm_kind, lw.scale_e, lw.shift_e, stream));
        }
        GN_TRY(mark(1, i, stream));
        // :111-114,124-137
        GN_TRY(mark(2, i, stream));
        GN_TRY(gnnome_node_aggregate_f32(e, H, N, A1, A2, A3, 5 * H, g->in_ptr, g->srt_src, g->out_ptr, g->out_pos, g->out_dst, h[cur], H, h[cur ^ 1],
                                         lw.norm_kind, lw.scale_h, lw.shift_h, stream));
        GN_TRY(mark(3, i, stream));
        cur ^= 1;
    }
    // score_predictor.py:12-17: the node halves of W1 once per node, then the per-edge tail
    GN_TRY(project(h[cur], N, H, m->W_nodes, m->W_nodes_planes, m->b_nodes, 2 * hs, PQ, stream));
    const float *Ps = PQ, *Qd = PQ + hs;
    if (g->transposed) {
        const float* t = Ps; Ps = Qd; Qd = t;
    }
    return gnnome_edge_score_f32(e, E, H, hs, Ps, Qd, 2 * hs, g->srt_src, g->srt_dst, g->srt_eid, m->W1e, m->ld_w1e, m->W2, m->b2, m->W3, m->b3, logits,
                                 nullptr, stream);
}
}  // namespace
}  // namespace gnnome
