// torch.ops.gnnome_hip.* as a COMPILED PyTorch-ROCm extension (north_star: "surfaced to Python through a PyTorch-ROCm C++/HIP
// extension").  The drop-in boundary stays the C ABI of libgnnome_hip.so (include/gnnome_hip.h: plain pointers, sizes and a stream); this file is
// the PyTorch side of it - TORCH_LIBRARY schemas, kernels under the dispatcher's CUDA key (= HIP on ROCm) that check their tensors, take the
// caller's current stream and call the C entries, and Meta kernels (shape inference: FakeTensor / torch.compile tracing).  No CPU kernels: on
// any other device the dispatcher raises.  Inference operators, no autograd formula (the training step is one autograd.Function,
// gnnome_amd/train.py).  What a maintainer calls from their own module code:
//
//     P      = torch.ops.gnnome_hip.linear(h, Wcat, bcat)                                                   # gated_gcn_full.py:91-96
//     e_new  = torch.ops.gnnome_hip.edge_gate(e, B1h, B2h, srt_src, srt_dst, W3, scale, shift, norm_kind)   # :97, 104-110
//     h_new  = torch.ops.gnnome_hip.node_aggregate(e_new, A1h, A2h, A3h, in_ptr, srt_src, out_ptr, out_pos, out_dst, h, scale, shift, norm_kind)
//     logits = torch.ops.gnnome_hip.edge_score(e, Ps, Qd, srt_src, srt_dst, srt_eid, W1e, W2, b2, W3, b3)   # score_predictor.py:12-24
//
// Built by gnnome_amd/csrc/Makefile (target torch_ext: g++ against this interpreter's torch headers, linked to libgnnome_hip.so beside it);
// gnnome_amd/torch_ops.py loads it.  It registers operators and nothing else: there is no Python object in here.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/library.h>

#include <algorithm>
#include <cstdint>
#include <tuple>

#include "gnnome_hip.h"

namespace {

using at::Tensor;
using OptTensor = std::optional<Tensor>;
using Guard = c10::hip::OptionalHIPGuardMasqueradingAsCUDA;

void ok(int rc, const char* entry) { TORCH_CHECK(rc == 0, "gnnome_hip: ", entry, " failed (", rc, "): ", gnnome_last_error()); }

void* stream_of(const Tensor& t) { return (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream(); }

const float* f32(const Tensor& t, const char* name) {
    TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kFloat, name, ": expected a float32 tensor on the GPU, got ", t.scalar_type(), " on ", t.device());
    return t.const_data_ptr<float>();
}

const float* f32_dense(const Tensor& t, const char* name) {
    TORCH_CHECK(t.is_contiguous(), name, ": expected a contiguous tensor");
    return f32(t, name);
}

const float* f32_opt(const OptTensor& t, const char* name) { return t.has_value() && t->defined() ? f32_dense(*t, name) : nullptr; }

const int32_t* i32(const Tensor& t, const char* name) {
    TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kInt && t.is_contiguous(), name, ": expected a contiguous int32 tensor on the GPU, got ",
                t.scalar_type(), " on ", t.device());
    return t.const_data_ptr<int32_t>();
}

// 2-D float32 with unit column stride -> row stride in elements (gnnome_amd.ops._rows)
int rows_ld(const Tensor& t, const char* name) {
    f32(t, name);
    TORCH_CHECK(t.dim() == 2 && (t.size(1) <= 1 || t.stride(1) == 1), name, ": expected a 2-D tensor with contiguous rows, got sizes ", t.sizes(),
                " strides ", t.strides());
    const int64_t ld = t.size(0) > 1 ? t.stride(0) : std::max<int64_t>(t.stride(0), t.size(1));
    TORCH_CHECK(ld >= t.size(1) && ld <= INT32_MAX, name, ": row stride ", ld, " outside what the C ABI takes (an int, >= the row's width)");
    return (int)ld;
}

bool aligned16(const Tensor& t) { return reinterpret_cast<uintptr_t>(t.const_data_ptr()) % 16 == 0; }

// ---- graph views (gnnome_build_graph_views; ops.GraphViews with validate = "now") -------------------------------------------------------------
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor> build_graph_views(const Tensor& src_in, const Tensor& dst_in, int64_t num_nodes) {
    TORCH_CHECK(src_in.is_cuda() && dst_in.is_cuda() && src_in.dim() == 1 && dst_in.dim() == 1, "build_graph_views: src, dst are 1-D tensors on the GPU");
    TORCH_CHECK(src_in.numel() == dst_in.numel(), "build_graph_views: src and dst differ in length");
    TORCH_CHECK(num_nodes >= 0, "build_graph_views: negative node count");
    Guard guard(src_in.device());
    const Tensor src = src_in.to(at::kInt).contiguous(), dst = dst_in.to(at::kInt).contiguous();
    const int64_t n = num_nodes, e = src.numel();
    if (e > 0) {   // endpoints are range-checked before anything is built (one host sync)
        const int lo = std::min(src.min().item<int>(), dst.min().item<int>()), hi = std::max(src.max().item<int>(), dst.max().item<int>());
        TORCH_CHECK_INDEX(lo >= 0 && hi < n, "edge endpoint out of range [0,", n, "): min ", lo, ", max ", hi);
    }
    const auto opt = src.options();
    Tensor in_ptr = at::empty({n + 1}, opt), out_ptr = at::empty({n + 1}, opt);
    Tensor srt_src = at::empty({e}, opt), srt_dst = at::empty({e}, opt), srt_eid = at::empty({e}, opt), out_pos = at::empty({e}, opt), out_dst = at::empty({e}, opt);
    size_t need = 0;
    ok(gnnome_graph_views_workspace_bytes(n, e, &need), "gnnome_graph_views_workspace_bytes");
    Tensor ws = at::empty({(int64_t)std::max<size_t>(need, 1)}, opt.dtype(at::kByte));   // freed on the stream it was used on (one stream: the caller's)
    ok(gnnome_build_graph_views(i32(src, "src"), i32(dst, "dst"), n, e, in_ptr.data_ptr<int32_t>(), srt_src.data_ptr<int32_t>(),
                                srt_dst.data_ptr<int32_t>(), srt_eid.data_ptr<int32_t>(), out_ptr.data_ptr<int32_t>(), out_pos.data_ptr<int32_t>(),
                                out_dst.data_ptr<int32_t>(), ws.data_ptr(), (size_t)ws.numel(), stream_of(src)),
       "gnnome_build_graph_views");
    return {in_ptr, srt_src, srt_dst, srt_eid, out_ptr, out_pos, out_dst};
}

// ---- encoders (gnnome_encode_f32; models/full_graph.py:26-27) -----------------------------------------------------------------------------------
Tensor encode(const Tensor& x, const Tensor& W1, const Tensor& b1, const Tensor& W2, const Tensor& b2, const OptTensor& gather) {
    TORCH_CHECK(x.dim() == 2 && W1.dim() == 2 && W2.dim() == 2 && W1.size(1) == x.size(1) && W2.size(1) == W1.size(0) && b1.numel() == W1.size(0) &&
                    b2.numel() == W2.size(0),
                "encode: shapes x[rows,F] W1[M,F] b1[M] W2[H,M] b2[H]");
    Guard guard(x.device());
    const bool g = gather.has_value() && gather->defined();
    const int64_t rows = g ? gather->numel() : x.size(0);
    Tensor out = at::empty({rows, W2.size(0)}, x.options());
    ok(gnnome_encode_f32(f32_dense(x, "encode.x"), rows, (int)x.size(1), g ? i32(*gather, "encode.gather") : nullptr, f32_dense(W1, "encode.W1"),
                         f32_dense(b1, "encode.b1"), (int)W1.size(0), f32_dense(W2, "encode.W2"), f32_dense(b2, "encode.b2"), (int)W2.size(0),
                         out.data_ptr<float>(), stream_of(x)),
       "gnnome_encode_f32");
    return out;
}

// ---- dense linear (gnnome_linear_f32 / gnnome_linear_planes_f32; gated_gcn_full.py:91-96, score_predictor.py:13-14) ----------------------------
Tensor linear(const Tensor& A, const Tensor& W, const OptTensor& bias) {
    const int lda = rows_ld(A, "linear.A"), ldw = rows_ld(W, "linear.W");
    TORCH_CHECK(A.size(1) == W.size(1), "linear: A[M,K] W[Nout,K]");
    const int64_t M = A.size(0);
    const int K = (int)A.size(1), Nout = (int)W.size(0);
    TORCH_CHECK(!bias.has_value() || !bias->defined() || bias->numel() == Nout, "linear: bias[Nout]");
    Guard guard(A.device());
    Tensor out = at::empty({M, (int64_t)Nout}, A.options());
    const float* b = f32_opt(bias, "linear.bias");
    // the node projections' shapes run on the fp16x3 planes of W (made here, per call: a module that keeps its weights calls the C ABI's
    // gnnome_weight_planes_f16 once and gnnome_linear_planes_f32 per forward, as gnnome_amd.engine does) - the library's own rule decides
    if (lda % 4 == 0 && ldw % 4 == 0 && aligned16(A) && aligned16(W) && gnnome_linear_planes_route(M, K, Nout, 0) == 1) {
        Tensor planes = at::empty({(int64_t)Nout * K * 2}, A.options().dtype(at::kHalf));
        ok(gnnome_weight_planes_f16(f32(W, "linear.W"), ldw, Nout, K, planes.data_ptr(), stream_of(A)), "gnnome_weight_planes_f16");
        ok(gnnome_linear_planes_f32(f32(A, "linear.A"), M, K, lda, planes.const_data_ptr(), b, Nout, out.data_ptr<float>(), Nout, stream_of(A)),
           "gnnome_linear_planes_f32");
        return out;
    }
    ok(gnnome_linear_f32(f32(A, "linear.A"), M, K, lda, f32(W, "linear.W"), ldw, b, Nout, out.data_ptr<float>(), Nout, stream_of(A)), "gnnome_linear_f32");
    return out;
}

// the same product in the reference's order of evaluation (gnnome_linear_ref_f32: torch's CPU nn.Linear bit for bit)
Tensor linear_ref(const Tensor& A, const Tensor& W, const OptTensor& bias) {
    const int lda = rows_ld(A, "linear_ref.A"), ldw = rows_ld(W, "linear_ref.W");
    TORCH_CHECK(A.size(1) == W.size(1), "linear_ref: A[M,K] W[Nout,K]");
    Guard guard(A.device());
    Tensor out = at::empty({A.size(0), W.size(0)}, A.options());
    ok(gnnome_linear_ref_f32(f32(A, "linear_ref.A"), A.size(0), (int)A.size(1), lda, f32(W, "linear_ref.W"), ldw, f32_opt(bias, "linear_ref.bias"),
                             (int)W.size(0), out.data_ptr<float>(), (int)W.size(0), stream_of(A)),
       "gnnome_linear_ref_f32");
    return out;
}

// ---- the gate (gnnome_edge_gate_f32; gated_gcn_full.py:97,104-110).  Functional: the operator does not modify e. --------------------------------
Tensor edge_gate(const Tensor& e_in, const Tensor& B1h, const Tensor& B2h, const Tensor& srt_src, const Tensor& srt_dst, const Tensor& W3,
                 const Tensor& scale, const Tensor& shift, int64_t norm_kind) {
    TORCH_CHECK(e_in.dim() == 2, "edge_gate: e[E,H]");
    Guard guard(e_in.device());
    const Tensor e = e_in.contiguous();
    const int ldn = rows_ld(B1h, "edge_gate.B1h"), ldn2 = rows_ld(B2h, "edge_gate.B2h"), ldw = rows_ld(W3, "edge_gate.W3");
    const int64_t E = e.size(0), H = e.size(1);
    TORCH_CHECK(ldn == ldn2, "edge_gate: B1h and B2h are column blocks of one projection (equal row strides)");
    TORCH_CHECK(B1h.size(1) == H && B2h.size(1) == H && W3.size(0) == H && W3.size(1) == H && scale.numel() == H && shift.numel() == H &&
                    srt_src.numel() == E && srt_dst.numel() == E,
                "edge_gate: shapes e[E,H] B1h,B2h[N,H] srt_src,srt_dst[E] W3[H,H] scale,shift[H]");
    Tensor out = at::empty_like(e);
    ok(gnnome_edge_gate_f32(f32_dense(e, "edge_gate.e"), out.data_ptr<float>(), E, (int)H, f32(B1h, "edge_gate.B1h"), f32(B2h, "edge_gate.B2h"), ldn,
                            i32(srt_src, "edge_gate.srt_src"), i32(srt_dst, "edge_gate.srt_dst"), f32(W3, "edge_gate.W3"), ldw, (int)norm_kind,
                            f32_dense(scale, "edge_gate.scale"), f32_dense(shift, "edge_gate.shift"), stream_of(e)),
       "gnnome_edge_gate_f32");
    return out;
}

// ---- both gated aggregations + node update (gnnome_node_aggregate_f32; gated_gcn_full.py:111-114,124-142) ---------------------------------------
Tensor node_aggregate(const Tensor& e, const Tensor& A1h, const Tensor& A2h, const Tensor& A3h, const Tensor& in_ptr, const Tensor& srt_src,
                      const Tensor& out_ptr, const Tensor& out_pos, const Tensor& out_dst, const Tensor& h_in, const Tensor& scale, const Tensor& shift,
                      int64_t norm_kind) {
    const int ldn = rows_ld(A1h, "node_aggregate.A1h"), l2 = rows_ld(A2h, "node_aggregate.A2h"), l3 = rows_ld(A3h, "node_aggregate.A3h");
    const int ldh = rows_ld(h_in, "node_aggregate.h_in");
    TORCH_CHECK(ldn == l2 && ldn == l3, "node_aggregate: A1h, A2h, A3h are column blocks of one projection (equal row strides)");
    const int64_t N = h_in.size(0), H = h_in.size(1), E = srt_src.numel();
    TORCH_CHECK(e.dim() == 2 && e.size(0) == E && e.size(1) == H && A1h.size(0) == N && A1h.size(1) == H && A2h.size(1) == H && A3h.size(1) == H &&
                    in_ptr.numel() == N + 1 && out_ptr.numel() == N + 1 && out_pos.numel() == E && out_dst.numel() == E && scale.numel() == H &&
                    shift.numel() == H,
                "node_aggregate: shapes e[E,H] A*h[N,H] in_ptr,out_ptr[N+1] srt_src,out_pos,out_dst[E] h_in[N,H] scale,shift[H]");
    Guard guard(h_in.device());
    Tensor out = at::empty({N, H}, h_in.options());
    ok(gnnome_node_aggregate_f32(f32_dense(e, "node_aggregate.e"), (int)H, N, f32(A1h, "A1h"), f32(A2h, "A2h"), f32(A3h, "A3h"), ldn,
                                 i32(in_ptr, "node_aggregate.in_ptr"), i32(srt_src, "node_aggregate.srt_src"), i32(out_ptr, "node_aggregate.out_ptr"),
                                 i32(out_pos, "node_aggregate.out_pos"), i32(out_dst, "node_aggregate.out_dst"), f32(h_in, "h_in"), ldh,
                                 out.data_ptr<float>(), (int)norm_kind, f32_dense(scale, "node_aggregate.scale"), f32_dense(shift, "node_aggregate.shift"),
                                 stream_of(h_in)),
       "gnnome_node_aggregate_f32");
    return out;
}

// ---- the edge scorer (gnnome_edge_score_f32; score_predictor.py:12-24), logits in edge-id order -------------------------------------------------
Tensor edge_score(const Tensor& e, const Tensor& Ps, const Tensor& Qd, const Tensor& srt_src, const Tensor& srt_dst, const Tensor& srt_eid,
                  const Tensor& W1e, const Tensor& W2, const Tensor& b2, const Tensor& W3, const Tensor& b3) {
    TORCH_CHECK(e.dim() == 2 && W2.dim() == 2, "edge_score: e[E,H] W2[32,hs]");
    const int ldn = rows_ld(Ps, "edge_score.Ps"), ldn2 = rows_ld(Qd, "edge_score.Qd"), ldw1 = rows_ld(W1e, "edge_score.W1e");
    TORCH_CHECK(ldn == ldn2, "edge_score: Ps and Qd are column blocks of one projection (equal row strides)");
    const int64_t E = e.size(0), H = e.size(1), hs = W2.size(1);
    TORCH_CHECK(Ps.size(1) == hs && Qd.size(1) == hs && W1e.size(0) == hs && W1e.size(1) == H && srt_src.numel() == E && srt_dst.numel() == E &&
                    srt_eid.numel() == E && b2.numel() == W2.size(0) && W3.numel() == W2.size(0) && b3.numel() == 1,
                "edge_score: shapes e[E,H] Ps,Qd[N,hs] srt_*[E] W1e[hs,H] W2[32,hs] b2[32] W3[32] b3[1]");
    Guard guard(e.device());
    Tensor logits = at::empty({E}, e.options());
    ok(gnnome_edge_score_f32(f32_dense(e, "edge_score.e"), E, (int)H, (int)hs, f32(Ps, "Ps"), f32(Qd, "Qd"), ldn, i32(srt_src, "edge_score.srt_src"),
                             i32(srt_dst, "edge_score.srt_dst"), i32(srt_eid, "edge_score.srt_eid"), f32(W1e, "W1e"), ldw1, f32_dense(W2, "edge_score.W2"),
                             f32_dense(b2, "edge_score.b2"), f32_dense(W3, "edge_score.W3"), f32_dense(b3, "edge_score.b3"), logits.data_ptr<float>(),
                             nullptr, stream_of(e)),
       "gnnome_edge_score_f32");
    return logits;
}

// ---- shape inference for tracing: no kernel runs ------------------------------------------------------------------------------------------------
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor> build_graph_views_meta(const Tensor& src, const Tensor&, int64_t n) {
    const auto opt = src.options().dtype(at::kInt);
    const int64_t e = src.size(0);
    auto mk = [&](int64_t k) { return at::empty({k}, opt); };
    return {mk(n + 1), mk(e), mk(e), mk(e), mk(n + 1), mk(e), mk(e)};
}
Tensor encode_meta(const Tensor& x, const Tensor&, const Tensor&, const Tensor& W2, const Tensor&, const OptTensor& gather) {
    return at::empty({gather.has_value() && gather->defined() ? gather->size(0) : x.size(0), W2.size(0)}, x.options());
}
Tensor linear_meta(const Tensor& A, const Tensor& W, const OptTensor&) { return at::empty({A.size(0), W.size(0)}, A.options()); }
Tensor edge_gate_meta(const Tensor& e, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, int64_t) {
    return at::empty_like(e);
}
Tensor node_aggregate_meta(const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&,
                           const Tensor&, const Tensor& h_in, const Tensor&, const Tensor&, int64_t) {
    return at::empty_like(h_in);
}
Tensor edge_score_meta(const Tensor& e, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&,
                       const Tensor&, const Tensor&, const Tensor&) {
    return at::empty({e.size(0)}, e.options());
}

}  // namespace

TORCH_LIBRARY(gnnome_hip, m) {
    m.def("build_graph_views(Tensor src, Tensor dst, int num_nodes) -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)");
    m.def("encode(Tensor x, Tensor W1, Tensor b1, Tensor W2, Tensor b2, Tensor? gather=None) -> Tensor");
    m.def("linear(Tensor A, Tensor W, Tensor? bias=None) -> Tensor");
    m.def("linear_ref(Tensor A, Tensor W, Tensor? bias=None) -> Tensor");
    m.def("edge_gate(Tensor e, Tensor B1h, Tensor B2h, Tensor srt_src, Tensor srt_dst, Tensor W3, Tensor scale, Tensor shift, int norm_kind=0) -> Tensor");
    m.def("node_aggregate(Tensor e, Tensor A1h, Tensor A2h, Tensor A3h, Tensor in_ptr, Tensor srt_src, Tensor out_ptr, Tensor out_pos, Tensor out_dst, "
          "Tensor h_in, Tensor scale, Tensor shift, int norm_kind=0) -> Tensor");
    m.def("edge_score(Tensor e, Tensor Ps, Tensor Qd, Tensor srt_src, Tensor srt_dst, Tensor srt_eid, Tensor W1e, Tensor W2, Tensor b2, Tensor W3, "
          "Tensor b3) -> Tensor");
    m.def("abi_version() -> int", []() -> int64_t { return gnnome_abi_version(); });   // the libgnnome_hip.so this extension is bound to
}

TORCH_LIBRARY_IMPL(gnnome_hip, CUDA, m) {   // the CUDA dispatch key is the HIP device on PyTorch-ROCm
    m.impl("build_graph_views", &build_graph_views);
    m.impl("encode", &encode);
    m.impl("linear", &linear);
    m.impl("linear_ref", &linear_ref);
    m.impl("edge_gate", &edge_gate);
    m.impl("node_aggregate", &node_aggregate);
    m.impl("edge_score", &edge_score);
}

TORCH_LIBRARY_IMPL(gnnome_hip, Meta, m) {
    m.impl("build_graph_views", &build_graph_views_meta);
    m.impl("encode", &encode_meta);
    m.impl("linear", &linear_meta);
    m.impl("linear_ref", &linear_meta);
    m.impl("edge_gate", &edge_gate_meta);
    m.impl("node_aggregate", &node_aggregate_meta);
    m.impl("edge_score", &edge_score_meta);
}
