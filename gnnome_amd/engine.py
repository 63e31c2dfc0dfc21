"""Host logic of the SymGatedGCN path: weight preparation, device staging and the kernel sequence.

Data layout in HBM for one forward (N nodes, E edges, H hidden):
    views        int32  in_ptr[N+1] srt_src[E] srt_dst[E] srt_eid[E] out_ptr[N+1] out_pos[E]
    h            fp32   [N,H]      node state, ping-pong per layer
    P            fp32   [N,5H]     A1h|A2h|A3h|B1h|B2h of the current layer (one GEMM)
    e            fp32   [E,H]      edge state in DESTINATION-SORTED order, updated in place
    Ps|Qd        fp32   [N,2*hs]   node halves of predictor.W1
    logits       fp32   [E]        written at the ORIGINAL edge id
The only [E,H]-sized tensor is `e`; it is written once by the edge encoder (already permuted),
read+written once per layer by edge_gate, read twice per layer by node_aggregate, read once by
edge_score.

The kernel sequence is written against an `ops` namespace (default: gnnome_amd.ops, the HIP
library) so that the partition / halo logic in dist.py can be unit-tested on CPU ranks with a
checker backend injected by the tests; the product never selects anything but the HIP backend.
"""
import torch
import torch.nn.functional as F

from . import ops as hip_ops
from ._lib import NORM_AFFINE, NORM_LAYER
from .graph import views_for


class LayerWeights:
    __slots__ = ("Wcat", "bcat", "W3", "b3", "norm", "scale_e", "shift_e", "scale_h", "shift_h", "ref", "gain_e", "planes")


# Which layers run in the reference's ORDER of evaluation (csrc/reference_order.hip) instead of on the bf16x6 matrix-core
# kernels.  Both are fp32-accurate; they produce DIFFERENT fp32 numbers (accumulation order), and an eval-mode BatchNorm
# with a large gain gamma / sqrt(running_var + eps) magnifies that difference.  The shipped checkpoint's layer 0 has a gain
# of 135 and dominates the distance between any two fp32 evaluations of the model (1.2e-4 in edge probability on an
# E. coli-sized graph); every other layer stays below 3.  "auto" sends a layer through the reference-order kernels when
# its bn_e gain exceeds this threshold, "reference" all layers (H in {64,128,256}), "fast" none.
REFERENCE_ORDER_GAIN = 16.0
ARITHMETIC_MODES = ("auto", "reference", "fast")

# Widths the kernels are built for.  The reference takes any hidden_features / hidden_edge_scores (configs/hyperparameters.py:22-24);
# an eval-mode BatchNorm model of another width runs on the next built width with ZERO-PADDED parameters, which is exact: a padded
# channel has zero weights, zero bias and a zero affine norm, so it stays 0.0 through relu + residual, enters every product as a
# zero addend at the END of the k-ascending sums, and the aggregation's padded columns are sums of 0.5 * 0.0.  LayerNorm's statistics run
# over the row: its kernels take the model's own width and leave the padded channels out (GNNOME_NORM_LAYER_OVER, round 5).  Train mode pads
# the same way on a twin model, train._padded_step.
BUILT_HIDDEN = (64, 128, 256)
BUILT_SCORE_HIDDEN = (32, 64, 128)


def padded_width(width, built=BUILT_HIDDEN, what="hidden_features"):
    for b in built:
        if width <= b:
            return b
    raise ValueError(f"{what}={width}: the HIP kernels are built for widths up to {built[-1]}")


def _pad(t, *shape):
    """t in the leading corner of a zero tensor of `shape` (t itself when it already has that shape)."""
    if tuple(t.shape) == tuple(shape):
        return t
    out = t.new_zeros(shape)
    out[tuple(slice(0, n) for n in t.shape)] = t
    return out


class Prepared:
    """Device-resident, kernel-ready copies of a model's parameters (eval semantics)."""

    def __init__(self, model, device):
        def dev(t):
            return t.detach().to(device=device, dtype=torch.float32).contiguous()

        arithmetic = getattr(model, "arithmetic", "auto")
        self.device = device
        self.model_hidden = model.linear2_node.out_features
        self.hidden = H = padded_width(self.model_hidden)    # the width every kernel of the stack runs at (see BUILT_HIDDEN)
        ne = model.linear2_node.in_features
        self.enc_node = tuple(dev(t) for t in (model.linear1_node.weight, model.linear1_node.bias,
                                               _pad(model.linear2_node.weight.detach(), H, ne), _pad(model.linear2_node.bias.detach(), H)))
        self.enc_edge = tuple(dev(t) for t in (model.linear1_edge.weight, model.linear1_edge.bias,
                                               _pad(model.linear2_edge.weight.detach(), H, ne), _pad(model.linear2_edge.bias.detach(), H)))
        self.layers = [prepare_layer(conv, device, arithmetic) for conv in model.gnn.convs]
        self.predictor = prepare_predictor(model.predictor, device)
        # fp16x3's operand range (|x| < 65504), checked ONCE for the weights: a model with a weight outside it (or a non-finite one) runs
        # its forward as bf16x6 from the start; the activations are checked per call (forward_in_range)
        weights = [t for lw in self.layers for t in (lw.Wcat, lw.W3)] + [self.predictor["_W1"], self.predictor["W2"]]
        amax = max((float(t.abs().max()) if t.numel() else 0.0) for t in weights) if weights else 0.0
        self.force_bf16x6 = not (amax < hip_ops_fp16_max())
        self.range_verified = self.range_failed = None
        self.block = None   # ops.ModelBlock: these parameters as gnnome_model_forward_f32 takes them, built on first use


def hip_ops_fp16_max():
    return getattr(hip_ops, "FP16_MAX", 65504.0)


def _inputs_key(views, x, e):
    """Identity of one set of inputs: the caller's tensor OBJECTS and their versions (an in-place change bumps the version, another
    tensor is another object) and the views."""
    import weakref
    return (weakref.ref(x), x._version, weakref.ref(e), e._version, weakref.ref(views))


def _same_inputs(key, views, x, e):
    return key is not None and key[0]() is x and key[1] == x._version and key[2]() is e and key[3] == e._version and key[4]() is views


def forward_in_range(ops, prep, views, x, e, xd, ed, check=True):
    """run_stack with the reference's DOMAIN (gated_gcn_full.py:97 is a plain fp32 nn.Linear): the forward's dense
    products run as fp16x3, whose operands must stay below 65504; an element beyond that leaves those kernels as a NaN row (never as a
    wrong finite value), so a forward whose logits are not all finite is run AGAIN as bf16x6 (fp32's range) and that is what the caller
    gets - without touching gnnome_set_tuning itself.  The check is one device reduction + one host sync after everything has been
    enqueued (the pattern of GraphViews(validate="lazy")), made once per set of inputs: a caller that scores the same tensors again
    (a benchmark loop, CapturedForward's warm-up and recording) is not synchronised again.  Inputs that are themselves non-finite cost
    one extra forward and come back non-finite, as from the reference.  check=False (`model.range_check = False`): no check, NaN rows
    stay NaN rows."""
    bf = getattr(ops, "bf16x6_arithmetic", None)
    if bf is None or getattr(ops, "_TUNING", {}).get(10, 0) == 1:   # the checker backend / bf16x6 already selected
        return run_stack(ops, prep, views, xd, ed)
    if prep.force_bf16x6 or _same_inputs(prep.range_failed, views, x, e):
        with bf():
            return run_stack(ops, prep, views, xd, ed)
    logits = run_stack(ops, prep, views, xd, ed)
    if not check or _same_inputs(prep.range_verified, views, x, e):
        return logits
    if bool(torch.isfinite(logits).all()):
        prep.range_verified = _inputs_key(views, x, e)
        return logits
    prep.range_failed = _inputs_key(views, x, e)
    with bf():
        return run_stack(ops, prep, views, xd, ed)


def _norm_affine(norm_module, device):
    """(kind, scale, shift) such that the kernels' y = norm(x) matches the torch module in eval mode."""
    def dev(t):
        return t.detach().to(device=device, dtype=torch.float32).contiguous()

    if isinstance(norm_module, torch.nn.BatchNorm1d):
        # eval BatchNorm1d exactly as torch's CPU kernel evaluates it (bit-exact against torch 2.10, see
        # tests/test_reference_order.py): alpha = gamma * (1 / sqrt(var + eps)) in fp32, beta = fma(-mean, alpha, bias),
        # y = fma(x, alpha, beta) - the kernels do the last step.  (A fold in fp64 is closer to the exact affine map but
        # is a DIFFERENT fp32 pair, 3.6e-5 in edge probability away from the reference on the shipped weights.)
        var = norm_module.running_var.detach().float().cpu()
        mean = norm_module.running_mean.detach().float().cpu()
        gamma = norm_module.weight.detach().float().cpu()
        beta = norm_module.bias.detach().float().cpu()
        scale = gamma * (1.0 / torch.sqrt(var + norm_module.eps))
        shift = (beta.double() - mean.double() * scale.double()).float()   # one rounding of the exact fma argument
        width = padded_width(scale.numel())
        return NORM_AFFINE, dev(_pad(scale, width)), dev(_pad(shift, width))
    if isinstance(norm_module, torch.nn.LayerNorm):
        if abs(norm_module.eps - 1e-5) > 1e-12:
            raise ValueError("LayerNorm eps other than 1e-5 is not supported by the HIP kernels")
        # a width between the built ones: zero-padded gamma / beta, and the row statistics over the model's OWN channels
        # (GNNOME_NORM_LAYER_OVER(w) in include/gnnome_hip.h - the padded channels hold exact zeros; round 5)
        own = int(norm_module.weight.numel())
        width = padded_width(own)
        kind = NORM_LAYER if width == own else (NORM_LAYER | (own << 8))
        return kind, dev(_pad(norm_module.weight.detach(), width)), dev(_pad(norm_module.bias.detach(), width))
    raise TypeError(type(norm_module))


def prepare_layer(conv, device, arithmetic=None):
    def dev(t):
        return t.detach().to(device=device, dtype=torch.float32).contiguous()

    arithmetic = getattr(conv, "arithmetic", "auto") if arithmetic is None else arithmetic
    if arithmetic not in ARITHMETIC_MODES:
        raise ValueError(f"arithmetic must be one of {ARITHMETIC_MODES}, got {arithmetic!r}")
    lw = LayerWeights()
    width = conv.B_3.weight.shape[0]
    hidden = padded_width(width)
    W = lambda lin: _pad(lin.weight.detach(), hidden, hidden)  # noqa: E731
    b = lambda lin: _pad(lin.bias.detach(), hidden)  # noqa: E731
    lw.Wcat = dev(torch.cat([W(conv.A_1), W(conv.A_2), W(conv.A_3), W(conv.B_1), W(conv.B_2)], 0))
    lw.W3 = dev(W(conv.B_3))
    lw.b3 = dev(b(conv.B_3))
    lw.norm, lw.scale_e, lw.shift_e = _norm_affine(conv.bn_e, device)
    kind_h, lw.scale_h, lw.shift_h = _norm_affine(conv.bn_h, device)
    assert kind_h == lw.norm
    lw.gain_e = float(lw.scale_e.abs().max()) if lw.norm == NORM_AFFINE and lw.scale_e.numel() else 0.0
    can = hip_ops.reference_order_supported(hidden, lw.norm)
    lw.ref = can and (arithmetic == "reference" or (arithmetic == "auto" and lw.gain_e > REFERENCE_ORDER_GAIN))
    # the projection's weights as fp16x3 planes, made once (ops.weight_planes -> gnnome_linear_planes_f32; round 6)
    planes_ok = getattr(hip_ops, "planes_supported", None)
    lw.planes = hip_ops.weight_planes(lw.Wcat) if (not lw.ref and planes_ok is not None and planes_ok(hidden, 5 * hidden)
                                                   and lw.Wcat.is_cuda) else None
    if lw.ref:
        lw.bcat = dev(torch.cat([b(conv.A_1), b(conv.A_2), b(conv.A_3), b(conv.B_1), b(conv.B_2)], 0))
    else:
        # B_3's bias rides on the B2h rows: B1h[src] + (B2h[dst] + b3) + e*W3^T
        lw.bcat = dev(torch.cat([b(conv.A_1), b(conv.A_2), b(conv.A_3), b(conv.B_1), b(conv.B_2) + b(conv.B_3)], 0))
    return lw


def prepare_predictor(pred, device):
    def dev(t):
        return t.detach().to(device=device, dtype=torch.float32).contiguous()

    hs_model, h3 = pred.W1.weight.shape
    width = h3 // 3
    if pred.W2.out_features != 32 or pred.W3.in_features != 32 or pred.W3.out_features != 1:
        raise ValueError("ScorePredictor tail must be hs -> 32 -> 1 (score_predictor.py:9-10)")
    hidden, hs = padded_width(width), padded_width(hs_model, BUILT_SCORE_HIDDEN, "hidden_edge_scores")
    # W1 acts on [x[src] | x[dst] | e]: each of its three [hs, H] blocks padded on its own (zero rows for the padded score
    # channels - relu(0) = 0 meets a zero column of W2 -, zero columns for the padded hidden channels)
    W1 = dev(torch.cat([_pad(pred.W1.weight.detach()[:, i * width:(i + 1) * width], hs, hidden) for i in range(3)], 1))
    b1 = _pad(pred.W1.bias.detach(), hs)
    # node halves stacked into one [2*hs, H] projection: rows 0..hs-1 act on x[src], rows hs.. on x[dst] (+ b1)
    W_nodes = torch.cat([W1[:, :hidden], W1[:, hidden:2 * hidden]], 0).contiguous()
    b_nodes = torch.cat([torch.zeros_like(b1), b1])
    # the node halves' projection on W_nodes' fp16x3 planes (made once) from 128 output columns on: at H = 128, hs = 64 26.5 against 35 us on the
    # bf16x6 kernel ops.linear picks for that shape by itself, and 9.6e-7 against 4.2e-6 from an fp64 product; at 2 hs = 64 the bf16x6 kernel is
    # level (20 against 21.5 us) and stays (tools/score_nodes_time.py)
    planes_ok = getattr(hip_ops, "planes_supported", None)
    planes = (hip_ops.weight_planes(W_nodes) if planes_ok is not None and W_nodes.is_cuda and planes_ok(hidden, 2 * hs) and 2 * hs >= 128 else None)
    return {
        "planes": planes,
        "hidden": hidden, "hs": hs, "model_hidden": width, "W_nodes": W_nodes, "b_nodes": dev(b_nodes), "W1_e": W1[:, 2 * hidden:],
        "W2": dev(_pad(pred.W2.weight.detach(), 32, hs)), "b2": dev(pred.W2.bias),
        "W3": dev(pred.W3.weight.reshape(-1)), "b3": dev(pred.W3.bias.reshape(-1)), "_W1": W1,
    }


class _StateProbe:
    """Has anything `Prepared` was built from changed?  The question is asked on every forward, and walking module.parameters() + buffers()
    (190 tensors behind generators) cost 0.26 ms of the 0.63 ms a forward spent on the host.  Kept instead: the flat list of (owning dict,
    name, tensor, version, storage address) - an in-place update bumps the version (optimizer step, load_state_dict), `.to()` / `.data = `
    moves the storage, a replaced Parameter is another object under the same name - and the sub-module counts (a layer added or removed)."""

    __slots__ = ("device", "arithmetic", "slots", "modules")

    def __init__(self, module, device):
        self.device, self.arithmetic = str(device), getattr(module, "arithmetic", "auto")
        self.slots, self.modules = [], []
        for sub in module.modules():
            self.modules.append((sub._modules, len(sub._modules)))
            for table in (sub._parameters, sub._buffers):
                for name, t in table.items():
                    if t is not None:
                        self.slots.append((table, name, t, t._version, t.data_ptr()))

    def unchanged(self, module, device):
        if str(device) != self.device or getattr(module, "arithmetic", "auto") != self.arithmetic:
            return False
        for table, count in self.modules:
            if len(table) != count:
                return False
        for table, name, t, version, ptr in self.slots:
            if table.get(name) is not t or t._version != version or t.data_ptr() != ptr:
                return False
        return True


def prepared_for(module, device, build):
    cached = module.__dict__.get("_gnnome_prepared")
    if cached is None or not cached[0].unchanged(module, device):
        probe = _StateProbe(module, device)   # (before the build: a change during it is seen next time)
        cached = (probe, build(module, device))
        module.__dict__["_gnnome_prepared"] = cached
    return cached[1]


def compute_device(*tensors):
    """The MI355X the call runs on: the inputs' device if they are already on one, else the current device."""
    for t in tensors:
        if t.is_cuda:
            return t.device
    if not torch.cuda.is_available():
        raise RuntimeError("gnnome_amd runs on an MI355X only: no HIP device is visible and there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _refuse_training(module):
    """The layer-level / predictor-level entry points are inference-only; training goes through the model
    (gnnome_amd/train.py differentiates the whole path as one autograd.Function)."""
    if module.training and (torch.is_grad_enabled() or any(isinstance(m, torch.nn.BatchNorm1d) for m in module.modules())):
        raise NotImplementedError("train mode is supported through SymGatedGCNModel.forward only; call .eval() for the "
                                  "layer-level API")
    if torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters()):
        # the layer-level entries return tensors WITHOUT autograd history: refuse instead of handing back None gradients
        # (a frozen-BatchNorm fine-tune through the layer API in eval mode would otherwise train nothing, silently)
        raise NotImplementedError("gnnome_amd layer-level forward is inference-only: its outputs carry no autograd history - wrap the "
                                  "call in torch.no_grad() (or freeze the parameters), or train through SymGatedGCNModel.forward")


# ---------------------------------------------------------------------------------------------------
# kernel sequences (ops = gnnome_amd.ops in the product)
# ---------------------------------------------------------------------------------------------------

def gate_update(ops, lw, views, e, B1, B2, scratch=None):
    """e <- gate(e): in place, except at H = 256 with the affine norm, where the streaming kernel needs separate input
    and output buffers (its column chunks run in different workgroups): two [E,H] buffers then take turns, the spare
    one lives in `scratch` for the duration of the forward."""
    if scratch is not None and e.shape[1] == 256 and lw.norm == 0 and getattr(ops, "edge_gate_out_of_place_at_256", False):
        spare = scratch.get("e_spare")
        if spare is None or spare.shape != e.shape or spare.device != e.device:
            spare = torch.empty_like(e)
        ops.edge_gate(e, B1, B2, views, lw.W3, lw.norm, lw.scale_e, lw.shift_e, out=spare)
        scratch["e_spare"] = e
        return spare
    ops.edge_gate(e, B1, B2, views, lw.W3, lw.norm, lw.scale_e, lw.shift_e)
    return e


def project(ops, lw, h, out=None):
    """P[rows,5H] = h Wcat^T + bcat: A1h|A2h|A3h|B1h|B2h (gated_gcn_full.py:91-96)."""
    if lw.ref:
        return ops.linear_ref(h, lw.Wcat, lw.bcat, out=out)
    return ops.linear(h, lw.Wcat, lw.bcat, out=out, planes=lw.planes)


def gate(ops, lw, views, e, B1, B2, raw_edges=None, scratch=None):
    """e' = relu(bn_e(B1h[src] + B2h[dst] + B_3(e))) + e on sorted-order rows (gated_gcn_full.py:97,104-110).  e = None:
    layer 0, the edge encoder's output is produced inside the gate kernel from raw_edges = (e_raw, encoder weights)."""
    if lw.ref:
        return ops.edge_gate_ref(e, B1, B2, views, lw.W3, lw.b3, lw.scale_e, lw.shift_e, raw_edges=raw_edges)
    if e is None:
        return ops.edge_gate_encode(raw_edges[0], raw_edges[1], B1, B2, views, lw.W3, lw.scale_e, lw.shift_e)
    return gate_update(ops, lw, views, e, B1, B2, scratch)


def aggregate_then_project(ops, lw, views, e, A1, A2, A3, h, then):
    """h' = the layer's node update, and `then(h'[rows], out=...)` - the NEXT consumer's node projection (the next layer's
    A1..B2, gated_gcn_full.py:91-96, or the scorer's node halves, score_predictor.py:13-14) - pipelined over node ranges:
    the aggregation runs as PIPELINE_CHUNKS consecutive launches on the current stream (gnnome_node_aggregate_range_f32), and
    as soon as a range of h' is complete its projection starts on a second, high-priority HIP stream, under the aggregation
    of the next range.  Values are those of the one-stream sequence bit for bit (both kernels compute every row independently
    of the launch's row count).

    MEASURED NEGATIVE, hence off by default (PIPELINE_CHUNKS = 1; tools/pipeline_ab.py reproduces it): at configs[1] the
    forward takes 4.89 ms on one stream and 5.20 / 5.25 / 5.54 ms with 2 / 4 / 8 ranges (10M edges: 47.4 against 49.0 ms;
    replayed from a hipGraph 6.2-9 ms - cross-stream edges are expensive graph nodes on ROCm).  The aggregation waits on
    memory for most of its cycles and the projection is bound by neither roof when it runs alone, but together they take
    LONGER than one after the other: both live off the same L2 / fabric (a likely mechanism, not isolated: the projection's
    256 MB of output displace the e' rows the aggregation's out-edge pass otherwise finds in L2; nontemporal stores in the
    projection change nothing)."""
    n = h.shape[0]
    side = ops.side_stream(h.device)
    main = torch.cuda.current_stream(h.device)
    h_out = torch.empty_like(h)
    P_next = torch.empty((n, then.width), dtype=torch.float32, device=h.device)   # both live on the main stream, which joins the side stream below
    bounds = [n * i // PIPELINE_CHUNKS // 32 * 32 for i in range(PIPELINE_CHUNKS)] + [n]
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        ops.node_aggregate(e, A1, A2, A3, views, h, lw.norm, lw.scale_h, lw.shift_h, node_range=(lo, hi), out=h_out)
        done = torch.cuda.Event()
        done.record(main)
        with torch.cuda.stream(side):
            side.wait_event(done)
            then(h_out[lo:hi], out=P_next[lo:hi])
    main.wait_stream(side)
    return h_out, P_next


import os as _os
ONE_CALL_FORWARD = _os.environ.get("GNNOME_ONE_CALL_FORWARD", "1") != "0"   # run_stack's default path through gnnome_model_forward_f32 (0: call by call)
PIPELINE_CHUNKS = 1   # off: measured slower, see aggregate_then_project
PIPELINE_MIN_NODES = 1 << 15   # below this a launch is a few microseconds and the extra launches + events cost more than they hide


class _Projection:
    """`then` of aggregate_then_project: rows -> rows @ W^T + b through the layer's own projection kernel."""

    def __init__(self, fn, W, b):
        self.fn, self.W, self.b, self.width = fn, W, b, W.shape[0]

    def __call__(self, rows, out=None):
        return self.fn(rows, self.W, self.b, out=out)


def layer_projection(ops, lw):
    if lw.ref:
        return _Projection(ops.linear_ref, lw.Wcat, lw.bcat)
    return _Projection(lambda rows, W, b, out=None: ops.linear(rows, W, b, out=out, planes=lw.planes), lw.Wcat, lw.bcat)


def layer_step(ops, lw, views, h, e, n_out=None, raw_edges=None, scratch=None, P=None, then=None):
    """One SymGatedGCN layer on sorted-order e (updated in place, see gate_update); returns (new h, e, then's output).  With
    e = None and raw_edges = (e_raw, encoder weights) the edge encoder is folded into the gate (layer 0).  P: this layer's
    projection if the previous step already produced it; then: the next consumer's projection, to be run pipelined with
    the aggregation (aggregate_then_project) - None: not computed here."""
    H = h.shape[1]
    if P is None:
        P = project(ops, lw, h)
    A1, A2, A3, B1, B2 = (P[:, i * H:(i + 1) * H] for i in range(5))
    if views.transposed:  # dgl.reverse(g): src <-> dst, see GraphViews.reversed
        A2, A3, B1, B2 = A3, A2, B2, B1
    e = gate(ops, lw, views, e, B1, B2, raw_edges, scratch)
    if then is not None:
        h_new, P_next = aggregate_then_project(ops, lw, views, e, A1, A2, A3, h, then)
        return h_new, e, P_next
    return ops.node_aggregate(e, A1, A2, A3, views, h, lw.norm, lw.scale_h, lw.shift_h, num_nodes_out=n_out), e, None


def encode_nodes(ops, views, x, enc):
    """h0 in the views' node numbering: x arrives in the caller's, and views over renumbered nodes (GraphViews.node_perm,
    gnnome_amd/node_order.py) read it through the encoder's gather argument - no permuted copy of x is made."""
    gather = getattr(views, "node_gather", None)
    if gather is None:
        return ops.encode(x, *enc)
    return ops.encode(x, *enc, gather=gather, rows=views.num_nodes)


def encode_edges(ops, prep, views, e_raw):
    """e0 in sorted order - or None when layer 0's gate kernel will produce it on the fly."""
    fuse = getattr(ops, "can_fuse_edge_encoder", None)
    if prep.layers and fuse is not None and fuse(e_raw, prep.enc_edge, prep.hidden, prep.layers[0].norm, prep.layers[0].Wcat):
        return None
    return ops.encode(e_raw, *prep.enc_edge, gather=views.srt_eid, rows=views.num_edges)


def predictor_projection(ops, pw):
    return _Projection(lambda rows, W, b, out=None: ops.linear(rows, W, b, out=out, planes=pw.get("planes")), pw["W_nodes"], pw["b_nodes"])


def score_step(ops, pw, views, h, e, logits, n_edges=None, PQ=None):
    hs = pw["hs"]
    if PQ is None:
        PQ = ops.linear(h, pw["W_nodes"], pw["b_nodes"], planes=pw.get("planes"))
    Ps, Qd = PQ[:, :hs], PQ[:, hs:]
    if views.transposed:
        # x[src'] | x[dst'] = x[dst] | x[src]; b1 is added once either way
        Ps, Qd = Qd, Ps
    return ops.edge_score(e, Ps, Qd, views, pw["W1_e"], pw["W2"], pw["b2"], pw["W3"], pw["b3"], logits, num_edges=n_edges)


def run_stack(ops, prep, views, x, e_raw, exchange=None, n_own=None, n_score=None, logits=None):
    """Encoders -> L layers -> scorer.  `exchange(h)` (optional) refreshes halo rows before every
    consumer of h; `n_own` limits the node update to the first rows, `n_score` the scorer to the first
    sorted positions (both used by the destination-range partition, dist.py).  With PIPELINE_CHUNKS > 1 (off by
    default: measured slower) every node projection after the first runs under the preceding aggregation on a second
    stream (aggregate_then_project)."""
    pipelined = (exchange is None and n_own is None and getattr(ops, "side_stream", None) is not None and PIPELINE_CHUNKS > 1
                 and views.num_nodes >= PIPELINE_MIN_NODES)
    one_call = getattr(ops, "model_forward", None)
    if (ONE_CALL_FORWARD and one_call is not None and exchange is None and n_own is None and n_score is None and not pipelined
            and not getattr(ops, "STREAM_AGGREGATE", False) and isinstance(views, ops.GraphViews) and x.is_cuda and x.is_contiguous()
            and e_raw.is_contiguous()):
        # the whole sequence below as ONE call into the library (gnnome_model_forward_f32: the same entries in the same order, same bits)
        block = getattr(prep, "block", None)
        if block is None:
            block = prep.block = ops.ModelBlock(prep)
        with ops.node_records_for(views, prep.hidden):   # (the aggregation's record form at the widths it pays for: ops.NODE_RECORDS_MAX_HIDDEN)
            return one_call(block, views, x, e_raw, logits)
    h = encode_nodes(ops, views, x, prep.enc_node)
    e = encode_edges(ops, prep, views, e_raw)
    scratch = {}
    P = None
    for i, lw in enumerate(prep.layers):
        if exchange is not None:
            h = exchange(h)
        then = None
        if pipelined:
            then = layer_projection(ops, prep.layers[i + 1]) if i + 1 < len(prep.layers) else predictor_projection(ops, prep.predictor)
        h, e, P = layer_step(ops, lw, views, h, e, n_out=n_own, raw_edges=(e_raw, prep.enc_edge), scratch=scratch, P=P, then=then)
    if exchange is not None:
        h = exchange(h)
    if logits is None:
        logits = torch.empty(views.num_edges if n_score is None else n_score, dtype=torch.float32, device=h.device)
    score_step(ops, prep.predictor, views, h, e, logits, n_edges=n_score, PQ=P)
    return logits


# ---------------------------------------------------------------------------------------------------
# module entry points
# ---------------------------------------------------------------------------------------------------

def model_forward(model, graph, x, e):
    """models/full_graph.py:22-30 on the MI355X."""
    if model.training:
        from .train import train_forward
        return train_forward(model, graph, x, e).to(x.device)
    out_device = x.device
    device = compute_device(x, e)
    prep = prepared_for(model, device, Prepared)
    views = views_for(graph, device, node_order=getattr(model, "node_order", "input"))
    if x.shape[0] != views.num_nodes or e.shape[0] != views.num_edges:
        raise ValueError(f"x has {x.shape[0]} rows for {views.num_nodes} nodes, e has {e.shape[0]} rows for {views.num_edges} edges")
    with torch.no_grad():
        xd = x.detach().to(device=device, dtype=torch.float32).contiguous()
        ed = e.detach().to(device=device, dtype=torch.float32).contiguous()
        logits = forward_in_range(hip_ops, prep, views, x, e, xd, ed, check=getattr(model, "range_check", True))
    views.check_range()   # a fresh graph's deferred endpoint check, after the whole forward has been enqueued
    return logits.unsqueeze(1).to(out_device)


def layer_forward_edge_id_order(conv, g, h, e):
    """gated_gcn_full.py:82-142 with e given and returned in edge-id order."""
    _refuse_training(conv)
    out_device = h.device
    device = compute_device(h, e)
    lw = prepared_for(conv, device, prepare_layer)
    views = views_for(g, device)
    with torch.no_grad():
        hd = h.detach().to(device=device, dtype=torch.float32).contiguous()
        ed = e.detach().to(device=device, dtype=torch.float32).contiguous()
        width, hidden = hd.shape[1], lw.W3.shape[0]
        if hidden != width:    # a width between the built ones: zero columns in, sliced off again below (BUILT_HIDDEN)
            hd, ed = F.pad(hd, (0, hidden - width)), F.pad(ed, (0, hidden - width))
        es = hip_ops.gather_rows(ed, views.srt_eid)
        if views.node_gather is not None:   # views over renumbered nodes: rows in, rows out in the caller's numbering
            hd = hip_ops.gather_rows(hd, views.node_gather)
        h_new, _, _ = layer_step(hip_ops, lw, views, hd, es)
        if views.node_perm is not None:
            h_new = h_new.index_select(0, views.node_perm)
        e_new = torch.empty_like(es)
        e_new[views.srt_eid.long()] = es
        if hidden != width:
            h_new, e_new = h_new[:, :width].contiguous(), e_new[:, :width].contiguous()
        h_new = F.dropout(h_new, conv.dropout, training=conv.training)
    views.check_range()   # deferred endpoint check of a fresh graph (GraphViews validate="lazy")
    return h_new.to(out_device), e_new.to(out_device)


def score_forward_edge_id_order(pred, graph, x, e):
    _refuse_training(pred)
    out_device = x.device
    device = compute_device(x, e)
    pw = prepared_for(pred, device, prepare_predictor)
    views = views_for(graph, device)
    with torch.no_grad():
        xd = x.detach().to(device=device, dtype=torch.float32).contiguous()
        ed = e.detach().to(device=device, dtype=torch.float32).contiguous()
        if pw["hidden"] != pw["model_hidden"]:
            xd, ed = (F.pad(t, (0, pw["hidden"] - pw["model_hidden"])) for t in (xd, ed))
        es = hip_ops.gather_rows(ed, views.srt_eid)
        if views.node_gather is not None:
            xd = hip_ops.gather_rows(xd, views.node_gather)
        logits = torch.empty(views.num_edges, dtype=torch.float32, device=device)
        score_step(hip_ops, pw, views, xd, es, logits)
    views.check_range()
    return logits.unsqueeze(1).to(out_device)
