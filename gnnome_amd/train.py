"""Training step of the SymGatedGCN path: train-mode forward + hand-written backward on the HIP kernels.

What the reference does: `train.py:138-145` runs `model(g, x, e)` in train mode, `train.py:328-330` calls
`loss.backward()` and lets torch autograd differentiate through `nn.Linear`, `BatchNorm1d` (batch
statistics), relu, sigmoid and DGL's gspmm / gsddmm.  Here the same function is differentiated by hand and
wired into autograd as ONE `torch.autograd.Function` whose inputs are the model's 142 parameters, so
`loss.backward()`, `optimizer.step()` and `state_dict()` in the caller keep working unchanged.

Forward, per layer (gated_gcn_full.py:82-142, single-gate form - see DESIGN.md):
    P   = h Wcat^T + bcat                                  gnnome_linear_f32
    xe  = B1h[src] + B2h[dst] + e W3^T                     gnnome_edge_gate_raw_f32
    e'  = relu(bn_e(xe)) + e          (batch statistics)   gnnome_colsum2_f32, gnnome_bn_relu_res_f32
    v   = A1h + fwd + bwd                                  gnnome_node_aggregate_raw_f32 (mode 1)
    h'  = dropout(relu(bn_h(v)) + h)                       gnnome_colsum2_f32, gnnome_bn_relu_res_f32
`bn_e` is applied twice per layer in the reference (:106 and :119, identical inputs): its running statistics
receive two momentum updates and `num_batches_tracked` advances by 2; the gradient reaches it through both
aggregation directions, which the single-gate form accounts for by summing d(sigma_f) + d(sigma_b).

Backward = the transposes, in reverse: scorer tail, node projections (dgrad / wgrad GEMMs), BatchNorm backward
(two-pass, per-channel sums over all rows), the per-edge gradient of both gated aggregations, segment sums for
the gathers.  Everything runs in libgnnome_hip.so; torch holds the memory, does [H]-sized vector arithmetic on
BatchNorm statistics and concatenates/slices views.

DOMAIN.  The forward's dense products and, since rounds 5-6, the backward's weight / data gradients at H >= 128 run as fp16x3: the
gradient operand is scaled by the power of two its maximum asks for (any magnitude from 1e-30 to 1e30 is fine), the OTHER operand - the
activations e, h and the weights - must lie inside fp16's range, |x| < 65504.  Beyond it the affected gradients are NaN (never wrong finite
values): `loss.backward()` then leaves non-finite `.grad`s, which `torch.nn.utils.clip_grad_norm_(..., error_if_nonfinite=True)` or a look at the
loss catches.  Inference re-runs such a forward as bf16x6 by itself (engine.forward_in_range); a training step does not - select fp32's range
for it with `GNNOME_SCALED_WGRAD=0 GNNOME_SCALED_NODE_WGRAD=0 GNNOME_SCALED_NODE_DGRAD=0` + `ops.bf16x6_arithmetic()` (gnnome_set_tuning(10, 1)).
"""
import torch

from . import ops as hip_ops
from .graph import views_for


class WholeGraph:
    """Where the rows of one training step live.  This is the single-process case: every node and edge of the graph
    is local and owned, nothing is exchanged.  gnnome_amd.dist.PartitionShard is the destination-range partition
    (owned nodes first, then halo; owned in-edges first, then the out-edges that end in the halo)."""

    world = 1
    alone = True      # no collective is issued (a PartitionShard of one rank under GNNOME_FORCE_COLLECTIVES=1 says False)

    def __init__(self, views, ops=hip_ops):
        self.views, self.ops = views, ops
        self.n_own = self.n_local = self.n_global = views.num_nodes
        self.e_own = self.e_local = self.e_global = views.num_edges
        self.score_views = views

    def halo_start(self, h):            # h[n_own:] <- the owners' rows (forward)
        pass

    def halo_finish(self):
        pass

    def halo_bwd(self, dh):             # owners' dh rows += dh[n_own:], then dh[n_own:] = 0 (backward)
        pass

    def halo_bwd_start(self, dh):       # the same in two halves: the rows leave ... (nothing may write dh in between)
        pass

    def halo_bwd_finish(self, dh):      # ... and are added where they belong
        pass

    def combine_stats(self, mean, var, rows):   # per-rank (mean, biased var, row count) -> statistics of the union
        return mean, var

    def sum_ranks(self, tensors):       # element-wise sum over ranks of a list of tensors (same shapes on every rank)
        return tensors

    def finish_logits(self, logits):    # assemble the per-rank disjoint pieces of logits[E_global]
        return logits


def dropout_mask(rows, cols, p, device):
    """Scaled keep-mask of F.dropout(h, p, training=True) (gated_gcn_full.py:139): Bernoulli(1 - p) / (1 - p), drawn from
    torch's generator of `device`.  A module-level function so that the tests can substitute known masks."""
    return torch.empty((rows, cols), dtype=torch.float32, device=device).bernoulli_(1.0 - p).div_(1.0 - p)


def _cat_layer(conv, ops=None):
    """(Wcat, bcat, WcatT, W3T): the stacked projection weight / bias of a layer and the two transposes its backward reads.  One
    launch where the backend packs them (gnnome_pack_layer_f32), otherwise torch operators (the transposes then come lazily: None)."""
    lins = (conv.A_1, conv.A_2, conv.A_3, conv.B_1, conv.B_2)
    H = conv.B_3.weight.shape[0]
    if ops is not None and hasattr(ops, "pack_layer") and H % 32 == 0 and conv.B_3.weight.is_cuda:
        return ops.pack_layer([m.weight for m in lins], [m.bias for m in lins], conv.B_3.weight, conv.B_3.bias)
    Wcat = torch.cat([m.weight for m in lins], 0).detach().contiguous()
    bcat = torch.cat([conv.A_1.bias, conv.A_2.bias, conv.A_3.bias, conv.B_1.bias, conv.B_2.bias + conv.B_3.bias], 0).detach().contiguous()
    return Wcat, bcat, None, None


def _roles(transposed):
    """column block of P used in each kernel role (dgl.reverse swaps src <-> dst: A2<->A3, B1<->B2)."""
    return dict(A1=0, A2=2, A3=1, B1=4, B2=3) if transposed else dict(A1=0, A2=1, A3=2, B1=3, B2=4)


def _bn_train(sh, bn, mean, var, n_local, rows, updates):
    """(mean, biased var) of this rank's n_local rows -> (mean, rstd, scale, shift) over the `rows` rows of the whole
    graph; running buffers advanced `updates` times like nn.BatchNorm1d."""
    mean, var = sh.combine_stats(mean, var, n_local)
    rstd = torch.rsqrt(var + bn.eps)
    scale = bn.weight.detach() * rstd
    shift = bn.bias.detach() - mean * scale
    if bn.track_running_stats:
        with torch.no_grad():
            unbiased = var * (rows / max(rows - 1, 1))
            for _ in range(updates):
                m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked + 1)
                bn.running_mean.mul_(1 - m).add_(mean, alpha=m)
                bn.running_var.mul_(1 - m).add_(unbiased, alpha=m)
                bn.num_batches_tracked += 1
    return mean.contiguous(), rstd.contiguous(), scale.contiguous(), shift.contiguous()


def _bn_train_fused(sh, bn, moments, updates):
    """The single-rank case in one launch (gnnome_bn_train_finish_f32): no per-channel torch arithmetic, no host work."""
    track = bn.track_running_stats
    return sh.ops.bn_train_finish(moments, bn.weight.detach(), bn.bias.detach(), bn.running_mean if track else None,
                                  bn.running_var if track else None, bn.num_batches_tracked if track else None, bn.momentum, bn.eps, updates)


def _can_fuse_bn(sh, bn):
    return sh.alone and bn.momentum is not None and hasattr(sh.ops, "bn_train_finish")


def _bn_bwd(sh, dy, x, scale, shift, mean, rstd, rows, n_once, out, stats=None, apply=True):
    """out <- d(input of bn); returns this rank's share of (d gamma, d beta), for out = relu(x*scale + shift) + res with
    scale = gamma*rstd.  `rows` = rows of the whole graph.  dy may hold PARTIAL gradients of rows that are replicated
    on another rank (cut edges): the per-channel sums run over all local rows - summed over ranks they are the true
    totals - while the mean-subtraction terms, which must enter once per row of the whole graph, are applied to the
    first n_once rows only (the owned ones)."""
    ops = sh.ops
    s1, s2 = ops.bn_bwd_stats(dy, x, scale, shift, mean) if stats is None else stats   # (stats: gathered by the producer of dy)
    if sh.alone and hasattr(ops, "bn_bwd_terms"):
        s2h, c1, c2 = ops.bn_bwd_terms(s1, s2, rstd, rows)   # the three vectors below in one launch
    else:
        s2h = rstd * s2                                     # sum dy*m*xhat
        t1, t2 = sh.sum_ranks([s1, s2h])
        c1, c2 = (t1 / rows).contiguous(), (t2 / rows).contiguous()
    if not apply:   # the caller fuses the apply pass into its consumer (ops.bn_bwd_dgrad): hand back the two mean terms
        return s2h, s1, c1, c2
    ops.bn_bwd_apply(dy[:n_once], x[:n_once], scale, shift, scale, c1, c2, mean, rstd, out=out[:n_once])
    if n_once < x.shape[0]:
        zero = torch.zeros_like(mean)
        ops.bn_bwd_apply(dy[n_once:], x[n_once:], scale, shift, scale, zero, zero, mean, rstd, out=out[n_once:])
    return s2h, s1


ACTIVATION_STORAGE = ("fp32", "bf16")
def _switch(name):   # GNNOME_<NAME>=0 in the environment turns a switch off for a whole process (bench.py A/B runs of one build)
    import os
    return os.environ.get("GNNOME_" + name, "1") != "0"


SCALED_NODE_DGRAD = _switch("SCALED_NODE_DGRAD")   # ... and dh += dP Wcat as one fp16x3 launch on the same scaled blocks (five bf16x6 residual GEMMs otherwise)
SCALED_NODE_WGRAD = _switch("SCALED_NODE_WGRAD")   # ... and the projection's [5H, H] weight gradient, on the five node gradients scaled by their common maximum
SCALED_WGRAD = _switch("SCALED_WGRAD")   # B_3's weight gradient as fp16x3 on dxe scaled by its maximum (gnnome_wgrad_scaled_f32) instead of bf16x6
FUSED_NODE_TABLES = _switch("FUSED_NODE_TABLES")   # bn_h's backward apply + the four node tables of the aggregation's backward as one launch (gnnome_bn_bwd_apply_tables_f32)
FUSED_AGG_BWD = _switch("FUSED_AGG_BWD")   # the aggregation's node sums and per-edge backward as one launch (gnnome_agg_bwd_fused_f32; tools/train_ab.py switches it)
TWO_PASS_GATE_WIDE = __import__("os").environ.get("GNNOME_TWO_PASS_GATE_WIDE", "0") == "1"   # ... also at hidden 256 (measured slower there)
TWO_PASS_GATE = _switch("TWO_PASS_GATE")   # the single-rank BatchNorm forward at hidden 128 as statistics pass + fused gate (tools/train_ab.py switches it for A/B runs)


def _storage_dtype(model):
    """`model.activation_storage`: "fp32" (the reference's arithmetic, the default) or "bf16" - the pre-normalisation gate output
    xe and its gradient dxe are kept in HBM as bfloat16 between the kernels of the training step (BASELINE configs[2]; a sixth of
    the step's traffic).  All arithmetic, every statistic and the residual streams e / e' / de / h stay fp32."""
    kind = getattr(model, "activation_storage", "fp32")
    if kind not in ACTIVATION_STORAGE:
        raise ValueError(f"activation_storage={kind!r} not in {ACTIVATION_STORAGE}")
    return torch.bfloat16 if kind == "bf16" else torch.float32


def _recompute_gate(model, storage):
    """`model.recompute_gate` (default False): the pre-normalisation gate output xe[E,H] of every layer is NOT kept for the
    backward; the backward launches the layer's raw gate again on the same operands (deterministic kernels: the same bits).
    One [E,H] tensor per layer less in HBM - 8 of the ~34 KB per edge a step holds at hidden 256 (BASELINE configs[4]: 6.25M
    local edges per rank) - for one more gate launch per layer and step (~10 % of a step)."""
    on = bool(getattr(model, "recompute_gate", False))
    if on and storage != torch.float32:
        raise ValueError('recompute_gate and activation_storage="bf16" are alternatives (nothing is stored when xe is recomputed)')
    return on


def _raw_gate(sh, conv, e, B1h, B2h, layer_norm, storage, path=None):
    """xe = B1h[src] + B2h[dst] + e W3^T of one layer on the kernel the shard's shape allows -> (path, xe, statistics):
    "plain" (LayerNorm: no statistics), "moments" (single rank: shifted column sums from the same pass, finished by
    gnnome_bn_train_finish_f32) or "stats" ((mean, biased var) of the owned rows).  The backward of a model with
    recompute_gate passes the forward's `path` back so that the same launch produces the same bits."""
    ops, views = sh.ops, sh.views
    W3 = conv.B_3.weight.detach().contiguous()
    if layer_norm:
        if storage != torch.float32:
            raise _no_bf16_storage()
        return "plain", ops.edge_gate_raw(e, B1h, B2h, views, W3), None
    recomputing = path is not None
    if path is None:
        path = "moments" if _can_fuse_bn(sh, conv.bn_e) and ops.can_fuse_gate_moments(e, B1h, B2h, storage) else "stats"
    if path == "moments":
        xe, mom = ops.edge_gate_raw_moments(e, B1h, B2h, views, W3, storage=storage)
        return path, xe, mom
    if storage != torch.float32:
        # bf16 storage off the fused single-rank path (round 4: partitions - BatchNorm statistics cross ranks, so they cannot come out of the
        # gate's own pass): the rows rounded by the same kernel, the statistics of the ROUNDED owned rows from a second pass over them
        if not ops.can_fuse_gate_moments(e, B1h, B2h, storage):
            raise _no_bf16_storage()
        xe, _ = ops.edge_gate_raw_moments(e, B1h, B2h, views, W3, storage=storage)
        m_e, v_e = ops.batch_stats(xe[:sh.e_own].float())   # (a transient fp32 copy of the owned rows; freed before the next layer)
        return path, xe, (m_e, v_e)
    if recomputing and sh.e_own != e.shape[0]:   # a partition's forward took edge_gate_raw + batch_stats(owned rows): the gate alone
        return path, ops.edge_gate_raw(e, B1h, B2h, views, W3), None
    xe, m_e, v_e = ops.edge_gate_raw_stats(e, B1h, B2h, views, W3, rows_stats=sh.e_own)
    return path, xe, (m_e, v_e)


def _no_bf16_storage():
    return ValueError('activation_storage="bf16" is built for BatchNorm models at hidden_features 64 / 128 / 256 (single rank: fused; partitions: round 4); '
                      "normalization='layer' and the recomputed gate take \"fp32\"")


class _TrainStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, sh, x, e_raw, names, *params):
        layer_norm = isinstance(model.gnn.convs[0].bn_e, torch.nn.LayerNorm)   # normalization='layer' (gated_gcn_full.py:40-42)
        ln_width = model.__dict__.get("_gnnome_norm_width")    # a zero-padded twin (_padded_step): LayerNorm over the model's own channels
        ops, views = sh.ops, sh.views
        H = model.linear2_node.out_features
        n_own, n_local, e_own, e_local = sh.n_own, sh.n_local, sh.e_own, sh.e_local
        r = _roles(views.transposed)
        blk = lambda P, k: P[:, r[k] * H:(r[k] + 1) * H]  # noqa: E731
        d = lambda t: t.detach().contiguous()  # noqa: E731
        new = lambda rows, cols: torch.empty((rows, cols), dtype=torch.float32, device=x.device)  # noqa: E731
        storage = _storage_dtype(model)
        recompute = _recompute_gate(model, storage)

        node_gather = getattr(views, "node_gather", None)   # views over renumbered nodes read x (caller's numbering) through it
        h = ops.encode(x, d(model.linear1_node.weight), d(model.linear1_node.bias), d(model.linear2_node.weight), d(model.linear2_node.bias),
                       **({} if node_gather is None else dict(gather=node_gather, rows=n_local)))
        e = ops.encode(e_raw, d(model.linear1_edge.weight), d(model.linear1_edge.bias), d(model.linear2_edge.weight),
                       d(model.linear2_edge.bias), gather=views.srt_eid, rows=e_local)
        saved = []
        for li, conv in enumerate(model.gnn.convs):
            Wcat, bcat, WcatT, W3T = _cat_layer(conv, ops)
            if li > 0:
                sh.halo_start(h)        # layer 0's halo rows come straight from the input features
            P = new(n_local, 5 * H)
            ops.linear(h[:n_own], Wcat, bcat, out=P[:n_own])    # owned rows while the halo rows are in flight
            sh.halo_finish()
            if n_local > n_own:
                ops.linear(h[n_own:], Wcat, bcat, out=P[n_own:])
            mean_e = rstd_e = sc_e = sh_e = mean_h = rstd_h = sc_h = sh_h = None
            if layer_norm and storage != torch.float32:
                raise _no_bf16_storage()
            # (fp32 storage only - measured, tools/train_two_pass_ab.py: 24.90 against 25.28 ms per step; with bf16 storage the third pass moves half
            #  the bytes and the two forms are level, 24.2 against 24.1)
            # (hidden = 256, round 5: the kernels exist - edge_tile_f16.hip - and measure 134.4 against 132.9 ms per step at the 2.5M-edge shard: the
            #  product formed twice costs more than the pass it saves at that width; GNNOME_TWO_PASS_GATE_WIDE=1 selects it)
            two_pass = (TWO_PASS_GATE and not layer_norm and not recompute and storage == torch.float32 and _can_fuse_bn(sh, conv.bn_e) and hasattr(ops, "edge_gate_bn") and
                        (H == 128 or TWO_PASS_GATE_WIDE) and ops.can_two_pass_gate(e, blk(P, "B1"), blk(P, "B2"), storage))
            if two_pass:
                # round 4: statistics alone (nothing stored), then the gate with the statistics folded in, which also leaves xe for the
                # backward - the [E,H] tensor is written once and not read back in the forward (2 GB per layer instead of 2.5 at configs[2])
                W3 = d(conv.B_3.weight)
                path = "moments"
                mean_e, rstd_e, sc_e, sh_e = _bn_train_fused(sh, conv.bn_e, ops.edge_gate_moments_only(e, blk(P, "B1"), blk(P, "B2"), views, W3, storage=storage),
                                                             updates=2)
                e_new, xe = ops.edge_gate_bn(e, blk(P, "B1"), blk(P, "B2"), views, W3, sc_e, sh_e, storage=storage)
            else:
                path, xe, stats = _raw_gate(sh, conv, e, blk(P, "B1"), blk(P, "B2"), layer_norm, storage)
            if two_pass:
                pass
            elif layer_norm:   # per-row statistics: nothing crosses rows (or ranks), and there are no running buffers
                e_new = ops.ln_relu_res(xe, d(conv.bn_e.weight), d(conv.bn_e.bias), e, width=ln_width)
            else:
                if path == "moments":
                    mean_e, rstd_e, sc_e, sh_e = _bn_train_fused(sh, conv.bn_e, stats, updates=2)
                else:
                    mean_e, rstd_e, sc_e, sh_e = _bn_train(sh, conv.bn_e, stats[0], stats[1], e_own, sh.e_global, updates=2)
                e_new = ops.bn_relu_res(xe, sc_e, sh_e, e)
            if recompute:
                xe = None   # the backward runs the same gate launch again (same kernel, same operands: the same bits)
            v, hf, rdf, hb, rdb = ops.node_aggregate_raw(e_new, blk(P, "A1"), blk(P, "A2"), blk(P, "A3"), views, 1, n_own,
                                                         rows_alloc=n_local)
            h_next = new(n_local, H)
            if layer_norm:
                ops.ln_relu_res(v[:n_own], d(conv.bn_h.weight), d(conv.bn_h.bias), h[:n_own], out=h_next[:n_own], width=ln_width)
            else:
                if _can_fuse_bn(sh, conv.bn_h):
                    mean_h, rstd_h, sc_h, sh_h = _bn_train_fused(sh, conv.bn_h, ops.batch_moments(v[:n_own]), updates=1)
                else:
                    m_h, v_h = ops.batch_stats(v[:n_own])
                    mean_h, rstd_h, sc_h, sh_h = _bn_train(sh, conv.bn_h, m_h, v_h, n_own, sh.n_global, updates=1)
                ops.bn_relu_res(v[:n_own], sc_h, sh_h, h[:n_own], out=h_next[:n_own])
            mask = None
            if conv.dropout > 0.0:
                mask = dropout_mask(n_own, H, conv.dropout, x.device)
                h_next[:n_own].copy_(ops.mul23(h_next[:n_own], mask, mask)[0])
            saved.append(dict(h=h, P=P, e=e, xe=xe, gate_path=(path, layer_norm, storage), e_new=e_new, mean_e=mean_e, rstd_e=rstd_e, v=v, hf=hf, rdf=rdf, hb=hb, rdb=rdb,
                              mean_h=mean_h, rstd_h=rstd_h, mask=mask, Wcat=Wcat, WcatT=WcatT, W3T=W3T, sc_e=sc_e, sh_e=sh_e, sc_h=sc_h, sh_h=sh_h))
            h, e = h_next, e_new

        pred = model.predictor
        hs = pred.W1.out_features
        W1 = d(pred.W1.weight)
        W_nodes = torch.cat([W1[:, :H], W1[:, H:2 * H]], 0).contiguous()
        b_nodes = torch.cat([torch.zeros_like(pred.W1.bias), pred.W1.bias]).detach().contiguous()
        sh.halo_start(h)
        sh.halo_finish()
        PQ = ops.linear(h, W_nodes, b_nodes)
        ps, qd = (PQ[:, hs:], PQ[:, :hs]) if views.transposed else (PQ[:, :hs], PQ[:, hs:])
        # every edge of the graph is scored once, by the rank that owns its destination
        z1 = new(e_own, hs)
        w_tail = (W1[:, 2 * H:], d(pred.W2.weight), d(pred.W2.bias), d(pred.W3.weight.reshape(-1)), d(pred.W3.bias.reshape(-1)))
        if not sh.alone:    # one contiguous piece per rank in sorted order; finish_logits all-gathers and un-permutes them
            piece = torch.empty(sh.part.score_pad, dtype=torch.float32, device=h.device)
            ops.edge_score(e, ps, qd, sh.score_views, *w_tail, piece, num_edges=e_own, scatter_to_edge_id=False, z1_out=z1)
            logits = sh.finish_logits(piece)
        else:
            logits = torch.empty(sh.e_global, dtype=torch.float32, device=h.device)
            ops.edge_score(e, ps, qd, sh.score_views, *w_tail, logits, num_edges=e_own, z1_out=z1)
        ctx.model, ctx.sh, ctx.names, ctx.saved = model, sh, names, saved
        ctx.tail = dict(h=h, e=e, z1=z1, W1=W1, W_nodes=W_nodes, x=x, e_raw=e_raw)
        return logits.unsqueeze(1)

    @staticmethod
    def backward(ctx, dlogits):
        model, sh, saved, tail = ctx.model, ctx.sh, ctx.saved, ctx.tail
        if saved is None:
            raise RuntimeError("the activations of this training step were released by its first backward(); "
                               "retain_graph=True is not supported - run the forward again")
        ops, views = sh.ops, sh.views
        H = model.linear2_node.out_features
        ln_width = model.__dict__.get("_gnnome_norm_width")
        n_own, n_local, e_own, e_local = sh.n_own, sh.n_local, sh.e_own, sh.e_local
        r = _roles(views.transposed)
        blk = lambda P, k: P[:, r[k] * H:(r[k] + 1) * H]  # noqa: E731
        d = lambda t: t.detach().contiguous()  # noqa: E731
        dev = dlogits.device
        g = {}
        # Rows that also live on another rank (cut edges, halo nodes) carry PARTIAL gradients here: everything below
        # is linear in the incoming gradient, so the per-rank parameter gradients sum to the whole graph's (sum_ranks
        # at the end), and halo_bwd returns the halo rows' share of dh to the owners once per layer.

        # ---- scorer (score_predictor.py:12-17): the owned in-edges, positions [0, e_own)
        pred = model.predictor
        hs = pred.W1.out_features
        dl = dlogits.reshape(-1).contiguous().float()
        dz1, dz2, u = ops.score_tail_bwd(tail["z1"], dl, sh.score_views, d(pred.W2.weight), d(pred.W2.bias), d(pred.W3.weight.reshape(-1)))
        g["predictor.W2.weight"] = ops.wgrad(dz2, tail["z1"])
        g["predictor.W2.bias"] = ops.colsum2(dz2)[0]
        g["predictor.W3.weight"] = ops.colsum2(u)[0].reshape(1, 32)
        g["predictor.W3.bias"] = (dl if sh.alone else dl[sh.score_views.srt_eid.long()]).sum().reshape(1)
        if e_local > e_own:
            dz1 = torch.cat([dz1, torch.zeros((e_local - e_own, hs), dtype=torch.float32, device=dev)], 0)
        W1 = tail["W1"]
        de = ops.linear(dz1, W1[:, 2 * H:].t().contiguous(), None)           # d e_final  [e_local,H]
        gW1e = ops.wgrad(dz1, tail["e"])
        d_ps = ops.segment_sum(dz1, views.out_ptr, views.out_pos, n_local)   # gathered by srt_src in the forward
        d_qd = ops.segment_sum(dz1, views.in_ptr, None, n_local)             # gathered by srt_dst
        dPQ = torch.cat([d_qd, d_ps], 1) if views.transposed else torch.cat([d_ps, d_qd], 1)
        g["predictor.W1.bias"] = ops.colsum2(dPQ[:, hs:].contiguous())[0]
        gWn = ops.wgrad(dPQ, tail["h"])                                      # [2hs, H]
        g["predictor.W1.weight"] = torch.cat([gWn[:hs], gWn[hs:], gW1e], 1)
        dh = ops.linear(dPQ, tail["W_nodes"].t().contiguous(), None)         # [n_local,H]

        # ---- layers, last to first (gated_gcn_full.py:82-142)
        # the halo rows of dh travel back to their owners while the kernels that do not read dh run (round 6): the predictor's and, per layer, the
        # weight gradients' - halo_bwd_start right after dh is complete, halo_bwd_finish where its owned rows are read next
        split_halo = hasattr(sh, "halo_bwd_start")
        if split_halo and saved:
            sh.halo_bwd_start(dh)
        for li in range(len(saved) - 1, -1, -1):
            s, conv, pfx = saved[li], model.gnn.convs[li], f"gnn.convs.{li}."
            if split_halo:
                sh.halo_bwd_finish(dh)
            else:
                sh.halo_bwd(dh)
            if s["mask"] is not None:
                dh[:n_own].copy_(ops.mul23(dh[:n_own], s["mask"], s["mask"])[0])
            # h' = relu(bn_h(v)) + h_in      (owned rows)
            dv = (torch.zeros if n_local > n_own else torch.empty)((n_local, H), dtype=torch.float32, device=dev)
            if s["sc_h"] is None:   # LayerNorm: per-row backward, linear in dy, so partial gradients of replicas simply add
                _, g[pfx + "bn_h.weight"], g[pfx + "bn_h.bias"] = ops.ln_bwd(dh[:n_own], s["v"][:n_own], d(conv.bn_h.weight),
                                                                             d(conv.bn_h.bias), out=dv[:n_own], width=ln_width)
                tables = None
            amax_nodes = None   # one slot for max |.| over the five node gradients dv, sum_out, sum_in, dB1, dB2 (raised by the kernels that write them)
            if s["sc_h"] is None:
                pass
            elif n_local == n_own and FUSED_NODE_TABLES and hasattr(ops, "bn_bwd_apply_tables"):
                if (SCALED_NODE_WGRAD and s["sc_e"] is not None and FUSED_AGG_BWD and hasattr(ops, "agg_bwd_fused") and getattr(ops, "NODE_AMAX", False)
                        and H % 128 == 0 and (s["xe"].dtype if s["xe"] is not None else s["gate_path"][2]) == torch.float32):
                    amax_nodes = torch.zeros(1, dtype=torch.int32, device=dev)
                kw = {"amax": amax_nodes} if amax_nodes is not None else {}
                # bn_h's backward and the four node tables of the aggregation's backward (Tf, Uf, Tb, Ub) in one pass over the node rows
                g[pfx + "bn_h.weight"], g[pfx + "bn_h.bias"], c1, c2 = _bn_bwd(sh, dh, s["v"], s["sc_h"], s["sh_h"], s["mean_h"], s["rstd_h"],
                                                                               sh.n_global, n_own, None, apply=False)
                _, *tables = ops.bn_bwd_apply_tables(dh, s["v"], s["sc_h"], s["sh_h"], s["sc_h"], c1, c2, s["mean_h"], s["rstd_h"], s["rdf"],
                                                     s["hf"], s["rdb"], s["hb"], out=dv, **kw)
            else:
                g[pfx + "bn_h.weight"], g[pfx + "bn_h.bias"] = _bn_bwd(sh, dh[:n_own], s["v"][:n_own], s["sc_h"], s["sh_h"], s["mean_h"],
                                                                       s["rstd_h"], sh.n_global, n_own, dv[:n_own])
                tables = None
            dh_in = dh
            # v = A1h + fwd + bwd
            if tables is not None:
                Tf, Uf, Tb, Ub = tables
            else:
                Tf, Uf = ops.mul23(dv, s["rdf"], s["hf"])
                Tb, Ub = ops.mul23(dv, s["rdb"], s["hb"])
            if s["xe"] is None:     # model.recompute_gate: xe was not kept - one more gate launch instead of an [E,H] tensor per layer
                _, layer_norm_, storage_ = s["gate_path"]
                _, s["xe"], _ = _raw_gate(sh, conv, s["e"], blk(s["P"], "B1"), blk(s["P"], "B2"), layer_norm_, storage_, path=s["gate_path"][0])
            stats_e = None
            amax_dxe = None
            fused = s["sc_e"] is not None and FUSED_AGG_BWD and hasattr(ops, "agg_bwd_fused")
            if fused:
                # the node sums (dA3 / dA2 by role), de += ... and bn_e's backward statistics in ONE pass over the e' rows
                sum_in, sum_out, _, s1_e, s2_e = ops.agg_bwd_fused(s["e_new"], Tf, Uf, Tb, Ub, blk(s["P"], "A2"), blk(s["P"], "A3"), views, de,
                                                                   s["xe"], s["sc_e"], s["sh_e"], s["mean_e"], n_local,
                                                                   **({"amax": amax_nodes} if amax_nodes is not None else {}))
                stats_e = (s1_e, s2_e)
            else:
                amax_nodes = None   # (these sums come without their maximum)
                sum_in, sum_out = ops.node_aggregate_raw(s["e_new"], None, Tb, Tf, views, 2, n_local)   # = dA3(role), dA2(role)
            if fused:
                pass
            elif s["sc_e"] is not None and hasattr(ops, "agg_edge_bwd_stats"):
                # de += ... and bn_e's backward statistics of the result in the same pass over the edges
                _, s1_e, s2_e = ops.agg_edge_bwd_stats(s["e_new"], Tf, Uf, Tb, Ub, blk(s["P"], "A2"), blk(s["P"], "A3"), views, de, s["xe"],
                                                       s["sc_e"], s["sh_e"], s["mean_e"])
                stats_e = (s1_e, s2_e)
            else:
                ops.agg_edge_bwd(s["e_new"], Tf, Uf, Tb, Ub, blk(s["P"], "A2"), blk(s["P"], "A3"), views, de)  # de += ...
            # e' = relu(bn_e(xe)) + e_in ;  xe = B1h[src] + B2h[dst] + e_in W3^T
            if s["sc_e"] is None:
                dxe = torch.empty_like(de)
                _, g[pfx + "bn_e.weight"], g[pfx + "bn_e.bias"] = ops.ln_bwd(de, s["xe"], d(conv.bn_e.weight), d(conv.bn_e.bias), out=dxe, width=ln_width)
            else:
                W3t = s["W3T"] if s["W3T"] is not None else d(conv.B_3.weight).t().contiguous()
                if hasattr(ops, "bn_bwd_dgrad") and ops.can_fuse_bn_bwd_dgrad(de, W3t, s["xe"]):
                    # BatchNorm backward and d e_in = d e' + dxe W3 in one pass over the edges (dxe computed by the load waves)
                    g[pfx + "bn_e.weight"], g[pfx + "bn_e.bias"], c1, c2 = _bn_bwd(sh, de, s["xe"], s["sc_e"], s["sh_e"], s["mean_e"], s["rstd_e"],
                                                                                   sh.e_global, e_own, None, stats=stats_e, apply=False)
                    if SCALED_WGRAD and hasattr(ops, "can_dgrad_amax") and ops.can_dgrad_amax(de, s["xe"]):
                        # the kernel that writes dxe also leaves max |dxe|: B_3's weight gradient below then runs as fp16x3 on the scaled rows
                        amax_dxe = torch.empty(1, dtype=torch.int32, device=dev)
                    dxe = ops.bn_bwd_dgrad(de, s["xe"], s["sc_e"], s["sh_e"], s["sc_e"], c1, c2, s["mean_e"], s["rstd_e"], W3t, rows_once=e_own,
                                           **({"amax": amax_dxe} if amax_dxe is not None else {}))
                    W3t = None
                elif s["xe"].dtype != torch.float32:
                    raise _no_bf16_storage()
                else:
                    dxe = torch.empty_like(de)
                    g[pfx + "bn_e.weight"], g[pfx + "bn_e.bias"] = _bn_bwd(sh, de, s["xe"], s["sc_e"], s["sh_e"], s["mean_e"], s["rstd_e"],
                                                                           sh.e_global, e_own, dxe, stats=stats_e)
            if s["sc_e"] is None or W3t is not None:
                ops.linear(dxe, s["W3T"] if s["W3T"] is not None else d(conv.B_3.weight).t().contiguous(), None, out=de,
                           accumulate=True)   # d e_in = d e' + dxe W3
            if hasattr(ops, "segment_sum2"):   # both gathers' transposes in one launch (the out-edge pass then hits L2)
                dB2, dB1 = ops.segment_sum2(dxe, views, n_local, **({"amax": amax_nodes} if amax_nodes is not None else {}))
            else:
                amax_nodes = None
                dB1 = ops.segment_sum(dxe, views.out_ptr, views.out_pos, n_local)
                dB2 = ops.segment_sum(dxe, views.in_ptr, None, n_local)
            parts = [None] * 5
            parts[r["A1"]], parts[r["A2"]], parts[r["A3"]], parts[r["B1"]], parts[r["B2"]] = dv, sum_out, sum_in, dB1, dB2
            names = ("A_1", "A_2", "A_3", "B_1", "B_2")
            WcatT = s["WcatT"] if s["WcatT"] is not None else s["Wcat"].t().contiguous()
            # dh first: its halo rows then travel (halo_bwd_start) under the weight gradients, which do not read it
            if hasattr(ops, "wgrad_blocks") and ops.can_use_blocks(parts):
                # the five [N,H] gradients stay where their kernels left them: weight gradients, bias gradients (column sums of
                # the same slabs) and dh += dP Wcat read them as column blocks
                # (amax_nodes: every block's producer raised it - the product runs as fp16x3 on the blocks scaled by their common maximum)
                dh = ops.linear_blocks(parts, WcatT, dh_in, accumulate=True, **({"amax": amax_nodes} if amax_nodes is not None and SCALED_NODE_DGRAD else {}))
                if split_halo and li > 0:
                    sh.halo_bwd_start(dh)
                gWcat, gbcat = ops.wgrad_blocks(parts, s["h"], **({"amax": amax_nodes} if amax_nodes is not None else {}))   # [5H, H], [5H]
                for k, name in enumerate(names):
                    g[pfx + name + ".bias"] = gbcat[k * H:(k + 1) * H]
            else:
                dP = torch.cat(parts, 1)
                dh = ops.linear(dP, WcatT, None, out=dh_in, accumulate=True)
                if split_halo and li > 0:
                    sh.halo_bwd_start(dh)
                for k, name in enumerate(names):
                    g[pfx + name + ".bias"] = ops.colsum2(parts[k])[0]
                gWcat = ops.wgrad(dP, s["h"])                                 # [5H, H]
            g[pfx + "B_3.weight"] = ops.wgrad(dxe, s["e"], amax=amax_dxe) if amax_dxe is not None else ops.wgrad(dxe, s["e"])
            # every edge has exactly one destination: sum_p dxe[p] = sum_i dB2[i], no second pass over [E,H]
            g[pfx + "B_3.bias"] = g[pfx + ("B_1" if views.transposed else "B_2") + ".bias"].clone()
            for k, name in enumerate(names):
                g[pfx + name + ".weight"] = gWcat[k * H:(k + 1) * H]
            saved[li] = s = None    # this layer's activations are done with: their memory serves the next layer's temporaries

        # ---- encoders (models/full_graph.py:26-27)
        def encoder_bwd(dout, inp, gather, rows, l1, l2, pfx1, pfx2):
            W1e_, b1e_, W2e_ = d(l1.weight), d(l1.bias), d(l2.weight)
            t = ops.encode_hidden(inp, W1e_, b1e_, gather=gather, rows=rows)
            g[pfx2 + ".weight"] = ops.wgrad(dout, t)
            g[pfx2 + ".bias"] = ops.colsum2(dout)[0]
            dt = ops.relu_bwd(ops.linear(dout, W2e_.t().contiguous(), None), t)
            F_ = inp.shape[1]
            src = inp if gather is None else inp[gather.long()]
            x4 = torch.zeros((rows, 4 * ((F_ + 3) // 4)), dtype=torch.float32, device=inp.device)
            x4[:, :F_] = src
            g[pfx1 + ".weight"] = ops.wgrad(dt, x4)[:, :F_].contiguous()
            g[pfx1 + ".bias"] = ops.colsum2(dt)[0] if dt.shape[1] in (16, 32, 64) else dt.sum(0)

        encoder_bwd(dh, tail["x"], getattr(views, "node_gather", None), n_local, model.linear1_node, model.linear2_node, "linear1_node", "linear2_node")
        encoder_bwd(de, tail["e_raw"], views.srt_eid, e_local, model.linear1_edge, model.linear2_edge, "linear1_edge", "linear2_edge")

        ctx.saved = ctx.tail = None
        grads = sh.sum_ranks([g[n].contiguous() for n in ctx.names])
        return (None, None, None, None, None) + tuple(grads)


def train_forward(model, graph, x, e):
    """`model(graph, x, e)` in train mode with autograd support: logits [E,1] on the compute device."""
    from .engine import compute_device
    device = compute_device(x, e)
    views = views_for(graph, device, node_order=getattr(model, "node_order", "input"))
    xd = x.detach().to(device=device, dtype=torch.float32).contiguous()
    ed = e.detach().to(device=device, dtype=torch.float32).contiguous()
    out = train_forward_on(model, WholeGraph(views), xd, ed)
    views.check_range()   # a fresh graph's deferred endpoint check (GraphViews validate="lazy")
    return out


def train_forward_on(model, shard, x_local, e_local):
    """The training step over `shard`'s rows (WholeGraph, or one rank's gnnome_amd.dist.PartitionShard): x_local /
    e_local are the input features of the shard's nodes / edges (local edge-id order).  Returns logits[E_global, 1],
    complete on every rank, differentiable w.r.t. the model's parameters; the parameter gradients that
    `backward()` leaves in `.grad` are already summed over ranks."""
    from .engine import BUILT_HIDDEN
    H, hs = model.linear2_node.out_features, model.predictor.W1.out_features
    names = [n for n, _ in model.named_parameters()]
    params = [p for _, p in model.named_parameters()]
    if any(p.device != x_local.device for p in params):
        raise RuntimeError("training needs the model on the compute device: call model.to(device) first")
    if H not in BUILT_HIDDEN or (hs not in TRAIN_SCORE_HIDDEN and hs <= TRAIN_SCORE_HIDDEN[-1]):
        return _padded_step(model, shard, x_local, e_local, names, params)
    return _TrainStep.apply(model, shard, x_local, e_local, names, *params)


TRAIN_SCORE_HIDDEN = (32, 64, 128)   # hidden_edge_scores the scorer's backward is built for (gnnome_score_tail_bwd_f32; 128: round 5)


def _pad_like(p, shape):
    """p in the leading corner of zeros of `shape` - a differentiable op, so gradients of the padded tensor flow back to p sliced."""
    pads = []
    for have, want in zip(reversed(p.shape), reversed(tuple(shape))):
        pads += [0, want - have]
    return torch.nn.functional.pad(p, pads) if any(pads) else p


def _padded_step(model, shard, x_local, e_local, names, params):
    """The training step of a model whose widths lie BETWEEN the built ones (the reference takes any hidden_features / hidden_edge_scores,
    configs/hyperparameters.py:22-24; engine.BUILT_HIDDEN): run on a twin of the next built widths whose parameters are zero-padded,
    DIFFERENTIABLE functions of the model's own - exact, like the padded inference path: a padded channel carries zero weights, gamma = beta = 0
    and constant-zero activations (batch mean 0, variance 0, x_hat = 0), so its relu mask is off, every gradient that reaches it is zero, and
    the gradients of the padded tensors arrive at the model's parameters sliced by autograd.  LayerNorm (round 5): the twin's kernels take the
    statistics over the model's own channels (`_gnnome_norm_width`; a padded channel gets xhat = 0, output 0 and dx = 0)."""
    from .engine import padded_width
    from .models import SymGatedGCNModel
    conv0 = model.gnn.convs[0]
    layer_norm = isinstance(conv0.bn_e, torch.nn.LayerNorm)
    H, hs = model.linear2_node.out_features, model.predictor.W1.out_features
    if hs > TRAIN_SCORE_HIDDEN[-1]:
        raise ValueError(f"train mode at hidden_edge_scores={hs}: the scorer's backward is built for widths up to {TRAIN_SCORE_HIDDEN[-1]}")
    Hp, hsp = padded_width(H), padded_width(hs, TRAIN_SCORE_HIDDEN, "hidden_edge_scores")
    dev = params[0].device
    twin = model.__dict__.get("_gnnome_padded_twin")
    if twin is None or next(twin.parameters()).device != dev:
        twin = SymGatedGCNModel(model.linear1_node.in_features, model.linear1_edge.in_features, Hp, model.linear1_node.out_features,
                                len(model.gnn.convs), hsp, "layer" if layer_norm else "batch", dropout=conv0.dropout).to(dev)
        model.__dict__["_gnnome_padded_twin"] = twin
    if layer_norm and Hp != H:
        twin.__dict__["_gnnome_norm_width"] = H
    twin.train()
    for attr in ("activation_storage", "recompute_gate", "node_order"):
        if hasattr(model, attr):
            setattr(twin, attr, getattr(model, attr))
    for conv, tconv in zip(model.gnn.convs, twin.gnn.convs):
        tconv.dropout = conv.dropout
    tparams = dict(twin.named_parameters())
    padded = []
    for name, p in zip(names, params):
        t = tparams[name]
        if name == "predictor.W1.weight":   # [hs, 3 H]: the x[src] | x[dst] | e blocks are padded one by one
            q = torch.cat([_pad_like(p[:, i * H:(i + 1) * H], (hsp, Hp)) for i in range(3)], 1)
        else:
            q = _pad_like(p, t.shape)
        with torch.no_grad():
            t.copy_(q)          # what the kernels read; autograd sees `q`
        padded.append(q)
    mbufs, tbufs = dict(model.named_buffers()), dict(twin.named_buffers())
    with torch.no_grad():
        for k, b in mbufs.items():
            tb = tbufs[k]
            if b.dim() == 0:
                tb.copy_(b)
            else:
                tb.fill_(1.0 if k.endswith("running_var") else 0.0)
                tb[:b.shape[0]].copy_(b)
    out = _TrainStep.apply(twin, shard, x_local, e_local, names, *padded)
    with torch.no_grad():       # the running statistics the step has just updated, back into the model's own buffers
        for k, b in mbufs.items():
            b.copy_(tbufs[k] if b.dim() == 0 else tbufs[k][:b.shape[0]])
    return out
