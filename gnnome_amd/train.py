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
"""
import torch

from . import ops
from .graph import views_for

EPS_BN = 1e-5


def _cat_layer(conv):
    Wcat = torch.cat([conv.A_1.weight, conv.A_2.weight, conv.A_3.weight, conv.B_1.weight, conv.B_2.weight], 0).detach().contiguous()
    bcat = torch.cat([conv.A_1.bias, conv.A_2.bias, conv.A_3.bias, conv.B_1.bias, conv.B_2.bias + conv.B_3.bias], 0).detach().contiguous()
    return Wcat, bcat


def _roles(transposed):
    """column block of P used in each kernel role (dgl.reverse swaps src <-> dst: A2<->A3, B1<->B2)."""
    return dict(A1=0, A2=2, A3=1, B1=4, B2=3) if transposed else dict(A1=0, A2=1, A3=2, B1=3, B2=4)


def _bn_train(bn, x, updates):
    """Batch statistics of x -> (mean, rstd, scale, shift); running buffers advanced `updates` times like nn.BatchNorm1d."""
    rows = x.shape[0]
    mean, var = ops.batch_stats(x)
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


def _bn_bwd(dy, x, scale, shift, mean, rstd, rows):
    """d(input of bn) and (d gamma, d beta) for out = relu(x*scale + shift) + res, scale = gamma*rstd."""
    s1, s2 = ops.bn_bwd_stats(dy, x, scale, shift, mean)
    s2h = rstd * s2                                     # sum dy*m*xhat
    dx = ops.bn_bwd_apply(dy, x, scale, shift, scale, (s1 / rows).contiguous(), (s2h / rows).contiguous(), mean, rstd)
    return dx, s2h, s1


class _TrainStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, views, x, e_raw, names, *params):
        conv0 = model.gnn.convs[0]
        if not isinstance(conv0.bn_e, torch.nn.BatchNorm1d):
            raise NotImplementedError("training is implemented for normalization='batch' (the reference default)")
        H = model.linear2_node.out_features
        N, E = views.num_nodes, views.num_edges
        r = _roles(views.transposed)
        blk = lambda P, k: P[:, r[k] * H:(r[k] + 1) * H]  # noqa: E731
        d = lambda t: t.detach().contiguous()  # noqa: E731

        h = ops.encode(x, d(model.linear1_node.weight), d(model.linear1_node.bias), d(model.linear2_node.weight), d(model.linear2_node.bias))
        e = ops.encode(e_raw, d(model.linear1_edge.weight), d(model.linear1_edge.bias), d(model.linear2_edge.weight),
                       d(model.linear2_edge.bias), gather=views.srt_eid, rows=E)
        saved = []
        for conv in model.gnn.convs:
            Wcat, bcat = _cat_layer(conv)
            P = ops.linear(h, Wcat, bcat)
            xe = ops.edge_gate_raw(e, blk(P, "B1"), blk(P, "B2"), views, d(conv.B_3.weight))
            mean_e, rstd_e, sc_e, sh_e = _bn_train(conv.bn_e, xe, updates=2)
            e_new = ops.bn_relu_res(xe, sc_e, sh_e, e)
            v, hf, rdf, hb, rdb = ops.node_aggregate_raw(e_new, blk(P, "A1"), blk(P, "A2"), blk(P, "A3"), views, 1, N)
            mean_h, rstd_h, sc_h, sh_h = _bn_train(conv.bn_h, v, updates=1)
            h_new = ops.bn_relu_res(v, sc_h, sh_h, h)
            mask = None
            h_next = h_new
            if conv.dropout > 0.0:
                mask = torch.empty_like(h_new).bernoulli_(1.0 - conv.dropout).div_(1.0 - conv.dropout)
                h_next, _ = ops.mul23(h_new, mask, mask)
            saved.append(dict(h=h, P=P, e=e, xe=xe, e_new=e_new, mean_e=mean_e, rstd_e=rstd_e, v=v, hf=hf, rdf=rdf, hb=hb, rdb=rdb,
                              mean_h=mean_h, rstd_h=rstd_h, mask=mask, Wcat=Wcat, sc_e=sc_e, sh_e=sh_e, sc_h=sc_h, sh_h=sh_h))
            h, e = h_next, e_new

        pred = model.predictor
        hs = pred.W1.out_features
        W1 = d(pred.W1.weight)
        W_nodes = torch.cat([W1[:, :H], W1[:, H:2 * H]], 0).contiguous()
        b_nodes = torch.cat([torch.zeros_like(pred.W1.bias), pred.W1.bias]).detach().contiguous()
        PQ = ops.linear(h, W_nodes, b_nodes)
        ps, qd = (PQ[:, hs:], PQ[:, :hs]) if views.transposed else (PQ[:, :hs], PQ[:, hs:])
        logits = torch.empty(E, dtype=torch.float32, device=h.device)
        z1 = torch.empty((E, hs), dtype=torch.float32, device=h.device)
        ops.edge_score(e, ps, qd, views, W1[:, 2 * H:], d(pred.W2.weight), d(pred.W2.bias), d(pred.W3.weight.reshape(-1)),
                       d(pred.W3.bias.reshape(-1)), logits, z1_out=z1)
        ctx.model, ctx.views, ctx.names, ctx.saved = model, views, names, saved
        ctx.tail = dict(h=h, e=e, z1=z1, W1=W1, W_nodes=W_nodes, x=x, e_raw=e_raw)
        return logits.unsqueeze(1)

    @staticmethod
    def backward(ctx, dlogits):
        model, views, saved, tail = ctx.model, ctx.views, ctx.saved, ctx.tail
        H = model.linear2_node.out_features
        N, E = views.num_nodes, views.num_edges
        r = _roles(views.transposed)
        blk = lambda P, k: P[:, r[k] * H:(r[k] + 1) * H]  # noqa: E731
        d = lambda t: t.detach().contiguous()  # noqa: E731
        g = {}

        # ---- scorer (score_predictor.py:12-17)
        pred = model.predictor
        hs = pred.W1.out_features
        dl = dlogits.reshape(-1).contiguous().float()
        dz1, dz2, u = ops.score_tail_bwd(tail["z1"], dl, views, d(pred.W2.weight), d(pred.W2.bias), d(pred.W3.weight.reshape(-1)))
        g["predictor.W2.weight"] = ops.wgrad(dz2, tail["z1"])
        g["predictor.W2.bias"] = ops.colsum2(dz2)[0]
        g["predictor.W3.weight"] = ops.colsum2(u)[0].reshape(1, 32)
        g["predictor.W3.bias"] = dl.sum().reshape(1)
        W1 = tail["W1"]
        de = ops.linear(dz1, W1[:, 2 * H:].t().contiguous(), None)           # d e_final  [E,H]
        gW1e = ops.wgrad(dz1, tail["e"])
        d_ps = ops.segment_sum(dz1, views.out_ptr, views.out_pos, N)         # gathered by srt_src in the forward
        d_qd = ops.segment_sum(dz1, views.in_ptr, None, N)                   # gathered by srt_dst
        dPQ = torch.cat([d_qd, d_ps], 1) if views.transposed else torch.cat([d_ps, d_qd], 1)
        g["predictor.W1.bias"] = ops.colsum2(dPQ[:, hs:].contiguous())[0]
        gWn = ops.wgrad(dPQ, tail["h"])                                      # [2hs, H]
        g["predictor.W1.weight"] = torch.cat([gWn[:hs], gWn[hs:], gW1e], 1)
        dh = ops.linear(dPQ, tail["W_nodes"].t().contiguous(), None)         # [N,H]

        # ---- layers, last to first (gated_gcn_full.py:82-142)
        for li in range(len(saved) - 1, -1, -1):
            s, conv, pfx = saved[li], model.gnn.convs[li], f"gnn.convs.{li}."
            if s["mask"] is not None:
                dh, _ = ops.mul23(dh, s["mask"], s["mask"])
            # h' = relu(bn_h(v)) + h_in
            dv, g[pfx + "bn_h.weight"], g[pfx + "bn_h.bias"] = _bn_bwd(dh, s["v"], s["sc_h"], s["sh_h"], s["mean_h"], s["rstd_h"], N)
            dh_in = dh
            # v = A1h + fwd + bwd
            Tf, Uf = ops.mul23(dv, s["rdf"], s["hf"])
            Tb, Ub = ops.mul23(dv, s["rdb"], s["hb"])
            sum_in, sum_out = ops.node_aggregate_raw(s["e_new"], None, Tb, Tf, views, 2, N)   # = dA3(role), dA2(role)
            ops.agg_edge_bwd(s["e_new"], Tf, Uf, Tb, Ub, blk(s["P"], "A2"), blk(s["P"], "A3"), views, de)  # de += ...
            # e' = relu(bn_e(xe)) + e_in ;  xe = B1h[src] + B2h[dst] + e_in W3^T
            dxe, g[pfx + "bn_e.weight"], g[pfx + "bn_e.bias"] = _bn_bwd(de, s["xe"], s["sc_e"], s["sh_e"], s["mean_e"], s["rstd_e"], E)
            g[pfx + "B_3.weight"] = ops.wgrad(dxe, s["e"])
            ops.linear(dxe, d(conv.B_3.weight).t().contiguous(), None, out=de, accumulate=True)   # d e_in = d e' + dxe W3
            dB1 = ops.segment_sum(dxe, views.out_ptr, views.out_pos, N)
            dB2 = ops.segment_sum(dxe, views.in_ptr, None, N)
            parts = [None] * 5
            parts[r["A1"]], parts[r["A2"]], parts[r["A3"]], parts[r["B1"]], parts[r["B2"]] = dv, sum_out, sum_in, dB1, dB2
            for k, name in enumerate(("A_1", "A_2", "A_3", "B_1", "B_2")):
                g[pfx + name + ".bias"] = ops.colsum2(parts[k])[0]
            # every edge has exactly one destination: sum_p dxe[p] = sum_i dB2[i], no second pass over [E,H]
            g[pfx + "B_3.bias"] = g[pfx + ("B_1" if views.transposed else "B_2") + ".bias"].clone()
            dP = torch.cat(parts, 1)
            gWcat = ops.wgrad(dP, s["h"])                                     # [5H, H]
            for k, name in enumerate(("A_1", "A_2", "A_3", "B_1", "B_2")):
                g[pfx + name + ".weight"] = gWcat[k * H:(k + 1) * H]
            dh = ops.linear(dP, s["Wcat"].t().contiguous(), None, out=dh_in, accumulate=True)

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

        encoder_bwd(dh, tail["x"], None, N, model.linear1_node, model.linear2_node, "linear1_node", "linear2_node")
        encoder_bwd(de, tail["e_raw"], views.srt_eid, E, model.linear1_edge, model.linear2_edge, "linear1_edge", "linear2_edge")

        ctx.saved = ctx.tail = None
        return (None, None, None, None, None) + tuple(g[n] for n in ctx.names)


def train_forward(model, graph, x, e):
    """`model(graph, x, e)` in train mode with autograd support: logits [E,1] on the compute device."""
    from .engine import compute_device
    device = compute_device(x, e)
    views = views_for(graph, device)
    names = [n for n, _ in model.named_parameters()]
    params = [p for _, p in model.named_parameters()]
    if any(p.device != device for p in params):
        raise RuntimeError("training needs the model on the compute device: call model.to(device) first")
    xd = x.detach().to(device=device, dtype=torch.float32).contiguous()
    ed = e.detach().to(device=device, dtype=torch.float32).contiguous()
    return _TrainStep.apply(model, views, xd, ed, names, *params)
