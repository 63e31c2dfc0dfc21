"""Whole-graph inference partitioned by destination-node range, one process per GPU.

The reference has no multi-GPU code at all (it scores whole graphs on the CPU "because GPU memory",
inference.py:388).  This is the north_star's scheme:

  * rank p OWNS a contiguous node range [lo_p, hi_p), chosen so that every rank has about the same
    number of incident edges;
  * it stores every edge INCIDENT to an owned node - in-edges and out-edges - plus the far endpoints
    of those edges as HALO nodes.  Local node ids put owned nodes first, so after the usual
    destination sort the in-edges of owned nodes are a PREFIX of the local edge storage;
  * a cut edge lives on two ranks; both compute the identical e' (it depends only on h[src], h[dst]
    and e), so edge state never travels;
  * once per layer the owners send the h rows their peers hold as halo: a packed all_to_all_single
    over RCCL/xGMI (direct peer-to-peer, per-link bound), received straight into the halo rows of h.
    The node projection of the owned rows runs while the exchange is in flight;
  * the scorer writes the logits of owned in-edges (every global edge exactly once) at their GLOBAL
    edge id into an [E] buffer; one all_reduce(SUM) over disjoint supports assembles the result.

All tensor math goes through an `ops` namespace exactly like engine.run_stack: the product passes
gnnome_amd.ops (HIP); the CPU/gloo tests pass the checker backend to exercise this host logic.
"""
import torch
import torch.distributed as dist

from . import engine
from . import ops as hip_ops


def split_by_incident_edges(src, dst, num_nodes, world):
    """Node-range boundaries [b_0=0, ..., b_world=N] balancing in-degree + out-degree per range."""
    deg = torch.bincount(dst.long(), minlength=num_nodes) + torch.bincount(src.long(), minlength=num_nodes)
    csum = torch.cumsum(deg + 1, 0)  # +1: isolated nodes still cost a node update
    total = int(csum[-1]) if num_nodes > 0 else 0
    bounds = [0]
    for p in range(1, world):
        target = total * p // world
        b = int(torch.searchsorted(csum, torch.tensor(target), right=False))
        bounds.append(max(bounds[-1], min(b, num_nodes)))
    bounds.append(num_nodes)
    return bounds


class PartitionedGraph:
    """Rank-local piece of a destination-range partition plus its halo-exchange plan."""

    def __init__(self):
        self.rank = self.world = 0
        self.lo = self.hi = self.n_own = self.n_local = self.num_edges_global = 0
        self.bounds = None
        self.node_gid = None        # int64 [n_local]  global id of local node (owned first, then halo ascending)
        self.edge_gid = None        # int64 [e_local]  global edge id of local edge (local edge-id order)
        self.views = None           # ops.GraphViews of the local graph
        self.n_score = 0            # in-edges of owned nodes = prefix of sorted positions
        self.srt_geid = None        # int32 [n_score]  global edge id of sorted position p
        self.send_idx = None        # int32 [sum(send_counts)] owned local rows to pack, grouped by peer
        self.send_counts = None     # python list, rows to send to each peer
        self.recv_counts = None     # python list, halo rows received from each peer (in halo order)

    @classmethod
    def from_global(cls, src, dst, num_nodes, rank, world, device, ops=hip_ops, group=None):
        """Build from the full edge list (every rank holds it in this harness, as inference.py holds the
        whole DGLGraph).  CPU torch preprocessing + one all_to_all of halo requests."""
        self = cls()
        src, dst = torch.as_tensor(src).long().cpu(), torch.as_tensor(dst).long().cpu()
        self.rank, self.world, self.num_edges_global = rank, world, int(src.numel())
        self.bounds = split_by_incident_edges(src, dst, num_nodes, world)
        lo, hi = self.bounds[rank], self.bounds[rank + 1]
        self.lo, self.hi, self.n_own = lo, hi, hi - lo

        own_d = (dst >= lo) & (dst < hi)
        own_s = (src >= lo) & (src < hi)
        keep = own_d | own_s
        self.edge_gid = torch.nonzero(keep).squeeze(1)
        ls, ld = src[keep], dst[keep]
        ends = torch.cat([ls, ld])
        halo = torch.unique(ends[(ends < lo) | (ends >= hi)])  # ascending => grouped by owner rank
        self.node_gid = torch.cat([torch.arange(lo, hi), halo])
        self.n_local = int(self.node_gid.numel())

        def to_local(g):
            own = (g >= lo) & (g < hi)
            pos = torch.searchsorted(halo, g.clamp(min=0)) if halo.numel() else torch.zeros_like(g)
            return torch.where(own, g - lo, self.n_own + pos)

        l_src, l_dst = to_local(ls).int(), to_local(ld).int()
        self.views = ops.GraphViews(l_src.to(device), l_dst.to(device), self.n_local)
        in_ptr = self.views.in_ptr
        self.n_score = int(in_ptr[self.n_own]) if self.n_own > 0 else 0
        assert self.n_score == int(own_d.sum())
        srt_eid = self.views.srt_eid[:self.n_score].long().cpu()
        self.srt_geid = self.edge_gid[srt_eid].int().to(device)

        # halo-exchange plan: tell each owner which of its rows I hold as halo
        bounds_t = torch.tensor(self.bounds)
        owner = torch.searchsorted(bounds_t, halo, right=True) - 1
        self.recv_counts = torch.bincount(owner, minlength=world).tolist() if halo.numel() else [0] * world
        # (the plan travels on `device` tensors: RCCL moves device memory only, gloo takes either)
        req_counts = torch.tensor(self.recv_counts, dtype=torch.int64, device=device)
        got_counts = torch.empty(world, dtype=torch.int64, device=device)
        if world > 1:
            dist.all_to_all_single(got_counts, req_counts, group=group)
        else:
            got_counts.copy_(req_counts)
        self.send_counts = got_counts.tolist()
        wanted = torch.empty(sum(self.send_counts), dtype=torch.int64, device=device)
        if world > 1:
            dist.all_to_all_single(wanted, halo.to(device).contiguous(), output_split_sizes=self.send_counts,
                                   input_split_sizes=self.recv_counts, group=group)
        assert wanted.numel() == 0 or (int(wanted.min()) >= lo and int(wanted.max()) < hi)
        self.send_idx = (wanted - lo).int().to(device)
        return self

    def local_node_rows(self, x_global):
        return x_global[self.node_gid]

    def local_edge_rows(self, e_global):
        return e_global[self.edge_gid]


class HaloExchange:
    """h[n_own:] <- owners' rows, via one packed all_to_all_single (RCCL on GPUs, gloo in the CPU tests)."""

    def __init__(self, part, ops, group=None):
        self.part, self.ops, self.group, self.work, self._keep = part, ops, group, None, None

    def start(self, h):
        p = self.part
        if p.world == 1:
            return
        packed = self.ops.gather_rows(h, p.send_idx)
        self._keep = packed
        self.work = dist.all_to_all_single(h[p.n_own:], packed, output_split_sizes=p.recv_counts,
                                           input_split_sizes=p.send_counts, group=self.group, async_op=True)

    def finish(self):
        if self.work is not None:
            self.work.wait()
            self.work, self._keep = None, None


def _project(ops, lw, h, n_own, xchg):
    """P = h * Wcat^T + bcat; the owned rows are projected while the halo rows are still in flight."""
    P = torch.empty((h.shape[0], lw.Wcat.shape[0]), dtype=torch.float32, device=h.device)
    if n_own > 0:
        ops.linear(h[:n_own], lw.Wcat, lw.bcat, out=P[:n_own])
    xchg.finish()
    if h.shape[0] > n_own:
        ops.linear(h[n_own:], lw.Wcat, lw.bcat, out=P[n_own:])
    return P


def run_partitioned(ops, prep, part, x_local, e_local, group=None, reduce_result=True):
    """The rank-local kernel sequence.  x_local: [n_local, F] features of owned+halo nodes, e_local:
    [e_local, F] features of local edges in local edge-id order.  Returns logits[E_global] (complete on
    every rank after the all_reduce when `reduce_result`)."""
    views, n_own = part.views, part.n_own
    xchg = HaloExchange(part, ops, group)
    H = prep.hidden
    h = ops.encode(x_local, *prep.enc_node)  # halo rows of layer 0 come straight from the input features
    e = engine.encode_edges(ops, prep, views, e_local)   # None: layer 0's gate encodes the edge tile itself
    for li, lw in enumerate(prep.layers):
        if li > 0:
            xchg.start(h)
        P = _project(ops, lw, h, n_own, xchg)
        A1, A2, A3, B1, B2 = (P[:, i * H:(i + 1) * H] for i in range(5))
        if e is None:
            e = ops.edge_gate_encode(e_local, prep.enc_edge, B1, B2, views, lw.W3, lw.scale_e, lw.shift_e)
        else:
            ops.edge_gate(e, B1, B2, views, lw.W3, lw.norm, lw.scale_e, lw.shift_e)
        h = ops.node_aggregate(e, A1, A2, A3, views, h, lw.norm, lw.scale_h, lw.shift_h, num_nodes_out=n_own)
    xchg.start(h)
    pw = prep.predictor
    hs = pw["hs"]
    xchg.finish()
    PQ = ops.linear(h, pw["W_nodes"], pw["b_nodes"])
    logits = torch.zeros(part.num_edges_global, dtype=torch.float32, device=h.device)
    if part.n_score > 0:
        # scatter straight to GLOBAL edge ids: a GraphViews-like shim whose srt_eid is the global map
        score_views = _ScoreViews(views, part.srt_geid)
        ops.edge_score(e, PQ[:, :hs], PQ[:, hs:], score_views, pw["W1_e"], pw["W2"], pw["b2"], pw["W3"], pw["b3"], logits,
                       num_edges=part.n_score)
    if reduce_result and part.world > 1:
        dist.all_reduce(logits, op=dist.ReduceOp.SUM, group=group)  # disjoint supports: a concatenation
    return logits


class _ScoreViews:
    """The scorer only reads srt_src / srt_dst / srt_eid."""

    def __init__(self, views, srt_geid):
        self.srt_src, self.srt_dst, self.srt_eid = views.srt_src, views.srt_dst, srt_geid


class PartitionedRunner:
    """Holds one rank's partition, weights and inputs on its GPU; forward() = one whole-graph scoring pass."""

    def __init__(self, model, part, x_global, e_global, device, ops=hip_ops, group=None):
        if model.training:
            raise NotImplementedError("partitioned execution is inference-only in this build")
        self.ops, self.part, self.group = ops, part, group
        self.prep = engine.prepared_for(model, device, engine.Prepared)
        self.x = part.local_node_rows(x_global).to(device=device, dtype=torch.float32).contiguous()
        self.e = part.local_edge_rows(e_global).to(device=device, dtype=torch.float32).contiguous()

    def forward(self):
        with torch.no_grad():
            return run_partitioned(self.ops, self.prep, self.part, self.x, self.e, self.group).unsqueeze(1)
