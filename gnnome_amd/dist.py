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
  * the scorer writes the logits of owned in-edges (every global edge exactly once) as one contiguous piece in
    sorted order; ONE all_gather of the pieces (E*4/G bytes each, padded to the largest) and one index_select
    through a map built with the plan put them into global edge-id order (SURVEY.md 8e "Scorer").

All tensor math goes through an `ops` namespace exactly like engine.run_stack: the product passes
gnnome_amd.ops (HIP); the CPU/gloo tests pass the checker backend to exercise this host logic.
"""
import os

import torch
import torch.distributed as dist

from . import engine
from . import ops as hip_ops


def force_collectives():
    """GNNOME_FORCE_COLLECTIVES=1: a single-rank process group still ISSUES every collective of the partitioned path (halo
    all_to_all both ways, BatchNorm statistics all_gather, BatchNorm-backward and gradient all_reduce, logits all_gather) instead
    of taking the world == 1 shortcuts - the way a one-GPU box runs this module's RCCL calls (tests/test_hip_partition.py).
    A one-rank collective is the identity, so the results must equal the plain path bit for bit."""
    return os.environ.get("GNNOME_FORCE_COLLECTIVES", "0") == "1"


def _alone(world):
    """True when the world == 1 shortcuts apply."""
    return world == 1 and not force_collectives()


def split_by_incident_edges(src, dst, num_nodes, world):
    """Node-range boundaries [b_0=0, ..., b_world=N] balancing in-degree + out-degree per range."""
    deg = torch.bincount(dst.long(), minlength=num_nodes) + torch.bincount(src.long(), minlength=num_nodes)
    return split_by_degree(deg, num_nodes, world)


def split_by_degree(deg, num_nodes, world):
    """The same boundaries from the per-node incident-edge counts (from_slices all-reduces them over the ranks' slices)."""
    csum = torch.cumsum(deg + 1, 0)  # +1: isolated nodes still cost a node update
    total = int(csum[-1]) if num_nodes > 0 else 0
    bounds = [0]
    for p in range(1, world):
        target = total * p // world
        b = int(torch.searchsorted(csum, torch.tensor(target, device=csum.device), right=False))
        bounds.append(max(bounds[-1], min(b, num_nodes)))
    bounds.append(num_nodes)
    return bounds


def partition_census(src, dst, num_nodes, world, bounds=None):
    """What PartitionedGraph.from_global would give every rank of a `world`-way destination-range partition, computed in one
    process without a process group (planning: which world size a graph wants; tools/partition_stats.py):
    per rank the owned nodes, local edges (incident to an owned node), owned in-edges (the ones it scores), halo rows it
    receives and rows it sends per layer.  Pure index arithmetic over the edge list, O(E) per rank."""
    src, dst = torch.as_tensor(src).long(), torch.as_tensor(dst).long()
    bounds = split_by_incident_edges(src, dst, num_nodes, world) if bounds is None else list(bounds)
    bt = torch.tensor(bounds[1:-1], device=src.device, dtype=torch.long)
    rs, rd = torch.bucketize(src, bt, right=True), torch.bucketize(dst, bt, right=True)   # owner rank of each endpoint
    cut = rs != rd
    ranks = []
    owned_in = torch.bincount(rd, minlength=world)
    local = owned_in + torch.bincount(rs[cut], minlength=world)
    # halo rows of rank p = distinct far endpoints of its cut edges: (p, node) pairs, counted once
    pair = torch.cat([rd[cut] * num_nodes + src[cut], rs[cut] * num_nodes + dst[cut]])
    pair = torch.unique(pair)
    holder, node = pair // num_nodes, pair % num_nodes
    owner = torch.bucketize(node, bt, right=True)
    recv = torch.bincount(holder, minlength=world)
    send = torch.bincount(owner, minlength=world)
    link = torch.bincount(owner * world + holder, minlength=world * world).view(world, world)   # rows owner -> holder
    for p in range(world):
        ranks.append({"rank": p, "owned_nodes": bounds[p + 1] - bounds[p], "local_edges": int(local[p]), "owned_in_edges": int(owned_in[p]),
                      "halo_rows": int(recv[p]), "rows_sent_per_layer": int(send[p]), "largest_link_rows": int(link[p].max())})
    e = int(src.numel())
    return {"world": world, "num_nodes": num_nodes, "num_edges": e, "bounds": bounds, "cut_edges": int(cut.sum()),
            "cut_fraction": float(cut.sum()) / max(e, 1), "edge_replication": float(local.sum()) / max(e, 1),
            "imbalance_local_edges": float(local.max()) * world / max(float(local.sum()), 1.0), "ranks": ranks}


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
        self.node_perm = self.node_gather = None   # optional renumbering the ranges were cut in (caller's id -> internal id, and back)
        self.score_pad = 0          # rows of the padded logits piece every rank contributes to the all_gather
        self.score_index = None     # int64 [E_global]: where global edge id k sits in the flattened [world, score_pad] gather

    @classmethod
    def from_global(cls, src, dst, num_nodes, rank, world, device, ops=hip_ops, group=None, node_perm=None):
        """Build from the full edge list (every rank holds it in this harness, as inference.py holds the
        whole DGLGraph).  The plan is computed on `device` (masks, unique, searchsorted over the E-sized edge list: 0.9 s per
        rank on the host at 10M edges) + one all_to_all of halo requests.
        node_perm (int64[N], the same on every rank; gnnome_amd.node_order.locality_order): the RANGES are cut in the renumbered
        node order - for graphs whose ids do not follow the layout; local_node_rows still takes rows in the caller's numbering."""
        self = cls()
        src, dst = torch.as_tensor(src).to(device).long(), torch.as_tensor(dst).to(device).long()
        if node_perm is not None:
            self._set_node_perm(node_perm, num_nodes, device)
            src, dst = self.node_perm[src], self.node_perm[dst]
        self.rank, self.world, self.num_edges_global = rank, world, int(src.numel())
        self.bounds = split_by_incident_edges(src, dst, num_nodes, world)
        lo, hi = self.bounds[rank], self.bounds[rank + 1]
        keep = ((dst >= lo) & (dst < hi)) | ((src >= lo) & (src < hi))
        self.edge_gid = torch.nonzero(keep).squeeze(1)
        return self._build(src[keep], dst[keep], device, ops, group)

    @classmethod
    def from_slices(cls, src_slice, dst_slice, num_nodes, rank, world, device, ops=hip_ops, group=None, node_perm=None):
        """Build from a rank's SLICE of the edge list: rank r holds the edges with global ids [off_r, off_r + m_r), the slices in
        rank order making up the whole list (a reader that splits its input file G ways; no rank ever holds all E endpoints).
        Degrees are all-reduced ([N] int64) for the range boundaries, then every edge travels once to the owner of its
        destination and, when that is another rank, once to the owner of its source (one all_to_all of (src, dst, id) triples).
        The ids a rank receives arrive ascending - peer blocks in rank order, each ascending - so the local edge order, and with
        it every later array, equals from_global's.  The same route moves edge features: shuffle_edge_rows(e_slice)."""
        self = cls()
        src = torch.as_tensor(src_slice).to(device).long()
        dst = torch.as_tensor(dst_slice).to(device).long()
        if node_perm is not None:   # (as from_global: ranges over renumbered nodes; the permutation itself needs the whole graph once)
            self._set_node_perm(node_perm, num_nodes, device)
            src, dst = self.node_perm[src], self.node_perm[dst]
        self.rank, self.world = rank, world
        alone = _alone(world)
        m = torch.tensor([src.numel()], dtype=torch.int64, device=device)
        counts = m.view(1, 1) if alone else all_gather_rows(m, world, group)
        counts = counts.view(-1).tolist()
        self.num_edges_global = int(sum(counts))
        offset = int(sum(counts[:rank]))
        deg = torch.stack([torch.bincount(dst, minlength=num_nodes), torch.bincount(src, minlength=num_nodes)])   # in | out, this slice
        if not alone:
            all_reduce_sum(deg, group)
        self.global_degrees = deg        # [2, N] int64 of the WHOLE graph: what inference.py:416-420 z-scores into x
        self.bounds = split_by_degree(deg[0] + deg[1], num_nodes, world)
        bt = torch.tensor(self.bounds[1:-1], device=device, dtype=torch.long)
        owner_d, owner_s = torch.bucketize(dst, bt, right=True), torch.bucketize(src, bt, right=True)
        gid = torch.arange(offset, offset + src.numel(), device=device)
        second = owner_s != owner_d
        to_rank = torch.cat([owner_d, owner_s[second]])
        which = torch.cat([torch.arange(src.numel(), device=device), torch.nonzero(second).squeeze(1)])
        order = torch.argsort(to_rank * (self.num_edges_global + 1) + gid[which])     # by receiving rank, then by global id
        self._slice_rows = which[order]                                                # slice row of every record sent
        self._slice_send = torch.bincount(to_rank, minlength=world).tolist()
        sc = torch.tensor(self._slice_send, dtype=torch.int64, device=device)
        rc = torch.empty(world, dtype=torch.int64, device=device)
        if alone:
            rc.copy_(sc)
        else:
            all_to_all_rows(rc, sc, None, None, group)
        self._slice_recv = rc.tolist()
        triples = torch.stack([src, dst, gid], 1)[self._slice_rows].contiguous()
        got = torch.empty((sum(self._slice_recv), 3), dtype=torch.int64, device=device)
        if alone:
            got.copy_(triples)
        else:
            all_to_all_rows(got, triples, self._slice_recv, self._slice_send, group)
        self.edge_gid = got[:, 2].contiguous()
        assert self.edge_gid.numel() < 2 or bool((self.edge_gid[1:] > self.edge_gid[:-1]).all()), "edge ids must arrive ascending"
        return self._build(got[:, 0].contiguous(), got[:, 1].contiguous(), device, ops, group)

    def shuffle_edge_rows(self, rows_slice):
        """[m_r, F] rows of this rank's edge slice (features, labels) -> the rows of its LOCAL edges, in local edge-id order (the
        route from_slices sent the endpoints along)."""
        if not hasattr(self, "_slice_rows"):
            raise RuntimeError("shuffle_edge_rows needs a plan built by from_slices")
        rows_slice = torch.as_tensor(rows_slice)
        flat = rows_slice.reshape(rows_slice.shape[0], -1).to(self._slice_rows.device)
        out = torch.empty((sum(self._slice_recv), flat.shape[1]), dtype=flat.dtype, device=flat.device)
        packed = flat[self._slice_rows].contiguous()
        if _alone(self.world):
            out.copy_(packed)
        else:
            all_to_all_rows(out, packed, self._slice_recv, self._slice_send, self._group)
        return out.reshape((out.shape[0],) + tuple(rows_slice.shape[1:]))

    def local_degree_features(self):
        """x rows of this rank's owned + halo nodes, [zscore(in_degree) | zscore(out_degree)] of the WHOLE graph (inference.py:416-420;
        mean and unbiased std in fp64, applied in fp32 like gnnome_degree_features_f32), from the degrees from_slices all-reduced."""
        if not hasattr(self, "global_degrees"):
            raise RuntimeError("local_degree_features needs a plan built by from_slices")
        d = self.global_degrees.double()
        mean, std = d.mean(1, keepdim=True), d.std(1, keepdim=True)
        local = self.global_degrees[:, self.node_gid].float()
        return ((local - mean.float()) / std.float()).t().contiguous()

    def _build(self, ls, ld, device, ops, group):
        """Everything after the rank knows its local edges (ls, ld: global endpoints, ascending global edge id)."""
        rank, world = self.rank, self.world
        self._group = group
        lo, hi = self.bounds[rank], self.bounds[rank + 1]
        self.lo, self.hi, self.n_own = lo, hi, hi - lo
        own_d = (ld >= lo) & (ld < hi)
        ends = torch.cat([ls, ld])
        halo = torch.unique(ends[(ends < lo) | (ends >= hi)])  # ascending => grouped by owner rank
        self.node_gid = torch.cat([torch.arange(lo, hi, device=halo.device), halo])
        self.n_local = int(self.node_gid.numel())

        def to_local(g):
            own = (g >= lo) & (g < hi)
            pos = torch.searchsorted(halo, g.clamp(min=0)) if halo.numel() else torch.zeros_like(g)
            return torch.where(own, g - lo, self.n_own + pos)

        l_src, l_dst = to_local(ls).int(), to_local(ld).int()
        self.views = ops.GraphViews(l_src, l_dst, self.n_local)
        in_ptr = self.views.in_ptr
        self.n_score = int(in_ptr[self.n_own]) if self.n_own > 0 else 0
        assert self.n_score == int(own_d.sum())
        srt_eid = self.views.srt_eid[:self.n_score].long()
        self.srt_geid = self.edge_gid[srt_eid].int()

        # halo-exchange plan: tell each owner which of its rows I hold as halo
        bounds_t = torch.tensor(self.bounds, device=halo.device)
        owner = torch.searchsorted(bounds_t, halo, right=True) - 1
        self.recv_counts = torch.bincount(owner, minlength=world).tolist() if halo.numel() else [0] * world
        # (the plan travels on `device` tensors: RCCL moves device memory only, gloo takes either)
        req_counts = torch.tensor(self.recv_counts, dtype=torch.int64, device=device)
        got_counts = torch.empty(world, dtype=torch.int64, device=device)
        if not _alone(world):
            all_to_all_rows(got_counts, req_counts, None, None, group)
        else:
            got_counts.copy_(req_counts)
        self.send_counts = got_counts.tolist()
        wanted = torch.empty(sum(self.send_counts), dtype=torch.int64, device=device)
        if not _alone(world):
            all_to_all_rows(wanted, halo.contiguous(), self.send_counts, self.recv_counts, group)
        assert wanted.numel() == 0 or (int(wanted.min()) >= lo and int(wanted.max()) < hi)
        self.send_idx = (wanted - lo).int()
        self._plan_logits(device, group)
        return self

    def _plan_logits(self, device, group):
        """Assembly of the logits: every rank scores its owned in-edges (a prefix of its sorted positions) into one piece;
        the pieces are all-gathered (padded to the largest) and un-permuted with one index_select.  The map from global
        edge id to its slot in the gathered buffer is exchanged here, once."""
        world = self.world
        if _alone(world):
            self.score_pad, self.score_index = self.n_score, None
            return
        counts = all_gather_rows(torch.tensor([self.n_score], dtype=torch.int64, device=device), world, group).view(-1).tolist()
        self.score_pad = pad = max(max(counts), 1)
        mine = torch.full((pad,), -1, dtype=torch.int64, device=device)
        mine[:self.n_score] = self.srt_geid.long()
        every = all_gather_rows(mine, world, group).view(-1)                 # [world * pad] global edge ids, -1 = padding
        slots = torch.nonzero(every >= 0).squeeze(1)
        index = torch.full((self.num_edges_global,), -1, dtype=torch.int64, device=device)
        index[every[slots]] = slots
        assert sum(counts) == self.num_edges_global and int(index.min()) >= 0, "every edge must be scored exactly once"
        self.score_index = index

    def assemble_logits(self, piece, group=None):
        """piece[score_pad] (this rank's owned in-edges in sorted order, tail unused) -> logits[E_global] in edge-id order,
        complete on every rank."""
        if _alone(self.world):
            raise RuntimeError("single rank: the scorer writes edge-id order itself")
        gathered = all_gather_rows(piece, self.world, group).view(-1)
        return gathered.index_select(0, self.score_index)

    def _set_node_perm(self, node_perm, num_nodes, device):
        self.node_perm = torch.as_tensor(node_perm).to(device=device, dtype=torch.int64)
        if self.node_perm.numel() != num_nodes:
            raise ValueError(f"node_perm has {self.node_perm.numel()} entries for {num_nodes} nodes")
        self.node_gather = torch.empty(num_nodes, dtype=torch.int64, device=device)
        self.node_gather[self.node_perm] = torch.arange(num_nodes, device=device)

    def local_node_rows(self, x_global):
        """Rows of this rank's owned + halo nodes out of a table in the CALLER's node numbering."""
        gid = self.node_gid if self.node_gather is None else self.node_gather[self.node_gid]
        return x_global[gid.to(x_global.device)]

    def local_edge_rows(self, e_global):
        return e_global[self.edge_gid.to(e_global.device)]


def _staged(t, group):
    """gloo moves host memory only: device tensors take a round trip through the host under it.  That is the
    two-processes-on-one-GPU test configuration; RCCL ("nccl") works on device memory directly."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_to_all_rows(out, inp, out_counts, in_counts, group=None, async_op=False):
    """out <- row blocks from every peer (out_counts rows each), inp's blocks (in_counts) go out."""
    if _staged(out, group):
        o, i = torch.empty(out.shape, dtype=out.dtype), inp.cpu()
        dist.all_to_all_single(o, i, output_split_sizes=out_counts, input_split_sizes=in_counts, group=group)
        out.copy_(o)
        return None
    return dist.all_to_all_single(out, inp, output_split_sizes=out_counts, input_split_sizes=in_counts, group=group, async_op=async_op)


def all_reduce_sum(t, group=None):
    if _staged(t, group):
        c = t.cpu()
        dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group)
        t.copy_(c)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def all_gather_rows(t, world, group=None):
    """[world, *t.shape]: every rank's t."""
    out = torch.empty(world * t.numel(), dtype=t.dtype, device="cpu" if _staged(t, group) else t.device)
    dist.all_gather_into_tensor(out, (t.cpu() if _staged(t, group) else t).reshape(-1).contiguous(), group=group)
    return out.to(t.device).view((world,) + tuple(t.shape))


class HaloExchange:
    """h[n_own:] <- owners' rows, via one packed all_to_all_single (RCCL on GPUs, gloo in the CPU tests)."""

    def __init__(self, part, ops, group=None):
        self.part, self.ops, self.group, self.work, self._keep, self._bwd = part, ops, group, None, None, None

    def start(self, h):
        p = self.part
        if _alone(p.world):
            return
        packed = self.ops.gather_rows(h, p.send_idx)
        self._keep = packed
        self.work = all_to_all_rows(h[p.n_own:], packed, p.recv_counts, p.send_counts, self.group, async_op=True)

    def finish(self):
        if self.work is not None:
            self.work.wait()
        self.work, self._keep = None, None

    def transpose(self, dh):
        """Backward of start/finish: the halo rows of dh return to their owners, which add them to their own rows
        (one peer's block at a time: within a block the rows are distinct, so the sum order is fixed), then
        dh[n_own:] = 0."""
        self.transpose_start(dh)
        self.transpose_finish(dh)

    def transpose_start(self, dh):
        """First half of `transpose`: the halo rows leave (RCCL: asynchronously, on the collective's stream).  Nothing may WRITE dh until
        transpose_finish; kernels that do not touch dh - the layer's weight gradients - run underneath."""
        p = self.part
        if _alone(p.world):
            return
        got = torch.empty((int(p.send_idx.numel()), dh.shape[1]), dtype=torch.float32, device=dh.device)
        sent = dh[p.n_own:].contiguous()   # (a view: dh is contiguous)
        work = all_to_all_rows(got, sent, p.send_counts, p.recv_counts, self.group, async_op=True)
        self._bwd = (work, got, sent)

    def transpose_finish(self, dh):
        p = self.part
        if _alone(p.world) or getattr(self, "_bwd", None) is None:
            return
        work, got, _ = self._bwd
        self._bwd = None
        if work is not None:
            work.wait()
        at = 0
        for cnt in p.send_counts:
            if cnt:
                self.ops.scatter_add_rows(got[at:at + cnt], p.send_idx[at:at + cnt], dh)
            at += cnt
        dh[p.n_own:].zero_()


class PartitionShard:
    """One rank's rows of a training step over a destination-range partition (see gnnome_amd.train.WholeGraph for
    the interface; SURVEY.md 8e "Training additions"):
      * BatchNorm batch statistics: every rank reduces its OWNED rows (each edge / node of the graph exactly once),
        the per-rank (count, mean, M2) are all-gathered and merged with the parallel-variance formula;
      * backward: per-channel BatchNorm sums and, at the end, all parameter gradients are all-reduced (one flat
        buffer); halo rows of dh are scatter-added back to their owners once per layer."""

    def __init__(self, part, ops=hip_ops, group=None):
        if part.views.transposed:
            raise NotImplementedError("partitioned training runs on the forward orientation of the graph")
        if part.n_own == 0 or part.n_score == 0:
            raise NotImplementedError("a rank without owned nodes or owned in-edges (graph too small for this world size)")
        self.part, self.ops, self.group, self.world = part, ops, group, part.world
        self.alone = _alone(part.world)     # False under GNNOME_FORCE_COLLECTIVES=1: one rank, every collective issued
        self.views = part.views
        self.n_own, self.n_local = part.n_own, part.n_local
        self.e_own, self.e_local = part.n_score, part.views.num_edges
        self.n_global, self.e_global = part.bounds[-1], part.num_edges_global
        self.score_views = _ScoreViews(part.views, part.srt_geid)
        self._xchg = HaloExchange(part, ops, group)

    def halo_start(self, h):
        self._xchg.start(h)

    def halo_finish(self):
        self._xchg.finish()

    def halo_bwd(self, dh):
        self._xchg.transpose(dh)

    def halo_bwd_start(self, dh):
        self._xchg.transpose_start(dh)

    def halo_bwd_finish(self, dh):
        self._xchg.transpose_finish(dh)

    def combine_stats(self, mean, var, rows):
        if self.alone:
            return mean, var
        H = mean.numel()
        mine = torch.cat([torch.full((1,), float(rows), dtype=torch.float64, device=mean.device), mean.double(), var.double() * rows])
        every = all_gather_rows(mine, self.world, self.group)           # [world, 1+2H], identical on every rank
        n, mu, m2 = every[:, :1], every[:, 1:1 + H], every[:, 1 + H:]
        total = n.sum()
        mean_all = (n * mu).sum(0) / total
        m2_all = (m2 + n * (mu - mean_all) ** 2).sum(0)
        return mean_all.float().contiguous(), (m2_all / total).float().contiguous()

    def sum_ranks(self, tensors):
        if self.alone:
            return tensors
        flat = torch.cat([t.reshape(-1) for t in tensors])
        all_reduce_sum(flat, self.group)
        out, at = [], 0
        for t in tensors:
            out.append(flat[at:at + t.numel()].view(t.shape))
            at += t.numel()
        return out

    def finish_logits(self, piece):
        """world > 1: this rank's piece (owned in-edges, sorted order) -> the complete logits in edge-id order."""
        return piece if self.alone else self.part.assemble_logits(piece, self.group)


def _project(ops, lw, h, n_own, xchg):
    """P = h * Wcat^T + bcat; the owned rows are projected while the halo rows are still in flight."""
    P = torch.empty((h.shape[0], lw.Wcat.shape[0]), dtype=torch.float32, device=h.device)
    if n_own > 0:
        engine.project(ops, lw, h[:n_own], out=P[:n_own])
    xchg.finish()
    if h.shape[0] > n_own:
        engine.project(ops, lw, h[n_own:], out=P[n_own:])
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
    scratch = {}
    for li, lw in enumerate(prep.layers):
        if li > 0:
            xchg.start(h)
        P = _project(ops, lw, h, n_own, xchg)
        A1, A2, A3, B1, B2 = (P[:, i * H:(i + 1) * H] for i in range(5))
        e = engine.gate(ops, lw, views, e, B1, B2, (e_local, prep.enc_edge), scratch)
        h = ops.node_aggregate(e, A1, A2, A3, views, h, lw.norm, lw.scale_h, lw.shift_h, num_nodes_out=n_own)
    xchg.start(h)
    pw = prep.predictor
    hs = pw["hs"]
    xchg.finish()
    PQ = ops.linear(h, pw["W_nodes"], pw["b_nodes"], planes=pw.get("planes"))
    score_views = _ScoreViews(views, part.srt_geid)
    if _alone(part.world) or not reduce_result:
        # scatter straight to GLOBAL edge ids: a GraphViews-like shim whose srt_eid is the global map
        logits = (torch.zeros if part.world > 1 else torch.empty)(part.num_edges_global, dtype=torch.float32, device=h.device)
        if part.n_score > 0:
            ops.edge_score(e, PQ[:, :hs], PQ[:, hs:], score_views, pw["W1_e"], pw["W2"], pw["b2"], pw["W3"], pw["b3"], logits,
                           num_edges=part.n_score)
        return logits
    piece = torch.empty(part.score_pad, dtype=torch.float32, device=h.device)
    if part.n_score > 0:
        ops.edge_score(e, PQ[:, :hs], PQ[:, hs:], score_views, pw["W1_e"], pw["W2"], pw["b2"], pw["W3"], pw["b3"], piece,
                       num_edges=part.n_score, scatter_to_edge_id=False)
    return part.assemble_logits(piece, group)


class CapturedPartitionedForward:
    """run_partitioned with every stretch of kernels BETWEEN two collectives recorded once as a hipGraph: one forward is then
    L + 1 graph replays + L halo exchanges + the logits all-gather instead of ~55 Python-issued launches - what a rank needs
    when its share of the graph is small enough for the host to become the limit (eight ranks on the 10M-edge graph: ~6 ms
    of kernels per rank).  The collectives themselves stay OUTSIDE the graphs (RCCL / gloo calls between replays, ordered on
    the stream), so nothing depends on capturing a communicator; the price is that the exchange no longer overlaps the
    projection of the owned rows (the halo rows of a layout-ordered graph are a few per cent of a rank's rows).
    Same kernels, same order: the logits equal run_partitioned's bit for bit.  Inputs are the runner's static x / e."""

    def __init__(self, runner):
        if runner.prep is None:
            raise RuntimeError("CapturedPartitionedForward is for eval-mode inference")
        ops, prep, part = runner.ops, runner.prep, runner.part
        self.part, self.group, self.ops = part, runner.group, ops
        self.alone = alone = _alone(part.world)
        dev = runner.x.device
        with torch.no_grad():
            for _ in range(2):   # eager warm-up: weight preparation, allocator pools, the aggregation's hub scratch
                run_partitioned(ops, prep, part, runner.x, runner.e, runner.group)
        torch.cuda.synchronize(dev)
        views, n_own, H = part.views, part.n_own, prep.hidden
        pool = torch.cuda.graph_pool_handle()
        self.graphs, self.h, self.packed = [], [], []
        state = {"h": None, "e": None, "scratch": {}}

        def layer(li):
            lw = prep.layers[li]
            if li == 0:
                state["h"] = ops.encode(runner.x, *prep.enc_node)
                state["e"] = engine.encode_edges(ops, prep, views, runner.e)
            h = state["h"]
            P = engine.project(ops, lw, h)
            A1, A2, A3, B1, B2 = (P[:, i * H:(i + 1) * H] for i in range(5))
            state["e"] = engine.gate(ops, lw, views, state["e"], B1, B2, (runner.e, prep.enc_edge), state["scratch"])
            state["h"] = ops.node_aggregate(state["e"], A1, A2, A3, views, h, lw.norm, lw.scale_h, lw.shift_h, num_nodes_out=n_own)
            self.h.append(state["h"])
            self.packed.append(None if alone else ops.gather_rows(state["h"], part.send_idx))

        def score():
            pw = prep.predictor
            hs = pw["hs"]
            PQ = ops.linear(state["h"], pw["W_nodes"], pw["b_nodes"], planes=pw.get("planes"))
            sv = _ScoreViews(views, part.srt_geid)
            if alone:
                self.piece = torch.empty(part.num_edges_global, dtype=torch.float32, device=dev)
            else:
                self.piece = torch.empty(part.score_pad, dtype=torch.float32, device=dev)
            if part.n_score > 0:
                ops.edge_score(state["e"], PQ[:, :hs], PQ[:, hs:], sv, pw["W1_e"], pw["W2"], pw["b2"], pw["W3"], pw["b3"], self.piece,
                               num_edges=part.n_score, scatter_to_edge_id=alone)

        with torch.no_grad():
            for fn in [lambda li=li: layer(li) for li in range(len(prep.layers))] + [score]:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool):
                    fn()
                self.graphs.append(g)
        self._keep = state   # the recorded kernels read and write these tensors at their recorded addresses

    def __call__(self):
        p = self.part
        for i, g in enumerate(self.graphs):
            g.replay()
            if i < len(self.h) and not self.alone:
                all_to_all_rows(self.h[i][p.n_own:], self.packed[i], p.recv_counts, p.send_counts, self.group)
        return (self.piece if self.alone else p.assemble_logits(self.piece, self.group)).unsqueeze(1)


class _ScoreViews:
    """The scorer only reads srt_src / srt_dst / srt_eid."""

    def __init__(self, views, srt_geid):
        self.srt_src, self.srt_dst, self.srt_eid = views.srt_src, views.srt_dst, srt_geid


class PartitionedRunner:
    """Holds one rank's partition, weights and inputs on its GPU; forward() = one whole-graph scoring pass."""

    def __init__(self, model, part, x_global, e_global, device, ops=hip_ops, group=None, x_local=None, e_local=None):
        """x_global[N,F]: the node features of the whole graph (this rank keeps its owned + halo rows), or x_local: those rows
        already selected (part.local_node_rows), for callers that never hold the [N,F] table on this device; e_global[E,F] or
        e_local (part.local_edge_rows / part.shuffle_edge_rows) likewise."""
        self.ops, self.part, self.group, self.model = ops, part, group, model
        self.prep = None if model.training else engine.prepared_for(model, device, engine.Prepared)
        rows = part.local_node_rows(x_global) if x_local is None else x_local
        self.x = rows.to(device=device, dtype=torch.float32).contiguous()
        erows = part.local_edge_rows(e_global) if e_local is None else e_local
        self.e = erows.to(device=device, dtype=torch.float32).contiguous()

    def capture(self):
        """Record the forward as hipGraph segments between the collectives (CapturedPartitionedForward); forward() replays them."""
        self._captured = CapturedPartitionedForward(self)
        return self

    def forward(self):
        """eval mode: logits[E,1] of one whole-graph scoring pass."""
        if self.prep is None:
            raise RuntimeError("built from a model in train mode: use train_forward()")
        if getattr(self, "_captured", None) is not None:
            return self._captured()
        with torch.no_grad():
            return run_partitioned(self.ops, self.prep, self.part, self.x, self.e, self.group).unsqueeze(1)

    def train_forward(self):
        """train mode: logits[E,1] with autograd history; after loss.backward() every rank holds the whole graph's
        parameter gradients (already summed over ranks), so identical optimizers stay in step without a DDP wrapper."""
        from .train import train_forward_on
        return train_forward_on(self.model, PartitionShard(self.part, self.ops, self.group), self.x, self.e)
