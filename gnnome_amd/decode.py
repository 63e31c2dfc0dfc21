"""Greedy decode of edge scores into contig walks on the MI355X - the consumer of the model's output
(inference.py:54-361).

    walks = get_contigs_greedy(g, succs, preds, edges, len_threshold, nb_paths=50)        # inference.py:167, same call

The reference walks Python dicts and sets, one candidate at a time (`ThreadPoolExecutor(1)`): with scoring at
milliseconds this loop IS the wall time of inference.py.  Here every iteration of the outer loop is

    sample nb_paths start edges (torch, as the reference: Categorical over the sigmoid scores of the remaining edges)
    gnnome_greedy_walks          one wavefront per candidate: forward walk, then the reverse-complement walk
    pick the longest contig      (first maximum, inference.py:304-305)
    gnnome_mark_walk_visited     walk nodes, their mates and the jumped-over nodes become visited (:311-334)

and only the chosen walk crosses PCIe.  Integer results (walks, contig lengths) equal the reference's exactly when the
same start edges are drawn: the walk kernel ranks successors on the reference's own fp32 `log(sigmoid(score))` values
(computed on the CPU like inference.py:184 unless `logprobs_on_device`), exact ties resolved as torch.topk(k=1) resolves
them on the CPU (csrc/decode.hip replays libstdc++'s nth_element), successor lists in edge-id order like
graph_parser.py:31-37.
"""
import ctypes
import logging
import math
import os
import pickle

import torch

from . import _lib
from .ops import _on, _ptr, _stream


def _i32_checked(t, name):
    t = torch.as_tensor(t)
    if t.numel() and (int(t.max()) >= 2 ** 31 or int(t.min()) < -2 ** 31):
        raise OverflowError(f"{name} does not fit int32")
    return t.to(torch.int32)


class DecodeGraph:
    """Device-resident arrays of one assembly graph for decoding: successor lists in edge-id order, lengths, scores."""

    def __init__(self, src, dst, num_nodes, prefix_length, read_length, device=None):
        if num_nodes % 2:
            raise ValueError("nodes come in (read, reverse complement) pairs: num_nodes must be even (graph_parser.py:174-181)")
        device = device or torch.device("cuda", torch.cuda.current_device())
        src, dst = torch.as_tensor(src).to(device).long(), torch.as_tensor(dst).to(device).long()
        e = int(src.numel())
        self.device, self.num_nodes, self.num_edges = device, int(num_nodes), e
        self.src, self.dst = src.int(), dst.int()
        order = torch.sort(src, stable=True).indices                      # successors of u in edge-id order
        self.succ_eid = order.int()
        self.succ_nbr = dst[order].int().contiguous()
        self.succ_ptr = torch.searchsorted(src[order].contiguous(), torch.arange(num_nodes + 1, device=device)).int()
        # the reference's `edges` dict maps a (u, v) PAIR to one id - the last one inserted (graph_parser.py:77-80);
        # with parallel edges every slot of the pair therefore carries that id
        if e:
            key = src[order] * num_nodes + dst[order]
            uniq, inv = torch.unique(key, return_inverse=True)
            if uniq.numel() != e:
                last = torch.zeros(uniq.numel(), dtype=torch.long, device=device).scatter_reduce(0, inv, order, "amax", include_self=False)
                self.succ_eid = last[inv].int()
        self.prefix_length = _i32_checked(prefix_length, "prefix_length").to(device).contiguous()
        self.read_length = _i32_checked(read_length, "read_length").to(device).contiguous()
        if self.prefix_length.numel() != e or self.read_length.numel() != num_nodes:
            raise ValueError("prefix_length is per edge, read_length per node")
        self.logp = None

    def set_scores(self, scores, logprobs_on_device=False, use_labels=False):
        """logProbs of inference.py:178-184.  Default: torch's CPU kernels, the reference's own bits; on the device the
        last bit of log / sigmoid may differ, which can move a near-tie."""
        s = torch.as_tensor(scores).reshape(-1).float()
        if s.numel() != self.num_edges:
            raise ValueError("one score per edge")
        where = self.device if logprobs_on_device else torch.device("cpu")
        s = s.to(where)
        if use_labels:
            s = s.masked_fill(s < 1e-9, 1e-9)
            self.logp, self.prob = torch.log(s).to(self.device).contiguous(), s.to(self.device)
        else:
            self.logp, self.prob = torch.log(torch.sigmoid(s)).to(self.device).contiguous(), torch.sigmoid(s).to(self.device)
        return self


class CandidateWalks:
    """Result of one gnnome_greedy_walks launch (device tensors; `contig(c)` assembles one walk on the host)."""

    def __init__(self, walks_f, walks_b, len_f, len_b, sum_f, sum_b, contig_len, status, src, dst):
        self.walks_f, self.walks_b, self.len_f, self.len_b = walks_f, walks_b, len_f, len_b
        self.sum_f, self.sum_b, self.contig_len, self.status, self.src, self.dst = sum_f, sum_b, contig_len, status, src, dst

    def contig_device(self, c, lf=None, lb=None):
        lf = int(self.len_f[c]) if lf is None else lf
        lb = int(self.len_b[c]) if lb is None else lb
        return torch.cat([torch.flip(self.walks_b[c, :lb], [0]) ^ 1, self.walks_f[c, :lf]])   # inference.py:157, :254

    def contig(self, c):
        return self.contig_device(c).tolist()


def greedy_walks(dg, visited, cand_eid, capacity=None):
    """All candidates of one outer iteration (inference.py:236-300): forward and reverse-complement greedy walks from the
    edges `cand_eid` (edge ids), not entering `visited` (uint8[N] on the device)."""
    lib = _lib.load()
    dev = dg.device
    cand_eid = torch.as_tensor(cand_eid).to(dev).int().contiguous()
    p = int(cand_eid.numel())
    cs, cd = dg.src[cand_eid.long()].contiguous(), dg.dst[cand_eid.long()].contiguous()
    cap = int(capacity or (dg.num_nodes // 2 + 2))
    mk = lambda dt, *shape: torch.empty(shape, dtype=dt, device=dev)  # noqa: E731
    walks_f, walks_b = mk(torch.int32, p, cap), mk(torch.int32, p, cap)
    len_f, len_b, status = mk(torch.int32, p), mk(torch.int32, p), mk(torch.int32, p)
    sum_f, sum_b, contig_len = mk(torch.float32, p), mk(torch.float32, p), mk(torch.int64, p)
    need = ctypes.c_size_t(0)
    _lib.check(lib.gnnome_greedy_walks_workspace_bytes(dg.num_nodes, p, ctypes.byref(need)), "greedy_walks_workspace_bytes")
    ws = torch.empty(int(need.value), dtype=torch.uint8, device=dev)
    with _on(dev):
        _lib.check(lib.gnnome_greedy_walks(_ptr(dg.succ_ptr), _ptr(dg.succ_nbr), _ptr(dg.succ_eid), _ptr(dg.logp), _ptr(dg.prefix_length),
                                           _ptr(dg.read_length), _ptr(visited), dg.num_nodes, _ptr(cs), _ptr(cd), _ptr(cand_eid), p,
                                           _ptr(walks_f), _ptr(walks_b), cap, _ptr(len_f), _ptr(len_b), _ptr(sum_f), _ptr(sum_b),
                                           _ptr(contig_len), _ptr(status), _ptr(ws), ws.numel(), _stream(dev)), "greedy_walks")
        ws.record_stream(torch.cuda.current_stream(dev))
    return CandidateWalks(walks_f, walks_b, len_f, len_b, sum_f, sum_b, contig_len, status, cs, cd)


def mark_walk_visited(dg, visited, walk):
    lib = _lib.load()
    walk = torch.as_tensor(walk).to(dg.device).int().contiguous()
    with _on(dg.device):
        _lib.check(lib.gnnome_mark_walk_visited(_ptr(dg.succ_ptr), _ptr(dg.succ_nbr), _ptr(walk), int(walk.numel()), _ptr(visited),
                                                _stream(dg.device)), "mark_walk_visited")


def sample_edges(prob_edges, nb_paths):
    """inference.py:54-67, restated: Categorical over the normalised edge probabilities, one draw per path, from torch's
    CPU generator - the reference's own random stream (it materialises prob_edges nb_paths times: 4 * nb_paths * E bytes
    and ~0.3 s per call at E = 1M)."""
    if prob_edges.shape[0] > 2 ** 24:
        prob_edges = prob_edges[:2 ** 24]   # the reference's own cut (torch's Categorical limit)
    prob_edges = prob_edges.masked_fill(prob_edges < 1e-9, 1e-9)
    prob_edges = prob_edges / prob_edges.sum()
    return torch.distributions.categorical.Categorical(prob_edges.repeat(nb_paths, 1)).sample()


def sample_edges_device(prob_edges, nb_paths):
    """The same distribution drawn on the device (with replacement) - no nb_paths-fold copy of the probabilities, no trip through the
    host; a different random stream than the reference's.  Inverse-CDF in float64: torch.multinomial builds its CDF with a float32
    device cumsum whose last bits are not reproducible from run to run, and with 10^6 edges x 100 draws x hundreds of iterations a draw
    now and then landed on the other side of a boundary (round 5: tests/test_decode.py's same-seed-same-walks check failed once in five
    runs).  In float64 the scan's reordering noise is 1e-16 of the total: two runs from one seed draw the same edges."""
    if prob_edges.shape[0] > 2 ** 24:
        prob_edges = prob_edges[:2 ** 24]
    p = prob_edges.double().clamp_min(1e-9)
    cdf = torch.cumsum(p, 0)
    u = torch.rand(nb_paths, dtype=torch.float64, device=p.device) * cdf[-1]
    return torch.searchsorted(cdf, u, right=True).clamp_(max=p.shape[0] - 1)


REFERENCE_SAMPLER_LIMIT = 2 ** 22   # remaining edges x nb_paths up to which the default sampler is the reference's own


def decode_contigs(dg, len_threshold, nb_paths=50, sampler=None, checkpoint_dir=None, load_checkpoint=False, visited=None,
                   stats=None):
    """The outer loop of get_contigs_greedy (inference.py:193-359) on a DecodeGraph with scores set.  `sampler(prob, k)`
    -> k indices into the remaining edges.  sampler=None (this function's default, NOT get_contigs_greedy's, which always
    passes the reference's sampler): the reference's own draw (sample_edges: torch's CPU generator, the same random stream)
    while remaining edges x nb_paths <= REFERENCE_SAMPLER_LIMIT (2^22: ~42k edges at 100 paths), sample_edges_device beyond -
    a DIFFERENT random stream; the switch is logged once per call.  Returns the list of walks (lists of node ids)."""
    dev = dg.device
    visited = torch.zeros(dg.num_nodes, dtype=torch.uint8, device=dev) if visited is None else visited
    all_contigs, all_walks_len, all_contigs_len = [], [], []
    ckpt = os.path.join(checkpoint_dir, "checkpoint.pkl") if checkpoint_dir else None
    if load_checkpoint and ckpt and os.path.isfile(ckpt):
        with open(ckpt, "rb") as f:
            state = pickle.load(f)
        all_contigs, all_walks_len, all_contigs_len = state["walks"], state["all_walks_len"], state["all_contigs_len"]
        if state["visited"]:
            visited[torch.tensor(sorted(state["visited"]), device=dev)] = 1
    src_l, dst_l = dg.src.long(), dg.dst.long()
    switched = False
    while True:
        remaining = torch.nonzero((visited[src_l] == 0) & (visited[dst_l] == 0)).squeeze(1)   # get_subgraph, :39-51
        if remaining.numel() == 0:
            break
        prob = dg.prob[remaining]
        draw = sampler or (sample_edges if remaining.numel() * nb_paths <= REFERENCE_SAMPLER_LIMIT else sample_edges_device)
        if sampler is None and draw is sample_edges_device and not switched:
            switched = True
            logging.getLogger(__name__).info("decode_contigs: %d remaining edges x %d paths > %d: start edges drawn by the device "
                                             "sampler (not the reference's random stream)", remaining.numel(), nb_paths, REFERENCE_SAMPLER_LIMIT)
        idx = draw(prob.cpu() if draw is sample_edges else prob, nb_paths)
        cand = remaining[torch.as_tensor(idx).to(dev).long()]
        res = greedy_walks(dg, visited, cand)
        lens = res.contig_len.clone()
        lens[res.src == res.dst] = 0                       # :262, :282-287 a self-loop counts as an empty contig
        host = torch.stack([lens, res.len_f.long(), res.len_b.long(), res.status.long()]).cpu()
        if int(host[3].max()) & 2:
            raise RuntimeError("an edge of a backward walk has no reverse-complement mate (the reference's DGL edge lookup "
                               "would raise): the graph is not strand-symmetric")
        if int(host[3].max()) & 1:
            raise RuntimeError("a walk exceeded the walk buffer: pass a larger capacity")
        best_len, best = torch.max(host[0], 0)               # first maximum, like list.index(max(...)) at :304-305
        best = int((host[0] == best_len).nonzero()[0])
        walk = res.contig_device(best, int(host[1][best]), int(host[2][best]))
        if stats is not None:
            stats.append({"contig_len": int(best_len), "walk_len": int(walk.numel()), "candidates": int(cand.numel()),
                          "sumLogProb": float(res.sum_f[best] + res.sum_b[best])})
        if int(best_len) < len_threshold:
            break
        mark_walk_visited(dg, visited, walk)
        all_contigs.append(walk.tolist())
        all_walks_len.append(int(walk.numel()))
        all_contigs_len.append(int(best_len))
        if ckpt and len(all_contigs) % 10 == 0:              # :340-355
            state = {"walks": all_contigs, "visited": set(torch.nonzero(visited).squeeze(1).tolist()), "all_walks_len": all_walks_len,
                     "all_contigs_len": all_contigs_len}
            tmp = os.path.join(checkpoint_dir, "checkpoint_tmp.pkl")
            with open(tmp, "wb") as f:
                pickle.dump(state, f)
            os.rename(tmp, ckpt)
    return all_contigs


def get_contigs_greedy(g, succs, preds, edges, len_threshold, nb_paths=50, use_labels=False, checkpoint_dir=None,
                       load_checkpoint=False, sampler=None, logprobs_on_device=False):
    """inference.py:167-359 with the same signature.  `g`: anything with edges() -> (src, dst), num_nodes(),
    edata['score'] (or ['y'] with use_labels), edata['prefix_length'], ndata['read_length'].  `succs`, `preds`, `edges`
    (the pickled dicts of graph_parser.py:409-411) are accepted for signature compatibility and NOT read: they are
    functions of g.edges() in edge-id order (graph_parser.py:31-37, :55-58, :77-80), which is what the device arrays are
    built from.  `sampler(prob, k)`: default = the reference's draw (sample_edges), so integer results equal the reference's
    under the same seed at every graph size; pass sample_edges_device for one device-side multinomial per contig."""
    del succs, preds, edges
    if sampler is None:
        # the reference's own Categorical draw from torch's CPU generator at EVERY size: with the same torch.manual_seed the
        # start edges, and therefore the contigs, are the reference's.  (It copies the probabilities nb_paths times - 0.3 s
        # per contig at 1M edges; `sampler=sample_edges_device` is the opt-in fast path with a different random stream,
        # which pipeline.assemble uses by default.)
        sampler = sample_edges
    src, dst = g.edges()
    dg = DecodeGraph(src, dst, int(g.num_nodes()), g.edata["prefix_length"], g.ndata["read_length"])
    dg.set_scores(g.edata["y"] if use_labels else g.edata["score"], logprobs_on_device=logprobs_on_device, use_labels=use_labels)
    return decode_contigs(dg, len_threshold, nb_paths, sampler, checkpoint_dir, load_checkpoint)


def mean_log_probs(sum_log_prob, len_walk, len_contig):
    """The per-candidate figures the reference prints (inference.py:264-288)."""
    if len_walk > 2:
        m = sum_log_prob / (len_walk - 2)
    else:
        m = 0.0
    return m, (m / math.sqrt(len_contig) if len_contig > 0 else 0.0)
