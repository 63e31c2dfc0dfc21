"""Thin torch-tensor front end of the C ABI: pointers, sizes and the current HIP stream go in,
nothing else.  torch is used for device memory and stream ownership only."""
import ctypes

import torch

from . import _lib
from ._lib import NORM_AFFINE, NORM_LAYER  # noqa: F401


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None and t.numel() > 0 else None)


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


_SIDE_STREAMS = {}


def side_stream(device):
    """The second HIP stream of `device` (high priority: its workgroups are placed ahead of the queued ones of the main
    stream) that engine.aggregate_then_project runs the next node projection on, under the aggregation."""
    s = _SIDE_STREAMS.get(device.index)
    if s is None:
        s = _SIDE_STREAMS[device.index] = torch.cuda.Stream(device=device, priority=-1)
    return s


class _on:
    """Make `device` current for the duration of a call - a no-op (no torch device-guard round trip, ~10 us
    per kernel launch otherwise) in the usual case that it already is."""

    __slots__ = ("idx", "prev")

    def __init__(self, device):
        self.idx, self.prev = device.index, -1

    def __enter__(self):
        if self.idx is not None:
            cur = torch.cuda.current_device()
            if cur != self.idx:
                torch.cuda.set_device(self.idx)
                self.prev = cur

    def __exit__(self, *exc):
        if self.prev >= 0:
            torch.cuda.set_device(self.prev)


def _f32(t, name):
    if t.dtype != torch.float32 or not t.is_cuda:
        raise TypeError(f"{name}: expected a CUDA float32 tensor, got {t.dtype} on {t.device}")
    return t


def _act(t, name):
    """A contiguous CUDA activation tensor in either storage type of the training step: float32, or bfloat16 (the xe / dxe
    rows of `activation_storage="bf16"`, read and written by the *_x16 entry points) -> (tensor, is_bf16)."""
    if t.dtype not in (torch.float32, torch.bfloat16) or not t.is_cuda or not t.is_contiguous():
        raise TypeError(f"{name}: expected a contiguous CUDA float32 or bfloat16 tensor, got {t.dtype} on {t.device}")
    return t, t.dtype == torch.bfloat16


def _i32(t, name):
    if t.dtype != torch.int32 or not t.is_cuda or not t.is_contiguous():
        raise TypeError(f"{name}: expected a contiguous CUDA int32 tensor, got {t.dtype} on {t.device}")
    return t


def _rows(t, name):
    """2-D float32 with unit column stride -> (tensor, row stride in elements)."""
    _f32(t, name)
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise ValueError(f"{name}: expected a 2-D tensor with contiguous rows, got shape {tuple(t.shape)} strides {t.stride()}")
    return t, (t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1]))


class GraphViews:
    """In-edge / out-edge orderings of one edge list (see include/gnnome_hip.h, "graph views")."""

    __slots__ = ("num_nodes", "num_edges", "in_ptr", "srt_src", "srt_dst", "srt_eid", "out_ptr", "out_pos", "out_dst", "device",
                 "transposed", "_range", "_bad", "node_perm", "node_gather", "_stream", "_c_block", "_records", "__weakref__")

    def __init__(self, src, dst, num_nodes, validate="now", node_perm=None):
        """node_perm (int64[N], optional): the views are built over RENUMBERED nodes, node_perm[caller's id] = internal id
        (gnnome_amd/node_order.py); node rows cross the boundary in the caller's numbering - x is gathered through
        node_gather (internal id -> caller's id) by the node encoder, degree_features returns rows in the caller's order,
        logits are per edge id and never notice.
        validate = "now": endpoints are range-checked before anything is built (one host sync).  "lazy": no host sync -
        the extremes are reduced on the device, the endpoints are clamped into range so that no kernel can fault, and
        check_range() raises later (the model call does it once everything is enqueued); for callers that hand over a
        fresh graph every step (train.py:96, :336).  False: trusted input."""
        lib = _lib.load()
        _i32(src, "src"), _i32(dst, "dst")
        dev = src.device
        n, e = int(num_nodes), int(src.numel())
        if dst.numel() != e:
            raise ValueError("src and dst differ in length")
        self._range, self._bad = None, None
        if e > 0 and validate:
            ext = torch.stack([torch.minimum(src.min(), dst.min()), torch.maximum(src.max(), dst.max())])
            self._range = (ext, n)
            if validate == "lazy":
                src, dst = src.clamp(0, max(n - 1, 0)), dst.clamp(0, max(n - 1, 0))
            else:
                self.check_range()
        self.num_nodes, self.num_edges, self.device, self.transposed = n, e, dev, False
        self.node_perm = self.node_gather = None
        self._stream = self._c_block = self._records = None
        if node_perm is not None:
            node_perm = node_perm.to(device=dev, dtype=torch.int64)
            if node_perm.numel() != n:
                raise ValueError(f"node_perm has {node_perm.numel()} entries for {n} nodes")
            gather = torch.empty(n, dtype=torch.int64, device=dev)
            gather[node_perm] = torch.arange(n, device=dev)
            self.node_perm, self.node_gather = node_perm, gather.int()
            if validate != "lazy":   # ("lazy" already clamped the endpoints into range)
                src, dst = src.clamp(0, max(n - 1, 0)), dst.clamp(0, max(n - 1, 0))
            src, dst = node_perm[src.long()].int(), node_perm[dst.long()].int()
        mk = lambda k: torch.empty(k, dtype=torch.int32, device=dev)  # noqa: E731
        self.in_ptr, self.out_ptr = mk(n + 1), mk(n + 1)
        self.srt_src, self.srt_dst, self.srt_eid, self.out_pos, self.out_dst = mk(e), mk(e), mk(e), mk(e), mk(e)
        need = ctypes.c_size_t(0)
        with _on(dev):
            _lib.check(lib.gnnome_graph_views_workspace_bytes(n, e, ctypes.byref(need)), "graph_views_workspace_bytes")
            ws = torch.empty(max(int(need.value), 1), dtype=torch.uint8, device=dev)
            _lib.check(lib.gnnome_build_graph_views(_ptr(src), _ptr(dst), n, e, _ptr(self.in_ptr), _ptr(self.srt_src),
                                                    _ptr(self.srt_dst), _ptr(self.srt_eid), _ptr(self.out_ptr),
                                                    _ptr(self.out_pos), _ptr(self.out_dst), _ptr(ws), ws.numel(),
                                                    _stream(dev)),
                       "build_graph_views")
            ws.record_stream(torch.cuda.current_stream(dev))

    def check_range(self):
        """Raise IndexError if an endpoint lay outside [0, N) (the deferred half of validate="lazy"; one host sync, once)."""
        if self._bad is not None:      # sticky: views built from a clamped (= different) edge list never become usable
            raise IndexError(self._bad)
        if self._range is not None:
            ext, n = self._range
            lo, hi = (int(v) for v in ext.tolist())
            if lo < 0 or hi >= n:
                self._bad = f"edge endpoint out of range [0,{n}): min {lo}, max {hi}"
                raise IndexError(self._bad)
            self._range = None

    def node_records(self):
        """records[N, 64] int32 (gnnome_build_node_records): per node its list pointers and the first 20 + 20 neighbour ids / out-edge positions in ONE
        256-byte row, what the aggregation's record form reads instead of three dependent trips (see NODE_RECORDS).  Built on first use, one small launch;
        shared with the reversed views (a function of the arrays, not of the roles the host gives them)."""
        if self._records is None:
            holder = [torch.empty((max(self.num_nodes, 1), 64), dtype=torch.int32, device=self.device)]
            with _on(self.device):
                _lib.check(_lib.load().gnnome_build_node_records(_ptr(self.in_ptr), _ptr(self.srt_src), _ptr(self.out_ptr), _ptr(self.out_pos),
                                                                 _ptr(self.out_dst), self.num_nodes, _ptr(holder[0]), _stream(self.device)),
                           "build_node_records")
            self._records = holder   # (a list: reversed() copies the slot, both views then share the one tensor)
        return self._records[0]

    def stream_schedule(self):
        """The streaming aggregation's schedule of this graph (StreamSchedule), built on first use - one host sync, once per graph;
        shared with the reversed views (the schedule is a function of the arrays, not of the roles the host gives them)."""
        if self._stream is None:
            self._stream = StreamSchedule(self)
        return self._stream

    def reversed(self):
        """Views of dgl.reverse(g, copy_ndata=True, copy_edata=True) - endpoints swapped, edge ids and edge
        storage order kept (gated_gcn_full.py:99, train.py:165).  Free: the same arrays, with the roles of
        the contiguous (by-dst) and permuted (by-src) runs exchanged by the caller (`transposed`)."""
        r = object.__new__(GraphViews)
        for k in GraphViews.__slots__:
            if k != "__weakref__":
                setattr(r, k, getattr(self, k))
        r.transposed = not self.transposed
        r._c_block = None
        return r


# The streaming aggregation (gnnome_node_aggregate_stream_f32): False = the one-wave-per-node kernel gnnome_node_aggregate_f32 everywhere (the
# default: the streaming form is a MEASURED NEGATIVE on the MI355X - 0.29 against 0.20 ms per launch at configs[1], profiles/r05_stream_aggregate_*,
# NOTES.md round 5); "auto" (or GNNOME_STREAM_AGGREGATE=auto) = the streaming kernel whenever the graph's schedule says its numbering has the
# locality the LDS slot window needs.
import os as _os
STREAM_AGGREGATE = "auto" if _os.environ.get("GNNOME_STREAM_AGGREGATE", "off").lower() in ("1", "on", "auto", "true") else False
STREAM_MIN_EDGES = 400_000      # below this the chunks get so short that their boundaries make a quarter of the rows far
STREAM_MAX_FAR = 0.20           # far rows / E above which the graph is left to the gathering kernel (a uniform random graph: ~1.0)
STREAM_ROWS_PER_CHUNK = 2048
STREAM_ROWS_PER_STEP = 16
STREAM_SLOTS = 62


class StreamSchedule:
    """include/gnnome_hip.h, "streaming aggregation": the per-graph schedule + the facts the host decides with."""

    __slots__ = ("chunks", "chunk_node", "chunk_steps", "steps", "edge_meta", "node_pend", "pend_nodes", "counters", "num_pending",
                 "num_far", "num_overflow", "max_live", "max_steps", "total_steps", "far_fraction", "usable", "why")

    def __init__(self, views, chunks=None, slots=None):
        lib = _lib.load()
        dev, n, e = views.device, views.num_nodes, views.num_edges
        self.usable, self.why = False, ""
        if e == 0 or n == 0:
            self.why = "empty graph"
            return
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        self.chunks = int(chunks) if chunks else max(2 * cus, -(-e // STREAM_ROWS_PER_CHUNK))
        slots = STREAM_SLOTS if slots is None else int(slots)
        cap, scratch = ctypes.c_int64(0), ctypes.c_size_t(0)
        _lib.check(lib.gnnome_stream_schedule_sizes(n, e, STREAM_ROWS_PER_STEP, ctypes.byref(cap), ctypes.byref(scratch)), "stream_schedule_sizes")
        mk = lambda k: torch.empty(k, dtype=torch.int32, device=dev)  # noqa: E731
        self.chunk_node, self.chunk_steps = mk(self.chunks + 1), mk(self.chunks)
        self.steps = torch.empty((int(cap.value), 4), dtype=torch.int32, device=dev)
        self.edge_meta = torch.empty(e, dtype=torch.uint8, device=dev)
        self.node_pend, self.pend_nodes, self.counters = mk(n), mk(n), mk(8)
        ws = torch.empty(int(scratch.value), dtype=torch.uint8, device=dev)
        with _on(dev):
            _lib.check(lib.gnnome_build_stream_schedule(n, e, _ptr(views.in_ptr), _ptr(views.srt_src), _ptr(views.out_ptr), _ptr(views.out_pos),
                                                        _ptr(views.out_dst), self.chunks, STREAM_ROWS_PER_STEP, slots, _ptr(self.chunk_node),
                                                        _ptr(self.chunk_steps), _ptr(self.steps), _ptr(self.edge_meta), _ptr(self.node_pend),
                                                        _ptr(self.pend_nodes), _ptr(self.counters), _ptr(ws), ws.numel(), _stream(dev)),
                       "build_stream_schedule")
            ws.record_stream(torch.cuda.current_stream(dev))
        c = self.counters.tolist()   # the one host sync
        self.num_pending, self.num_far, self.num_overflow, self.max_live, self.max_steps = c[0], c[1], c[2], c[3], c[4]
        self.total_steps = int(self.chunk_steps.sum())
        self.far_fraction = self.num_far / e
        if self.far_fraction > STREAM_MAX_FAR:
            self.why = f"{self.far_fraction:.0%} of the rows are far: the numbering has no locality (or the graph is too small for its chunks)"
        elif self.max_steps > 4 * (self.total_steps // self.chunks + 16):
            self.why = f"a chunk of {self.max_steps} steps against {self.total_steps // self.chunks} on average: a hub"
        else:
            self.usable = True


def stream_schedule_for(views, e_rows, num_nodes_out, node_range, norm_kind):
    """The schedule the streaming aggregation would run this call with, or None: whole-graph inference updates with the affine norm
    on graphs whose schedule is usable (STREAM_AGGREGATE)."""
    if (not STREAM_AGGREGATE or norm_kind != NORM_AFFINE or node_range is not None or views.num_edges < STREAM_MIN_EDGES
            or e_rows != views.num_edges or (num_nodes_out is not None and num_nodes_out != views.num_nodes) or not hasattr(views, "stream_schedule")):
        return None
    sched = views.stream_schedule()
    return sched if sched.usable else None


edge_gate_out_of_place_at_256 = True   # engine.gate_update: the H = 256 streaming gate needs out != e


_TUNING = {}   # what set_tuning was last told, per key (the library keeps no getter)


def set_tuning(key, value):
    """Select a kernel variant for A/B measurements (see gnnome_set_tuning in include/gnnome_hip.h)."""
    _lib.check(_lib.load().gnnome_set_tuning(int(key), int(value)), "set_tuning")
    _TUNING[int(key)] = int(value)


FP16_MAX = 65504.0   # fp16x3's operand range (include/gnnome_hip.h, gnnome_linear_f32): beyond it an output row is NaN


class bf16x6_arithmetic:
    """`with ops.bf16x6_arithmetic():` - the forward's dense products as bf16x6 (fp32's range; round 3's arithmetic) instead of fp16x3 for
    the duration: gnnome_set_tuning(10, 1), restored on exit.  engine.model_forward re-runs a forward under it when fp16x3's range did
    not hold (an operand >= 65504 leaves the fp16x3 kernels as NaN rows, never as wrong finite values)."""

    def __enter__(self):
        self.prev = _TUNING.get(10, 0)
        if self.prev != 1:
            set_tuning(10, 1)
        return self

    def __exit__(self, *exc):
        if self.prev != 1:
            set_tuning(10, self.prev)


def encode(x, W1, b1, W2, b2, gather=None, rows=None):
    lib = _lib.load()
    x, _ = _rows(x.contiguous(), "encode.in")
    hidden, hidden_ne = W2.shape[0], W1.shape[0]
    rows = int(x.shape[0] if rows is None else rows)
    out = torch.empty((rows, hidden), dtype=torch.float32, device=x.device)
    with _on(x.device):
        _lib.check(lib.gnnome_encode_f32(_ptr(x), rows, x.shape[1], _ptr(gather), _ptr(W1), _ptr(b1), hidden_ne, _ptr(W2),
                                         _ptr(b2), hidden, _ptr(out), _stream(x.device)), "encode_f32")
    return out


def planes_supported(K, Nout):
    """Shapes gnnome_linear_planes_f32 is built for (the node projections: K = H in {64, 128, 256}, Nout = 5H or 2 hs)."""
    return K in (64, 128, 256) and Nout % (32 if K == 256 else 64) == 0 and 0 < Nout <= 1536


def weight_planes(W):
    """The two fp16 planes of W[Nout, K] in MFMA fragment order (gnnome_weight_planes_f16): what `linear(..., planes=)` multiplies by.
    Made once per weight matrix - engine.Prepared keeps them beside Wcat; `linear` makes them per call when it is not given any."""
    lib = _lib.load()
    W, ldw = _rows(W, "weight_planes.W")
    Nout, K = W.shape
    planes = torch.empty(Nout * K * 2, dtype=torch.float16, device=W.device)
    with _on(W.device):
        _lib.check(lib.gnnome_weight_planes_f16(_ptr(W), ldw, Nout, K, _ptr(planes), _stream(W.device)), "weight_planes_f16")
    return planes


def _planes_route(A, lda, W, ldw, out, ldc, accumulate, given):
    """Whether this product runs on gnnome_linear_planes_f32: the library's own rule (gnnome_linear_planes_route - shapes, tuning switches,
    GNNOME_PLANES_LINEAR=0; without planes from the caller exactly the shapes round 4 / 5 sent to the fp16x3 edge-tile kernels, so a narrower
    product keeps its bf16x6 kernel unless the caller hands in planes) and, here, what the kernel asks of its pointers."""
    return (not accumulate and lda % 4 == 0 and ldw % 4 == 0 and ldc % 4 == 0 and A.data_ptr() % 16 == 0 and W.data_ptr() % 16 == 0
            and out.data_ptr() % 16 == 0 and A.data_ptr() != out.data_ptr()
            and _lib.load().gnnome_linear_planes_route(A.shape[0], A.shape[1], W.shape[0], 1 if given else 0) == 1)


def linear(A, W, bias, out=None, accumulate=False, planes=None):
    """out[M,Nout] = A @ W.T + bias on the matrix cores, fp32-faithful (accumulate: out += ...).  A, W, out may be row-strided.
    The node projections' shapes (planes_supported) run on gnnome_linear_planes_f32 from W's fp16x3 planes - `planes` if the
    caller keeps them (weight_planes(W); they must belong to W), otherwise made here."""
    lib = _lib.load()
    A, lda = _rows(A, "linear.A")
    W, ldw = _rows(W, "linear.W")
    M, K = A.shape
    Nout = W.shape[0]
    if out is None:
        out = torch.empty((M, Nout), dtype=torch.float32, device=A.device)
    out, ldc = _rows(out, "linear.out")
    if _planes_route(A, lda, W, ldw, out, ldc, accumulate, planes is not None):
        if planes is None:
            planes = weight_planes(W)
        with _on(A.device):
            _lib.check(lib.gnnome_linear_planes_f32(_ptr(A), M, K, lda, _ptr(planes), _ptr(bias), Nout, _ptr(out), ldc, _stream(A.device)),
                       "linear_planes_f32")
        return out
    with _on(A.device):
        fn = lib.gnnome_linear_acc_f32 if accumulate else lib.gnnome_linear_f32
        _lib.check(fn(_ptr(A), M, K, lda, _ptr(W), ldw, _ptr(bias), Nout, _ptr(out), ldc, _stream(A.device)), "linear_f32")
    return out


def _dptr(t):
    return t.data_ptr() if t is not None and t.numel() > 0 else None


class ModelBlock:
    """A model's prepared parameters as gnnome_model_forward_f32 takes them (gnnome_model_params + its host array of gnnome_layer_params).
    Holds references to every tensor it points at."""

    def __init__(self, prep):
        pw = prep.predictor
        self.keep = [prep.enc_node, prep.enc_edge, prep.layers, pw]
        n = len(prep.layers)
        self.layers = (_lib.LayerParams * max(n, 1))()
        for lp, lw in zip(self.layers, prep.layers):
            for name in ("Wcat", "bcat", "W3", "b3", "scale_e", "shift_e", "scale_h", "shift_h"):
                t = getattr(lw, name)
                if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                    raise TypeError(f"model_forward: layer tensor {name} must be a contiguous CUDA float32 tensor")
                setattr(lp, name, _dptr(t))
            lp.Wcat_planes = _dptr(lw.planes)
            lp.norm_kind, lp.reference_order = int(lw.norm), int(bool(lw.ref))
        m = self.params = _lib.ModelParams()
        m.hidden, m.num_layers, m.score_hidden = prep.hidden, n, pw["hs"]
        (W1, b1, W2, b2), (V1, c1, V2, c2) = prep.enc_node, prep.enc_edge
        m.hidden_ne, m.node_features, m.edge_features = W1.shape[0], W1.shape[1], V1.shape[1]
        if V1.shape[0] != W1.shape[0]:
            raise ValueError("model_forward: the two encoders share hidden_ne (models/full_graph.py:13-16)")
        m.node_W1, m.node_b1, m.node_W2, m.node_b2 = (_dptr(t) for t in (W1, b1, W2, b2))
        m.edge_W1, m.edge_b1, m.edge_W2, m.edge_b2 = (_dptr(t) for t in (V1, c1, V2, c2))
        m.layers_host = ctypes.cast(self.layers, ctypes.POINTER(_lib.LayerParams))
        W1e = pw["W1_e"]
        if W1e.stride(1) != 1:
            raise ValueError("model_forward: predictor W1's edge block must have contiguous rows")
        m.W_nodes, m.W_nodes_planes, m.b_nodes = _dptr(pw["W_nodes"]), _dptr(pw.get("planes")), _dptr(pw["b_nodes"])
        m.W1e, m.ld_w1e = _dptr(W1e), W1e.stride(0)
        m.W2, m.b2, m.W3, m.b3 = (_dptr(pw[k]) for k in ("W2", "b2", "W3", "b3"))


def _views_block(views):
    blk = getattr(views, "_c_block", None)
    if blk is None:
        blk = _lib.Views()
        blk.num_nodes, blk.num_edges = views.num_nodes, views.num_edges
        for name in ("in_ptr", "srt_src", "srt_dst", "srt_eid", "out_ptr", "out_pos", "out_dst"):
            setattr(blk, name, _dptr(getattr(views, name)))
        blk.node_gather = _dptr(getattr(views, "node_gather", None))
        blk.transposed = int(bool(views.transposed))
        try:
            views._c_block = blk
        except AttributeError:   # duck-typed views that do not take attributes: built per call
            pass
    return blk


_WS_BYTES = {}


# Where the forward's buffers lie in HBM (round 6).  The same build on the same box runs configs[1]'s forward at one of three levels - 4.06-4.11,
# 4.16-4.22 or 4.24-4.26 ms - and the level is a property of the BLOCKS the driver handed out: it does not move while a block is reused
# (tools/placement_probe.py "same"), it changes when the forward gets a fresh block ("hold"), it depends neither on the buffers' offsets inside a
# block nor on virtual addresses, blocks of >= 1 GiB all land on one level, and ONE block holding h, P, e and PQ comes out slow far more often than
# the same buffers as allocations of their own (4.24 against 4.10 ms on one box, `tools/ab_one_call.sh`; profiles/r06_placement_*.txt).  So the one-call
# forward takes its buffers one by one (gnnome_model_forward_buffers_f32); "block" = one workspace (gnnome_model_forward_f32), for A/B runs.
FORWARD_BUFFERS = _os.environ.get("GNNOME_FORWARD_BUFFERS", "separate")


# The aggregation's record form (round 6: csrc/node_aggregate.hip, REC): a node's pointers AND its first 20 + 20 neighbour ids from one 256-byte load.
# Same bits; 0.120 against 0.135 ms per launch at H = 64 (1M edges), 0.2019 against 0.2054 at H = 128, level at 256 (profiles/r06_aggregation_node_records.txt):
# on for the widths up to NODE_RECORDS_MAX_HIDDEN in whole-graph BatchNorm inference, 256 bytes per node of extra views.  GNNOME_NODE_RECORDS=0: off.
NODE_RECORDS_MAX_HIDDEN = int(_os.environ.get("GNNOME_NODE_RECORDS", "64"))


class node_records_for:
    """`with ops.node_records_for(views, hidden):` - the calling thread's whole-range BatchNorm aggregations read `views.node_records()` for the duration
    (gnnome_debug_node_records: a thread-local switch of the library), where the width qualifies; a no-op otherwise."""

    def __init__(self, views, hidden):
        self.on = (0 < hidden <= NODE_RECORDS_MAX_HIDDEN and isinstance(views, GraphViews) and views.num_nodes > 0 and views.num_edges > 0
                   and not torch.cuda.is_current_stream_capturing())
        self.views = views

    def __enter__(self):
        if self.on:
            _lib.check(_lib.load().gnnome_debug_node_records(_ptr(self.views.node_records())), "debug_node_records")
        return self

    def __exit__(self, *exc):
        if self.on:
            _lib.load().gnnome_debug_node_records(None)


def model_forward(block, views, x, e_raw, logits=None):
    """models/full_graph.py:22-30 as ONE library call (gnnome_model_forward_buffers_f32): x[N, node_features], e_raw[E, edge_features] contiguous CUDA
    float32 in the caller's numbering -> logits[E] in edge-id order.  `block` = ModelBlock(prepared parameters)."""
    lib = _lib.load()
    dev = x.device
    m = block.params
    n, e, H, hs = views.num_nodes, views.num_edges, m.hidden, m.score_hidden
    if logits is None:
        logits = torch.empty(e, dtype=torch.float32, device=dev)
    if FORWARD_BUFFERS == "block":
        key = (n, e, H, hs)
        need = _WS_BYTES.get(key)
        if need is None:
            nb = ctypes.c_size_t(0)
            _lib.check(lib.gnnome_model_forward_workspace_bytes(n, e, H, hs, ctypes.byref(nb)), "model_forward_workspace_bytes")
            need = _WS_BYTES[key] = nb.value
        ws = torch.empty(max(need, 256), dtype=torch.uint8, device=dev)   # (torch's blocks are 512-byte aligned)
        with _on(dev):
            _lib.check(lib.gnnome_model_forward_f32(ctypes.byref(m), ctypes.byref(_views_block(views)), _ptr(x), _ptr(e_raw), _ptr(logits), _ptr(ws), need,
                                                    _stream(dev)), "model_forward_f32")
        return logits
    shapes = {"h0": (n, H), "h1": (n, H), "P": (n, 5 * H), "PQ": (n, 2 * hs), "e0": (e, H)}
    if H == 256:
        shapes["e1"] = (e, H)
    vb = _views_block(views)

    def run(t):
        bufs = _lib.ForwardBuffers()
        bufs.h[0], bufs.h[1], bufs.P, bufs.PQ = t["h0"].data_ptr(), t["h1"].data_ptr(), t["P"].data_ptr(), t["PQ"].data_ptr()
        bufs.e[0], bufs.e[1] = t["e0"].data_ptr(), (t["e1"].data_ptr() if "e1" in t else None)
        _lib.check(lib.gnnome_model_forward_buffers_f32(ctypes.byref(m), ctypes.byref(vb), _ptr(x), _ptr(e_raw), _ptr(logits), ctypes.byref(bufs),
                                                        _stream(dev)), "model_forward_buffers_f32")

    with _on(dev):
        run(_placed_buffers(dev, shapes, run))
    return logits


# Placement of the forward's buffers (see FORWARD_BUFFERS above): a forward that is asked for again and again with the same shapes gets its buffers PLACED
# once - on its PLACEMENT_AFTER_USES-th use, for the edge buffer(s), then P, then the three [N,*] buffers, a few fresh allocations of that group are timed
# with the forward itself (everything else held fixed; same bits from every candidate) and the fastest is kept for that (device, stream, shapes); the rest
# goes back to torch's allocator.  Only where placement varies: every buffer below 1 GiB (larger blocks all land on one level), at least 64 MiB in all.
# A graph scored once (inference.py:440) never gets here; a captured forward (hipGraph) keeps the graph pool's buffers.  GNNOME_TUNE_PLACEMENT=0: off.
TUNE_PLACEMENT = _os.environ.get("GNNOME_TUNE_PLACEMENT", "1") != "0"
PLACEMENT_AFTER_USES = 3
PLACEMENT_CANDIDATES = 4
PLACEMENT_MIN_TOTAL, PLACEMENT_MAX_BUFFER = 64 << 20, 1 << 30
PLACEMENT_KEPT = 3          # (device, stream, shapes) entries that keep their buffers; the oldest placement is let go beyond that


class _Placed:
    __slots__ = ("uses", "tried", "bufs", "log")

    def __init__(self):
        self.uses, self.tried, self.bufs, self.log = 0, False, None, None


_PLACED = {}


def _placed_buffers(dev, shapes, run):
    """name -> tensor for one forward: the kept set of this (device, stream, shapes) if there is one, transient allocations otherwise (freed, stream-ordered,
    when the caller returns) - and on the PLACEMENT_AFTER_USES-th use of shapes whose placement varies, the set chosen as described above."""
    mk = lambda name: torch.empty((max(shapes[name][0], 1), shapes[name][1]), dtype=torch.float32, device=dev)  # noqa: E731
    sizes = [max(r, 1) * c * 4 for r, c in shapes.values()]
    if not TUNE_PLACEMENT or sum(sizes) < PLACEMENT_MIN_TOTAL or max(sizes) >= PLACEMENT_MAX_BUFFER or torch.cuda.is_current_stream_capturing():
        return {name: mk(name) for name in shapes}
    stream = torch.cuda.current_stream(dev)
    key = (dev.index, stream.cuda_stream) + tuple(sorted(shapes.items()))
    st = _PLACED.get(key)
    if st is None:
        st = _PLACED[key] = _Placed()
    if st.bufs is not None:
        return st.bufs
    st.uses += 1
    cur = {name: mk(name) for name in shapes}
    free = torch.cuda.mem_get_info(dev)[0]
    if st.tried or st.uses < PLACEMENT_AFTER_USES or free < 2 * (PLACEMENT_CANDIDATES + 1) * sum(sizes):
        return cur
    st.tried = True

    def time_of(t):
        run(t)
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record(stream)
        run(t)
        run(t)
        t1.record(stream)
        t1.synchronize()
        return t0.elapsed_time(t1) / 2

    for _ in range(3):
        run(cur)                                 # (the first forwards of a process run below the clock the later ones get)
    best, log, rejected = time_of(cur), [], []
    log.append(("start", best))
    for group in (tuple(k for k in ("e0", "e1") if k in shapes), ("P",), ("h0", "h1", "PQ")):
        for _ in range(PLACEMENT_CANDIDATES):
            cand = dict(cur)
            for name in group:
                cand[name] = mk(name)            # fresh: the blocks of `cur` and of every rejected candidate are still held
            t = time_of(cand)
            log.append((group, t))
            if t < best * 0.997:
                rejected.extend(cur[name] for name in group)
                cur, best = cand, t
            else:
                rejected.extend(cand[name] for name in group)
    st.bufs, st.log = cur, log
    del rejected, cand
    try:
        torch.cuda.empty_cache()                 # the candidates that lost go back to the driver, not into torch's cache (one device synchronisation, once)
    except RuntimeError:                         # (another thread is capturing a graph: the blocks stay in torch's cache)
        pass
    kept = [k for k, v in _PLACED.items() if v.bufs is not None]
    for k in kept[:-PLACEMENT_KEPT]:
        _PLACED[k].bufs = None                   # (a dict keeps insertion order: the oldest placements first)
    return cur


def edge_gate(e, B1h, B2h, views, W3, norm_kind, scale, shift, out=None, num_edges=None):
    lib = _lib.load()
    e, _ = _rows(e, "edge_gate.e")
    B1h, ldn = _rows(B1h, "edge_gate.B1h")
    B2h, ldn2 = _rows(B2h, "edge_gate.B2h")
    assert ldn == ldn2 and e.is_contiguous()
    W3, ldw = _rows(W3, "edge_gate.W3")
    out = e if out is None else out
    E = int(e.shape[0] if num_edges is None else num_edges)
    with _on(e.device):
        _lib.check(lib.gnnome_edge_gate_f32(_ptr(e), _ptr(out), E, e.shape[1], _ptr(B1h), _ptr(B2h), ldn,
                                            _ptr(views.srt_src), _ptr(views.srt_dst), _ptr(W3), ldw, norm_kind, _ptr(scale),
                                            _ptr(shift), _stream(e.device)), "edge_gate_f32")
    return out


def can_fuse_edge_encoder(e_raw, enc, hidden, norm_kind, B1h):
    """The layer-0 gate can produce the encoded edge tile itself (gnnome_edge_gate_encode_f32)."""
    widths = (64, 128, 256) if _TUNING.get(10, 0) == 0 and _TUNING.get(0, 0) == 0 else (64, 128)   # 256: round 4, the fp16x3 kernel's mode 5
    return (e_raw.dim() == 2 and e_raw.shape[1] == 2 and enc[0].shape == (16, 2) and hidden in widths
            and norm_kind == NORM_AFFINE and e_raw.shape[0] > 0 and B1h.stride(0) % 4 == 0 and B1h.data_ptr() % 16 == 0)


def edge_gate_encode(e_raw, enc, B1h, B2h, views, W3, scale, shift):
    """Layer 0: encoder + gate in one kernel; returns e'[E,H] in sorted order."""
    lib = _lib.load()
    e_raw = _dense(e_raw, "edge_gate_encode.e_raw")
    B1h, ldn = _rows(B1h, "edge_gate_encode.B1h")
    B2h, _ = _rows(B2h, "edge_gate_encode.B2h")
    W3, ldw = _rows(W3, "edge_gate_encode.W3")
    E, H = views.num_edges, W3.shape[0]
    out = torch.empty((E, H), dtype=torch.float32, device=e_raw.device)
    W1, b1, W2, b2 = enc
    with _on(e_raw.device):
        _lib.check(lib.gnnome_edge_gate_encode_f32(_ptr(e_raw), _ptr(views.srt_eid), _ptr(W1), _ptr(b1), _ptr(W2), _ptr(b2),
                                                   _ptr(out), E, H, _ptr(B1h), _ptr(B2h), ldn, _ptr(views.srt_src),
                                                   _ptr(views.srt_dst), _ptr(W3), ldw, _ptr(scale), _ptr(shift),
                                                   _stream(e_raw.device)), "edge_gate_encode_f32")
    return out


def linear_ref(A, W, bias, out=None):
    """out[M,Nout] = A @ W.T + bias in the REFERENCE'S ORDER of evaluation (k-ascending fma chain from zero, bias added
    afterwards: what torch's CPU nn.Linear computes, bit for bit).  K in {64,128,256}, Nout % 8 == 0 (K = 256: Nout % 32 == 0)."""
    lib = _lib.load()
    A, lda = _rows(A, "linear_ref.A")
    W, ldw = _rows(W, "linear_ref.W")
    M, K = A.shape
    Nout = W.shape[0]
    if out is None:
        out = torch.empty((M, Nout), dtype=torch.float32, device=A.device)
    out, ldc = _rows(out, "linear_ref.out")
    with _on(A.device):
        _lib.check(lib.gnnome_linear_ref_f32(_ptr(A), M, K, lda, _ptr(W), ldw, _ptr(bias), Nout, _ptr(out), ldc, _stream(A.device)),
                   "linear_ref_f32")
    return out


def reference_order_supported(hidden, norm_kind, B1h=None):
    """Shapes the reference-order kernels take (gnnome_linear_ref_f32 / gnnome_edge_gate_ref_f32)."""
    return hidden in (64, 128, 256) and norm_kind == NORM_AFFINE   # 256: round 4 (the matrix-core form, reference_order_mfma.hip)


def edge_gate_ref(e, B1h, B2h, views, W3, b3, scale, shift, raw_edges=None, num_edges=None):
    """The gate in the reference's order of evaluation (see gnnome_edge_gate_ref_f32); b3 separate, NOT folded into B2h.
    e[E,H] is updated in place and returned; with e = None and raw_edges = (e_raw, (W1, b1, W2, b2)) the edge encoder's
    output is computed on the fly (layer 0) and a fresh e'[E,H] is returned."""
    lib = _lib.load()
    B1h, ldn = _rows(B1h, "edge_gate_ref.B1h")
    B2h, ldn2 = _rows(B2h, "edge_gate_ref.B2h")
    W3, ldw = _rows(W3, "edge_gate_ref.W3")
    assert ldn == ldn2
    H = W3.shape[0]
    if e is None:
        e_raw, (W1, b1, W2, b2) = raw_edges
        e_raw = _dense(e_raw, "edge_gate_ref.e_raw")
        E = views.num_edges if num_edges is None else int(num_edges)
        out = torch.empty((views.num_edges, H), dtype=torch.float32, device=e_raw.device)
        enc = (_ptr(e_raw), _ptr(views.srt_eid), _ptr(W1), _ptr(b1), _ptr(W2), _ptr(b2))
        e_in = _ptr(None)
    else:
        e = _dense(e, "edge_gate_ref.e")
        E = int(e.shape[0] if num_edges is None else num_edges)
        out, e_in = e, _ptr(e)
        enc = (_ptr(None),) * 6
    with _on(out.device):
        _lib.check(lib.gnnome_edge_gate_ref_f32(e_in, _ptr(out), E, H, _ptr(B1h), _ptr(B2h), ldn, _ptr(views.srt_src),
                                                _ptr(views.srt_dst), _ptr(W3), ldw, _ptr(b3), _ptr(scale), _ptr(shift), *enc,
                                                _stream(out.device)), "edge_gate_ref_f32")
    return out


def node_aggregate(e, A1h, A2h, A3h, views, h_in, norm_kind, scale, shift, num_nodes_out=None, node_range=None, out=None):
    """node_range = (begin, end): only those rows of `out` (required then) are computed - gnnome_node_aggregate_range_f32;
    the ranges of one aggregation go in ascending order from node 0."""
    lib = _lib.load()
    A1h, ldn = _rows(A1h, "node_aggregate.A1h")
    A2h, l2 = _rows(A2h, "node_aggregate.A2h")
    A3h, l3 = _rows(A3h, "node_aggregate.A3h")
    h_in, ldh = _rows(h_in, "node_aggregate.h_in")
    assert ldn == l2 == l3
    hidden = h_in.shape[1]
    n_out = int(h_in.shape[0] if num_nodes_out is None else num_nodes_out)
    h_out = torch.empty((h_in.shape[0], hidden), dtype=torch.float32, device=h_in.device) if out is None else out
    assert h_out.is_contiguous() and h_out.shape == (h_in.shape[0], hidden) and h_out.dtype == torch.float32
    sched = None
    if STREAM_AGGREGATE and getattr(views, "num_nodes", None) == h_in.shape[0]:   # (opt-in; duck-typed views without the counts take the gathering kernel)
        sched = stream_schedule_for(views, e.shape[0], num_nodes_out, node_range, norm_kind)
    with _on(h_in.device):
        if sched is not None:
            pend = torch.empty((max(sched.num_pending, 1), 3, hidden), dtype=torch.float32, device=h_in.device)
            _lib.check(lib.gnnome_node_aggregate_stream_f32(_ptr(e), hidden, n_out, views.num_edges, _ptr(A1h), _ptr(A2h), _ptr(A3h), ldn, _ptr(views.in_ptr),
                                                            _ptr(views.srt_src), _ptr(views.out_ptr), _ptr(views.out_pos), _ptr(views.out_dst),
                                                            _ptr(h_in), ldh, _ptr(h_out), _ptr(scale), _ptr(shift), sched.chunks,
                                                            STREAM_ROWS_PER_STEP, STREAM_SLOTS, _ptr(sched.chunk_node), _ptr(sched.chunk_steps),
                                                            _ptr(sched.steps), _ptr(sched.edge_meta), _ptr(sched.node_pend), _ptr(sched.pend_nodes),
                                                            _ptr(sched.counters), sched.num_pending, _ptr(pend), _stream(h_in.device)),
                       "node_aggregate_stream_f32")
        elif node_range is None:
            _lib.check(lib.gnnome_node_aggregate_f32(_ptr(e), hidden, n_out, _ptr(A1h), _ptr(A2h), _ptr(A3h), ldn,
                                                     _ptr(views.in_ptr), _ptr(views.srt_src), _ptr(views.out_ptr),
                                                     _ptr(views.out_pos), _ptr(views.out_dst), _ptr(h_in), ldh, _ptr(h_out),
                                                     norm_kind, _ptr(scale), _ptr(shift), _stream(h_in.device)),
                       "node_aggregate_f32")
        else:
            assert out is not None, "node_aggregate(node_range=...) writes rows of a caller-owned `out`"
            _lib.check(lib.gnnome_node_aggregate_range_f32(_ptr(e), hidden, n_out, int(node_range[0]), int(node_range[1]), _ptr(A1h),
                                                           _ptr(A2h), _ptr(A3h), ldn, _ptr(views.in_ptr), _ptr(views.srt_src),
                                                           _ptr(views.out_ptr), _ptr(views.out_pos), _ptr(views.out_dst), _ptr(h_in),
                                                           ldh, _ptr(h_out), norm_kind, _ptr(scale), _ptr(shift),
                                                           _stream(h_in.device)), "node_aggregate_range_f32")
    return h_out


def edge_score(e, Ps, Qd, views, W1e, W2, b2, W3, b3, logits, num_edges=None, scatter_to_edge_id=True, z1_out=None):
    lib = _lib.load()
    e, _ = _rows(e, "edge_score.e")
    Ps, ldn = _rows(Ps, "edge_score.Ps")
    Qd, ldn2 = _rows(Qd, "edge_score.Qd")
    assert ldn == ldn2 and e.is_contiguous()
    W1e, ldw1 = _rows(W1e, "edge_score.W1e")
    E = int(e.shape[0] if num_edges is None else num_edges)
    eid = views.srt_eid if scatter_to_edge_id else None
    with _on(e.device):
        _lib.check(lib.gnnome_edge_score_f32(_ptr(e), E, e.shape[1], W2.shape[1], _ptr(Ps), _ptr(Qd), ldn,
                                             _ptr(views.srt_src), _ptr(views.srt_dst), _ptr(eid), _ptr(W1e), ldw1,
                                             _ptr(W2), _ptr(b2), _ptr(W3), _ptr(b3), _ptr(logits), _ptr(z1_out),
                                             _stream(e.device)), "edge_score_f32")
    return logits


def gather_rows(table, idx, out=None):
    lib = _lib.load()
    table, ld_in = _rows(table, "gather_rows.in")
    _i32(idx, "gather_rows.idx")
    rows, width = int(idx.numel()), table.shape[1]
    if out is None:
        out = torch.empty((rows, width), dtype=torch.float32, device=table.device)
    out, ld_out = _rows(out, "gather_rows.out")
    with _on(table.device):
        _lib.check(lib.gnnome_gather_rows_f32(_ptr(table), ld_in, _ptr(idx), rows, width, _ptr(out), ld_out,
                                              _stream(table.device)), "gather_rows_f32")
    return out


def scatter_add_rows(src, idx, out):
    """out[idx[r]] += src[r] for DISTINCT idx (transpose of gather_rows; halo gradients returning to their owner)."""
    src, ld_in = _rows(src, "scatter_add_rows.in")
    out, ld_out = _rows(out, "scatter_add_rows.out")
    _i32(idx, "scatter_add_rows.idx")
    if int(idx.numel()) != src.shape[0] or src.shape[1] != out.shape[1]:
        raise ValueError("scatter_add_rows: shape mismatch")
    _call("gnnome_scatter_add_rows_f32", src.device, _ptr(src), ld_in, _ptr(idx), int(idx.numel()), src.shape[1], _ptr(out), ld_out)
    return out


# ---------------------------------------------------------------------------------------------------
# training-step entries (include/gnnome_hip.h, "Training step")
# ---------------------------------------------------------------------------------------------------

def _call(name, device, *args):
    lib = _lib.load()
    with _on(device):
        _lib.check(getattr(lib, name)(*args, _stream(device)), name)


def _dense(t, name):
    _f32(t, name)
    if not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous tensor")
    return t


def edge_gate_raw(e, B1h, B2h, views, W3):
    e = _dense(e, "edge_gate_raw.e")
    B1h, ldn = _rows(B1h, "edge_gate_raw.B1h")
    B2h, _ = _rows(B2h, "edge_gate_raw.B2h")
    W3, ldw = _rows(W3, "edge_gate_raw.W3")
    out = torch.empty_like(e)
    _call("gnnome_edge_gate_raw_f32", e.device, _ptr(e), _ptr(out), e.shape[0], e.shape[1], _ptr(B1h), _ptr(B2h), ldn,
          _ptr(views.srt_src), _ptr(views.srt_dst), _ptr(W3), ldw)
    return out


def edge_gate_raw_stats(e, B1h, B2h, views, W3, rows_stats=None):
    """-> (xe, mean, biased var): the raw gate and the batch statistics of its rows (of the first rows_stats rows when
    given).  One pass where possible: the kernel leaves per-workgroup shifted column sums and colsum2 adds them up in
    a fixed order.  Shapes the fused kernel does not take (H = 256; statistics over a prefix of the rows, as on a
    partition) go through edge_gate_raw + batch_stats."""
    H, E = e.shape[1], e.shape[0]
    fused = H in (64, 128, 256) and E > 0 and (rows_stats is None or rows_stats == E) and B1h.stride(0) % 4 == 0 and \
        B1h.data_ptr() % 16 == 0 and B2h.data_ptr() % 16 == 0
    if not fused:
        xe = edge_gate_raw(e, B1h, B2h, views, W3)
        mean, var = batch_stats(xe if rows_stats is None else xe[:rows_stats])
        return xe, mean, var
    e = _dense(e, "edge_gate_raw_stats.e")
    B1h, ldn = _rows(B1h, "edge_gate_raw_stats.B1h")
    B2h, _ = _rows(B2h, "edge_gate_raw_stats.B2h")
    W3, ldw = _rows(W3, "edge_gate_raw_stats.W3")
    # centre = row 0 of the output, from three [1,H] torch ops (no host sync: the indices stay on the device)
    s0, d0 = views.srt_src[:1].long(), views.srt_dst[:1].long()
    center = (B1h.index_select(0, s0) + B2h.index_select(0, d0) + e[:1] @ W3.t()).reshape(-1).contiguous()
    rows = ctypes.c_int(0)
    _lib.check(_lib.load().gnnome_edge_gate_raw_stats_rows(H, ctypes.byref(rows)), "edge_gate_raw_stats_rows")
    partial = torch.empty((rows.value, 2 * H), dtype=torch.float32, device=e.device)
    out = torch.empty_like(e)
    _call("gnnome_edge_gate_raw_stats_f32", e.device, _ptr(e), _ptr(out), E, H, _ptr(B1h), _ptr(B2h), ldn, _ptr(views.srt_src),
          _ptr(views.srt_dst), _ptr(W3), ldw, _ptr(center), _ptr(partial))
    sums = _partial_sums(partial, H)
    m1 = sums[:H] / E
    return out, (center + m1).contiguous(), (sums[H:] / E - m1 * m1).clamp_min_(0.0)


def node_aggregate_raw(e, A1h, A2h, A3h, views, mode, num_nodes, rows_alloc=None):
    """mode 1 -> (v, fwd, rden_f, bwd, rden_b); mode 2 -> (sum_in s*A2h[src], sum_out s*A3h[dst]).
    rows_alloc > num_nodes: outputs get that many rows, the ones past num_nodes zero (halo rows of a partition)."""
    A2h, ldn = _rows(A2h, "node_aggregate_raw.A2h")
    A3h, l3 = _rows(A3h, "node_aggregate_raw.A3h")
    assert ldn == l3
    H, dev = A2h.shape[1], A2h.device
    if rows_alloc is not None and rows_alloc > num_nodes:
        mk = lambda: torch.zeros((rows_alloc, H), dtype=torch.float32, device=dev)  # noqa: E731
    else:
        mk = lambda: torch.empty((num_nodes, H), dtype=torch.float32, device=dev)  # noqa: E731
    if mode == 1:
        A1h, l1 = _rows(A1h, "node_aggregate_raw.A1h")
        assert l1 == ldn
        v, a0, a1, a2, a3 = mk(), mk(), mk(), mk(), mk()
    else:
        v, a0, a1, a2, a3 = None, mk(), None, mk(), None
    _call("gnnome_node_aggregate_raw_f32", dev, _ptr(e), H, num_nodes, mode, _ptr(A1h) if mode == 1 else _ptr(None), _ptr(A2h),
          _ptr(A3h), ldn, _ptr(views.in_ptr), _ptr(views.srt_src), _ptr(views.out_ptr), _ptr(views.out_pos), _ptr(views.out_dst),
          _ptr(v), _ptr(a0), _ptr(a1), _ptr(a2), _ptr(a3))
    return (v, a0, a1, a2, a3) if mode == 1 else (a0, a2)


_COL_WS = {}


def _col_workspace(device):
    """Scratch of the deterministic column sums, one per device.  Launches on one device are stream-ordered in this
    package, so one buffer serves them all."""
    ws = _COL_WS.get(device.index)
    if ws is None:
        need = ctypes.c_size_t(0)
        _lib.check(_lib.load().gnnome_colsum_workspace_bytes(ctypes.byref(need)), "colsum_workspace_bytes")
        ws = _COL_WS[device.index] = torch.empty(int(need.value), dtype=torch.uint8, device=device)
    return ws


def colsum2(x, y=None, center=None):
    """(sum_r x', sum_r x'*y') per column with x' = x - center; y=None gives the (centred) sum of squares."""
    x = _dense(x, "colsum2.x")
    H = x.shape[1]
    s = (torch.empty if x.shape[0] > 0 else torch.zeros)((2, H), dtype=torch.float32, device=x.device)   # the kernel overwrites both rows
    ws = _col_workspace(x.device)
    _call("gnnome_colsum2_f32", x.device, _ptr(x), _ptr(y), x.shape[0], H, _ptr(center), _ptr(s[0]), _ptr(s[1]), _ptr(ws), ws.numel())
    return s[0], s[1]


def batch_stats(x):
    """Per-column (mean, biased variance) of x[rows, H] in ONE pass over x, shifted by its first row:
    var = E[(x-c)^2] - E[x-c]^2 with c = x[0] is as accurate as the two-pass formula whenever c lies within a few
    standard deviations of the mean (it is a sample), unlike the unshifted E[x^2] - E[x]^2."""
    rows = x.shape[0]
    c = x[0].contiguous()
    d1, d2 = colsum2(x, center=c)
    m1 = d1 / rows
    return (c + m1).contiguous(), (d2 / rows - m1 * m1).clamp_min_(0.0)


def batch_moments(x):
    """(d1, d2, center, rows): shifted column sums of x[rows,H] - d1 = sum(x - c), d2 = sum((x - c)^2), c = x[0] - in one
    pass; what bn_train_finish (and batch_stats) turn into mean / variance."""
    c = x[0]
    d1, d2 = colsum2(x, center=c)
    return d1, d2, c, x.shape[0]


def edge_gate_raw_moments(e, B1h, B2h, views, W3, storage=torch.float32):
    """-> (xe, (d1, d2, center, E)): the raw gate and the shifted column sums of its rows, ONE pass over [E,H] (the kernel
    leaves per-workgroup partial sums, colsum2 adds them up in a fixed order); H in {64,128}, all E rows.
    storage=torch.bfloat16: xe is rounded to bf16 on the way out and the sums are those of the rounded values."""
    H, E = e.shape[1], e.shape[0]
    e = _dense(e, "edge_gate_raw_moments.e")
    x16 = storage == torch.bfloat16
    B1h, ldn = _rows(B1h, "edge_gate_raw_moments.B1h")
    B2h, _ = _rows(B2h, "edge_gate_raw_moments.B2h")
    W3, ldw = _rows(W3, "edge_gate_raw_moments.W3")
    center = torch.empty(H, dtype=torch.float32, device=e.device)
    _call("gnnome_gate_center_f32", e.device, _ptr(e), E, H, _ptr(B1h), _ptr(B2h), ldn, _ptr(views.srt_src), _ptr(views.srt_dst), _ptr(W3), ldw,
          _ptr(center))
    rows = ctypes.c_int(0)
    _lib.check(_lib.load().gnnome_edge_gate_raw_stats_rows(H, ctypes.byref(rows)), "edge_gate_raw_stats_rows")
    partial = torch.empty((rows.value, 2 * H), dtype=torch.float32, device=e.device)
    out = torch.empty_like(e, dtype=torch.bfloat16 if x16 else torch.float32)
    _call("gnnome_edge_gate_raw_stats_x16" if x16 else "gnnome_edge_gate_raw_stats_f32", e.device, _ptr(e), _ptr(out), E, H, _ptr(B1h),
          _ptr(B2h), ldn, _ptr(views.srt_src), _ptr(views.srt_dst), _ptr(W3), ldw, _ptr(center), _ptr(partial))
    sums = _partial_sums(partial, H)
    return out, (sums[:H], sums[H:], center, E)


def can_two_pass_gate(e, B1h, B2h, storage=None):
    """The training forward's gate in two passes instead of three (round 4): statistics alone, then gate + xe out.  hidden = 128, the
    default kernels (gnnome_set_tuning key 0 untouched), 16-byte aligned row tables."""
    wide = e.shape[1] == 256 and storage in (None, torch.float32) and _TUNING.get(10, 0) == 0   # round 5: the fp16x3 kernel, fp32 storage
    return ((e.shape[1] == 128 or wide) and e.shape[0] > 0 and e.is_contiguous() and B1h.stride(0) % 4 == 0 and B2h.stride(0) % 4 == 0 and
            B1h.data_ptr() % 16 == 0 and B2h.data_ptr() % 16 == 0 and _TUNING.get(0, 0) == 0)


def edge_gate_moments_only(e, B1h, B2h, views, W3, storage=torch.float32):
    """-> (d1, d2, center, E): edge_gate_raw_moments' statistics WITHOUT the [E,H] output (gnnome_edge_gate_raw_stats_f32 with x_out = NULL)."""
    H, E = e.shape[1], e.shape[0]
    e = _dense(e, "edge_gate_moments_only.e")
    x16 = storage == torch.bfloat16
    B1h, ldn = _rows(B1h, "edge_gate_moments_only.B1h")
    B2h, _ = _rows(B2h, "edge_gate_moments_only.B2h")
    W3, ldw = _rows(W3, "edge_gate_moments_only.W3")
    center = torch.empty(H, dtype=torch.float32, device=e.device)
    _call("gnnome_gate_center_f32", e.device, _ptr(e), E, H, _ptr(B1h), _ptr(B2h), ldn, _ptr(views.srt_src), _ptr(views.srt_dst), _ptr(W3), ldw,
          _ptr(center))
    rows = ctypes.c_int(0)
    _lib.check(_lib.load().gnnome_edge_gate_raw_stats_rows(H, ctypes.byref(rows)), "edge_gate_raw_stats_rows")
    partial = torch.empty((rows.value, 2 * H), dtype=torch.float32, device=e.device)
    _call("gnnome_edge_gate_raw_stats_x16" if x16 else "gnnome_edge_gate_raw_stats_f32", e.device, _ptr(e), None, E, H, _ptr(B1h),
          _ptr(B2h), ldn, _ptr(views.srt_src), _ptr(views.srt_dst), _ptr(W3), ldw, _ptr(center), _ptr(partial))
    sums = _partial_sums(partial, H)
    return sums[:H], sums[H:], center, E


def edge_gate_bn(e, B1h, B2h, views, W3, scale, shift, storage=torch.float32):
    """-> (e_new, xe): e_new = relu(xe * scale + shift) + e and xe = e W3^T + B1h[src] + B2h[dst] in ONE pass (gnnome_edge_gate_bn_f32):
    the train-mode gate once the batch statistics are known.  storage = bfloat16: xe stored as bf16, e_new computed from the rounded rows."""
    H, E = e.shape[1], e.shape[0]
    e = _dense(e, "edge_gate_bn.e")
    x16 = storage == torch.bfloat16
    B1h, ldn = _rows(B1h, "edge_gate_bn.B1h")
    B2h, _ = _rows(B2h, "edge_gate_bn.B2h")
    W3, ldw = _rows(W3, "edge_gate_bn.W3")
    e_new = torch.empty_like(e)
    xe = torch.empty_like(e, dtype=torch.bfloat16 if x16 else torch.float32)
    _call("gnnome_edge_gate_bn_x16" if x16 else "gnnome_edge_gate_bn_f32", e.device, _ptr(e), _ptr(e_new), _ptr(xe), E, H, _ptr(B1h), _ptr(B2h), ldn,
          _ptr(views.srt_src), _ptr(views.srt_dst), _ptr(W3), ldw, _ptr(scale), _ptr(shift))
    return e_new, xe


def _partial_sums(partial, H):
    """[2H] = (sum | sum of squares) from the edge-tile kernel's per-wave rows: [rows][2H] at H <= 128, two stacked [rows][H]
    matrices at H = 256 (edge_gate_pl256.hip); added up by the deterministic column-sum kernel."""
    if H == 256:
        rows = partial.shape[0]
        stacked = partial.view(2, rows, H)
        return torch.cat([colsum2(stacked[0])[0], colsum2(stacked[1])[0]])
    return colsum2(partial)[0]


def can_fuse_gate_moments(e, B1h, B2h, storage=torch.float32):
    """H in {64, 128, 256}; bf16 storage at 256 (round 4) runs on the fp16x3 kernel only (not under set_tuning(10, 1))."""
    widths = (64, 128, 256) if storage == torch.float32 or _TUNING.get(10, 0) == 0 else (64, 128)
    return (e.shape[1] in widths and e.shape[0] > 0 and B1h.stride(0) % 4 == 0 and B1h.data_ptr() % 16 == 0 and
            B2h.data_ptr() % 16 == 0)


def bn_train_finish(moments, weight, bias, running_mean, running_var, num_batches_tracked, momentum, eps, updates):
    """One launch for the per-channel arithmetic of a train-mode BatchNorm1d call (see gnnome_bn_train_finish_f32):
    -> (mean, rstd, scale, shift); running_mean / running_var / num_batches_tracked advance in place `updates` times."""
    d1, d2, center, rows = moments
    H = d1.numel()
    mean, rstd, scale, shift = (torch.empty(H, dtype=torch.float32, device=d1.device) for _ in range(4))
    _call("gnnome_bn_train_finish_f32", d1.device, _ptr(d1), _ptr(d2), _ptr(center), int(rows), H, _ptr(weight), _ptr(bias), _ptr(running_mean),
          _ptr(running_var), _ptr(num_batches_tracked), float(momentum), float(eps), int(updates), _ptr(mean), _ptr(rstd), _ptr(scale),
          _ptr(shift))
    return mean, rstd, scale, shift


def bn_bwd_terms(s1, s2, rstd, rows):
    """(rstd * s2, s1 / rows, rstd * s2 / rows) in one launch: the per-channel vectors of BatchNorm's backward (one rank)."""
    H = s1.numel()
    s2h, c1, c2 = (torch.empty(H, dtype=torch.float32, device=s1.device) for _ in range(3))
    _call("gnnome_bn_bwd_terms_f32", s1.device, _ptr(s1), _ptr(s2), _ptr(rstd), int(rows), H, _ptr(s2h), _ptr(c1), _ptr(c2))
    return s2h, c1, c2


def pack_layer(weights5, biases5, B3_weight, B3_bias):
    """(Wcat[5H,H], bcat[5H], WcatT[H,5H], W3T[H,H]) of one SymGatedGCN layer in one launch (see gnnome_pack_layer_f32)."""
    H, dev = B3_weight.shape[0], B3_weight.device
    ws = [_dense(w.detach(), "pack_layer.weight") for w in weights5]
    bs = [_dense(b.detach(), "pack_layer.bias") for b in biases5]
    w3, b3 = _dense(B3_weight.detach(), "pack_layer.B3_weight"), _dense(B3_bias.detach(), "pack_layer.B3_bias")
    wt = (ctypes.c_void_p * 5)(*[w.data_ptr() for w in ws])
    bt = (ctypes.c_void_p * 5)(*[b.data_ptr() for b in bs])
    Wcat = torch.empty((5 * H, H), dtype=torch.float32, device=dev)
    bcat = torch.empty(5 * H, dtype=torch.float32, device=dev)
    WcatT = torch.empty((H, 5 * H), dtype=torch.float32, device=dev)
    W3T = torch.empty((H, H), dtype=torch.float32, device=dev)
    _call("gnnome_pack_layer_f32", dev, wt, bt, _ptr(w3), _ptr(b3), H, _ptr(Wcat), _ptr(bcat), _ptr(WcatT), _ptr(W3T))
    return Wcat, bcat, WcatT, W3T


def bn_relu_res(x, scale, shift, res, out=None):
    (x, x16), res = _act(x, "bn_relu_res.x"), _dense(res, "bn_relu_res.res")
    out = torch.empty_like(res) if out is None else _dense(out, "bn_relu_res.out")
    _call("gnnome_bn_relu_res_x16" if x16 else "gnnome_bn_relu_res_f32", x.device, _ptr(x), _ptr(scale), _ptr(shift), _ptr(res), x.shape[0],
          x.shape[1], _ptr(out))
    return out


def bn_bwd_stats(dy, x, scale, shift, mean):
    H = x.shape[1]
    s = (torch.empty if x.shape[0] > 0 else torch.zeros)((2, H), dtype=torch.float32, device=x.device)   # overwritten by the kernel
    ws = _col_workspace(x.device)
    _call("gnnome_bn_bwd_stats_f32", x.device, _ptr(_dense(dy, "dy")), _ptr(_dense(x, "x")), _ptr(scale), _ptr(shift), _ptr(mean),
          x.shape[0], H, _ptr(s[0]), _ptr(s[1]), _ptr(ws), ws.numel())
    return s[0], s[1]


def bn_bwd_apply(dy, x, scale, shift, a, c1, c2, mean, rstd, out=None):
    dx = torch.empty_like(x) if out is None else _dense(out, "bn_bwd_apply.out")
    if x.shape[0] == 0:
        return dx
    _call("gnnome_bn_bwd_apply_f32", x.device, _ptr(_dense(dy, "dy")), _ptr(_dense(x, "x")), _ptr(scale), _ptr(shift), x.shape[0],
          x.shape[1], _ptr(a), _ptr(c1), _ptr(c2), _ptr(mean), _ptr(rstd), _ptr(dx))
    return dx


def bn_bwd_apply_tables(dy, x, scale, shift, a, c1, c2, mean, rstd, rdf, hf, rdb, hb, out=None, amax=None):
    """bn_bwd_apply and the two mul23 calls that follow it for a layer's node BatchNorm, one launch: -> (dx, Tf, Uf, Tb, Ub) with
    Tf = dx*rdf, Uf = Tf*hf, Tb = dx*rdb, Ub = Tb*hb."""
    dx = torch.empty_like(x) if out is None else _dense(out, "bn_bwd_apply_tables.out")
    t = torch.empty((4,) + tuple(x.shape), dtype=torch.float32, device=x.device)
    if x.shape[0] == 0:
        return dx, t[0], t[1], t[2], t[3]
    for name, v in (("rdf", rdf), ("hf", hf), ("rdb", rdb), ("hb", hb)):
        if v.shape != x.shape:
            raise ValueError(f"bn_bwd_apply_tables.{name}: shape {tuple(v.shape)} != {tuple(x.shape)}")
    _call("gnnome_bn_bwd_apply_tables_f32", x.device, _ptr(_dense(dy, "dy")), _ptr(_dense(x, "x")), _ptr(scale), _ptr(shift), x.shape[0],
          x.shape[1], _ptr(a), _ptr(c1), _ptr(c2), _ptr(mean), _ptr(rstd), _ptr(_dense(rdf, "rdf")), _ptr(_dense(hf, "hf")),
          _ptr(_dense(rdb, "rdb")), _ptr(_dense(hb, "hb")), _ptr(dx), _ptr(t[0]), _ptr(t[1]), _ptr(t[2]), _ptr(t[3]),
          _ptr(_amax_slot(amax, x.device, "bn_bwd_apply_tables")))   # amax: raised to max |dx|
    return dx, t[0], t[1], t[2], t[3]


def ln_relu_res(x, gamma, beta, res, out=None, width=None):
    """relu(LayerNorm(x) * gamma + beta) + res (normalization='layer' in train mode).  width: the statistics run over the first `width`
    channels (a zero-padded narrower model; default: all)."""
    x, res = _dense(x, "ln_relu_res.x"), _dense(res, "ln_relu_res.res")
    out = torch.empty_like(x) if out is None else _dense(out, "ln_relu_res.out")
    _call("gnnome_ln_relu_res_f32", x.device, _ptr(x), _ptr(gamma), _ptr(beta), _ptr(res), x.shape[0], x.shape[1], int(width or 0), _ptr(out))
    return out


def ln_bwd(dy, x, gamma, beta, out=None, width=None):
    """-> (dx, dgamma, dbeta) of out = relu(LayerNorm(x) * gamma + beta) + res."""
    dy, x = _dense(dy, "ln_bwd.dy"), _dense(x, "ln_bwd.x")
    H = x.shape[1]
    dx = torch.empty_like(x) if out is None else _dense(out, "ln_bwd.out")
    s = (torch.empty if x.shape[0] > 0 else torch.zeros)((2, H), dtype=torch.float32, device=x.device)
    ws = _col_workspace(x.device)
    _call("gnnome_ln_bwd_f32", x.device, _ptr(dy), _ptr(x), _ptr(gamma), _ptr(beta), x.shape[0], H, int(width or 0), _ptr(dx), _ptr(s[0]), _ptr(s[1]),
          _ptr(ws), ws.numel())
    return dx, s[1], s[0]


def mul23(a, b, c):
    o1, o2 = torch.empty_like(a), torch.empty_like(a)
    _call("gnnome_mul23_f32", a.device, _ptr(_dense(a, "a")), _ptr(_dense(b, "b")), _ptr(_dense(c, "c")), a.numel(), _ptr(o1), _ptr(o2))
    return o1, o2


def add(a, b, out=None):
    out = torch.empty_like(a) if out is None else out
    _call("gnnome_add_f32", a.device, _ptr(_dense(a, "a")), _ptr(_dense(b, "b")), a.numel(), _ptr(out))
    return out


def relu_bwd(dy, y):
    dx = torch.empty_like(dy)
    _call("gnnome_relu_bwd_f32", dy.device, _ptr(_dense(dy, "dy")), _ptr(_dense(y, "y")), dy.numel(), _ptr(dx))
    return dx


def segment_sum(X, ptr, pos, num_nodes, out=None):
    X = _dense(X, "segment_sum.X")
    W = X.shape[1]
    if out is None:
        out = torch.empty((num_nodes, W), dtype=torch.float32, device=X.device)
    out, ld = _rows(out, "segment_sum.out")
    _call("gnnome_segment_sum_f32", X.device, _ptr(X), W, _ptr(ptr), _ptr(pos), num_nodes, _ptr(out), ld)
    return out


def segment_sum2(X, views, num_nodes, out_in=None, out_out=None, amax=None):
    """(sum over in-edge rows, sum over out-edge rows) of X[E,W] per node, one launch; outputs may be column blocks of a
    wider table (row-strided)."""
    X, x16 = _act(X, "segment_sum2.X")
    W = X.shape[1]
    mk = lambda: torch.empty((num_nodes, W), dtype=torch.float32, device=X.device)  # noqa: E731
    out_in = mk() if out_in is None else out_in
    out_out = mk() if out_out is None else out_out
    out_in, ld_in = _rows(out_in, "segment_sum2.out_in")
    out_out, ld_out = _rows(out_out, "segment_sum2.out_out")
    if amax is not None and x16:
        # a PRODUCER that cannot raise the slot must say so: a consumer (wgrad_blocks / linear_blocks with the same amax) would scale the
        # blocks by a maximum that leaves these sums out - fp16x3 operands beyond 65504, NaN gradients.  The consumers' own
        # fall-backs (a width they are not built for) are safe: unscaled bf16x6 needs no maximum.
        raise TypeError("segment_sum2(amax=) is built for float32 rows: with bfloat16 rows pass amax=None (and keep the consumers on bf16x6)")
    if amax is not None and num_nodes > 0:   # (amax: raised to max |out_in|, |out_out|)
        _call("gnnome_segment_sum2_amax_f32", X.device, _ptr(X), W, _ptr(views.in_ptr), _ptr(views.out_ptr), _ptr(views.out_pos), num_nodes,
              _ptr(out_in), ld_in, _ptr(out_out), ld_out, _ptr(_amax_slot(amax, X.device, "segment_sum2")))
        return out_in, out_out
    _call("gnnome_segment_sum2_x16" if x16 else "gnnome_segment_sum2_f32", X.device, _ptr(X), W, _ptr(views.in_ptr), _ptr(views.out_ptr), _ptr(views.out_pos), num_nodes,
          _ptr(out_in), ld_in, _ptr(out_out), ld_out)
    return out_in, out_out


def wgrad(A, B, out=None, amax=None):
    """out[Ka,Kb] = A^T @ B over the rows (nn.Linear weight gradient dW = dY^T X).  A may be a contiguous bfloat16 tensor
    (the dxe rows of the bf16-storage training step).  amax: a one-element int32 device tensor holding the bits of max |A| (what
    bn_bwd_dgrad(..., amax=) leaves) - the product then runs as fp16x3 with A scaled into fp16's range (gnnome_wgrad_scaled_f32)."""
    x16 = A.dtype == torch.bfloat16
    A, lda = (_act(A, "wgrad.A")[0], A.shape[1]) if x16 else _rows(A, "wgrad.A")
    B, ldb = _rows(B, "wgrad.B")
    rows, Ka, Kb = A.shape[0], A.shape[1], B.shape[1]
    if out is None:
        out = torch.empty((Ka, Kb), dtype=torch.float32, device=A.device)
    out, ldc = _rows(out, "wgrad.out")
    need = ctypes.c_size_t(0)
    _lib.check(_lib.load().gnnome_wgrad_workspace_bytes(rows, Ka, Kb, ctypes.byref(need)), "wgrad_workspace_bytes")
    ws = torch.empty(max(int(need.value), 4), dtype=torch.uint8, device=A.device)
    if amax is not None and not x16 and rows > 0:
        if amax.dtype != torch.int32 or amax.numel() != 1 or amax.device != A.device:
            raise ValueError("wgrad.amax: a one-element int32 tensor on A's device (the bits of max |A|)")
        _call("gnnome_wgrad_scaled_f32", A.device, _ptr(A), lda, Ka, _ptr(B), ldb, Kb, rows, _ptr(amax), _ptr(out), ldc, _ptr(ws), ws.numel())
        return out
    _call("gnnome_wgrad_x16" if x16 else "gnnome_wgrad_f32", A.device, _ptr(A), lda, Ka, _ptr(B), ldb, Kb, rows, _ptr(out), ldc, _ptr(ws),
          ws.numel())
    return out


def _block_table(blocks, name):
    """A list of [rows, width] tensors with one common row stride -> (host array of device pointers, rows, width, lda)."""
    rows, width = blocks[0].shape
    lda = None
    for k, b in enumerate(blocks):
        b, ld = _rows(b, f"{name}[{k}]")
        if tuple(b.shape) != (rows, width) or b.device != blocks[0].device or (lda is not None and ld != lda):
            raise ValueError(f"{name}: the column blocks must share shape, device and row stride")
        lda = ld
    table = (ctypes.c_void_p * len(blocks))(*[b.data_ptr() for b in blocks])
    return table, rows, width, lda


def wgrad_blocks(blocks, B, colsum=True, amax=None):
    """(C, s): C[len(blocks)*width, Kb] = [blocks[0] | blocks[1] | ...]^T @ B without concatenating, and the column sums s of
    the blocks (the bias gradients) from the same pass (None with colsum=False).  amax: the slot the blocks' producers raised to the
    largest |element| of all blocks (agg_bwd_fused / bn_bwd_apply_tables / segment_sum2 with amax=) - the product then runs as fp16x3."""
    table, rows, width, lda = _block_table(blocks, "wgrad_blocks.A")
    B, ldb = _rows(B, "wgrad_blocks.B")
    Ka, Kb, dev = len(blocks) * width, B.shape[1], B.device
    out = torch.empty((Ka, Kb), dtype=torch.float32, device=dev)
    sums = torch.empty(Ka, dtype=torch.float32, device=dev) if colsum else None
    need = ctypes.c_size_t(0)
    _lib.check(_lib.load().gnnome_wgrad_workspace_bytes(rows, Ka, Kb, ctypes.byref(need)), "wgrad_workspace_bytes")
    ws = torch.empty(max(int(need.value), 4), dtype=torch.uint8, device=dev)
    if amax is not None and rows > 0 and width % 128 == 0:
        _call("gnnome_wgrad_blocks_scaled_f32", dev, table, len(blocks), width, lda, _ptr(B), ldb, Kb, rows, _ptr(_amax_slot(amax, dev, "wgrad_blocks")),
              _ptr(out), Kb, _ptr(sums), _ptr(ws), ws.numel())
        return out, sums
    _call("gnnome_wgrad_blocks_f32", dev, table, len(blocks), width, lda, _ptr(B), ldb, Kb, rows, _ptr(out), Kb, _ptr(sums), _ptr(ws),
          ws.numel())
    return out, sums


def linear_blocks(blocks, W, out, accumulate=False, amax=None):
    """out[M,Nout] (+)= [blocks[0] | blocks[1] | ...] @ W.T without concatenating (W[Nout, len(blocks)*width]).  amax: the slot the blocks'
    producers raised to their largest |element| - ONE fp16x3 launch on the scaled blocks (gnnome_linear_blocks_scaled_f32)."""
    table, M, width, lda = _block_table(blocks, "linear_blocks.A")
    W, ldw = _rows(W, "linear_blocks.W")
    out, ldc = _rows(out, "linear_blocks.out")
    assert W.shape[1] == len(blocks) * width and out.shape[0] == M
    if amax is not None and M > 0 and width % 32 == 0:
        _call("gnnome_linear_blocks_scaled_f32", out.device, table, len(blocks), width, M, lda, _ptr(W), ldw, W.shape[0],
              _ptr(_amax_slot(amax, out.device, "linear_blocks")), _ptr(out), ldc, 1 if accumulate else 0)
        return out
    _call("gnnome_linear_blocks_f32", out.device, table, len(blocks), width, M, lda, _ptr(W), ldw, W.shape[0], _ptr(out), ldc,
          1 if accumulate else 0)
    return out


def can_use_blocks(blocks):
    b0 = blocks[0]
    return (len(blocks) <= 8 and b0.dim() == 2 and b0.shape[1] % 32 == 0 and
            all(b.data_ptr() % 16 == 0 and b.shape == b0.shape and b.stride() == b0.stride() and b.stride(1) == 1 for b in blocks))


def score_tail_bwd(z1, dscore, views, W2, b2, W3):
    E, hs = z1.shape
    dz1 = torch.empty_like(z1)
    dz2 = torch.empty((E, 32), dtype=torch.float32, device=z1.device)
    u = torch.empty_like(dz2)
    _call("gnnome_score_tail_bwd_f32", z1.device, _ptr(_dense(z1, "z1")), _ptr(_dense(dscore, "dscore")), _ptr(views.srt_eid), E, hs,
          _ptr(W2), _ptr(b2), _ptr(W3), _ptr(dz1), _ptr(dz2), _ptr(u))
    return dz1, dz2, u


def agg_edge_bwd(e, Tf, Uf, Tb, Ub, A2h, A3h, views, de):
    A2h, ldn = _rows(A2h, "agg_edge_bwd.A2h")
    A3h, _ = _rows(A3h, "agg_edge_bwd.A3h")
    _call("gnnome_agg_edge_bwd_f32", e.device, _ptr(_dense(e, "e")), e.shape[0], e.shape[1], _ptr(Tf), _ptr(Uf), _ptr(Tb), _ptr(Ub),
          _ptr(A2h), _ptr(A3h), ldn, _ptr(views.srt_src), _ptr(views.srt_dst), _ptr(_dense(de, "de")))
    return de


def can_fuse_bn_bwd_dgrad(de, W, xe=None):
    """H = 256 (edge_gate_pl256.hip mode 3; bf16 storage since round 4): the update is out of place under the hood."""
    return de.shape[1] in (64, 128, 256) and de.shape[0] > 0 and de.is_contiguous() and W.stride(0) % 4 == 0


def can_dgrad_amax(de, xe):
    """bn_bwd_dgrad can leave max |dxe| (amax=): hidden = 128 (the plane-form kernel) or 256 (round 6), fp32 storage."""
    return de.shape[1] in (128, 256) and de.shape[0] > 0 and xe.dtype == torch.float32 and _TUNING.get(0, 0) != 8


def bn_bwd_dgrad(de, xe, scale, shift, a, c1, c2, mean, rstd, Wt, rows_once=None, amax=None):
    """dxe = BatchNorm-backward(de, xe) (as bn_bwd_apply) and de += dxe @ Wt.T in ONE pass (gnnome_bn_bwd_dgrad_f32): the
    edge-tile kernel's load waves compute the A tile instead of reading it.  Returns dxe; de is updated in place.
    rows_once: the mean terms c1, c2 enter the first rows_once rows only (a partition's owned in-edges); default all rows."""
    de, (xe, x16) = _dense(de, "bn_bwd_dgrad.de"), _act(xe, "bn_bwd_dgrad.xe")
    Wt, ldw = _rows(Wt, "bn_bwd_dgrad.W")
    dxe = torch.empty_like(xe)     # dxe is stored the way xe is
    once = de.shape[0] if rows_once is None else int(rows_once)
    if de.shape[1] == 256:
        # two workgroups per row at this width: the kernel writes the updated rows to a fresh buffer, which then BECOMES de
        # (Tensor.set_: same tensor object, new storage - the caller's `de` is updated "in place" as at the other widths)
        out = torch.empty_like(de)
        if amax is not None:
            if not can_dgrad_amax(de, xe) or amax.dtype != torch.int32 or amax.numel() != 1 or amax.device != de.device:
                raise ValueError("bn_bwd_dgrad.amax: needs fp32 storage and a one-element int32 tensor on the same device")
            _call("gnnome_bn_bwd_dgrad_out_amax_f32", de.device, _ptr(de), _ptr(out), _ptr(xe), de.shape[0], once, 256, _ptr(scale), _ptr(shift), _ptr(a),
                  _ptr(c1), _ptr(c2), _ptr(mean), _ptr(rstd), _ptr(Wt), ldw, _ptr(dxe), _ptr(amax))
        else:
            _call("gnnome_bn_bwd_dgrad_out_x16" if x16 else "gnnome_bn_bwd_dgrad_out_f32", de.device, _ptr(de), _ptr(out), _ptr(xe), de.shape[0], once, 256, _ptr(scale), _ptr(shift), _ptr(a),
                  _ptr(c1), _ptr(c2), _ptr(mean), _ptr(rstd), _ptr(Wt), ldw, _ptr(dxe))
        if de._base is not None or de.storage_offset() != 0:   # a view of somebody else's storage: set_ would leave that storage unchanged
            de.copy_(out)
        else:
            de.set_(out)
        return dxe
    if amax is not None:   # (a one-element int32 tensor: the bits of max |dxe| as a non-negative float, for wgrad(dxe, ., amax=))
        if not can_dgrad_amax(de, xe) or amax.dtype != torch.int32 or amax.numel() != 1 or amax.device != de.device:
            raise ValueError("bn_bwd_dgrad.amax: needs hidden = 128, fp32 storage and a one-element int32 tensor on the same device")
        _call("gnnome_bn_bwd_dgrad_amax_f32", de.device, _ptr(de), _ptr(xe), de.shape[0], once, de.shape[1], _ptr(scale), _ptr(shift),
              _ptr(a), _ptr(c1), _ptr(c2), _ptr(mean), _ptr(rstd), _ptr(Wt), ldw, _ptr(dxe), _ptr(amax))
        return dxe
    _call("gnnome_bn_bwd_dgrad_x16" if x16 else "gnnome_bn_bwd_dgrad_f32", de.device, _ptr(de), _ptr(xe), de.shape[0], once, de.shape[1], _ptr(scale), _ptr(shift),
          _ptr(a), _ptr(c1), _ptr(c2), _ptr(mean), _ptr(rstd), _ptr(Wt), ldw, _ptr(dxe))
    return dxe


def agg_edge_bwd_stats(e, Tf, Uf, Tb, Ub, A2h, A3h, views, de, xe, scale, shift, mean):
    """agg_edge_bwd + the BatchNorm-backward statistics (s1, s2) of the updated de in one pass (see the header)."""
    A2h, ldn = _rows(A2h, "agg_edge_bwd_stats.A2h")
    A3h, _ = _rows(A3h, "agg_edge_bwd_stats.A3h")
    H = e.shape[1]
    s = (torch.empty if e.shape[0] > 0 else torch.zeros)((2, H), dtype=torch.float32, device=e.device)
    ws = _col_workspace(e.device)
    xe, x16 = _act(xe, "agg_edge_bwd_stats.xe")
    _call("gnnome_agg_edge_bwd_stats_x16" if x16 else "gnnome_agg_edge_bwd_stats_f32", e.device, _ptr(_dense(e, "e")), e.shape[0], H, _ptr(Tf),
          _ptr(Uf), _ptr(Tb), _ptr(Ub), _ptr(A2h), _ptr(A3h), ldn, _ptr(views.srt_src), _ptr(views.srt_dst), _ptr(_dense(de, "de")), _ptr(xe),
          _ptr(scale), _ptr(shift),
          _ptr(mean), _ptr(s[0]), _ptr(s[1]), _ptr(ws), ws.numel())
    return de, s[0], s[1]


NODE_AMAX = True   # agg_bwd_fused / bn_bwd_apply_tables / segment_sum2 / wgrad_blocks take amax= (train.py asks before it builds the slot)


def _amax_slot(amax, device, what):
    if amax is not None and (amax.dtype != torch.int32 or amax.numel() != 1 or amax.device != device):
        raise ValueError(f"{what}.amax: a one-element int32 tensor on the operands' device (the bits of a non-negative float)")
    return amax


def agg_bwd_fused(e, Tf, Uf, Tb, Ub, A2h, A3h, views, de, xe, scale, shift, mean, num_nodes, amax=None):
    """node_aggregate_raw(e, None, Tb, Tf, views, 2, num_nodes) and agg_edge_bwd_stats(...) in one launch (gnnome_agg_bwd_fused_f32: one
    read of e from HBM instead of two) -> (sum_in, sum_out, de, s1, s2); de is updated in place."""
    A2h, ldn = _rows(A2h, "agg_bwd_fused.A2h")
    A3h, l3 = _rows(A3h, "agg_bwd_fused.A3h")
    assert ldn == l3
    H, dev = e.shape[1], e.device
    for name, t in (("Tf", Tf), ("Uf", Uf), ("Tb", Tb), ("Ub", Ub)):
        if t.shape != (num_nodes, H) or not t.is_contiguous() or t.dtype != torch.float32:
            raise ValueError(f"agg_bwd_fused.{name}: expected a contiguous float32 [{num_nodes}, {H}] tensor, got {tuple(t.shape)} {t.dtype}")
    if A2h.shape[0] < num_nodes or A3h.shape[0] < num_nodes:
        raise ValueError("agg_bwd_fused: A2h / A3h have fewer rows than num_nodes")
    sums = torch.empty((2, num_nodes, H), dtype=torch.float32, device=dev)
    s = torch.empty((2, H), dtype=torch.float32, device=dev)
    ws = _col_workspace(dev)
    xe, x16 = _act(xe, "agg_bwd_fused.xe")
    _call("gnnome_agg_bwd_fused_x16" if x16 else "gnnome_agg_bwd_fused_f32", dev, _ptr(_dense(e, "e")), num_nodes, e.shape[0], H, _ptr(Tf), _ptr(Uf),
          _ptr(Tb), _ptr(Ub), _ptr(A2h), _ptr(A3h), ldn, _ptr(views.in_ptr), _ptr(views.srt_src), _ptr(views.out_ptr), _ptr(views.out_pos),
          _ptr(views.out_dst), _ptr(_dense(de, "de")), _ptr(xe), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(sums[0]), _ptr(sums[1]), _ptr(s[0]),
          _ptr(s[1]), _ptr(_amax_slot(amax, dev, "agg_bwd_fused")), _ptr(ws), ws.numel())   # amax: RAISED to max |sum_in|, |sum_out| (never zeroed here)
    return sums[0], sums[1], de, s[0], s[1]


def encode_hidden(x, W1, b1, gather=None, rows=None):
    x = x.contiguous()
    rows = int(x.shape[0] if rows is None else rows)
    t = torch.empty((rows, W1.shape[0]), dtype=torch.float32, device=x.device)
    _call("gnnome_encode_hidden_f32", x.device, _ptr(x), rows, x.shape[1], _ptr(gather), _ptr(W1), _ptr(b1), W1.shape[0], _ptr(t))
    return t


# ---------------------------------------------------------------------------------------------------
# callers' closure (include/gnnome_hip.h, "The callers' arithmetic either side of the model call")
# ---------------------------------------------------------------------------------------------------

def _closure_workspace(device):
    need = ctypes.c_size_t(0)
    _lib.check(_lib.load().gnnome_closure_workspace_bytes(ctypes.byref(need)), "closure_workspace_bytes")
    return torch.empty(int(need.value), dtype=torch.uint8, device=device)


def degree_features(views, reverse=False):
    """x[N,2] = [zscore(in_degree) | zscore(out_degree)] read off the views' CSR pointers; `reverse` = the reference's
    argument (train.py:112-122): columns swapped.  The degrees are those of the edge list the views were BUILT from - like
    the reference's stored ndata['in_deg'] / ['out_deg'], which dgl.reverse(g, True, True) copies and does not recompute
    (train.py:165-166) - so `views.reversed()` gives the same features as `views`, and the reference's second pass is
    `degree_features(views.reversed(), reverse=True)`, literally."""
    x = torch.empty((views.num_nodes, 2), dtype=torch.float32, device=views.device)
    ws = _closure_workspace(views.device)
    _call("gnnome_degree_features_f32", views.device, _ptr(views.in_ptr), _ptr(views.out_ptr), views.num_nodes, int(bool(reverse)), _ptr(x),
          _ptr(ws), ws.numel())
    if views.node_perm is not None:   # rows in the caller's node numbering (the z-score is over all nodes: order-independent)
        x = x.index_select(0, views.node_perm)
    return x


def edge_features(overlap_length, overlap_similarity):
    """e[E,2] = [zscore(overlap_length) | overlap_similarity] (utils/data_utils.py:31-41)."""
    ol = _dense(overlap_length.float(), "edge_features.overlap_length")
    sim = _dense(overlap_similarity.float(), "edge_features.overlap_similarity")
    if ol.shape != sim.shape or ol.dim() != 1:
        raise ValueError("edge_features: overlap_length and overlap_similarity must be 1-d and equally long")
    e = torch.empty((ol.numel(), 2), dtype=torch.float32, device=ol.device)
    ws = _closure_workspace(ol.device)
    _call("gnnome_edge_features_f32", ol.device, _ptr(ol), _ptr(sim), ol.numel(), _ptr(e), _ptr(ws), ws.numel())
    return e


def edge_loss(logits, logits_rev, labels, pos_weight, alpha=0.0, need_grad=True, need_counts=False):
    """-> (loss[1], d loss/d logits, d loss/d logits_rev, tfpn int64[4]); entries not asked for are None."""
    a = _dense(logits, "edge_loss.logits")
    b = None if logits_rev is None else _dense(logits_rev, "edge_loss.logits_rev")
    y = _dense(labels, "edge_loss.labels")
    E = a.numel()
    if a.dim() != 1 or y.shape != a.shape or (b is not None and b.shape != a.shape):
        raise ValueError("edge_loss: logits, logits_rev and labels must be 1-d and equally long")
    pw = torch.as_tensor(pos_weight, dtype=torch.float32).reshape(1).to(a.device)
    loss = torch.empty(1, dtype=torch.float32, device=a.device)
    da = torch.empty_like(a) if need_grad else None
    db = torch.empty_like(a) if need_grad and b is not None else None
    tfpn = torch.empty(4, dtype=torch.int64, device=a.device) if need_counts else None
    ws = _closure_workspace(a.device)
    _call("gnnome_edge_loss_f32", a.device, _ptr(a), _ptr(b), _ptr(y), E, _ptr(pw), float(alpha), 1.0 / max(E, 1), _ptr(loss),
          _ptr(da), _ptr(db), _ptr(tfpn), _ptr(ws), ws.numel())
    return loss, da, db, tfpn
