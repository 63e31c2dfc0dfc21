"""Test double for the eight DGL 0.8.1 calls GNNome's hot path makes.

This is NOT DGL and NOT part of the product.  DGL 0.8.1 (requirements_cpu.txt:23
of the reference) is an un-vendored third-party wheel that cannot be installed
offline, so `tests/golden/make_golden.py` puts this package ahead of the
reference on sys.path purely to let the reference's own, unmodified
`models/` + `layers/` classes run at fixture-generation time.  Each function
restates the *documented* semantics of the DGL call:

  apply_edges(fn.u_add_v(a, b, out))      out[k] = ndata[a][src_k] + ndata[b][dst_k]
  update_all(fn.u_mul_e(a, s, m), fn.sum(m, out))
                                          out[i] = sum_{k: dst_k = i} ndata[a][src_k] * edata[s][k]
  update_all(fn.copy_e(s, m), fn.sum(m, out))
                                          out[i] = sum_{k: dst_k = i} edata[s][k]
  apply_edges(udf)                        udf sees edges.src[f] = ndata[f][src], edges.dst[f] = ndata[f][dst]
  reverse(g, copy_ndata, copy_edata)      endpoints swapped, edge ids kept, feature tensors shared
  g.local_scope()                         feature writes inside are dropped on exit

Reduction order over a node's edges is edge-id order (DGL leaves it unspecified).
"""
import contextlib
import torch

from . import function  # noqa: F401
from . import nn  # noqa: F401


class _EdgeBatch:
    def __init__(self, g):
        self.src = {k: v[g._src] for k, v in g.ndata.items()}
        self.dst = {k: v[g._dst] for k, v in g.ndata.items()}
        self.data = g.edata


class DGLGraph:
    def __init__(self, src, dst, num_nodes):
        self._src = torch.as_tensor(src, dtype=torch.int64)
        self._dst = torch.as_tensor(dst, dtype=torch.int64)
        self._n = int(num_nodes)
        self.ndata = {}
        self.edata = {}

    def num_nodes(self):
        return self._n

    def num_edges(self):
        return int(self._src.numel())

    def edges(self):
        return self._src, self._dst

    def to(self, device):
        return self

    @contextlib.contextmanager
    def local_scope(self):
        nd, ed = dict(self.ndata), dict(self.edata)
        try:
            yield
        finally:
            self.ndata, self.edata = nd, ed

    def apply_edges(self, func):
        if isinstance(func, function._UAddV):
            self.edata[func.out] = self.ndata[func.u][self._src] + self.ndata[func.v][self._dst]
        else:
            self.edata.update(func(_EdgeBatch(self)))

    def update_all(self, msg, red):
        assert isinstance(red, function._Sum) and red.msg == msg.out
        if isinstance(msg, function._UMulE):
            m = self.ndata[msg.u][self._src] * self.edata[msg.e]
        elif isinstance(msg, function._CopyE):
            m = self.edata[msg.e]
        else:
            raise NotImplementedError(type(msg))
        out = torch.zeros((self._n,) + tuple(m.shape[1:]), dtype=m.dtype)
        out.index_add_(0, self._dst, m)
        self.ndata[red.out] = out


def graph(data, num_nodes=None):
    src, dst = data
    return DGLGraph(src, dst, num_nodes)


def reverse(g, copy_ndata=True, copy_edata=False):
    r = DGLGraph(g._dst, g._src, g._n)
    if copy_ndata:
        r.ndata = dict(g.ndata)
    if copy_edata:
        r.edata = dict(g.edata)
    return r


def seed(_):
    pass
