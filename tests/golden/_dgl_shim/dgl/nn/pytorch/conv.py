"""layers/processor.py:1 of the reference imports these unconditionally; the hot path never builds them."""


class _Unused:
    def __init__(self, *a, **k):
        raise NotImplementedError("not on the SymGatedGCN path")


GraphConv = GATConv = SAGEConv = _Unused
