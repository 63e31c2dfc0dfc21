"""dgl.function builtins used by layers/gated_gcn_full.py:104-126 of the reference."""


class _UAddV:
    def __init__(self, u, v, out):
        self.u, self.v, self.out = u, v, out


class _UMulE:
    def __init__(self, u, e, out):
        self.u, self.e, self.out = u, e, out


class _CopyE:
    def __init__(self, e, out):
        self.e, self.out = e, out


class _Sum:
    def __init__(self, msg, out):
        self.msg, self.out = msg, out


def u_add_v(u, v, out):
    return _UAddV(u, v, out)


def u_mul_e(u, e, out):
    return _UMulE(u, e, out)


def copy_e(e, out):
    return _CopyE(e, out)


def sum(msg, out):  # noqa: A001 - mirrors dgl.function.sum
    return _Sum(msg, out)
