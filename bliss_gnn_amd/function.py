"""The ``dgl.function`` builtins the reference's models pass to ``apply_edges`` / ``update_all`` (model.py:82, :98; DGL's
SAGEConv / GraphConv use copy_u / u_mul_e with sum / mean internally): descriptors only, the work is done by
``Block.apply_edges`` / ``Block.update_all`` on the gfx950 kernels."""


class _Message:
    def __init__(self, kind, lhs, rhs, out):
        self.kind, self.lhs, self.rhs, self.out = kind, lhs, rhs, out


class _Reduce:
    def __init__(self, kind, msg, out):
        self.kind, self.msg, self.out = kind, msg, out


def u_add_v(lhs_field, rhs_field, out):
    """edge[out] = src[lhs_field] + dst[rhs_field]   (model.py:82)."""
    return _Message("u_add_v", lhs_field, rhs_field, out)


def u_mul_e(lhs_field, rhs_field, out):
    """message[out] = src[lhs_field] * edge[rhs_field]   (model.py:98)."""
    return _Message("u_mul_e", lhs_field, rhs_field, out)


def copy_u(u, out):
    return _Message("copy_u", u, None, out)


def sum(msg, out):       # noqa: A001  (dgl.function.sum)
    return _Reduce("sum", msg, out)


def mean(msg, out):
    return _Reduce("mean", msg, out)
