"""Import the reference's model files UNCHANGED (test infrastructure; works only where
/root/reference exists, i.e. in the build container — never on the GPU box).

large/ours.py and 100M/ours.py import two un-vendored third-party packages
(large/requirements.txt:8,10): `torch_sparse.{SparseTensor, matmul}` and
`torch_geometric.utils.degree`.  Neither is installed and there is no network, so minimal
stand-ins restating their published semantics are registered in
sys.modules before the import:
  * SparseTensor(row, col, value, sparse_sizes): COO entries sorted by row*ncols+col (stable),
    duplicates kept (torch_sparse 0.6.10 SparseStorage);
  * matmul(adj, x): sum-reduce SpMM, autograd flows to x only (here: index_add over the entries);
  * degree(index, N): zeros(N).scatter_add_(0, index, ones).
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import torch

REFERENCE_ROOT = os.environ.get("SGF_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "large", "ours.py"))


class _SparseTensor:
    def __init__(self, row=None, col=None, value=None, sparse_sizes=None, is_sorted=False):
        n_rows, n_cols = sparse_sizes
        if not is_sorted:
            perm = torch.argsort(row * n_cols + col, stable=True)
            row, col = row[perm], col[perm]
            value = value[perm] if value is not None else None
        self.row, self.col, self.value, self.sizes = row, col, value, (n_rows, n_cols)

    def sparse_sizes(self):
        return self.sizes


def _matmul(adj, x, reduce="sum"):
    assert reduce == "sum"
    out = torch.zeros((adj.sizes[0], x.shape[1]), dtype=x.dtype, device=x.device)
    return out.index_add(0, adj.row, adj.value.to(x.dtype).unsqueeze(1) * x[adj.col])


def _degree(index, num_nodes=None, dtype=None):
    n = int(index.max()) + 1 if num_nodes is None else num_nodes
    out = torch.zeros((n,), dtype=dtype if dtype is not None else torch.get_default_dtype(),
                      device=index.device)
    return out.scatter_add_(0, index, torch.ones((index.numel(),), dtype=out.dtype, device=index.device))


def _gcn_norm(edge_index, edge_weight=None, num_nodes=None, improved=False, add_self_loops=True,
              dtype=None):
    """torch_geometric 1.7.2 nn/conv/gcn_conv.py gcn_norm for a LongTensor edge_index (SURVEY.md
    App. D vii): add_remaining_self_loops(fill 1 / 2), deg = scatter_add(w, col),
    norm = deg^-1/2[row] * w * deg^-1/2[col] with inf -> 0."""
    n = int(edge_index.max()) + 1 if num_nodes is None else num_nodes
    if edge_weight is None:
        edge_weight = torch.ones((edge_index.size(1),), dtype=dtype or torch.get_default_dtype(),
                                 device=edge_index.device)
    if add_self_loops:
        fill = 2.0 if improved else 1.0
        row, col = edge_index[0], edge_index[1]
        mask = row != col
        loop_w = torch.full((n,), fill, dtype=edge_weight.dtype, device=edge_index.device)
        loop_w[row[~mask]] = edge_weight[~mask]            # an existing self-loop keeps its weight
        loops = torch.arange(n, dtype=edge_index.dtype, device=edge_index.device)
        edge_index = torch.cat([edge_index[:, mask], torch.stack([loops, loops])], dim=1)
        edge_weight = torch.cat([edge_weight[mask], loop_w])
    row, col = edge_index[0], edge_index[1]
    deg = torch.zeros(n, dtype=edge_weight.dtype, device=edge_index.device).scatter_add_(0, col, edge_weight)
    dis = deg.pow(-0.5)
    dis.masked_fill_(dis == float("inf"), 0)
    return edge_index, dis[row] * edge_weight * dis[col]


class _MessagePassing(torch.nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()


class _GCNConv(_MessagePassing):
    """torch_geometric 1.7.2 GCNConv (defaults: improved=False, add_self_loops=True, normalize=True,
    bias=True): out = A_norm (x @ weight) + bias, weight [in, out] glorot, bias zeros."""

    def __init__(self, in_channels, out_channels, improved=False, cached=False, add_self_loops=True,
                 normalize=True, bias=True, **kwargs):
        super().__init__()
        assert normalize and add_self_loops and not improved and bias
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = torch.nn.Parameter(torch.empty(in_channels, out_channels))
        self.bias = torch.nn.Parameter(torch.empty(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        a = (6.0 / (self.weight.size(-2) + self.weight.size(-1))) ** 0.5
        self.weight.data.uniform_(-a, a)
        self.bias.data.fill_(0)

    def forward(self, x, edge_index, edge_weight=None):
        ei, w = _gcn_norm(edge_index, edge_weight, x.size(0), dtype=x.dtype)
        x = x @ self.weight
        out = torch.zeros_like(x).index_add(0, ei[1], w.to(x.dtype).unsqueeze(1) * x[ei[0]])
        return out + self.bias


def _placeholder(name):
    class _P(torch.nn.Module):
        def __init__(self, *a, **k):
            raise NotImplementedError(f"stand-in: torch_geometric.nn.{name} is not on the sgformer path")
    _P.__name__ = name
    return _P


def install_stand_ins():
    if "torch_sparse" not in sys.modules:
        ts = types.ModuleType("torch_sparse")
        ts.SparseTensor, ts.matmul = _SparseTensor, _matmul
        sys.modules["torch_sparse"] = ts
    if "torch_geometric" not in sys.modules:
        tg = types.ModuleType("torch_geometric")
        tgu = types.ModuleType("torch_geometric.utils")
        tgu.degree = _degree
        tg.utils = tgu
        sys.modules["torch_geometric"] = tg
        sys.modules["torch_geometric.utils"] = tgu
    if "torch_geometric.nn" not in sys.modules:   # medium/models.py:6-7 imports these at module level
        tg = sys.modules["torch_geometric"]
        tgn = types.ModuleType("torch_geometric.nn")
        tgn.GCNConv, tgn.MessagePassing = _GCNConv, _MessagePassing
        for nm in ("SGConv", "GATConv", "JumpingKnowledge", "APPNP"):
            setattr(tgn, nm, _placeholder(nm))
        tgc = types.ModuleType("torch_geometric.nn.conv")
        tgg = types.ModuleType("torch_geometric.nn.conv.gcn_conv")
        tgg.gcn_norm = _gcn_norm
        tgc.gcn_conv = tgg
        tgn.conv = tgc
        if not hasattr(tg, "nn"):
            tg.nn = tgn
        sys.modules.setdefault("torch_geometric.nn", tgn)
        sys.modules.setdefault("torch_geometric.nn.conv", tgc)
        sys.modules.setdefault("torch_geometric.nn.conv.gcn_conv", tgg)


def _exec(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_reference(variant: str = "large"):
    """Return the reference's `ours` module for variant 'large', '100M' or 'medium', executed
    unchanged.  medium/ours.py:9 does `from models import GCN`: medium/models.py is executed unchanged
    too (over the GCNConv stand-in) and registered as `models` for the duration of the import."""
    if not reference_available():
        raise FileNotFoundError(f"reference not found under {REFERENCE_ROOT}")
    install_stand_ins()
    if variant == "difformer":      # medium/difformer.py (self-contained: torch_sparse + PyG degree only)
        return _exec(os.path.join(REFERENCE_ROOT, "medium", "difformer.py"), "_sgf_reference_medium_difformer")
    path = os.path.join(REFERENCE_ROOT, variant, "ours.py")
    if variant != "medium":
        return _exec(path, f"_sgf_reference_{variant}_ours")
    saved = sys.modules.get("models")
    sys.modules["models"] = _exec(os.path.join(REFERENCE_ROOT, "medium", "models.py"), "_sgf_reference_medium_models")
    try:
        mod = _exec(path, "_sgf_reference_medium_ours")
        mod.models = sys.modules["models"]
        return mod
    finally:
        if saved is None:
            del sys.modules["models"]
        else:
            sys.modules["models"] = saved
