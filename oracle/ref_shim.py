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


def load_reference(variant: str = "large"):
    """Return the reference's `ours` module for variant 'large' or '100M', executed unchanged."""
    if not reference_available():
        raise FileNotFoundError(f"reference not found under {REFERENCE_ROOT}")
    install_stand_ins()
    path = os.path.join(REFERENCE_ROOT, variant, "ours.py")
    spec = importlib.util.spec_from_file_location(f"_sgf_reference_{variant}_ours", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
