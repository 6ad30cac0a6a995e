"""Row N1 of SURVEY.md §8f: the induced-subgraph step of the reference's random-partition mini-batch
trainer (large/main-batch.py:134-141, evaluator twin large/eval.py:80-96) on the GPU.

`subgraph` has the signature and semantics of torch_geometric.utils.subgraph (PyG 1.7.2) — keep the
edges whose two endpoints are in `subset`, in their original order, optionally relabel node
subset[j] -> j — but runs sgf_subgraph_* (sgformer_amd/csrc/subgraph.hip).  The reference calls the
PyG function on the HOST for every batch: an O(E) pass over all 126 M edges of ogbn-products per
100 k-node batch.  Here the full edge_index is staged to the GPU once (cached per tensor), each call
is two streaming passes over it at HBM speed, and the result is returned ON THE GPU (the trainer's
next line is `.to(device)`, a no-op then).  `sgformer_amd.launch` installs it over
`torch_geometric.utils.subgraph` for the `main-batch.py` trainer, which stays byte-for-byte
unchanged.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Optional

import torch

from . import ops

_resident: "OrderedDict[tuple, tuple]" = OrderedDict()


def _on_gpu(t: torch.Tensor, device) -> torch.Tensor:
    """Device copy of a (host) tensor, cached on identity + version so the full edge_index crosses
    PCIe once, not once per batch.  The cache entry pins the source tensor."""
    if t.is_cuda:
        return t.contiguous()
    key = (t.data_ptr(), t._version, tuple(t.shape), str(device))
    hit = _resident.get(key)
    if hit is not None:
        _resident.move_to_end(key)
        return hit[1]
    d = t.contiguous().to(device)
    _resident[key] = (t, d)
    while len(_resident) > 2:
        _resident.popitem(last=False)
    return d


class _Parent:
    """What is kept per PARENT edge list for the mini-batch path (one or two per process): its CSR (sgf_csr_build, once),
    whether A^T == A (checked once) and the node -> local-id table of sgf_subgraph_csr_*."""

    def __init__(self, ei_dev: torch.Tensor, n: int):
        g = ops.CSRGraph(ei_dev, n, validate=True)
        g.transposed()                                    # one-off: A^T == A ?
        self.rowptr, self.colind, self.n, self.nnz = g.rowptr, g.colind, n, g.nnz
        self.symmetric = bool(g.symmetric)
        self.local_of = torch.full((max(n, 1),), -1, dtype=torch.int32, device=ei_dev.device)


_parents: "OrderedDict[tuple, tuple]" = OrderedDict()


def _parent_of(edge_index: torch.Tensor, ei_dev: torch.Tensor, n: int) -> _Parent:
    key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), str(ei_dev.device), n)
    hit = _parents.get(key)
    if hit is not None:
        _parents.move_to_end(key)
        return hit[1]
    p = _Parent(ei_dev, n)
    _parents[key] = (edge_index, p)                       # (pins the key tensor: a recycled data_ptr cannot alias it)
    while len(_parents) > 2:
        _parents.popitem(last=False)
    return p


def _use_parent_csr() -> bool:
    import os
    return hasattr(ops.K, "subgraph_csr") and os.environ.get("SGF_SUBGRAPH_CSR", "1") != "0"


def subgraph(subset, edge_index, edge_attr=None, relabel_nodes: bool = False, num_nodes: Optional[int] = None):
    """torch_geometric.utils.subgraph — see _subgraph.  On the GPU the whole call (the index's H2D copy, the kernels, the one
    device -> host read of the edge count) runs on the PREP stream (sgformer_amd/staging.py): the host read then waits for
    the batch's own few kernels, not for the previous batch's backward; the compute stream takes the results through an event."""
    from . import staging
    if (ops.K.name == "hip" and staging.enabled() and torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing()):
        dev = edge_index.device if edge_index.is_cuda else torch.device("cuda", torch.cuda.current_device())
        with staging.on_prep(dev) as hand_over:
            out, attr = _subgraph(subset, edge_index, edge_attr, relabel_nodes, num_nodes)
            hand_over(out, attr, *(getattr(out, "_sgf_csr", None) or ()))
        return out, attr
    return _subgraph(subset, edge_index, edge_attr, relabel_nodes, num_nodes)


def _subgraph(subset, edge_index, edge_attr=None, relabel_nodes: bool = False,
              num_nodes: Optional[int] = None):
    """torch_geometric.utils.subgraph(subset, edge_index, edge_attr, relabel_nodes, num_nodes).

    relabel_nodes=True without edge attributes — the mini-batch trainer's call (large/main-batch.py:139) — takes the parent's
    cached CSR (sgf_subgraph_csr_*): the batch's m parent rows are read instead of all parent edges, and the result carries its
    normalised CSR (`_sgf_csr`, adopted by ops.CSRGraph instead of a second sort) and, for a parent with A^T == A, the promise
    that it is symmetric too.  Its edges come in (target, source) order — the same multiset as the reference's, whose order
    (the parent's edge order) nothing on the path depends on; SGF_SUBGRAPH_CSR=0 keeps the edge-order-preserving passes."""
    if ops.K.name == "hip":
        if not torch.cuda.is_available():
            ops.K.check(edge_index)   # raises the standard no-CPU-path error
        device = edge_index.device if edge_index.is_cuda else torch.device("cuda", torch.cuda.current_device())
    else:
        device = edge_index.device
    n = int(num_nodes) if num_nodes is not None else (int(edge_index.max()) + 1 if edge_index.numel() else 0)
    if isinstance(subset, (list, tuple)):
        subset = torch.tensor(subset, dtype=torch.long)
    subset = subset.to(device)
    if subset.dtype == torch.bool:
        subset = subset.nonzero().view(-1)
    ei = _on_gpu(edge_index, device) if ops.K.name == "hip" else edge_index
    if relabel_nodes and edge_attr is None and _use_parent_csr() and ei.shape[1] > 0:
        parent = _parent_of(edge_index, ei, n)
        sub = subset.contiguous().long()
        got = ops.K.subgraph_csr(parent.rowptr, parent.colind, n, sub, parent.local_of, True)
        if got is not None:
            rowptr_b, colind_b, val_b, deg_b, out, longest = got
            out._sgf_trusted = True
            out._sgf_max_in_degree = longest              # (read with the batch's size: no long-row path for most batches)
            out._sgf_csr = (rowptr_b, colind_b, val_b, deg_b)
            out._sgf_symmetric = parent.symmetric
            return out, None
    out, eid = ops.K.subgraph(ei, n, subset.contiguous().long(), bool(relabel_nodes), edge_attr is not None)
    if relabel_nodes:
        out._sgf_trusted = True       # ids are positions in `subset`: CSRGraph skips its range check (a host sync)
    if edge_attr is not None:
        edge_attr = edge_attr.to(device)[eid]
    return out, edge_attr


# ------------------------------------------------------------------------------------------------
# Row N2 of SURVEY.md §8f: the trainer's graph prologue (large/main.py:75-79, 100M/nb-sample.py:79-80)
# ------------------------------------------------------------------------------------------------
def _device_for(edge_index):
    if ops.K.name == "hip":
        if not torch.cuda.is_available():
            ops.K.check(edge_index)   # raises the standard no-CPU-path error
        return edge_index.device if edge_index.is_cuda else torch.device("cuda", torch.cuda.current_device())
    return edge_index.device


def _nodes(edge_index, num_nodes):
    if num_nodes is not None:
        return int(num_nodes)
    return int(edge_index.max()) + 1 if edge_index.numel() else 0


def graph_prologue(edge_index, num_nodes=None, undirected=True, remove_loops=True, add_loops=True):
    """to_undirected -> remove_self_loops -> add_self_loops (each optional) in ONE device pass
    (sgf_graph_prologue_*: one radix sort of the symmetrised keys + a flag scan).  Returns the new
    edge_index ON THE GPU."""
    dev = _device_for(edge_index)
    n = _nodes(edge_index, num_nodes)
    ei = edge_index.to(dev).contiguous()
    if ei.dtype != torch.int64:
        ei = ei.long()
    return ops.K.graph_prologue(ei, n, bool(undirected), bool(remove_loops), bool(add_loops))


def to_undirected(edge_index, num_nodes=None):
    """torch_geometric.utils.to_undirected (PyG 1.7.2): both directions, coalesced (sorted by row*N+col)."""
    return graph_prologue(edge_index, num_nodes, True, False, False)


def remove_self_loops(edge_index, edge_attr=None):
    """torch_geometric.utils.remove_self_loops: entries with row != col, order kept."""
    if edge_attr is not None:          # attributes ride along: plain mask (not on the sgformer recipes' path)
        mask = edge_index[0] != edge_index[1]
        return edge_index[:, mask], edge_attr[mask]
    return graph_prologue(edge_index, 0, False, True, False), None


def add_self_loops(edge_index, edge_weight=None, fill_value=1.0, num_nodes=None):
    """torch_geometric.utils.add_self_loops: (i, i) for every node appended at the end."""
    n = _nodes(edge_index, num_nodes)
    if edge_weight is not None:
        loops = torch.arange(n, dtype=edge_index.dtype, device=edge_index.device)
        return (torch.cat([edge_index, torch.stack([loops, loops])], dim=1),
                torch.cat([edge_weight, edge_weight.new_full((n,), fill_value)]))
    return graph_prologue(edge_index, n, False, False, True), None
