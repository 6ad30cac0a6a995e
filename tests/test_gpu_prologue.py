"""N2: the trainer prologue on the GPU — to_undirected / remove_self_loops /
add_self_loops of large/main.py:75-79 and 100M/nb-sample.py:79-80 — bit-exact against the numpy
restatement (oracle/graph_oracle.py) and against the PyG-semantics stand-ins the unchanged trainers run
with here (tests/standins/torch_geometric/utils)."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import graph_oracle as G

pytestmark = pytest.mark.gpu

STANDINS = os.path.join(os.path.dirname(__file__), "standins")


def _raw_edges(n, m, seed, loops=True, dups=True):
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(0, n, (m,), generator=g)
    dst = torch.randint(0, n, (m,), generator=g)
    if loops:
        dst[:: 7] = src[:: 7]                      # self-loops to drop
    ei = torch.stack([src, dst])
    if dups:
        ei = torch.cat([ei, ei[:, : m // 5]], dim=1)   # duplicate edges to coalesce
    return ei


@pytest.mark.parametrize("n,m,seed", [(50, 400, 0), (1000, 20000, 1), (30000, 700000, 2), (7, 0, 3)])
def test_full_prologue_bit_exact(cuda, n, m, seed):
    from sgformer_amd import batching
    ei = _raw_edges(n, m, seed)
    out = batching.graph_prologue(ei.to(cuda), n)                    # undirected + drop loops + add loops
    ref = G.graph_prologue(ei.numpy(), n, undirected=True)
    assert out.dtype == torch.int64 and out.is_cuda
    assert np.array_equal(out.cpu().numpy(), ref)
    # --directed runs (large/parse.py:57): no symmetrisation, the caller's edge order is kept
    out_d = batching.graph_prologue(ei.to(cuda), n, undirected=False)
    assert np.array_equal(out_d.cpu().numpy(), G.graph_prologue(ei.numpy(), n, undirected=False))


def test_pyg_signatures_match_the_standins(cuda):
    """The three functions the launcher installs over torch_geometric.utils, one by one, as the trainer
    calls them (to_undirected WITHOUT num_nodes, large/main.py:76)."""
    from sgformer_amd import batching
    sys.path.insert(0, STANDINS)
    try:
        import importlib
        tgu = importlib.import_module("torch_geometric.utils")
        ei = _raw_edges(500, 6000, 11)
        a = batching.to_undirected(ei.to(cuda))
        assert torch.equal(a.cpu(), tgu.to_undirected(ei))
        b, none_b = batching.remove_self_loops(a)
        assert none_b is None and torch.equal(b.cpu(), tgu.remove_self_loops(a.cpu())[0])
        c, none_c = batching.add_self_loops(b, num_nodes=500)
        assert none_c is None and torch.equal(c.cpu(), tgu.add_self_loops(b.cpu(), num_nodes=500)[0])
        # 100M/nb-sample.py:79-80: to_undirected then add_self_loops, existing self-loops KEPT (duplicates later)
        d, _ = batching.add_self_loops(batching.to_undirected(ei.to(cuda)), num_nodes=500)
        assert torch.equal(d.cpu(), tgu.add_self_loops(tgu.to_undirected(ei), num_nodes=500)[0])
        # edge attributes ride along on the plain path
        w = torch.rand(ei.shape[1])
        e2, w2 = batching.remove_self_loops(ei, w)
        r2, rw2 = tgu.remove_self_loops(ei, w)
        assert torch.equal(e2, r2) and torch.equal(w2, rw2)
    finally:
        sys.path.remove(STANDINS)


def test_prologue_feeds_the_csr_build(cuda):
    """The prologue's output is what model(x, edge_index) receives: the CSR built from it equals the CSR of
    the host-side prologue (T1 bit-exact contract downstream of N2)."""
    from sgformer_amd import batching, ops
    from oracle import sgformer_oracle as O
    n = 4000
    ei = _raw_edges(n, 50000, 5)
    out = batching.graph_prologue(ei.to(cuda), n)
    g = ops.CSRGraph(out, n)
    rowptr, colind, val, _ = O.csr_build(G.graph_prologue(ei.numpy(), n), n)
    assert np.array_equal(g.rowptr.cpu().numpy(), rowptr) and np.array_equal(g.colind.cpu().numpy(), colind)
    assert np.array_equal(g.val.cpu().numpy().view(np.uint32), val.view(np.uint32))
    g.transposed()
    assert g.symmetric
