"""r05 odds and ends on the GPU: the C-ABI collectives (sgf_comm_*) on a real RCCL communicator, a small MULTIGRAPH whose row is
longer than the long-row threshold, the general sampler fallback."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_sgf_comm_on_a_one_rank_rccl_communicator(cuda):
    """sgf_comm_* (csrc/comm.hip) with world = 1: librccl.so is dlopen'ed, a real communicator is created on this GPU, and the
    three collectives run on the caller's stream — all-reduce(sum) of one rank is the identity, all-gather / all-to-all of one
    rank are copies.  (More ranks need more GPUs: tests/test_gpu_rccl.py runs torch.distributed's RCCL path when they exist.)"""
    from sgformer_amd import _lib
    lib = _lib.load()
    if not lib.sgf_comm_available():
        pytest.skip("librccl.so not loadable on this box")
    torch.cuda.set_device(cuda)
    idb = ctypes.create_string_buffer(lib.sgf_comm_unique_id_bytes())
    _lib.call("sgf_comm_unique_id", idb)
    comm = ctypes.c_void_p()
    _lib.call("sgf_comm_create", ctypes.byref(comm), 1, 0, idb)
    try:
        st = ctypes.c_void_p(torch.cuda.current_stream(cuda).cuda_stream)
        x = torch.randn(65794, device=cuda)                      # d^2 + d + 2 floats at d = 256: the attention payload
        want = x.clone()
        _lib.call("sgf_comm_all_reduce_f32", comm, ctypes.c_void_p(x.data_ptr()), x.numel(), st)
        torch.cuda.synchronize()
        assert torch.equal(x, want)
        src = torch.randn(1000, 256, device=cuda).bfloat16()
        dst = torch.empty_like(src)
        nbytes = src.numel() * 2
        _lib.call("sgf_comm_all_gather", comm, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), nbytes, st)
        torch.cuda.synchronize()
        assert torch.equal(dst, src)
        dst.zero_()
        off = (ctypes.c_int64 * 1)(0)
        cnt = (ctypes.c_int64 * 1)(nbytes)
        _lib.call("sgf_comm_all_to_all", comm, ctypes.c_void_p(src.data_ptr()), off, cnt, ctypes.c_void_p(dst.data_ptr()), off, cnt, st)
        torch.cuda.synchronize()
        assert torch.equal(dst, src)
    finally:
        _lib.call("sgf_comm_destroy", comm)


def test_small_multigraph_with_a_row_longer_than_the_long_row_threshold(cuda):
    """sgf_csr_build keeps duplicate edges (large/ours.py:33 does not coalesce), so a graph of fewer nodes than the
    long-row threshold can still have a row LONGER than it; the segment count must not be short-circuited by the node count."""
    from oracle import sgformer_oracle as O
    from sgformer_amd import ops
    n = 300
    g = torch.Generator().manual_seed(1)
    src = torch.cat([torch.randint(0, n, (3000,), generator=g), torch.randint(0, n, (2500,), generator=g)])
    dst = torch.cat([torch.randint(0, n, (3000,), generator=g), torch.full((2500,), 7)])       # node 7: a row of > 2500 entries
    ei = torch.stack([src, dst])
    graph = ops.CSRGraph(ei.to(cuda), n)
    assert int((graph.rowptr[1:] - graph.rowptr[:-1]).max()) > ops.LONG_ROW and graph.long_segments > 0
    for dtype, tol in ((torch.float32, 2e-6), (torch.bfloat16, 1e-2)):
        x = torch.randn(n, 64, generator=g).to(dtype)
        y = ops.spmm_on(graph, x.to(cuda), False)
        rowptr, colind, val, _ = O.csr_build(ei.numpy(), n)
        ref = np.zeros((n, 64))
        xn = x.double().numpy()
        for i in range(n):
            for e in range(rowptr[i], rowptr[i + 1]):
                ref[i] += float(val[e]) * xn[colind[e]]
        err = np.abs(y.double().cpu().numpy() - ref).max() / max(1.0, np.abs(ref).max())
        assert err <= tol, (dtype, err)


def test_sampler_falls_back_to_hops_for_huge_fanouts(cuda, monkeypatch):
    """a fan-out list whose worst-case batch does not fit the one-call buffers is sampled hop by hop (buffers sized
    from the real frontier) — same contract: seeds first, edges point to nodes of earlier hops, local ids in range."""
    from sgformer_amd import sampling, synth
    n = 20000
    ei = synth.synthetic_graph(n, 12.0, seed=2)
    monkeypatch.setattr(sampling, "_ONE_CALL_MAX_ENTRIES", 1000)
    s = sampling.NeighborSampler(ei.to(cuda), n, [5, 4, 3], seed=3, device=cuda)
    seeds = torch.arange(0, 64, device=cuda)
    n_id, e, bs = s.sample(seeds)
    assert bs == 64 and torch.equal(n_id[:64].cpu(), torch.arange(64))
    assert int(e.max()) < n_id.numel() and int(e.min()) >= 0 and n_id.unique().numel() == n_id.numel()
    deg = torch.bincount(e[1].cpu(), minlength=n_id.numel())
    assert int(deg.max()) <= 5
    n_id2, e2, _ = s.sample(seeds)          # the node table was reset: a second batch is consistent too
    assert int(e2.max()) < n_id2.numel()


@pytest.mark.parametrize("kind", ["undirected", "directed_multigraph", "no_edges_kept"])
def test_subgraph_from_the_parent_csr_is_bit_exact(cuda, kind):
    """sgf_subgraph_csr_* (r05): the induced subgraph + its normalised CSR straight from the parent's CSR == bit for bit what
    sgf_csr_build / the oracle's csr_build give for torch_geometric's subgraph(subset, edge_index, relabel_nodes=True); the
    emitted edge list is the same multiset of edges as the edge-order-preserving sgf_subgraph_* result; the node table is
    restored; a parent with A^T == A promises symmetric batches, a directed one does not."""
    from oracle import sgformer_oracle as O
    from sgformer_amd import batching, ops, synth
    n = 5000
    g = torch.Generator().manual_seed(3)
    if kind == "undirected":
        ei = synth.synthetic_graph(n, 14.0, seed=5)
    elif kind == "directed_multigraph":
        src, dst = torch.randint(0, n, (40000,), generator=g), torch.randint(0, n, (40000,), generator=g)
        ei = torch.stack([torch.cat([src, src[:5000]]), torch.cat([dst, dst[:5000]])])       # duplicates, self-loops, no symmetry
    else:
        ei = torch.stack([torch.arange(0, n - 1), torch.arange(1, n)])
    subset = torch.randperm(n, generator=g)[: (1200 if kind != "no_edges_kept" else 40)]
    if kind == "no_edges_kept":
        subset = subset[(subset % 7) == 0][:20] * 1          # a few scattered nodes of a path: (almost) no edge survives
    eid, subd = ei.to(cuda), subset.to(cuda)
    batching._parents.clear()
    out, _ = batching.subgraph(subd, eid, num_nodes=n, relabel_nodes=True)
    assert hasattr(out, "_sgf_csr") and out._sgf_trusted
    m = subset.numel()
    # the reference semantics on the host: mask + relabel, original order
    pos = torch.full((n,), -1, dtype=torch.long)
    pos[subset] = torch.arange(m)
    keep = (pos[ei[0]] >= 0) & (pos[ei[1]] >= 0)
    ref_ei = pos[ei[:, keep]]
    assert out.shape == ref_ei.shape
    key = lambda e: torch.sort(e[1] * m + e[0])[0]       # noqa: E731
    assert torch.equal(key(out.cpu()), key(ref_ei))
    rowptr, colind, val, deg = O.csr_build(ref_ei.numpy(), m)
    rb, cb, vb, db = (t.cpu().numpy() for t in out._sgf_csr)
    assert np.array_equal(rb, rowptr) and np.array_equal(cb, colind.astype(np.int32)) and np.array_equal(db, deg.astype(np.int32))
    assert np.array_equal(vb.view(np.uint32), val.view(np.uint32))
    assert out._sgf_max_in_degree == (int(np.diff(rowptr).max()) if m else 0)     # total[2] of sgf_subgraph_csr_plan
    # the emitted edge list is in (target, source) order: exactly the CSR's
    assert np.array_equal(out[1].cpu().numpy(), np.repeat(np.arange(m), np.diff(rowptr))) and np.array_equal(out[0].cpu().numpy(), colind)
    parent = next(iter(batching._parents.values()))[1]
    assert int((parent.local_of != -1).sum()) == 0
    assert parent.symmetric == (kind == "undirected") and bool(getattr(out, "_sgf_symmetric", False)) == (kind == "undirected")
    # ops.CSRGraph adopts the arrays (no second sort) and the SpMM on them equals the SpMM on a freshly built CSR
    gr = ops.CSRGraph(out, m)
    assert gr.rowptr.data_ptr() == out._sgf_csr[0].data_ptr()
    fresh = ops.CSRGraph(ref_ei.to(cuda), m)
    x = torch.randn(m, 64, generator=g).to(cuda)
    assert torch.equal(ops.spmm_on(gr, x, False), ops.spmm_on(fresh, x, False))
    if kind == "undirected":
        assert gr.symmetric is True and gr.transposed()[0].data_ptr() == gr.rowptr.data_ptr()
    # a subset that repeats a node takes the general path (and leaves the table clean)
    dup = torch.cat([subd[:10], subd[:3]])
    out2, _ = batching.subgraph(dup, eid, num_nodes=n, relabel_nodes=True)
    assert not hasattr(out2, "_sgf_csr") and int((parent.local_of != -1).sum()) == 0
