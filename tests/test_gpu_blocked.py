"""T2 with on-chip reuse: sgf_reorder, sgf_spmm_plan, sgf_spmm_blocked, sgf_gather_rows.

The planning kernels are deterministic, so they are compared ENTRY FOR ENTRY with the plain numpy
restatement in oracle/graph_oracle.py; the product they serve is the reference's
torch_sparse.matmul(adj, x) (large/ours.py:34), so the row-block kernel is compared with the fp64
oracle SpMM on the un-planned CSR (1e-6 relative, fp32 — the bar of tests/test_gpu_kernels.py), and
the module run on a re-ordered graph with the oracle model in the CALLER's node order.
"""
import numpy as np
import pytest
import torch

from oracle import graph_oracle as G
from oracle import sgformer_oracle as O

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


def _graphs():
    from sgformer_amd import synth
    out = {
        "community": synth.synthetic_graph_community(3000, 24.0, seed=2, comm_size=(24, 96), comms_per_super=8),
        "uniform": synth.synthetic_graph(1500, 9.0, seed=4),
        "directed": synth.synthetic_graph(1200, 7.0, seed=5, directed=True),
    }
    # isolated nodes, duplicate edges, a hub with > 1024 in-edges (long-row path), no self-loops on some nodes
    g = torch.Generator().manual_seed(9)
    hub_src = torch.randint(0, 2500, (1500,), generator=g)
    base = synth.synthetic_graph_community(2500, 12.0, seed=7, comm_size=(16, 64), comms_per_super=6)
    extra = torch.stack([torch.cat([hub_src, torch.tensor([5, 5, 5])]),
                         torch.cat([torch.full((1500,), 17), torch.tensor([9, 9, 9])])])
    ei = torch.cat([base, extra], dim=1)
    keep = (ei[0] < 2400) & (ei[1] < 2400) | (ei[1] == 17)         # nodes 2400.. become isolated
    out["hub_dups_isolated"] = ei[:, keep]
    return out


@pytest.mark.parametrize("name", ["community", "uniform", "directed", "hub_dups_isolated"])
def test_reorder_matches_oracle(cuda, name):
    from sgformer_amd import ops
    ei = _graphs()[name]
    n = int(ei.max()) + 1 if name != "hub_dups_isolated" else 2500
    perm, inv, comm = ops.K.reorder(ei.to(cuda), n, 6, 6)
    p_ref, i_ref, c_ref = G.reorder(ei.numpy(), n, 6, 6)
    assert np.array_equal(comm.cpu().numpy(), c_ref)
    assert np.array_equal(perm.cpu().numpy(), p_ref)
    assert np.array_equal(inv.cpu().numpy(), i_ref)
    assert np.array_equal(np.sort(perm.cpu().numpy()), np.arange(n))          # a permutation
    if name == "community":          # the planted communities (24-96 nodes) are recovered, not shattered or merged
        sizes = np.bincount(c_ref)
        assert 20 <= np.median(sizes) <= 120 and sizes.max() <= 400


@pytest.mark.parametrize("name,rpb,lds_rows", [("community", 128, 288), ("community", 64, 40), ("community", 8, 3),
                                               ("hub_dups_isolated", 128, 144), ("uniform", 32, 16)])
def test_spmm_plan_matches_oracle(cuda, name, rpb, lds_rows):
    from sgformer_amd import ops
    ei = _graphs()[name]
    n = 2500 if name == "hub_dups_isolated" else int(ei.max()) + 1
    _, inv, _ = G.reorder(ei.numpy(), n, 6, 6)
    ei2 = torch.from_numpy(inv.astype(np.int64))[ei]
    g = ops.CSRGraph(ei2.to(cuda), n)
    ecode, ev, nlds, sh_ptr, sh_cols, st = ops.K.spmm_plan(g.rowptr, g.colind, g.val, n, rpb, lds_rows, ops.LONG_ROW)
    r = G.spmm_plan(g.rowptr.cpu().numpy(), g.colind.cpu().numpy(), g.val.cpu().numpy(), n, rpb, lds_rows, ops.LONG_ROW)
    assert np.array_equal(sh_ptr.cpu().numpy(), r[3])
    assert np.array_equal(sh_cols.cpu().numpy()[: r[4].size], r[4])
    assert np.array_equal(nlds.cpu().numpy()[:n], r[2])
    assert np.array_equal(ecode.cpu().numpy(), r[0])
    assert np.array_equal(ev.cpu().numpy().view(np.uint32), r[1].view(np.uint32))
    assert st[0] == int(r[5][0]) and st[1] == int(r[5][1]) and st[3] == int(r[5][3])
    if name == "hub_dups_isolated":            # the hub row is left to the long-row path: plain source ids
        lens = np.diff(g.rowptr.cpu().numpy())
        hub = int(np.argmax(lens))
        assert lens[hub] > ops.LONG_ROW and r[2][hub] == 0
    if name == "community" and lds_rows >= 144:
        assert st[0] / st[3] > 0.5             # most entries of a community graph are served from LDS


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("name,rpb,lds_rows,d", [("community", 128, 144, 256), ("community", 64, 64, 128),
                                                 ("community", 8, 5, 100), ("hub_dups_isolated", 128, 144, 256),
                                                 ("uniform", 128, 144, 64), ("directed", 32, 20, 36),
                                                 # <= 64 rows per block, bf16, d % 8 == 0: k_spmm_blk2 (two entries per
                                                 # instruction, two blocks per CU)
                                                 ("community", 64, 144, 256), ("hub_dups_isolated", 64, 96, 256),
                                                 ("uniform", 16, 8, 64), ("directed", 64, 40, 200)])
def test_spmm_blocked_vs_oracle(cuda, dtype, name, rpb, lds_rows, d):
    """large/ours.py:34 — Y = A X through the LDS-staged kernel vs the fp64 oracle on the plain CSR."""
    from sgformer_amd import ops
    ei = _graphs()[name]
    n = 2500 if name == "hub_dups_isolated" else int(ei.max()) + 1
    g = ops.CSRGraph(ei.to(cuda), n)
    g.blocked = True
    plan = ops.BlockedPlan(g.rowptr, g.colind, g.val, n, dtype, rows_per_block=rpb, lds_rows=lds_rows)
    torch.manual_seed(1)
    x = torch.randn(n, d)
    xs = x.to(dtype)
    y = ops.K.spmm_blocked(g.rowptr, plan, xs.to(cuda), n, long_segments=g.long_segments)
    rowptr, colind, val, _ = O.csr_build(ei.numpy(), n)
    ref = O.spmm_csr(rowptr, colind, val, xs.double()) if hasattr(O, "spmm_csr") else None
    if ref is None:
        rows = np.repeat(np.arange(n), np.diff(rowptr))
        ref_np = np.zeros((n, d))
        np.add.at(ref_np, rows, val[:, None].astype(np.float64) * xs.double().numpy()[colind])
        ref = torch.from_numpy(ref_np)
    assert _rel(y.float(), ref) <= (1e-6 if dtype == torch.float32 else 3e-3)
    # and the plain kernel on the same CSR agrees to rounding (same entries, different summation order)
    y0 = ops.K.spmm(g.rowptr, g.colind, g.val, xs.to(cuda), n, long_segments=g.long_segments)
    assert _rel(y.float(), y0.float()) <= (1e-6 if dtype == torch.float32 else 6e-3)
    # into a column slice of a wider buffer (the sharded path's `out=`)
    if d % 8 == 0:
        wide = torch.zeros(n, 2 * d, dtype=dtype, device=cuda)
        ops.K.spmm_blocked(g.rowptr, plan, xs.to(cuda), n, out=wide[:, d:], long_segments=g.long_segments)
        assert torch.equal(wide[:, d:], y) and float(wide[:, :d].abs().max()) == 0.0


def test_spmm_blocked_follows_the_plan(cuda):
    """The kernel evaluated entry by entry through the plan (slot -> staged row -> source) in fp64."""
    from sgformer_amd import ops
    ei = _graphs()["community"][:, ::3]
    n = 3000
    g = ops.CSRGraph(ei.to(cuda), n)
    plan = ops.BlockedPlan(g.rowptr, g.colind, g.val, n, torch.float32, rows_per_block=16, lds_rows=9)
    x = torch.randn(n, 8)
    y = ops.K.spmm_blocked(g.rowptr, plan, x.to(cuda), n)
    ref = G.spmm_blocked(g.rowptr.cpu().numpy(), plan.ecode.cpu().numpy(), plan.eval.cpu().numpy(),
                         plan.nlds.cpu().numpy(), plan.sh_ptr.cpu().numpy(), plan.sh_cols.cpu().numpy(),
                         x.numpy(), 16)
    assert _rel(y, torch.from_numpy(ref)) <= 1e-6


@pytest.mark.parametrize("idx_dtype", [torch.int32, torch.int64])
@pytest.mark.parametrize("d,src_dtype,dst_dtype", [(100, torch.float32, torch.bfloat16), (65, torch.float32, torch.float32),
                                                   (256, torch.bfloat16, torch.bfloat16), (47, torch.bfloat16, torch.float32)])
def test_gather_rows(cuda, idx_dtype, d, src_dtype, dst_dtype):
    """x[idx] of large/main-batch.py:138 / the permutation at the module boundary: exact (a copy + cast)."""
    from sgformer_amd import ops
    g = torch.Generator().manual_seed(d)
    src = torch.randn(1000, d, generator=g).to(src_dtype)
    idx = torch.randint(0, 1000, (777,), generator=g)
    out = ops.gather_rows(src.to(cuda), idx.to(idx_dtype).to(cuda), dst_dtype)
    assert out.dtype == dst_dtype and torch.equal(out.cpu(), src[idx].to(dst_dtype))
    # permutation + autograd: the gradient is the gather by the inverse
    perm = torch.randperm(1000, generator=g)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(1000)
    xg = src.float().to(cuda).requires_grad_(True)
    w = torch.randn(1000, d, generator=g)
    y = ops.permute_rows(xg, perm.to(torch.int32).to(cuda), inv.to(torch.int32).to(cuda))
    (y * w.to(cuda)).sum().backward()
    assert torch.equal(y.detach().cpu(), src.float()[perm]) and torch.equal(xg.grad.cpu(), w[inv])


@pytest.mark.parametrize("blocked", ["0", "1"])
@pytest.mark.parametrize("name,dtype", [("community", torch.float32), ("directed", torch.float32),
                                        ("community", torch.bfloat16)])
def test_module_on_reordered_graph_matches_oracle(cuda, name, dtype, blocked, monkeypatch):
    """The whole module with the node order + row-block plan ADOPTED (mode 'always') against the fp64
    oracle in the caller's node order: logits 1e-4, gradients relative (large/ours.py:265-276)."""
    from sgformer_amd import ops
    from sgformer_amd.ours import SGFormer
    monkeypatch.setenv("SGF_SPMM_BLOCKED", blocked)      # the stream kernels / the LDS-staged row blocks
    cfg = dict(trans_num_layers=1, trans_num_heads=1, trans_use_act=False, gnn_num_layers=2, gnn_use_init=True,
               graph_weight=0.5)
    ei = _graphs()[name]
    n, f, d, c = int(ei.max()) + 1, 20, 64, 6
    torch.manual_seed(3)
    x = torch.randn(n, f)
    y = torch.randint(0, c, (n,))
    idx = torch.randperm(n)[: n // 2]
    p = O.init_params(cfg, f, d, c, seed=1)
    m = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0,
                 compute_dtype=None if dtype == torch.float32 else dtype, **cfg)
    m.load_state_dict({**m.state_dict(), **p})
    m = m.to(cuda).train()
    prev = ops.set_reorder_mode("always")
    try:
        ops.graph_cache.clear()
        eig = ei.to(cuda)
        logits = m(x.to(cuda), eig)
        view = ops.graph_cache.get(eig, n).view()
        assert view.perm is not None and view.stats["reordered"]
        loss = O.nll_loss(logits, y.to(cuda), idx.to(cuda))
        loss.backward()
    finally:
        ops.set_reorder_mode(prev)
        ops.graph_cache.clear()
    if dtype != torch.float32:
        # bf16: the re-ordered run against the SAME module on the caller's order (both bf16; what differs is
        # the row order of every reduction), logits to 2 bf16 ulps of their scale, gradients 3e-2 relative
        grads = {k: prm.grad.detach().clone() for k, prm in m.named_parameters() if prm.grad is not None}
        m.zero_grad(set_to_none=True)
        prev = ops.set_reorder_mode("never")
        try:
            logits0 = m(x.to(cuda), eig)
            O.nll_loss(logits0, y.to(cuda), idx.to(cuda)).backward()
        finally:
            ops.set_reorder_mode(prev)
            ops.graph_cache.clear()
        scale = float(logits0.detach().abs().max())
        assert float((logits.detach() - logits0.detach()).abs().max()) <= 2.0 ** -6 * max(scale, 1.0)
        gmax = max(float(g.norm()) for g in grads.values())
        for k, prm in m.named_parameters():
            if prm.grad is not None:
                assert float((grads[k] - prm.grad).norm()) <= 3e-2 * float(prm.grad.norm()) + 3e-3 * gmax, k
        return
    p64 = {k: v.double().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in p.items()}
    ref = O.sgformer_forward(p64, x.double(), ei, cfg, training=True)
    O.nll_loss(ref, y, idx).backward()
    assert float((logits.detach().double().cpu() - ref.detach()).abs().max()) <= 1e-4
    gmax = max(float(v.grad.norm()) for v in p64.values() if v.grad is not None)
    for k, prm in m.named_parameters():
        g = p64[k].grad
        if g is None:
            continue
        assert float((prm.grad.double().cpu() - g).norm()) <= 5e-4 * float(g.norm()) + 1e-6 * gmax, k


def test_uniform_graph_keeps_the_plain_kernel(cuda):
    """Policy: a graph without reuse (an expander) is not re-ordered — the row-block plan must serve a
    quarter of the entries from LDS to be adopted."""
    from sgformer_amd import ops, synth
    n = 120_000
    ei = synth.synthetic_graph(n, 12.0, seed=1, device=cuda)
    ops.graph_cache.clear()
    v = ops.prepare_graph(ei, n)
    assert v.perm is None and v.stats.get("why") == "no reuse to exploit" and v.stats["lds_fraction"] < 0.25
    ei2 = synth.synthetic_graph_community(n, 20.0, seed=1).to(cuda)
    v2 = ops.prepare_graph(ei2, n)
    assert v2.perm is not None and v2.stats["lds_fraction"] > 0.5
    ops.graph_cache.clear()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("name,d", [("community", 256), ("hub_dups_isolated", 256), ("directed", 136), ("uniform", 200),
                                    ("community", 64)])
def test_spmm_stream_vs_oracle(cuda, dtype, name, d):
    """sgf_spmm_stream (flattened stream; bf16: two rows per 16-byte-per-lane load, k_spmm_seg_bf16x2) vs the
    fp64 oracle SpMM of large/ours.py:34 — long rows, empty rows, odd row boundaries inside a pair included."""
    from sgformer_amd import ops
    ei = _graphs()[name]
    n = 2500 if name == "hub_dups_isolated" else int(ei.max()) + 1
    g = ops.CSRGraph(ei.to(cuda), n)
    torch.manual_seed(2)
    xs = torch.randn(n, d).to(dtype)
    y = ops.K.spmm(g.rowptr, g.colind, g.val, xs.to(cuda), n, long_segments=g.long_segments, stream_hint=True)
    rowptr, colind, val, _ = O.csr_build(ei.numpy(), n)
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    ref = np.zeros((n, d))
    np.add.at(ref, rows, val[:, None].astype(np.float64) * xs.double().numpy()[colind])
    assert _rel(y.float(), torch.from_numpy(ref)) <= (1e-6 if dtype == torch.float32 else 3e-3)
    y2 = ops.K.spmm(g.rowptr, g.colind, g.val, xs.to(cuda), n, long_segments=g.long_segments, stream_hint=True)
    assert torch.equal(y, y2)                                # deterministic
