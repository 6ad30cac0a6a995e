"""T2 on a re-ordered graph, matrix-core form: sgf_spmm_tile_blocks / _plan / _fill / sgf_spmm_tile (csrc/spmm_tile.hip).

The plan is deterministic, so its arrays are compared ENTRY FOR ENTRY (tiles: bit for bit) with the numpy
restatement in oracle/graph_oracle.py; the product it serves is the reference's torch_sparse.matmul(adj, x)
(large/ours.py:34), so the kernel is compared with the fp64 oracle SpMM on the un-planned CSR and the SAME bf16
inputs.  Tolerance: the output is rounded to bf16 once (2^-9 relative per element, ~2.3e-3 in Frobenius norm for
random data); the tile values carry 2^-17.
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
    from tests.test_gpu_blocked import _graphs as g
    return g()


def _reordered(name, cuda):
    """(CSRGraph in sgf_reorder order, n, comm_sorted numpy) from the ORACLE's order (so the test is independent of
    sgf_reorder)."""
    from sgformer_amd import ops
    ei = _graphs()[name]
    n = 2500 if name == "hub_dups_isolated" else int(ei.max()) + 1
    perm, inv, comm = G.reorder(ei.numpy(), n, 6, 6)
    ei2 = torch.from_numpy(inv.astype(np.int64))[ei]
    return ops.CSRGraph(ei2.to(cuda), n), n, comm[perm], ei2


@pytest.mark.parametrize("name,max_rows", [("community", 128), ("community", 64), ("community", 256), ("hub_dups_isolated", 128),
                                           ("uniform", 96), ("directed", 32)])
def test_tile_blocks_match_oracle(cuda, name, max_rows):
    from sgformer_amd import ops
    g, n, cs, _ = _reordered(name, cuda)
    blk = ops.K.tile_blocks(torch.from_numpy(cs.astype(np.int32)).to(cuda), n, max_rows, cuda).cpu().numpy()
    ref = G.tile_blocks(cs, n, max_rows)
    assert np.array_equal(blk, ref)
    rows = np.diff(blk)
    assert blk[0] == 0 and blk[-1] == n and rows.min() >= 1 and rows.max() <= max_rows
    fixed = ops.K.tile_blocks(None, n, max_rows, cuda).cpu().numpy()
    assert np.array_equal(fixed, G.tile_blocks(None, n, max_rows))


@pytest.mark.parametrize("name,max_rows,cap,min_count", [("community", 128, 512, 2), ("community", 64, 64, 3),
                                                         ("community", 256, 512, 3), ("hub_dups_isolated", 256, 256, 2),
                                                         ("hub_dups_isolated", 128, 256, 2), ("uniform", 128, 128, 2),
                                                         ("directed", 32, 32, 1)])
def test_tile_plan_matches_oracle(cuda, name, max_rows, cap, min_count):
    from sgformer_amd import ops
    g, n, cs, _ = _reordered(name, cuda)
    blk = G.tile_blocks(cs, n, max_rows)
    blk_d = torch.from_numpy(blk).to(cuda)
    sh_ptr, sh_cols, tile_ptr, tiles, rem_rowptr, rem_col, rem_val, st = ops.K.tile_plan(
        g.rowptr, g.colind, g.val, n, blk_d, cap, min_count, ops.LONG_ROW)
    r = G.tile_plan(g.rowptr.cpu().numpy(), g.colind.cpu().numpy(), g.val.cpu().numpy(), n, blk, cap, min_count,
                    ops.LONG_ROW)
    assert st[:6] == [int(v) for v in r[7][:6]] and st[7] == 0
    assert np.array_equal(sh_ptr.cpu().numpy(), r[0])
    assert np.array_equal(sh_cols.cpu().numpy()[: r[1].size], r[1])
    assert np.array_equal(tile_ptr.cpu().numpy(), r[2])
    nfrag = int(r[2][-1])
    got = tiles.cpu().numpy().view(np.uint16)[: nfrag * 1024]
    assert np.array_equal(got, r[3])                                  # hi / lo bf16 fragments, bit for bit
    assert np.array_equal(rem_rowptr.cpu().numpy(), r[4])
    nrem = int(r[4][-1])
    assert np.array_equal(rem_col.cpu().numpy()[:nrem], r[5])
    assert np.array_equal(rem_val.cpu().numpy()[:nrem].view(np.uint32), r[6].view(np.uint32))
    assert np.all(np.diff(r[0]) % 32 == 0)
    if name == "community" and cap >= 256:
        assert st[0] / st[3] > 0.6            # most entries of a community graph fall into the tiles
    if name == "hub_dups_isolated":           # the hub row stays on the gather path entirely
        lens = np.diff(g.rowptr.cpu().numpy())
        hub = int(np.argmax(lens))
        assert lens[hub] > ops.LONG_ROW and lens[hub] <= r[4][hub + 1] - r[4][hub] <= lens[hub] + 1


def _oracle_product(ei2, n, xs):
    rowptr, colind, val, _ = O.csr_build(ei2.numpy(), n)
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    ref = np.zeros((n, xs.shape[1]))
    np.add.at(ref, rows, val[:, None].astype(np.float64) * xs.double().numpy()[colind])
    return torch.from_numpy(ref)


@pytest.mark.parametrize("d", [256, 128])
@pytest.mark.parametrize("name,max_rows,cap,min_count", [("community", 128, 512, 2), ("community", 64, 64, 3),
                                                         ("community", 96, 32, 2), ("hub_dups_isolated", 128, 256, 2),
                                                         ("community", 256, 512, 3), ("hub_dups_isolated", 256, 128, 2),
                                                         ("uniform", 224, 64, 2),
                                                         ("uniform", 128, 128, 2), ("directed", 32, 32, 1),
                                                         ("directed", 128, 1024, 1)])
def test_spmm_tile_vs_oracle(cuda, name, max_rows, cap, min_count, d):
    """large/ours.py:34 — Y = A X through tiles + remainder vs the fp64 oracle on the plain CSR, bf16 storage."""
    from sgformer_amd import ops
    g, n, cs, ei2 = _reordered(name, cuda)
    g.blk_row = torch.from_numpy(G.tile_blocks(cs, n, max_rows)).to(cuda)
    plan = ops.TilePlan(g.rowptr, g.colind, g.val, n, g.blk_row, cap=cap, min_count=min_count)
    torch.manual_seed(1)
    xs = torch.randn(n, d).to(torch.bfloat16)
    y = ops.K.spmm_tile(plan, xs.to(cuda), n)
    ref = _oracle_product(ei2, n, xs)
    # one bf16 rounding of the result; compare with the same sum rounded the same way
    assert _rel(y.float(), ref) <= 3e-3
    assert _rel(y.float(), ref.to(torch.bfloat16).float()) <= 1.5e-3          # differs from the exactly rounded sum in few elements
    err = (y.double().cpu() - ref).abs()
    assert float((err / (ref.abs() + 1e-3)).max()) <= 2.0 ** -7              # no element is off by more than bf16 rounding + slack
    y0 = ops.K.spmm(g.rowptr, g.colind, g.val, xs.to(cuda), n, long_segments=g.long_segments)
    assert _rel(y.float(), y0.float()) <= 3e-3
    # into a column slice of a wider buffer
    wide = torch.zeros(n, 2 * d, dtype=torch.bfloat16, device=cuda)
    ops.K.spmm_tile(plan, xs.to(cuda), n, out=wide[:, d:])
    assert torch.equal(wide[:, d:], y) and float(wide[:, :d].abs().max()) == 0.0
    # deterministic
    assert torch.equal(ops.K.spmm_tile(plan, xs.to(cuda), n), y)


def test_spmm_tile_follows_the_plan(cuda):
    """The kernel vs the plan evaluated cell by cell (fragment -> slot -> staged source; remainder CSR) in fp64, on
    an operand whose rows are all different (detects any transposition of the fragment / staging layouts)."""
    from sgformer_amd import ops
    g, n, cs, _ = _reordered("community", cuda)
    blk = G.tile_blocks(cs, n, 128)
    g.blk_row = torch.from_numpy(blk).to(cuda)
    plan = ops.TilePlan(g.rowptr, g.colind, g.val, n, g.blk_row, cap=512, min_count=2, keep_dense=True)
    gen = torch.Generator().manual_seed(3)
    xs = torch.randn(n, 128, generator=gen).to(torch.bfloat16)
    y = ops.K.spmm_tile(plan, xs.to(cuda), n)
    ref = G.spmm_tile(blk, plan.sh_ptr.cpu().numpy(), plan.sh_cols.cpu().numpy(), plan.tile_ptr.cpu().numpy(),
                      plan.tiles.cpu().numpy().view(np.uint16)[: plan.fragments * 1024],
                      plan.rem_rowptr.cpu().numpy(), plan.rem_col.cpu().numpy(), plan.rem_val.cpu().numpy(),
                      xs.double().numpy())
    assert _rel(y.float(), torch.from_numpy(ref)) <= 3e-3
    assert _rel(y.float(), torch.from_numpy(ref).to(torch.bfloat16).float()) <= 1e-3


@pytest.mark.parametrize("name,max_rows,cap,min_count", [("community", 128, 512, 2), ("community", 64, 128, 3),
                                                          ("hub_dups_isolated", 128, 512, 2), ("uniform", 128, 256, 2),
                                                          ("community", 256, 512, 2)])
def test_tile_pack_matches_oracle(cuda, name, max_rows, cap, min_count):
    """The packed form of the fragments (sgf_spmm_tile_pack*: sparse groups as 8-byte entries, dense ones copied; what the
    kernel streams) against the numpy restatement, bit for bit, and unpacked again = the plan's dense fragments."""
    from sgformer_amd import ops
    g, n, cs, _ = _reordered(name, cuda)
    blk = G.tile_blocks(cs, n, max_rows)
    g.blk_row = torch.from_numpy(blk).to(cuda)
    plan = ops.TilePlan(g.rowptr, g.colind, g.val, n, g.blk_row, cap=cap, min_count=min_count, keep_dense=True)
    tiles = plan.tiles.cpu().numpy().view(np.uint16)[: plan.fragments * 1024]
    tile_ptr = plan.tile_ptr.cpu().numpy()
    grp, pool = G.tile_pack(blk, tile_ptr, tiles)
    ng = plan.fragments // 2
    assert np.array_equal(plan.grp.cpu().numpy()[:ng], grp[:ng])
    assert plan.tile_bytes == pool.size and np.array_equal(plan.pool.cpu().numpy()[: pool.size], pool)
    assert np.array_equal(G.tile_unpack(grp, pool, plan.fragments, blk, tile_ptr), tiles)
    if name == "community":
        assert (grp[:ng, 1] >= 0).any() and plan.tile_bytes < plan.fragments * 2048     # packing pays on this graph
        from sgformer_amd import _lib
        assert _lib.load().sgf_spmm_tile_sparse_len() == G.TILE_SPARSE_MAX


def test_spmm_tile_exact_on_small_integers(cuda):
    """Integer-valued X and a unit-weight adjacency: every product and partial sum is exact in fp32 and in bf16, so the
    result must EQUAL the integer sum — whatever path (tile or gather) an entry takes."""
    from sgformer_amd import ops
    g, n, cs, ei2 = _reordered("community", cuda)
    ones = torch.ones_like(g.val)
    g.blk_row = torch.from_numpy(G.tile_blocks(cs, n, 128)).to(cuda)
    plan = ops.TilePlan(g.rowptr, g.colind, ones, n, g.blk_row, cap=512, min_count=2)
    gen = torch.Generator().manual_seed(5)
    xi = torch.randint(-2, 3, (n, 256), generator=gen)
    y = ops.K.spmm_tile(plan, xi.to(torch.bfloat16).to(cuda), n)
    rowptr = g.rowptr.cpu().numpy()
    colind = g.colind.cpu().numpy()
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    ref = np.zeros((n, 256), dtype=np.int64)
    np.add.at(ref, rows, xi.numpy()[colind])
    assert int(np.abs(ref).max()) < 256                       # exactly representable in bf16
    assert np.array_equal(y.float().cpu().numpy().astype(np.int64), ref)


@pytest.mark.parametrize("d", [256, 128])
def test_module_on_tiled_graph_matches_oracle(cuda, d):
    """SGFormer (products recipe, bf16) on a graph the policy re-orders and multiplies with sgf_spmm_tile, against the
    fp64 oracle in the CALLER's node order (tolerances of tests/test_gpu_model.py::test_bf16_activation_mode), and
    against the same bf16 module on the caller's order (what differs: the row order of every reduction and the SpMM
    kernel) — the bf16 error of the tiled run must not exceed twice that of the plain run."""
    from sgformer_amd import ops
    from sgformer_amd.ours import SGFormer
    cfg = dict(trans_num_layers=1, trans_num_heads=1, trans_use_act=False, gnn_num_layers=3, gnn_use_init=True,
               graph_weight=0.5)
    ei = _graphs()["community"]
    n, f, c = int(ei.max()) + 1, 40, 6
    torch.manual_seed(3)
    x = torch.randn(n, f)
    y = torch.randint(0, c, (n,))
    idx = torch.randperm(n)[: n // 2]
    p = O.init_params(cfg, f, d, c, seed=1)
    m = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, compute_dtype=torch.bfloat16, **cfg)
    m.load_state_dict({**m.state_dict(), **p})
    m = m.to(cuda).train()
    eig = ei.to(cuda)

    def run(mode):
        prev = ops.set_reorder_mode(mode)
        try:
            ops.graph_cache.clear()
            m.zero_grad(set_to_none=True)
            logits = m(x.to(cuda), eig)
            view = ops.graph_cache.get(eig, n).view()
            O.nll_loss(logits, y.to(cuda), idx.to(cuda)).backward()
            grads = {k: prm.grad.detach().double().cpu() for k, prm in m.named_parameters() if prm.grad is not None}
            return logits.detach().double().cpu(), grads, view
        finally:
            ops.set_reorder_mode(prev)
            ops.graph_cache.clear()

    lt, gt, view = run("always")
    assert view.perm is not None and view.graph.tiled and "tiles" in view.stats["kernel"]
    lp, gp, view0 = run("never")
    assert view0.perm is None
    p64 = {k: v.double().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in p.items()}
    ref = O.sgformer_forward(p64, x.double(), ei, cfg, training=True)
    O.nll_loss(ref, y, idx).backward()
    ref = ref.detach()
    e_t = float((lt - ref).norm() / ref.norm())
    e_p = float((lp - ref).norm() / ref.norm())
    assert e_t <= 3e-2 and e_t <= 2.0 * e_p + 1e-3, (e_t, e_p)
    for k in ["fc.weight", "graph_conv.convs.2.W.weight", "graph_conv.fcs.0.weight", "trans_conv.fcs.0.weight"]:
        gr = p64[k].grad
        r_t = float((gt[k] - gr).norm() / gr.norm())
        r_p = float((gp[k] - gr).norm() / gr.norm())
        assert r_t <= 0.12 and r_t <= 2.0 * r_p + 1e-2, (k, r_t, r_p)


def test_spmm_tile_hubs_and_heavy_rows_in_one_tile(cuda):
    """Hub rows (> LONG_ROW gathered entries: reduced by the long-row kernels) next to rows just BELOW the threshold, rows
    without entries and ordinary rows, all inside the same 32-row tiles — the wave numbers its stream positions over the
    non-hub rows and steps over the hubs' entries (csrc/spmm_tile.hip), several stash epochs per half.  Hubs first / last /
    adjacent in a tile, two tiles of one block affected, one block without any staged source.  vs fp64 on the same inputs."""
    from sgformer_amd import ops
    rng = np.random.default_rng(11)
    n, d = 2048, 256
    lens = rng.integers(0, 12, size=n)
    lens[0], lens[1], lens[2] = 1500, 900, 0             # hub first in its tile, a heavy neighbour, an empty row
    lens[40], lens[41], lens[42], lens[63] = 2100, 1300, 1000, 1100      # adjacent hubs, heavy row, hub last in the tile
    lens[300:316] = 700                                   # 16 heavy rows: 11 200 positions in one half
    lens[1000] = 5000
    rowptr = np.zeros(n + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(lens)
    colind = np.concatenate([np.sort(rng.choice(n, size=k, replace=bool(k > n))) for k in lens]).astype(np.int32)   # (duplicates in the longest rows)
    val = rng.standard_normal(colind.size).astype(np.float32)
    rp, ci, va = (torch.from_numpy(a).to(cuda) for a in (rowptr, colind, val))
    blk = ops.K.tile_blocks(None, n, 128, cuda)
    plan = ops.TilePlan(rp, ci, va, n, blk, cap=512, min_count=2)
    assert plan.long_segments >= 5
    x = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).to(torch.bfloat16)
    y = ops.K.spmm_tile(plan, x.to(cuda), n)
    ref = np.zeros((n, d))
    np.add.at(ref, np.repeat(np.arange(n), lens), val.astype(np.float64)[:, None] * x.double().numpy()[colind])
    ref = torch.from_numpy(ref)
    err = (y.double().cpu() - ref).abs()
    scale = ref.abs().max(dim=1, keepdim=True).values.clamp_min(1e-3)
    assert float((err / scale).max()) <= 8e-3, float((err / scale).max())          # one bf16 rounding per element
    assert torch.equal(ops.K.spmm_tile(plan, x.to(cuda), n), y)
    assert float(y[2].abs().max()) == 0.0                                            # the empty row


def test_permuted_features_are_cached_per_tensor_version(cuda):
    """Full-graph training passes the same feature tensor every step: on a re-ordered graph its permuted copy is kept with
    the graph view (sgformer_amd/ours.py) — reused while the tensor is unchanged, rebuilt after an in-place update or for
    another tensor, never used for a tensor that requires grad."""
    from sgformer_amd import ops
    from sgformer_amd.ours import SGFormer
    cfg = dict(trans_num_layers=1, trans_num_heads=1, trans_use_act=False, gnn_num_layers=2, gnn_use_init=True, graph_weight=0.5)
    ei = _graphs()["community"].to(cuda)
    n, f, c = int(ei.max()) + 1, 40, 6
    torch.manual_seed(5)
    m = SGFormer(f, 128, c, trans_dropout=0.0, gnn_dropout=0.0, compute_dtype=torch.bfloat16, **cfg).to(cuda).eval()
    x = torch.randn(n, f, device=cuda)
    prev = ops.set_reorder_mode("always")
    try:
        ops.graph_cache.clear()
        with torch.no_grad():
            a = m(x, ei)
            view = ops.graph_cache.get(ei, n).view()
            assert view.perm is not None
            kept = view._x_cache[1]
            b = m(x, ei)
            assert view._x_cache[1] is kept and torch.equal(a, b)              # reused
            x.mul_(2.0)                                                         # same storage, new version
            c2 = m(x, ei)
            assert view._x_cache[1] is not kept and not torch.equal(c2, a)
            fresh = m(x.clone(), ei)                                            # another tensor with the same values
            assert torch.equal(fresh, c2)
        xg = x.clone().requires_grad_(True)
        kept = view._x_cache[1]
        m(xg, ei).sum().backward()
        assert view._x_cache[1] is kept and xg.grad is not None and bool(torch.isfinite(xg.grad).all())
    finally:
        ops.set_reorder_mode(prev)
        ops.graph_cache.clear()
