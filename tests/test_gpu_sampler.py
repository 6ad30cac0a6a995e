"""N2 (second half): device neighbour sampling, sgf_neighbor_sample_* / sgformer_amd.sampling.

The reference samples with PyG's NeighborLoader (100M/nb-sample.py:125-151: num_neighbors [15, 10, 5], replace=False,
directed=True) on host workers; its std::mt19937 stream cannot be reproduced, so parity is STRUCTURAL — the
properties that define the loader's output, whatever the random numbers:
  * the seeds are the first batch_size nodes (the trainer slices [:batch_size], 100M/nb-sample.py:29-30,41-42);
  * nodes appear in order of first discovery, hop after hop, without repetition;
  * every node that entered in hop h - 1 has exactly min(in-degree, fanout_h) incoming sampled edges, all distinct
    stored entries of the graph (sampling WITHOUT replacement), pointing neighbour -> node; nodes of the last hop have none;
and the whole procedure, draw included, equals the plain restatement in oracle/graph_oracle.py bit for bit.
Plus: the draw is uniform (chi-square on a hub), and a training step of the 100M model on a sampled batch matches the
fp64 oracle on the same (directed) batch.
"""
import numpy as np
import pytest
import torch

from oracle import graph_oracle as G
from oracle import sgformer_oracle as O

pytestmark = pytest.mark.gpu


def _graph(n=4000, deg=14.0, seed=3):
    from sgformer_amd import synth
    return synth.synthetic_graph_skewed(n, deg, gamma=2.5, seed=seed), n


@pytest.mark.parametrize("fanouts", [[15, 10, 5], [3, 2], [-1, 4], [32], [40, 3], [100], [0, 5], [5, 0, 3]])
def test_sampler_matches_oracle_and_loader_semantics(cuda, fanouts):
    from sgformer_amd.sampling import NeighborSampler
    ei, n = _graph()
    s = NeighborSampler(ei.to(cuda), n, fanouts, seed=1234)
    rowptr, colind = s.rowptr.cpu().numpy(), s.colind.cpu().numpy().astype(np.int64)
    deg = np.diff(rowptr)
    g = torch.Generator().manual_seed(5)
    for b in range(3):
        seeds = torch.randperm(n, generator=g)[:200]
        n_id, e, bs = s.sample(seeds.to(cuda))
        n_id, e = n_id.cpu().numpy(), e.cpu().numpy()
        r_id, r_src, r_dst = G.neighbor_sample(rowptr, colind, seeds.numpy(), fanouts, 1234, b)
        assert np.array_equal(n_id, r_id) and np.array_equal(e[0], r_src) and np.array_equal(e[1], r_dst)
        # --- loader semantics, independent of the oracle ---
        assert bs == 200 and np.array_equal(n_id[:bs], seeds.numpy())
        assert np.unique(n_id).size == n_id.size
        src_g, dst_g = n_id[e[0]], n_id[e[1]]
        # every sampled edge is a stored entry (source in the in-neighbour list of its target), none twice
        stored = set(zip(np.repeat(np.arange(n), deg).tolist(), colind.tolist()))
        pairs = list(zip(dst_g.tolist(), src_g.tolist()))
        assert all(p in stored for p in pairs) and len(set(pairs)) == len(pairs)
        # fan-out per frontier node, hop by hop; discovery order
        lo, hi, pos = 0, bs, 0
        for k in fanouts:
            want = np.minimum(deg[n_id[lo:hi]], k if k >= 0 else deg.max())
            cnt = int(want.sum())
            blk_dst, blk_src = e[1][pos:pos + cnt], e[0][pos:pos + cnt]
            assert np.array_equal(np.bincount(blk_dst - lo, minlength=hi - lo), want)
            assert np.all(np.diff(blk_dst) >= 0)                               # frontier nodes in order
            newly = blk_src[blk_src >= hi]
            first = newly[np.sort(np.unique(newly, return_index=True)[1])]     # order of first appearance
            assert np.array_equal(first, np.arange(hi, hi + first.size))
            pos += cnt
            lo, hi = hi, hi + first.size
            if lo == hi:
                break
        assert pos == e.shape[1] and hi == n_id.size
        assert bool((s.local_of == torch.iinfo(torch.int32).min).all())        # state reset for the next batch


@pytest.mark.parametrize("fanouts", [[15, 10, 5], [40, 6], [7]])
def test_whole_batch_call_equals_hop_by_hop(cuda, fanouts):
    """sgf_neighbor_sample_batch (hop bookkeeping on the device, ONE host read per batch) == sgf_neighbor_sample_mark + one
    sgf_neighbor_sample_hop per hop (two reads per hop), bit for bit: node list, edges, and the state left behind."""
    from sgformer_amd.sampling import NeighborSampler
    ei, n = _graph(n=5000, deg=18.0, seed=4)
    a = NeighborSampler(ei.to(cuda), n, fanouts, seed=99)
    b = NeighborSampler(ei.to(cuda), n, fanouts, seed=99)
    assert a._fan_dev is not None
    b._fan_dev = None                                   # the hop-by-hop path
    g = torch.Generator().manual_seed(1)
    for t in range(4):
        seeds = torch.randperm(n, generator=g)[: [1, 64, 700, 1024][t]].to(cuda)
        ra, rb = a.sample(seeds), b.sample(seeds)
        assert torch.equal(ra[0], rb[0]) and torch.equal(ra[1], rb[1]) and ra[2] == rb[2]
        assert bool((a.local_of == torch.iinfo(torch.int32).min).all())
    assert a.host_reads == 4 and b.host_reads == 4 * 2 * len(fanouts)
    n_id, e, bs = a.sample(torch.zeros(0, dtype=torch.int64, device=cuda))      # an empty batch
    assert n_id.numel() == 0 and e.shape == (2, 0) and bs == 0


@pytest.mark.parametrize("k", [15, 40])
def test_sampler_is_reproducible_and_uniform(cuda, k):
    from sgformer_amd.sampling import NeighborSampler
    ei, n = _graph(n=3000, deg=20.0, seed=8)
    a = NeighborSampler(ei.to(cuda), n, [k, 10, 5], seed=7)
    b = NeighborSampler(ei.to(cuda), n, [k, 10, 5], seed=7)
    c = NeighborSampler(ei.to(cuda), n, [k, 10, 5], seed=8)
    seeds = torch.arange(100, 400)
    ra, rb, rc = a.sample(seeds.to(cuda)), b.sample(seeds.to(cuda)), c.sample(seeds.to(cuda))
    assert torch.equal(ra[0], rb[0]) and torch.equal(ra[1], rb[1])
    assert not (ra[1].shape == rc[1].shape and torch.equal(ra[1], rc[1]))
    # uniformity: the hub (node 0) has hundreds of in-neighbours; over many batches every one of them is drawn with
    # probability k / deg (k = 15: Floyd's subset sampling, k = 40: selection sampling)
    deg = int(a.rowptr[1] - a.rowptr[0])
    assert deg > 150
    nbrs = a.colind[: deg].cpu().numpy()
    hits = np.zeros(n, dtype=np.int64)
    trials = 400
    for t in range(trials):
        n_id, e, _ = a.sample(torch.tensor([0], device=cuda))
        first = e[:, e[1] == 0][0]
        assert first.numel() == k
        hits[n_id[first].cpu().numpy()] += 1
    assert hits.sum() == k * trials and np.all(hits[np.setdiff1d(np.arange(n), nbrs)] == 0)
    exp = k * trials / deg
    chi2 = float(((hits[nbrs] - exp) ** 2 / exp).sum())
    assert chi2 < deg + 6 * np.sqrt(2 * deg), (chi2, deg)                      # mean deg-1, sd sqrt(2 deg)


def test_loader_feeds_the_100m_model(cuda):
    """One training step of sgformer_amd.ours_100m.SGFormer on a sampled batch — graph.x, graph.edge_index,
    output[:batch_size] vs graph.y[:batch_size], CrossEntropyLoss (100M/nb-sample.py:27-35) — against the fp64 oracle
    on the same directed batch."""
    from sgformer_amd.ours_100m import SGFormer
    from sgformer_amd.sampling import NeighborLoader
    ei, n = _graph(n=5000, deg=12.0, seed=4)
    f, c, d = 24, 9, 64
    torch.manual_seed(0)

    class Data:
        pass
    data = Data()
    data.x, data.y, data.edge_index = torch.randn(n, f), torch.randint(0, c, (n,)), ei
    loader = NeighborLoader(data, input_nodes=torch.arange(0, 600), num_neighbors=[15, 10, 5], batch_size=256,
                            shuffle=True, num_workers=12, persistent_workers=True, seed=3)
    assert len(loader) == 3
    cfg = dict(alpha=0.5, trans_num_layers=1, gnn_num_layers=3, gnn_use_init=True, graph_weight=0.8)
    p = O.init_params(cfg, f, d, c, seed=1)
    m = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, **cfg)
    m.load_state_dict({**m.state_dict(), **p})
    m = m.to(cuda).train()
    seen = 0
    for graph in loader:
        graph = graph.to(cuda)
        bs = graph.batch_size
        assert graph.x.shape[0] == graph.y.shape[0] == graph.n_id.numel() and graph.x.is_cuda
        assert torch.equal(graph.x.cpu(), data.x[graph.n_id.cpu()]) and torch.equal(graph.y.cpu(), data.y[graph.n_id.cpu()])
        out = m(graph.x, graph.edge_index)[:bs]
        loss = torch.nn.functional.cross_entropy(out, graph.y[:bs])
        m.zero_grad(set_to_none=True)
        loss.backward()
        p64 = {k: v.double().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in p.items()}
        ref = O.sgformer_forward(p64, graph.x.double().cpu(), graph.edge_index.cpu(), cfg, training=True)[:bs]
        lref = torch.nn.functional.cross_entropy(ref, graph.y[:bs].cpu())
        lref.backward()
        assert float((out.detach().double().cpu() - ref.detach()).abs().max()) <= 1e-4
        assert abs(float(loss.detach()) - float(lref.detach())) <= 1e-5
        gmax = max(float(v.grad.norm()) for v in p64.values() if v.grad is not None)
        for k, prm in m.named_parameters():
            g = p64[k].grad
            if g is not None:
                assert float((prm.grad.double().cpu() - g).norm()) <= 5e-4 * float(g.norm()) + 1e-6 * gmax, k
        seen += bs
    assert seen == 600
