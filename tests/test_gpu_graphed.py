"""The mini-batch step as hipGraph replays (sgformer_amd/graphed.py) against the same step issued launch by launch."""

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _run(cuda, monkeypatch, graphs: bool, dtype, batches: int, move_at=None):
    """`batches` training steps of large/main-batch.py:134-151 on induced subgraphs of ONE node count; returns the per-step
    logits, the final parameters and BatchNorm buffers."""
    from sgformer_amd import batching, graphed, launch, ops, synth
    from sgformer_amd.ours import SGFormer
    monkeypatch.setenv("SGF_BATCH_GRAPH", "1" if graphs else "0")
    n, f, c, d, m = 30000, 100, 47, 64, 6144
    ei = synth.synthetic_graph(n, 14.0, seed=11)
    x, y, _ = synth.synthetic_task(n, f, c, seed=11)
    x, y = x.to(cuda), y.to(cuda)
    torch.manual_seed(5)
    model = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, compute_dtype=dtype, **synth.RECIPES["ogbn-products"]).to(cuda)
    opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=1e-5)
    gen = torch.Generator().manual_seed(17)
    before = dict(graphed.counters)
    logits = []
    batching._parents.clear()
    for step in range(batches):
        if move_at is not None and step == move_at:
            # evaluate_large's round trip (large/eval.py:41 + main-batch.py:131): the parameters come back at NEW addresses
            model.to("cpu")
            model.to(cuda)
            model.train()
        idx = torch.randperm(n, generator=gen)[:m]
        ei_i, _ = batching.subgraph(idx, ei, num_nodes=n, relabel_nodes=True)
        model.train()
        opt.zero_grad()
        out = model(x[idx.to(cuda)], ei_i)
        logits.append(out.detach().float().clone())
        loss = F.nll_loss(F.log_softmax(out.float(), dim=1), y[idx.to(cuda)])
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    used = {k: graphed.counters[k] - before[k] for k in before}
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    # evaluation after training runs the eager path on whatever graph it is given
    model.eval()
    with torch.no_grad():
        idx = torch.arange(m)
        ei_i, _ = batching.subgraph(idx, ei, num_nodes=n, relabel_nodes=True)
        ev = model(x[:m], ei_i).float().clone()
    ops.graph_cache.clear()
    return logits, state, used, ev


@pytest.mark.parametrize("dtype", [torch.bfloat16, None], ids=["bf16", "f32"])
def test_replayed_steps_equal_the_eager_steps(cuda, monkeypatch, dtype):
    """7 Adam steps on batches of one size: step 1 eager, step 2 captures (and already returns the captured graphs' result),
    steps 3-7 replay — every step's logits, the final parameters and the BatchNorm running statistics equal the all-eager
    run's bit for bit (same kernels, same order, same arithmetic)."""
    eager = _run(cuda, monkeypatch, False, dtype, 7)
    graph = _run(cuda, monkeypatch, True, dtype, 7)
    assert eager[2] == {"captures": 0, "replays": 0}
    assert graph[2] == {"captures": 1, "replays": 6}
    for i, (a, b) in enumerate(zip(eager[0], graph[0])):
        assert torch.equal(a, b), f"step {i}: max |diff| {float((a - b).abs().max())}"
    for k in eager[1]:
        assert torch.equal(eager[1][k], graph[1][k]), k
    assert torch.equal(eager[3], graph[3])


def test_parameters_that_moved_are_recaptured(cuda, monkeypatch):
    """The trainer's evaluation moves the model to the host and back (large/eval.py:41): the captured launches hold the old
    addresses, so the next step captures again — and stays equal to the eager run."""
    eager = _run(cuda, monkeypatch, False, torch.bfloat16, 6, move_at=4)
    graph = _run(cuda, monkeypatch, True, torch.bfloat16, 6, move_at=4)
    assert graph[2] == {"captures": 2, "replays": 5}
    for i, (a, b) in enumerate(zip(eager[0], graph[0])):
        assert torch.equal(a, b), f"step {i}: max |diff| {float((a - b).abs().max())}"
    for k in eager[1]:
        assert torch.equal(eager[1][k], graph[1][k]), k


def test_a_parameter_frozen_after_the_capture_is_recaptured(cuda, monkeypatch):
    """The captured backward computes gradients for the parameters that required one AT CAPTURE: freezing / unfreezing a
    parameter afterwards must capture again, not return stale (or no) gradients."""
    from sgformer_amd import batching, graphed, ops, synth
    from sgformer_amd.ours import SGFormer
    monkeypatch.setenv("SGF_BATCH_GRAPH", "1")
    n, f, c, d, m = 20000, 100, 47, 64, 5000
    ei = synth.synthetic_graph(n, 10.0, seed=2)
    x = torch.randn(m, f, device=cuda)
    batching._parents.clear()
    ei_i, _ = batching.subgraph(torch.arange(m), ei, num_nodes=n, relabel_nodes=True)
    torch.manual_seed(1)
    model = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, compute_dtype=torch.bfloat16, **synth.RECIPES["ogbn-products"]).to(cuda).train()
    before = dict(graphed.counters)

    def step():
        model.zero_grad()
        model(x, ei_i).float().sum().backward()
        return {k: (None if p.grad is None else p.grad.clone()) for k, p in model.named_parameters()}

    step()
    full = step()                                        # capture 1
    model.fc.weight.requires_grad_(False)
    frozen = step()                                      # capture 2: fc.weight gets no gradient any more
    model.fc.weight.requires_grad_(True)
    again = step()                                       # capture 3: and gets it again
    assert {k: graphed.counters[k] - before[k] for k in before} == {"captures": 3, "replays": 3}
    assert full["fc.weight"] is not None and frozen["fc.weight"] is None
    for k in full:
        assert torch.equal(full[k], again[k]), k
        if k != "fc.weight":
            assert torch.equal(full[k], frozen[k]), k
    ops.graph_cache.clear()


def test_ineligible_calls_keep_the_eager_path(cuda, monkeypatch):
    """Active dropout, features that require a gradient, evaluation and small batches never capture."""
    from sgformer_amd import batching, graphed, synth
    from sgformer_amd.ours import SGFormer
    monkeypatch.setenv("SGF_BATCH_GRAPH", "1")
    n, f, c, d, m = 20000, 100, 47, 64, 5000
    ei = synth.synthetic_graph(n, 10.0, seed=2)
    x = torch.randn(m, f, device=cuda)
    batching._parents.clear()
    ei_i, _ = batching.subgraph(torch.arange(m), ei, num_nodes=n, relabel_nodes=True)
    before = dict(graphed.counters)
    drop = SGFormer(f, d, c, trans_dropout=0.5, gnn_dropout=0.5, compute_dtype=torch.bfloat16, **synth.RECIPES["ogbn-products"]).to(cuda)
    plain = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, compute_dtype=torch.bfloat16, **synth.RECIPES["ogbn-products"]).to(cuda)
    for _ in range(3):
        drop.train()(x, ei_i).sum().backward()
        plain.train()(x.clone().requires_grad_(True), ei_i).sum().backward()
        plain.eval()
        with torch.no_grad():
            plain(x, ei_i)
        small, _ = batching.subgraph(torch.arange(1000), ei, num_nodes=n, relabel_nodes=True)
        plain.train()(x[:1000], small).sum().backward()
    assert graphed.counters == before


def test_two_forwards_before_one_backward_and_kept_logits(cuda, monkeypatch):
    """What a replay must not break: logits a caller keeps across batches stay what they were (the graphs' output buffer is
    static: a copy is handed out), and a second forward of the same size BEFORE the first one's backward — its logits
    dropped, only the loss kept — runs eager instead of overwriting the activations that backward still needs."""
    from sgformer_amd import batching, graphed, ops, synth
    from sgformer_amd.ours import SGFormer
    n, f, c, d, m = 30000, 100, 47, 64, 6144
    ei = synth.synthetic_graph(n, 14.0, seed=11)
    x, y, _ = synth.synthetic_task(n, f, c, seed=11)
    x, y = x.to(cuda), y.to(cuda)
    gen = torch.Generator().manual_seed(3)
    batches = []
    batching._parents.clear()
    for _ in range(4):
        idx = torch.randperm(n, generator=gen)[:m]
        batches.append((x[idx.to(cuda)], batching.subgraph(idx, ei, num_nodes=n, relabel_nodes=True)[0], y[idx.to(cuda)]))

    def run(graphs):
        monkeypatch.setenv("SGF_BATCH_GRAPH", "1" if graphs else "0")
        torch.manual_seed(5)
        model = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, compute_dtype=torch.bfloat16,
                         **synth.RECIPES["ogbn-products"]).to(cuda).train()
        before = dict(graphed.counters)
        kept = []
        for xb, eb, yb in batches[:2]:                       # eager, then capture + replay
            model.zero_grad()
            out = model(xb, eb)
            kept.append(out)
            F.nll_loss(F.log_softmax(out.float(), dim=1), yb).backward()
        snapshot = kept[1].detach().clone()
        model.zero_grad()
        loss_a = F.nll_loss(F.log_softmax(model(*batches[2][:2]).float(), dim=1), batches[2][2])      # replay; logits dropped
        loss_b = F.nll_loss(F.log_softmax(model(*batches[3][:2]).float(), dim=1), batches[3][2])      # must NOT replay
        (loss_a + loss_b).backward()
        grads = [p.grad.detach().clone() for p in model.parameters()]
        used = {k: graphed.counters[k] - before[k] for k in before}
        assert torch.equal(kept[1].detach(), snapshot), "logits kept from an earlier batch changed under a later replay"
        ops.graph_cache.clear()
        return grads, used, float(loss_a), float(loss_b)

    g_eager, used_e, la_e, lb_e = run(False)
    g_graph, used_g, la_g, lb_g = run(True)
    assert used_e == {"captures": 0, "replays": 0} and used_g == {"captures": 1, "replays": 2}
    assert (la_e, lb_e) == (la_g, lb_g)
    for a, b in zip(g_eager, g_graph):
        assert torch.equal(a, b)


def test_cached_batches_survive_replays_and_accumulation_stays_eager(cuda, monkeypatch):
    """Two things a replay must not do to a caller that is not the reference trainer:
    (1) a trainer that keeps same-sized DEVICE batches across epochs: the capture's static input is a private buffer, so
        the cached feature tensors are what they were after any number of replays, and the logits of epoch 2 equal epoch 1's
        eager logits for the same parameters;
    (2) gradient accumulation over two batches without zero_grad: a replayed backward hands out static gradient buffers, so
        with a gradient in place the step runs eager and p.grad == g1 + g2 of the eager run, bit for bit."""
    from sgformer_amd import batching, graphed, ops, synth
    from sgformer_amd.ours import SGFormer
    n, f, c, d, m = 20000, 100, 47, 64, 5000
    ei = synth.synthetic_graph(n, 10.0, seed=3)
    gen = torch.Generator().manual_seed(4)
    batching._parents.clear()
    batches = []
    for _ in range(2):
        idx = torch.randperm(n, generator=gen)[:m]
        ei_i, _ = batching.subgraph(idx, ei, num_nodes=n, relabel_nodes=True)
        batches.append((torch.randn(m, f, generator=gen).to(cuda), ei_i))
    kept = [b[0].clone() for b in batches]

    def run(graphs: bool):
        monkeypatch.setenv("SGF_BATCH_GRAPH", "1" if graphs else "0")
        torch.manual_seed(1)
        model = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, compute_dtype=torch.bfloat16,
                         **synth.RECIPES["ogbn-products"]).to(cuda).train()
        before = dict(graphed.counters)
        outs = []
        for epoch in range(3):                           # the SAME two cached batches, three epochs, parameters untouched
            for xb, eb in batches:
                model.zero_grad()
                out = model(xb, eb)
                out.float().sum().backward()
                outs.append(out.detach().float().clone())
        # accumulation: two backward passes into the same .grad
        model.zero_grad()
        for xb, eb in batches:
            model(xb, eb).float().sum().backward()
        acc = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
        used = {k: graphed.counters[k] - before[k] for k in before}
        ops.graph_cache.clear()
        return outs, acc, used

    eager = run(False)
    graph = run(True)
    assert graph[2]["captures"] == 1 and graph[2]["replays"] >= 5
    for xb, k in zip(batches, kept):
        assert torch.equal(xb[0], k)                     # (1) the caller's tensors are untouched
    for i, (a, b) in enumerate(zip(eager[0], graph[0])):
        assert torch.equal(a, b), f"call {i}"
    for i in range(2):                                   # same parameters, same batch: every epoch returns the same logits
        assert torch.equal(graph[0][i], graph[0][i + 2]) and torch.equal(graph[0][i], graph[0][i + 4])
    assert eager[1].keys() == graph[1].keys()
    for k in eager[1]:                                   # (2)
        assert torch.equal(eager[1][k], graph[1][k]), k


def _two_region_graph(n, seed):
    """sparse uniform graph over n nodes + a DENSE block among nodes [10, 8202) + node 0 as a hub of nodes [20000, 26000),
    symmetric, coalesced, one self-loop per node (the trainer's prologue)."""
    from sgformer_amd import synth
    g = torch.Generator().manual_seed(seed)
    sparse = synth.synthetic_graph(n, 6.0, seed=seed)
    a = torch.randint(10, 8202, (8192 * 20,), generator=g)
    b = torch.randint(10, 8202, (8192 * 20,), generator=g)
    hub = torch.arange(20000, 26000)
    src = torch.cat([sparse[0], a, b, torch.zeros_like(hub), hub])
    dst = torch.cat([sparse[1], b, a, hub, torch.zeros_like(hub)])
    key = torch.unique(src * n + dst)
    return torch.stack([key // n, key % n])


def test_a_batch_beyond_the_captured_capacity_and_a_first_long_row(cuda, monkeypatch):
    """(a) a later batch of a captured size whose nnz exceeds StaticCSR.cap (the fixed-capacity arrays the
    captured SpMM launches read) is a new capture, not a truncated graph; (b) a batch with a row of 2 000 entries that FITS a
    capture made without the long-row path (graphed._long_bound: the batches seen until then had no such row) replays it
    and stays correct — the captured row kernel handles any row length.  Every step against the all-eager run: bit for bit
    where both run the same kernels, within fp32 summation-order noise for the hub batches (the eager run reduces the hub row
    through the long-row queue)."""
    from sgformer_amd import batching, graphed, ops, synth
    from sgformer_amd.ours import SGFormer
    n, f, c, d, m = 40000, 100, 47, 64, 8192
    ei = _two_region_graph(n, 4)
    gen = torch.Generator().manual_seed(9)
    tail = torch.arange(12000, n)

    def sparse():
        return tail[torch.randperm(tail.numel(), generator=gen)[:m]]

    def hubfit():                                      # node 0, 2 000 of its neighbours, the rest from the sparse region
        rest = torch.arange(26000, n)
        return torch.cat([torch.zeros(1, dtype=torch.long), torch.arange(20000, 22000),
                          rest[torch.randperm(rest.numel(), generator=gen)[:m - 2001]]])

    batches = [sparse(), sparse(), sparse(), hubfit(), torch.arange(10, 10 + m), sparse(), hubfit()]
    x = torch.randn(n, f, generator=gen).to(cuda)

    def run(graphs: bool):
        monkeypatch.setenv("SGF_BATCH_GRAPH", "1" if graphs else "0")
        batching._parents.clear()
        torch.manual_seed(3)
        model = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, **synth.RECIPES["ogbn-products"]).to(cuda).train()   # fp32
        before = dict(graphed.counters)
        outs, info = [], []
        for idx in batches:
            ei_i, _ = batching.subgraph(idx, ei, num_nodes=n, relabel_nodes=True)
            model.zero_grad()
            out = model(x[idx.to(cuda)], ei_i)
            out.float().sum().backward()
            outs.append(out.detach().float().clone())
            info.append((int(ei_i.shape[1]), int(getattr(ei_i, "_sgf_max_in_degree", 0))))
        used = {k: graphed.counters[k] - before[k] for k in before}
        ops.graph_cache.clear()
        return outs, used, info

    eager, _, info = run(False)
    graph, used, _ = run(True)
    nnz, longest = [i[0] for i in info], [i[1] for i in info]
    cap1 = int(nnz[1] * 1.25) + 4096                             # graphed._capture's capacity rule, from the capture batch
    assert nnz[3] <= cap1 < nnz[4], (nnz, cap1)                  # the hub batch fits the first capture, the dense one does not
    assert longest[3] > ops.LONG_ROW and longest[6] > ops.LONG_ROW and max(longest[:3] + [longest[4]]) <= ops.LONG_ROW, info
    assert used == {"captures": 2, "replays": 6}, used
    for i in (0, 1, 2, 4, 5):                                    # same kernels, same order: bit-identical
        assert torch.equal(eager[i], graph[i]), i
    for i in (3, 6):                                             # the hub row: long-row queue (eager) vs row kernel (captured)
        scale = float(eager[i].abs().max())
        assert float((eager[i] - graph[i]).abs().max()) <= 2e-5 * max(1.0, scale), i
