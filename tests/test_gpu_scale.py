"""Parity AT SCALE on the code paths bench.py and a real trainer actually take.

The small-shape tests compare every kernel with the fp64 oracle; what they cannot see is (a) the AUTO re-order
policy (graphs of >= 100 k nodes, decided at the second forward: ops.GraphView.decide), (b) the matrix-core tile SpMM
that policy then selects for bf16 rows (sgf_spmm_tile), (c) composition effects of bf16 storage over ~25 ops at a
size where every reduction spans 10^5 rows.  Here the module runs the way a trainer runs it — no set_reorder_mode,
two forwards — on graphs with community structure hidden behind shuffled ids, against the fp64 oracle in the CALLER's
node order:

  * products recipe (large/run.sh:15-19), d = 256, fp32: logits 1e-4 abs, loss 1e-5, gradients as in
    tests/test_gpu_golden.py (5e-4 relative or 4 x the fp32 CPU oracle's own error); bf16: the same run must not be
    more than 2 x as far from the oracle as the bf16 run on the UN-re-ordered graph (which differs only in SpMM kernel
    and row order), and within the bf16 bounds of tests/test_gpu_model.py::test_bf16_activation_mode;
  * pokec recipe (large/run.sh:22-26: two GCN layers + use_init, C = 2, labels -1 = unlabeled as in
    large/data_utils.py:15-16), fp32, d = 256 (BASELINE.json config 4);
  * 100M recipe (100M/run.sh:3-7: alpha residual), d = 128, C = 172, bf16 (BASELINE.json config 5's model) — the
    d = 128 form of the tile kernel.
"""
import json
import os

import pytest
import torch

from oracle import sgformer_oracle as O

pytestmark = pytest.mark.gpu


def _oracle(cfg, p, x, ei, y, idx, dtype=torch.float64):
    pp = {k: v.to(dtype).requires_grad_(v.is_floating_point() and "running" not in k) for k, v in p.items()}
    ref = O.sgformer_forward(pp, x.to(dtype), ei, cfg, training=True)
    loss = O.nll_loss(ref, y, idx)
    loss.backward()
    return ref.detach(), float(loss.detach()), {k: v.grad for k, v in pp.items() if v.grad is not None}


def _run(m, x, ei, y, idx, cuda, forwards=2):
    """What a trainer does: `forwards` training-mode forwards on the same edge_index (the policy decides at the
    second), backward of the last."""
    from sgformer_amd import ops
    ops.graph_cache.clear()
    eig, xg = ei.to(cuda), x.to(cuda)
    for _ in range(forwards - 1):
        with torch.no_grad():
            m(xg, eig)
    m.zero_grad(set_to_none=True)
    logits = m(xg, eig)
    view = ops.graph_cache.get(eig, x.shape[0]).view()
    loss = O.nll_loss(logits, y.to(cuda), idx.to(cuda))
    loss.backward()
    torch.cuda.synchronize()
    grads = {k: prm.grad.detach().double().cpu() for k, prm in m.named_parameters() if prm.grad is not None}
    ops.graph_cache.clear()
    return logits.detach().double().cpu(), float(loss.detach()), grads, view


def _check_fp32(tag, logits, loss, grads, ref, loss_ref, g64, g32):
    err = float((logits - ref).abs().max())
    report = {"logits_max_abs_err": err, "loss_err": abs(loss - loss_ref), "logits_scale": float(ref.abs().max())}
    gmax = max(float(g.norm()) for g in g64.values())
    bad = []
    for k, g in g64.items():
        num = float((grads[k] - g).norm())
        num32 = float((g32[k].double() - g).norm())
        amax = float((grads[k] - g).abs().max())
        report["grad/" + k] = {"rel": num / max(float(g.norm()), 1e-300), "cpu_fp32_rel": num32 / max(float(g.norm()), 1e-300)}
        if amax > 1e-4 or num > max(5e-4 * float(g.norm()) + 1e-6 * gmax, 4.0 * num32):
            bad.append(k)
    print(tag, json.dumps(report))
    assert err <= 1e-4, report
    assert report["loss_err"] <= 1e-5, report
    assert not bad, (bad, report)


@pytest.fixture(scope="module")
def products_case():
    from sgformer_amd import synth
    n, f, c, d = 170_000, 100, 47, 256
    cfg = dict(synth.RECIPES["ogbn-products"])
    ei = synth.synthetic_graph_community(n, 22.0, seed=11)           # ids shuffled: the locality has to be recovered
    x, y, idx = synth.synthetic_task(n, f, c, seed=11)
    p = O.init_params(cfg, f, d, c, seed=0)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ref, loss_ref, g64 = _oracle(cfg, p, x, ei, y, idx)
    _, _, g32 = _oracle(cfg, p, x, ei, y, idx, torch.float32)
    return dict(n=n, f=f, c=c, d=d, cfg=cfg, ei=ei, x=x, y=y, idx=idx, p=p, ref=ref, loss_ref=loss_ref, g64=g64, g32=g32)


def _module(case, cuda, dtype, cls=None):
    from sgformer_amd.ours import SGFormer
    cls = cls or SGFormer
    m = cls(case["f"], case["d"], case["c"], trans_dropout=0.0, gnn_dropout=0.0,
            compute_dtype=None if dtype == torch.float32 else dtype, **case["cfg"])
    m.load_state_dict({**m.state_dict(), **case["p"]})
    return m.to(cuda).train()


def test_products_recipe_auto_policy_fp32(cuda, products_case):
    """170 k nodes, community graph with shuffled ids, AUTO policy: the graph is re-ordered at the second forward and
    the fp32 run (stream kernels on the re-ordered CSR) stays within BASELINE's 1e-4 of the fp64 oracle."""
    k = products_case
    m = _module(k, cuda, torch.float32)
    logits, loss, grads, view = _run(m, k["x"], k["ei"], k["y"], k["idx"], cuda)
    assert view.perm is not None and view.stats["reordered"] and view.stats["lds_fraction"] > 0.5
    _check_fp32("products-170k fp32 auto-reorder:", logits, loss, grads, k["ref"], k["loss_ref"], k["g64"], k["g32"])


def test_products_recipe_auto_policy_bf16(cuda, products_case):
    """The same in bf16 (BASELINE.json config 3's dtype): the policy picks the matrix-core tile SpMM; its distance to
    the fp64 oracle is bounded by the bf16 module bounds AND by twice the distance of the same bf16 module on the
    graph as given (plain kernels, caller's row order)."""
    from sgformer_amd import ops
    k = products_case
    m = _module(k, cuda, torch.bfloat16)
    lt, loss_t, gt, view = _run(m, k["x"], k["ei"], k["y"], k["idx"], cuda)
    assert view.perm is not None and view.graph.tiled and "tiles" in view.stats["kernel"]
    prev = ops.set_reorder_mode("never")
    try:
        lp, loss_p, gp, view0 = _run(m, k["x"], k["ei"], k["y"], k["idx"], cuda)
    finally:
        ops.set_reorder_mode(prev)
    assert view0.perm is None
    ref = k["ref"]
    e_t = float((lt - ref).norm() / ref.norm())
    e_p = float((lp - ref).norm() / ref.norm())
    report = {"logits_rel_tiled": e_t, "logits_rel_plain": e_p, "loss": [loss_t, loss_p, k["loss_ref"]]}
    for name in ["fc.weight", "graph_conv.convs.2.W.weight", "graph_conv.convs.0.W.weight", "graph_conv.fcs.0.weight",
                 "trans_conv.fcs.0.weight", "trans_conv.convs.0.Wv.weight"]:
        g = k["g64"][name]
        report["grad/" + name] = [float((gt[name] - g).norm() / g.norm()), float((gp[name] - g).norm() / g.norm())]
    _report("products-170k bf16 auto-reorder (tiled vs plain):", report)
    _check_bf16(report, e_t, e_p, loss_t, k["loss_ref"])


def test_products_recipe_powerlaw_graph_with_hubs_bf16(cuda):
    """The same comparison on a power-law community graph with global hubs (SURVEY §8d input class (b): skewed community
    sizes, rows of thousands of entries next to ordinary ones; ids shuffled), 150 k nodes, bf16, AUTO policy: the tile
    SpMM with its hub rows on the long-row queue and their waves stepping over them — bounded by twice the distance of
    the plain kernels on the graph as given."""
    from sgformer_amd import ops, synth
    n, f, c, d = 150_000, 100, 47, 256
    cfg = dict(synth.RECIPES["ogbn-products"])
    ei = synth.synthetic_graph_community_powerlaw(n, 24.0, seed=13, p_hub=0.1, hub_gamma=6.0)   # 5 rows beyond LONG_ROW (up to 22 813 entries), 6 just below
    x, y, idx = synth.synthetic_task(n, f, c, seed=13)
    p = O.init_params(cfg, f, d, c, seed=2)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ref, loss_ref, g64 = _oracle(cfg, p, x, ei, y, idx)
    case = dict(f=f, d=d, c=c, cfg=cfg, p=p)
    m = _module(case, cuda, torch.bfloat16)
    lt, loss_t, gt, view = _run(m, x, ei, y, idx, cuda)
    assert view.perm is not None and view.graph.tiled
    plan = view.graph.tile_plan(False)
    assert plan.long_segments > 0                                     # hubs are really there, on the queue
    prev = ops.set_reorder_mode("never")
    try:
        lp, loss_p, gp, view0 = _run(m, x, ei, y, idx, cuda)
    finally:
        ops.set_reorder_mode(prev)
    assert view0.perm is None
    e_t = float((lt - ref).norm() / ref.norm())
    e_p = float((lp - ref).norm() / ref.norm())
    report = {"logits_rel_tiled": e_t, "logits_rel_plain": e_p, "loss": [loss_t, loss_p, loss_ref],
              "long_segments": plan.long_segments, "tile_fraction": plan.tile_fraction}
    for name in ["fc.weight", "graph_conv.convs.2.W.weight", "graph_conv.fcs.0.weight", "trans_conv.fcs.0.weight"]:
        g = g64[name]
        report["grad/" + name] = [float((gt[name] - g).norm() / g.norm()), float((gp[name] - g).norm() / g.norm())]
    _report("products-recipe power-law 150k bf16 (tiled vs plain):", report)
    _check_bf16(report, e_t, e_p, loss_t, loss_ref, grad_rel=BF16_GRAD_REL_POWERLAW)


def test_pokec_recipe_fp32_with_unlabeled_nodes(cuda):
    """BASELINE.json config 4's model: large/run.sh:22-26 (two GCN layers + use_init, one attention layer, gw 0.5),
    hidden 256, C = 2, fp32; 30 % of the labels are -1 = unlabeled and the training rows are drawn from the labelled
    ones only (large/data_utils.py:15-16).  120 k nodes, uniform graph of pokec's degree: the policy tries the
    re-ordering, finds nothing, keeps the plain kernels."""
    from sgformer_amd import synth
    n, f, c, d = 120_000, 65, 2, 256
    cfg = dict(synth.RECIPES["pokec"])
    ei = synth.synthetic_graph(n, 27.3, seed=5)
    x, y, _ = synth.synthetic_task(n, f, c, seed=5)
    g = torch.Generator().manual_seed(17)
    y = torch.where(torch.rand(n, generator=g) < 0.3, torch.full_like(y, -1), y)
    labelled = torch.nonzero(y != -1).squeeze(1)
    idx = labelled[torch.randperm(labelled.numel(), generator=g)[: labelled.numel() // 2]]
    p = O.init_params(cfg, f, d, c, seed=3)
    case = dict(f=f, d=d, c=c, cfg=cfg, p=p)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ref, loss_ref, g64 = _oracle(cfg, p, x, ei, y, idx)
    _, _, g32 = _oracle(cfg, p, x, ei, y, idx, torch.float32)
    m = _module(case, cuda, torch.float32)
    logits, loss, grads, view = _run(m, x, ei, y, idx, cuda)
    assert view.perm is None and view.stats.get("why") == "no reuse to exploit"
    _check_fp32("pokec-recipe 120k fp32:", logits, loss, grads, ref, loss_ref, g64, g32)
    # the fused loss (sgformer_amd.loss, what bench.py times) on the same rows: labels -1 never reach it
    from sgformer_amd import loss as L
    lf = L.log_softmax_nll(m(x.to(cuda), ei.to(cuda)), y.to(cuda), idx.to(cuda))
    assert abs(float(lf) - loss_ref) <= 1e-5


def test_100m_recipe_bf16_d128(cuda):
    """BASELINE.json config 5's model: 100M/ours.py (alpha residual) with the 100M/run.sh:3-7 flags, hidden 128,
    172 classes, 128 features, bf16 — on a 120 k-node community graph with shuffled ids, AUTO policy: the d = 128
    form of the tile SpMM and the unfused head (C > 64) against the fp64 oracle."""
    from sgformer_amd import ops, synth
    from sgformer_amd.ours_100m import SGFormer
    n, f, c, d = 120_000, 128, 172, 128
    cfg = dict(synth.RECIPES["papers100M-shard8"])
    ei = synth.synthetic_graph_community(n, 24.0, seed=9, comm_size=(48, 200))
    x, y, idx = synth.synthetic_task(n, f, c, seed=9)
    p = O.init_params(cfg, f, d, c, seed=2)
    case = dict(f=f, d=d, c=c, cfg=cfg, p=p)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ref, loss_ref, g64 = _oracle(cfg, p, x, ei, y, idx)
    m = _module(case, cuda, torch.bfloat16, cls=SGFormer)
    lt, loss_t, gt, view = _run(m, x, ei, y, idx, cuda)
    assert view.perm is not None and view.graph.tiled
    prev = ops.set_reorder_mode("never")
    try:
        lp, loss_p, gp, _ = _run(m, x, ei, y, idx, cuda)
    finally:
        ops.set_reorder_mode(prev)
    e_t = float((lt - ref).norm() / ref.norm())
    e_p = float((lp - ref).norm() / ref.norm())
    report = {"logits_rel_tiled": e_t, "logits_rel_plain": e_p, "loss": [loss_t, loss_p, loss_ref]}
    for name in ["fc.weight", "graph_conv.convs.2.W.weight", "graph_conv.fcs.0.weight", "trans_conv.fcs.0.weight"]:
        g = g64[name]
        report["grad/" + name] = [float((gt[name] - g).norm() / g.norm()), float((gp[name] - g).norm() / g.norm())]
    _report("100M-recipe 120k bf16 d=128:", report)
    _check_bf16(report, e_t, e_p, loss_t, loss_ref, grad_rel=BF16_GRAD_REL_100M)


# bf16 bounds = 2 x the error measured on MI355X in round 4 (gpurun_out/scale_reports.jsonl -> profiles/r04_scale_reports.jsonl),
# relative Frobenius distance to the fp64 oracle; VERDICT r03 item 3c.  The earlier blanket bounds were 3e-2 / 0.15.
BF16_LOGITS_REL = 1.0e-2            # measured 4.8e-3 (170 k), 5.1e-3 (2.45 M)
BF16_LOSS_REL = 2.0e-5              # measured 1.8e-6 .. 3.0e-6
BF16_GRAD_REL = {                   # measured (tiled, plain) per test in the report file
    "fc.weight": 4.0e-3, "graph_conv.convs.2.W.weight": 9.0e-2, "graph_conv.convs.0.W.weight": 1.7e-1,
    "graph_conv.fcs.0.weight": 9.5e-2, "trans_conv.fcs.0.weight": 5.2e-2, "trans_conv.convs.0.Wv.weight": 4.2e-3,
}


BF16_GRAD_REL_POWERLAW = {"fc.weight": 4.8e-3, "graph_conv.convs.2.W.weight": 9.1e-2, "graph_conv.fcs.0.weight": 9.7e-2,
                          "trans_conv.fcs.0.weight": 5.3e-2}
BF16_GRAD_REL_100M = {"fc.weight": 4.0e-3, "graph_conv.convs.2.W.weight": 9.7e-2, "graph_conv.fcs.0.weight": 1.17e-1,
                      "trans_conv.fcs.0.weight": 6.8e-2}


def _check_bf16(report, e_t, e_p, loss_t, loss_ref, logits_rel=BF16_LOGITS_REL, grad_rel=None):
    grad_rel = grad_rel or BF16_GRAD_REL
    assert e_t <= logits_rel and e_t <= 2.0 * e_p + 1e-3, report
    assert abs(loss_t - loss_ref) <= BF16_LOSS_REL * abs(loss_ref), report
    for name, (r_t, r_p) in ((n_, v) for n_, v in report.items() if n_.startswith("grad/")):
        assert r_t <= grad_rel[name[5:]] and r_t <= 2.0 * r_p + 2e-2, (name, report)


def _report(tag, report):
    """Measured errors of the scale tests, kept next to the run (gpurun_out/ is merged back from the GPU box)."""
    print(tag, json.dumps(report))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "scale_reports.jsonl"), "a") as f:
            f.write(json.dumps({"test": tag, **report}) + "\n")


@pytest.mark.skipif(os.environ.get("SGF_SKIP_FULLSIZE") == "1", reason="SGF_SKIP_FULLSIZE=1")
@pytest.mark.parametrize("graph", ["uniform", "community"])
def test_products_recipe_at_full_size_bf16(cuda, graph):
    """BASELINE.json config 3 AT ITS OWN SIZE (N = 2 449 029, nnz = 126 M, d = 256, bf16 — the workload bench.py times): one
    training-mode forward of the HIP module against the fp32 CPU oracle (oracle/sgformer_oracle.py with a prebuilt sparse
    adjacency; ~1 min of host time) on every row.  What only this size exercises: 32-bit byte offsets next to 2^32 (X is
    1.25 GB), the XCD-remapped tails of the row kernels, 19 k row blocks / 1 GB of tile plan on the re-ordered community
    graph (`community`: sgf_reorder + sgf_spmm_tile through the auto policy), BatchNorm / attention sums over 2.4 M rows.
    Bounds: bf16 storage over ~25 ops — relative Frobenius error of the logits 2e-2, no row off by more than 0.25 of the
    logits' range, loss within 2 %."""
    from sgformer_amd import ops, synth
    from sgformer_amd.ours import SGFormer
    n, avg_deg, f, c, d = synth.SHAPES["ogbn-products"]
    cfg = dict(synth.RECIPES["ogbn-products"])
    gen = synth.synthetic_graph if graph == "uniform" else synth.synthetic_graph_community
    ei = gen(n, avg_deg, seed=123)
    x, y, idx = synth.synthetic_task(n, f, c, seed=123)
    p = O.init_params(cfg, f, d, c, seed=0)
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    adj = O.build_adj(ei, n)
    with torch.no_grad():
        ref = O.sgformer_forward({k: v.clone() for k, v in p.items()}, x, ei, cfg, training=True, adj=adj)
    del adj
    loss_ref = float(O.nll_loss(ref, y, idx))
    m = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, compute_dtype=torch.bfloat16, **cfg)
    m.load_state_dict({**m.state_dict(), **p})
    m = m.to(cuda).train()
    ops.graph_cache.clear()
    eig, xg = ei.to(cuda), x.to(cuda).bfloat16()
    with torch.no_grad():
        m(xg, eig)                                   # the auto policy decides at the second forward
        logits = m(xg, eig).float()
    view = ops.graph_cache.get(eig, n).view()
    loss = float(O.nll_loss(logits, y.to(cuda), idx.to(cuda)))
    lg = logits.cpu()
    ops.graph_cache.clear()
    rel = float((lg - ref).norm() / ref.norm())
    worst = float((lg - ref).abs().max())
    scale = float(ref.max() - ref.min())
    report = {"graph": graph, "reordered": view.perm is not None, "kernel": view.stats.get("kernel", "k_spmm_row"),
              "logits_rel_frobenius": rel, "worst_abs": worst, "logits_range": scale, "loss": loss, "loss_ref_fp32": loss_ref,
              "finite": bool(torch.isfinite(lg).all())}
    _report("products-2.45M bf16 full size:", report)
    assert report["finite"]
    assert (view.perm is not None) == (graph == "community"), report
    assert rel <= BF16_LOGITS_REL and worst <= 6e-3 * scale, report     # measured 5.06e-3 and 0.0147 / 5.09 = 2.9e-3
    assert abs(loss - loss_ref) <= BF16_LOSS_REL * abs(loss_ref), report   # measured 2.5e-6 / 3.0e-6


@pytest.mark.skipif(os.environ.get("SGF_SKIP_FULLSIZE") == "1", reason="SGF_SKIP_FULLSIZE=1")
def test_products_recipe_gradients_at_full_size(cuda):
    """VERDICT r04 "parity reach" (c): GRADIENTS at BASELINE config 3's own size (N = 2 449 029, nnz = 126 M, d = 256), where the
    fp32 reductions are longest — through a size-independent property, since a CPU backward at this size does not fit a test:

      * fp32 module: the backward must be the derivative of the forward that test_products_recipe_at_full_size_bf16 / the
        arxiv-size tests pin against the oracle — central finite differences of the loss (taken in fp64 from the fp32 logits)
        along a random direction of each of five parameter tensors agree with <grad, direction> to 2 %;
      * bf16 module: every parameter gradient within the bf16 bounds of the smaller scale tests (BF16_GRAD_REL) of the fp32
        module's gradient, on the same 50 000-row training sample.
    The loss is taken over a 50 k-row sample of the training rows (a sparse cotangent: every reduction still spans all rows)."""
    from sgformer_amd import ops, synth
    from sgformer_amd.ours import SGFormer
    n, avg_deg, f, c, d = synth.SHAPES["ogbn-products"]
    cfg = dict(synth.RECIPES["ogbn-products"])
    ei = synth.synthetic_graph(n, avg_deg, seed=123, device=cuda)
    x, y, idx = synth.synthetic_task(n, f, c, seed=123)
    idx = idx[torch.randperm(idx.numel(), generator=torch.Generator().manual_seed(5))[:50000]].to(cuda)
    x, y = x.to(cuda), y.to(cuda)
    p = O.init_params(cfg, f, d, c, seed=0)

    def build(dtype):
        m = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, compute_dtype=dtype, **cfg)
        m.load_state_dict({**m.state_dict(), **p})
        m = m.to(cuda).train()
        for mod in m.modules():                                   # the forwards below must not move the running statistics
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.momentum = 0.0
        return m

    def loss64(m):
        with torch.no_grad():
            lg = m(x, ei).double()
        return float(torch.nn.functional.nll_loss(torch.log_softmax(lg, dim=1)[idx], y[idx]))

    m = build(None)
    logits = m(x, ei)
    O.nll_loss(logits, y, idx).backward()
    g32 = {k: prm.grad.detach().clone() for k, prm in m.named_parameters() if prm.grad is not None}
    report = {}
    gen = torch.Generator().manual_seed(11)
    for name in ["fc.weight", "graph_conv.convs.2.W.weight", "graph_conv.convs.0.W.weight", "graph_conv.fcs.0.weight",
                 "trans_conv.convs.0.Wq.weight"]:
        prm = dict(m.named_parameters())[name]
        v = torch.randn(prm.shape, generator=gen).to(cuda)
        v = v / v.norm()
        # a step along the gradient's own direction mixed with a random one, so that the derivative is not tiny
        gdir = g32[name] / g32[name].norm().clamp_min(1e-30)
        v = (gdir + 0.5 * v)
        v = v / v.norm()
        eps = 2e-3 * float(prm.detach().norm())
        with torch.no_grad():
            prm.add_(v, alpha=eps)
            lp = loss64(m)
            prm.add_(v, alpha=-2 * eps)
            lm = loss64(m)
            prm.add_(v, alpha=eps)
        fd = (lp - lm) / (2 * eps)
        an = float((g32[name].double() * v.double()).sum())
        report["fd/" + name] = [fd, an]
    del m, logits
    torch.cuda.empty_cache()
    mb = build(torch.bfloat16)
    O.nll_loss(mb(x, ei).float(), y, idx).backward()
    for k, prm in mb.named_parameters():
        if prm.grad is not None and k in BF16_GRAD_REL:
            report["bf16_vs_fp32/" + k] = float((prm.grad.double() - g32[k].double()).norm() / g32[k].double().norm())
    ops.graph_cache.clear()
    _report("products-2.45M gradients at full size:", report)
    for k, val in report.items():
        if k.startswith("fd/"):
            fd, an = val
            assert abs(fd - an) <= 2e-2 * abs(an) + 1e-7, (k, report)
        else:
            assert val <= BF16_GRAD_REL[k.split("/", 1)[1]], (k, report)


def test_bf16_training_trajectory_tracks_fp32(cuda):
    """VERDICT r04 "parity reach" (b): 20 Adam steps of the products recipe (d = 256, the trainer's two parameter groups,
    large/main.py:114-119) in bf16 mode against the SAME module in fp32 mode (itself pinned to the oracle, step by step, in
    tests/test_gpu_model.py::test_training_trajectory): the bf16 loss curve stays within 1 % of the fp32 curve at every
    step and both fall.  (The final parameters are reported, not bounded: Adam moves every weight by ~lr per step whatever
    the gradient's size, so weights whose gradient is bf16 noise drift apart without the loss noticing.)"""
    from sgformer_amd import synth
    from sgformer_amd.ours import SGFormer
    cfg = dict(synth.RECIPES["ogbn-products"])
    n, f, d, c = 30000, 100, 256, 47
    ei = synth.synthetic_graph_community(n, 20.0, seed=3).to(cuda)
    x, y, idx = synth.synthetic_task(n, f, c, seed=3)
    # learnable labels: a linear teacher on the features, so that 20 steps move the loss visibly
    teacher = torch.randn(f, c, generator=torch.Generator().manual_seed(1))
    y = (x @ teacher).argmax(1)
    x, y, idx = x.to(cuda), y.to(cuda), idx.to(cuda)
    p = O.init_params(cfg, f, d, c, seed=2)
    curves, finals = {}, {}
    for tag, dtype in (("fp32", None), ("bf16", torch.bfloat16)):
        m = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, compute_dtype=dtype, **cfg)
        m.load_state_dict({**m.state_dict(), **p})
        m = m.to(cuda).train()
        opt = torch.optim.Adam([{"params": m.params1, "weight_decay": 1e-5}, {"params": m.params2, "weight_decay": 1e-5}],
                               lr=0.01)
        losses = []
        for _ in range(20):
            opt.zero_grad(set_to_none=True)
            loss = O.nll_loss(m(x, ei).float(), y, idx)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        curves[tag] = losses
        finals[tag] = {k: v.detach().double().clone() for k, v in m.named_parameters()}
    report = {"fp32": curves["fp32"], "bf16": curves["bf16"]}
    rel = {k: float((finals["bf16"][k] - v).norm() / v.norm().clamp_min(1e-30)) for k, v in finals["fp32"].items()}
    report["worst_param_rel"] = max(rel.items(), key=lambda kv: kv[1])
    _report("bf16 vs fp32 trajectory, 20 Adam steps:", report)
    assert curves["fp32"][-1] < 0.8 * curves["fp32"][0] and curves["bf16"][-1] < 0.8 * curves["bf16"][0], report
    for a, b in zip(curves["bf16"], curves["fp32"]):
        assert abs(a - b) <= 1e-2 * abs(b), report


def test_bf16_training_trajectory_100_steps_with_adam_state(cuda):
    """the trajectory test at 100 Adam steps (large/main.py:125-143's loop: zero_grad, forward, the three
    loss lines, backward, the two-group Adam step), bf16 mode against fp32 mode of the same module from the same initial
    parameters, and the OPTIMIZER STATE compared at the end: the first moments of every weight matrix point the same way
    (cosine), the second moments agree in scale.  The loss curves stay together the whole way: 1 % while the loss is O(1),
    2e-3 absolute once it has fallen below 0.2 (the bf16 run is not expected to track the fp32 run's fourth digit there)."""
    from sgformer_amd import synth
    from sgformer_amd.ours import SGFormer
    cfg = dict(synth.RECIPES["ogbn-products"])
    n, f, d, c = 30000, 100, 256, 47
    ei = synth.synthetic_graph_community(n, 20.0, seed=3).to(cuda)
    x, y, idx = synth.synthetic_task(n, f, c, seed=3)
    teacher = torch.randn(f, c, generator=torch.Generator().manual_seed(1))
    y = (x @ teacher).argmax(1)
    x, y, idx = x.to(cuda), y.to(cuda), idx.to(cuda)
    p = O.init_params(cfg, f, d, c, seed=2)
    curves, moments = {}, {}
    for tag, dtype in (("fp32", None), ("bf16", torch.bfloat16)):
        m = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, compute_dtype=dtype, **cfg)
        m.load_state_dict({**m.state_dict(), **p})
        m = m.to(cuda).train()
        opt = torch.optim.Adam([{"params": m.params1, "weight_decay": 1e-5}, {"params": m.params2, "weight_decay": 1e-5}],
                               lr=0.01)
        losses = []
        for _ in range(100):
            opt.zero_grad(set_to_none=True)
            loss = O.nll_loss(m(x, ei).float(), y, idx)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        curves[tag] = losses
        names = {id(prm): k for k, prm in m.named_parameters()}
        moments[tag] = {names[id(prm)]: (st["exp_avg"].double().clone(), st["exp_avg_sq"].double().clone())
                        for prm, st in opt.state.items()}
    report = {"fp32": curves["fp32"][::10] + [curves["fp32"][-1]], "bf16": curves["bf16"][::10] + [curves["bf16"][-1]]}
    worst_loss = 0.0
    for a, b in zip(curves["bf16"], curves["fp32"]):
        worst_loss = max(worst_loss, abs(a - b) / (1e-2 * abs(b) + 2e-3))
    report["worst_loss_gap_over_bound"] = worst_loss
    cos, ratio = {}, {}
    for k, (m1, v1) in moments["fp32"].items():
        if not k.endswith("weight") or m1.dim() != 2:
            continue
        m2, v2 = moments["bf16"][k]
        cos[k] = float((m1 * m2).sum() / (m1.norm() * m2.norm()).clamp_min(1e-300))
        ratio[k] = float(v2.sum() / v1.sum().clamp_min(1e-300))
    report["exp_avg_cosine"], report["exp_avg_sq_ratio"] = cos, ratio
    _report("bf16 vs fp32 trajectory, 100 Adam steps + optimizer state:", report)
    assert curves["fp32"][-1] < 0.05 * curves["fp32"][0] and curves["bf16"][-1] < 0.05 * curves["bf16"][0], report
    assert worst_loss <= 1.0, report
    for k in cos:
        assert cos[k] >= 0.5 and 0.25 <= ratio[k] <= 4.0, (k, report)
