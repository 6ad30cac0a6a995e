"""sgformer_amd.launch rewires a few GLOBAL functions for the unchanged trainers (F.log_softmax, F.nll_loss,
torch.optim.Adam.__init__, torch_geometric.utils.*).  These tests drive each patch with callers that are NOT the reference's
trainers — shapes, arguments and index tensors a third-party library in the same process might use — and require the
un-patched function's result, value and gradient; plus the `--sgf-patches minimal` mode, which installs the module drop-in
and nothing else.  CPU only (the CPU kernel table of tests/cpu_kernels.py stands in for libsgf)."""
import os
import sys
import types

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def patched_loss():
    from sgformer_amd import launch, ops
    from tests.cpu_kernels import CpuKernels
    prev = ops.set_kernels(CpuKernels())
    ls0, nll0 = F.log_softmax, F.nll_loss
    launch.patch_nll_loss()
    try:
        yield ls0, nll0
    finally:
        launch.unpatch_nll_loss()
        ops.set_kernels(prev)
    assert F.log_softmax is ls0 and F.nll_loss is nll0


def _same(a, b, tol=1e-6):
    return a.shape == b.shape and float((a.detach().double() - b.detach().double()).abs().max()) <= tol if a.numel() else a.shape == b.shape


@pytest.mark.parametrize("case", ["3d_dim1", "3d_dim2", "dim0", "dim_neg1", "dtype_arg", "wide", "transposed", "empty", "1d", "f64"])
def test_log_softmax_patch_third_party_calls(patched_loss, case):
    """every call shape a third party may make returns ATen's values and gradients; only the trainers' 2-D / dim=1 / <= 64
    classes form becomes lazy — and a lazy result used in ANY other way than `out[idx]` + nll_loss materialises to ATen's."""
    from sgformer_amd.loss import LazyLogSoftmax
    ls0, _ = patched_loss
    g = torch.Generator().manual_seed(1)
    kw = {}
    if case == "3d_dim1":
        x, kw = torch.randn(4, 5, 6, generator=g), dict(dim=1)
    elif case == "3d_dim2":
        x, kw = torch.randn(4, 5, 6, generator=g), dict(dim=2)
    elif case == "dim0":
        x, kw = torch.randn(7, 5, generator=g), dict(dim=0)
    elif case == "dim_neg1":
        x, kw = torch.randn(7, 5, generator=g), dict(dim=-1)
    elif case == "dtype_arg":
        x, kw = torch.randn(7, 5, generator=g), dict(dim=1, dtype=torch.float64)
    elif case == "wide":
        x, kw = torch.randn(3, 100, generator=g), dict(dim=1)
    elif case == "transposed":
        x, kw = torch.randn(5, 7, generator=g).t(), dict(dim=1)
    elif case == "empty":
        x, kw = torch.randn(0, 5), dict(dim=1)
    elif case == "1d":
        x, kw = torch.randn(9, generator=g), dict(dim=0)
    else:
        x, kw = torch.randn(7, 5, generator=g, dtype=torch.float64), dict(dim=1)
    x = x.clone().requires_grad_(True)
    got = F.log_softmax(x, **kw)
    ref = ls0(x, **kw)
    lazy_expected = case == "dim_neg1"
    assert isinstance(got, LazyLogSoftmax) == lazy_expected
    assert got.dtype == ref.dtype and _same(got + 0, ref)
    if x.numel():
        w = torch.randn(ref.shape, generator=g, dtype=ref.dtype)
        gg, = torch.autograd.grad((F.log_softmax(x, **kw) * w).sum(), x)
        gr, = torch.autograd.grad((ls0(x, **kw) * w).sum(), x)
        assert _same(gg, gr)


@pytest.mark.parametrize("case", ["3d_target2d", "reduction_none", "reduction_sum", "weights", "ignore_5", "ignore_all", "float_target",
                                  "label_smoothing_ce", "size_mismatch"])
def test_nll_loss_patch_third_party_calls(patched_loss, case):
    """F.nll_loss behind the patch == ATen's for what a third party may pass (K-dimensional input, other reductions, class
    weights, a non-negative ignore_index, every target ignored, errors raised for bad arguments)."""
    ls0, nll0 = patched_loss
    g = torch.Generator().manual_seed(2)
    n, c = 40, 6
    logits = torch.randn(n, c, generator=g, requires_grad=True)
    tgt = torch.randint(0, c, (n,), generator=g)
    kw = {}
    if case == "3d_target2d":
        logits = torch.randn(4, c, 5, generator=g, requires_grad=True)
        tgt = torch.randint(0, c, (4, 5), generator=g)
    elif case == "reduction_none":
        kw = dict(reduction="none")
    elif case == "reduction_sum":
        kw = dict(reduction="sum")
    elif case == "weights":
        kw = dict(weight=torch.rand(c, generator=g))
    elif case == "ignore_5":
        kw = dict(ignore_index=5)
    elif case == "ignore_all":
        tgt = torch.full((n,), -100)
    elif case == "float_target":
        with pytest.raises(Exception):
            F.nll_loss(ls0(logits, dim=1), tgt.float())
        return
    elif case == "size_mismatch":
        with pytest.raises(Exception):
            F.nll_loss(ls0(logits, dim=1), tgt[:-1])
        return
    elif case == "label_smoothing_ce":          # F.cross_entropy does not go through F.nll_loss: untouched
        got = F.cross_entropy(logits, tgt, label_smoothing=0.1)
        ref = -(0.9 * ls0(logits, dim=1).gather(1, tgt[:, None]).squeeze(1) + 0.1 * ls0(logits, dim=1).mean(1)).mean()
        assert abs(float(got) - float(ref)) <= 1e-6
        return
    dim = 1
    for lazy in (False, True):                   # an ordinary log-prob tensor, and the patched log_softmax's own result
        lp = F.log_softmax(logits, dim=dim) if lazy else ls0(logits, dim=dim)
        got = F.nll_loss(lp, tgt, **kw)
        ref = nll0(ls0(logits, dim=dim), tgt, **kw)
        if case == "ignore_all":
            assert torch.isnan(got) and torch.isnan(ref)
            continue
        assert _same(got, ref)
        gg, = torch.autograd.grad(got.sum(), logits)
        gr, = torch.autograd.grad(ref.sum(), logits)
        assert _same(gg, gr)


def test_lazy_rows_with_a_reused_and_then_mutated_index(patched_loss):
    """the uniqueness verdict of an index tensor is cached on (data_ptr, _version): an in-place edit that introduces a
    duplicate must be seen (ADVICE r04's silent-wrong-gradient case, through the cache)."""
    ls0, nll0 = patched_loss
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(30, 5, generator=g, requires_grad=True)
    y = torch.randint(0, 5, (30,), generator=g)
    idx = torch.tensor([0, 4, 9, 17])
    for _ in range(2):
        got = nn.NLLLoss()(F.log_softmax(logits, dim=1)[idx], y[idx])
        ref = nll0(ls0(logits, dim=1)[idx], y[idx])
        gg, = torch.autograd.grad(got, logits)
        gr, = torch.autograd.grad(ref, logits)
        assert _same(got, ref) and _same(gg, gr)
        idx[1] = 9                               # now [0, 9, 9, 17]: the same tensor object, a new _version


@pytest.fixture
def patched_adam(monkeypatch):
    from sgformer_amd import launch
    monkeypatch.delenv("SGF_FUSED_ADAM", raising=False)
    orig = torch.optim.Adam.__init__
    was = getattr(torch.optim.Adam, "_sgf_patched", False)
    torch.optim.Adam._sgf_patched = False
    launch.patch_adam()
    try:
        yield orig
    finally:
        torch.optim.Adam.__init__ = orig
        torch.optim.Adam._sgf_patched = was


@pytest.mark.parametrize("case", ["plain", "generator_group", "tensor_group", "capturable", "foreach_true", "fused_false", "amsgrad",
                                  "positional", "differentiable", "empty"])
def test_adam_patch_third_party_constructions(patched_adam, case):
    """torch.optim.Adam built the ways other code builds it: the patched constructor produces the optimizer the original
    does (same groups, same defaults, explicit fused / foreach respected) and one step moves the parameters identically.
    (CPU parameters: the patch must not turn `fused` on for them.)"""
    orig_init = patched_adam
    torch.manual_seed(0)
    m = nn.Linear(4, 3)
    m2 = nn.Linear(4, 3)
    m2.load_state_dict(m.state_dict())

    def build(mod, cls_init):
        args, kw = (), dict(lr=0.05)
        if case == "generator_group":
            params = [{"params": (p for p in mod.parameters()), "weight_decay": 1e-3}]
        elif case == "tensor_group":
            params = [{"params": mod.weight}, {"params": [mod.bias], "lr": 0.01}]
        elif case == "positional":
            params, args, kw = mod.parameters(), (0.05, (0.8, 0.9), 1e-7, 1e-4, False), {}
        elif case == "empty":
            params = []
        else:
            params = mod.parameters()
        if case == "capturable":
            kw["capturable"] = False             # (True needs CUDA parameters; the keyword itself must pass through)
        if case == "foreach_true":
            kw["foreach"] = True
        if case == "fused_false":
            kw["fused"] = False
        if case == "amsgrad":
            kw["amsgrad"] = True
        if case == "differentiable":
            kw["differentiable"] = False
        opt = torch.optim.Adam.__new__(torch.optim.Adam)
        cls_init(opt, params, *args, **kw)
        return opt

    if case == "empty":
        with pytest.raises(ValueError):
            build(m, torch.optim.Adam.__init__)
        return
    a, b = build(m, torch.optim.Adam.__init__), build(m2, orig_init)
    assert [len(g["params"]) for g in a.param_groups] == [len(g["params"]) for g in b.param_groups]
    for ga, gb in zip(a.param_groups, b.param_groups):
        for k in gb:
            if k != "params":
                assert ga[k] == gb[k], (k, ga[k], gb[k])
    x = torch.randn(5, 4)
    for mod, opt in ((m, a), (m2, b)):
        mod(x).pow(2).sum().backward()
        opt.step()
    assert torch.equal(m.weight, m2.weight) and torch.equal(m.bias, m2.bias)


def test_prologue_and_subgraph_patches_with_third_party_arguments(monkeypatch):
    """torch_geometric.utils.{subgraph, to_undirected, remove_self_loops, add_self_loops} as served by sgformer_amd.batching
    (PyG 1.7.2 signatures, the reference's pin): edge attributes, boolean / list subsets, num_nodes omitted, empty graphs —
    against plain restatements of PyG's semantics."""
    from sgformer_amd import batching, ops
    from tests.cpu_kernels import CpuKernels
    prev = ops.set_kernels(CpuKernels())
    try:
        g = torch.Generator().manual_seed(5)
        n = 30
        ei = torch.randint(0, n, (2, 120), generator=g)
        w = torch.rand(120, generator=g)
        # subgraph: list subset, boolean subset, edge attributes, no relabelling, num_nodes omitted
        keep = sorted(set(torch.randperm(n, generator=g)[:12].tolist()))
        mask = torch.zeros(n, dtype=torch.bool)
        mask[keep] = True
        em = mask[ei[0]] & mask[ei[1]]
        for subset in (keep, mask, torch.tensor(keep)):
            out, attr = batching.subgraph(subset, ei, edge_attr=w, relabel_nodes=False)
            assert torch.equal(out, ei[:, em]) and torch.equal(attr, w[em])
        relabel = torch.full((n,), -1, dtype=torch.long)
        relabel[torch.tensor(keep)] = torch.arange(len(keep))
        out, attr = batching.subgraph(torch.tensor(keep), ei, None, True, n)
        ref = relabel[ei[:, em]]
        assert attr is None and sorted(map(tuple, out.t().tolist())) == sorted(map(tuple, ref.t().tolist()))
        out, _ = batching.subgraph([], ei, relabel_nodes=True, num_nodes=n)
        assert out.shape == (2, 0)
        # remove_self_loops / add_self_loops with weights (the attribute-carrying forms)
        e2, w2 = batching.remove_self_loops(ei, w)
        m2 = ei[0] != ei[1]
        assert torch.equal(e2, ei[:, m2]) and torch.equal(w2, w[m2])
        e3, w3 = batching.add_self_loops(ei, w, fill_value=2.0, num_nodes=n)
        assert e3.shape[1] == 120 + n and torch.equal(e3[:, 120:], torch.arange(n).repeat(2, 1)) and bool((w3[120:] == 2.0).all())
        e4, none = batching.add_self_loops(ei)          # num_nodes omitted: max id + 1
        assert none is None and e4.shape[1] == 120 + int(ei.max()) + 1
        # to_undirected: both directions, coalesced, sorted by (row, col)
        und = batching.to_undirected(ei, n)
        key = torch.unique(torch.cat([ei[0] * n + ei[1], ei[1] * n + ei[0]]))
        assert torch.equal(und, torch.stack([key // n, key % n]))
        with pytest.raises(TypeError):                  # PyG >= 2 keywords are not silently swallowed
            batching.to_undirected(ei, edge_attr=w)
    finally:
        ops.set_kernels(prev)
        batching._parents.clear()


def test_minimal_mode_installs_the_module_and_nothing_else(tmp_path, monkeypatch):
    """`--sgf-patches minimal`: the trainer imports the drop-in as `ours`; F.log_softmax, F.nll_loss, torch.optim.Adam.__init__,
    torch_geometric.utils.* and the host thread count are exactly what they were."""
    from sgformer_amd import launch
    tdir = tmp_path / "large"
    tdir.mkdir()
    trainer = tdir / "main-batch.py"
    trainer.write_text(
        "import json, sys, torch, torch.nn.functional as F\n"
        "import ours, torch_geometric.utils as U\n"
        "json.dump({'ours': ours.__name__, 'nll': F.nll_loss.__module__, 'ls': F.log_softmax.__module__,\n"
        "           'adam': torch.optim.Adam.__init__.__module__, 'subgraph': U.subgraph, 'und': U.to_undirected,\n"
        "           'threads': torch.get_num_threads(), 'argv': sys.argv[1:]}, open(sys.argv[1], 'w'))\n")
    tg = types.ModuleType("torch_geometric")
    tgu = types.ModuleType("torch_geometric.utils")
    tgu.subgraph, tgu.to_undirected, tgu.remove_self_loops, tgu.add_self_loops = "pyg-subgraph", "pyg-und", "pyg-rsl", "pyg-asl"
    tg.utils = tgu
    monkeypatch.setitem(sys.modules, "torch_geometric", tg)
    monkeypatch.setitem(sys.modules, "torch_geometric.utils", tgu)
    monkeypatch.setitem(sys.modules, "ours", None)
    monkeypatch.setattr(sys, "argv", list(sys.argv))
    monkeypatch.setattr(sys, "path", list(sys.path))
    monkeypatch.delenv("OMP_NUM_THREADS", raising=False)
    threads = torch.get_num_threads()
    adam0, nll0, ls0 = torch.optim.Adam.__init__, F.nll_loss, F.log_softmax
    out = tmp_path / "out.json"
    try:
        launch.main(["--sgf-patches", "minimal", str(trainer), str(out), "--foo"])
        import json
        r = json.load(open(out))
        assert r["ours"] == "sgformer_amd.ours" and r["argv"] == [str(out), "--foo"]
        assert r["subgraph"] == "pyg-subgraph" and r["und"] == "pyg-und" and r["threads"] == threads
        assert torch.optim.Adam.__init__ is adam0 and F.nll_loss is nll0 and F.log_softmax is ls0
        with pytest.raises(SystemExit):
            launch.main(["--sgf-patches", "some", str(trainer), str(out)])
    finally:
        torch.optim.Adam.__init__ = adam0
        launch.unpatch_nll_loss()
        torch.set_num_threads(threads)


def test_staging_wrappers_are_plain_tensors_without_a_gpu():
    """sgformer_amd/staging.py on a host without CUDA (or with SGF_PREP_STREAM=0): the wrappers the launcher puts around the
    mini-batch trainer's features / labels behave as the tensors they wrap — same values, plain results, host semantics of
    the labels untouched (large/main-batch.py:45-46, :66, :102-107, :115-116 all run on them)."""
    from sgformer_amd import staging
    lab = torch.randint(0, 5, (50,))
    ls = staging.staged(lab)
    assert isinstance(ls, staging.StagedHost) and not ls.is_cuda and torch.equal(ls.as_subclass(torch.Tensor), lab)
    ls = ls.unsqueeze(1)                                        # :46
    assert isinstance(ls, staging.StagedHost) and ls.shape == (50, 1)
    assert int(max(ls.max().item() + 1, ls.shape[1])) == int(lab.max()) + 1      # :66
    one_hot = F.one_hot(ls, int(ls.max()) + 1).squeeze(1)      # :103
    assert type(one_hot) is torch.Tensor and one_hot.shape == (50, int(lab.max()) + 1)
    idx = torch.randperm(50)[:20]
    rows = ls[idx]
    assert isinstance(rows, staging.StagedHost) and torch.equal(rows.as_subclass(torch.Tensor), lab.unsqueeze(1)[idx])
    assert type(rows.to(torch.float)) is torch.Tensor and type(rows.to("cpu")) in (torch.Tensor, staging.StagedHost)
    mask = torch.zeros(50, dtype=torch.bool)
    mask[ls.squeeze(1) == 2] = True                             # comparisons / mask building on the host
    assert int(mask.sum()) == int((lab == 2).sum())
    assert rows.numpy().shape == (20, 1) and type(ls + 1) is torch.Tensor
    x = torch.randn(50, 8)
    assert staging.resident(x) is x                             # a host tensor is left alone
    assert staging._cuda_target((ls, torch.device("cuda", 0)), {}) == torch.device("cuda", 0)
    assert staging._cuda_target((ls, "cuda:1"), {"non_blocking": True}) == torch.device("cuda", 1)
    assert staging._cuda_target((ls, torch.float32), {}) is None and staging._cuda_target((ls, "cpu"), {}) is None
    assert staging._cuda_target((ls, torch.device("cuda", 0), torch.float16), {}) is None
