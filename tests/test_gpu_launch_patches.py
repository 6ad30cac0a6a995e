"""The launcher's global patches on the GPU with callers that are not the reference's trainers: the
one-pass loss kernels (sgf_nll_fwd / _bwd) behind F.log_softmax / F.nll_loss, and the fused-Adam default — against ATen /
the unpatched optimizer on the same CUDA tensors.  (tests/test_launch_patches.py holds the CPU-side cases.)"""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture
def patched(cuda):
    from sgformer_amd import launch
    ls0, nll0 = F.log_softmax, F.nll_loss
    launch.patch_nll_loss()
    try:
        yield ls0, nll0
    finally:
        launch.unpatch_nll_loss()


def _close(a, b, tol=2e-6):
    return float((a.detach().double() - b.detach().double()).abs().max()) <= tol


@pytest.mark.parametrize("case", ["trainer", "duplicates", "negative", "ignore_index", "bool_mask", "weights", "sum", "3d", "dim0",
                                  "wide", "bf16_logits", "two_uses"])
def test_loss_patches_on_cuda(patched, cuda, case):
    from sgformer_amd.loss import LazyLogSoftmax
    ls0, nll0 = patched
    g = torch.Generator().manual_seed(11)
    n, c = 3000, 47
    logits = torch.randn(n, c, generator=g).to(cuda).requires_grad_(True)
    y = torch.randint(0, c, (n,), generator=g).to(cuda)
    idx = torch.randperm(n, generator=g)[:1200].to(cuda)
    crit = nn.NLLLoss()
    kw = {}
    if case == "duplicates":
        idx = torch.tensor([5, 9, 9, 700, 5, 2999], device=cuda)
    elif case == "negative":
        idx = torch.tensor([-1, 3, 17, -2999], device=cuda)
    elif case == "bool_mask":
        m = torch.zeros(n, dtype=torch.bool, device=cuda)
        m[idx] = True
        idx = m
    elif case == "weights":
        kw = dict(weight=torch.rand(c, generator=g).to(cuda))
    elif case == "sum":
        kw = dict(reduction="sum")
    if case == "3d":
        x = torch.randn(6, c, 9, generator=g).to(cuda).requires_grad_(True)
        t = torch.randint(0, c, (6, 9), generator=g).to(cuda)
        got, ref = F.nll_loss(F.log_softmax(x, dim=1), t), nll0(ls0(x, dim=1), t)
        gg, = torch.autograd.grad(got, x)
        gr, = torch.autograd.grad(ref, x)
        assert _close(got, ref) and _close(gg, gr)
        return
    if case == "dim0":
        out = F.log_softmax(logits, dim=0)
        assert not isinstance(out, LazyLogSoftmax) and _close(out, ls0(logits, dim=0))
        return
    if case == "wide":
        x = torch.randn(50, 172, generator=g).to(cuda)
        assert not isinstance(F.log_softmax(x, dim=1), LazyLogSoftmax) and _close(F.log_softmax(x, dim=1), ls0(x, dim=1))
        return
    if case == "bf16_logits":
        lb = logits.detach().bfloat16().requires_grad_(True)
        got = crit(F.log_softmax(lb, dim=1)[idx], y[idx])
        ref = nll0(ls0(lb.float(), dim=1)[idx], y[idx])
        assert abs(float(got) - float(ref)) <= 2e-2
        return
    t = y[idx]
    if case == "ignore_index":
        t = t.clone()
        t[::4] = -100
    out = F.log_softmax(logits, dim=1)
    rows = out[idx]
    got = F.nll_loss(rows, t, **kw) if kw else crit(rows, t)
    ref = nll0(ls0(logits, dim=1)[idx], t, **kw)
    gg, = torch.autograd.grad(got, logits, retain_graph=True)
    gr, = torch.autograd.grad(ref, logits)
    assert _close(got, ref, 2e-6 * max(1.0, abs(float(ref)))) and _close(gg, gr, 1e-6 * max(1.0, float(gr.abs().max())))
    if case == "two_uses":                       # the same lazy object used twice: argmax (materialised) after the loss
        assert torch.equal(out.argmax(1), ls0(logits, dim=1).argmax(1))
        assert _close(out.exp().sum(1), torch.ones(n, device=cuda), 1e-4)


@pytest.mark.parametrize("case", ["two_groups", "generator_group", "capturable", "foreach", "amsgrad", "mixed_devices"])
def test_adam_patch_on_cuda(cuda, monkeypatch, case):
    """CUDA parameters: the patched constructor turns torch's single-kernel form on (same arithmetic) unless the caller
    said otherwise; five steps move the parameters as the unpatched optimizer does (1e-6), whatever the construction."""
    from sgformer_amd import launch
    monkeypatch.delenv("SGF_FUSED_ADAM", raising=False)
    orig = torch.optim.Adam.__init__
    was = getattr(torch.optim.Adam, "_sgf_patched", False)
    torch.optim.Adam._sgf_patched = False
    launch.patch_adam()
    patched_init = torch.optim.Adam.__init__
    try:
        torch.manual_seed(0)
        a = nn.Sequential(nn.Linear(8, 16), nn.Linear(16, 4)).to(cuda)
        b = nn.Sequential(nn.Linear(8, 16), nn.Linear(16, 4)).to(cuda)
        b.load_state_dict(a.state_dict())
        extra_a, extra_b = nn.Linear(4, 4), nn.Linear(4, 4)          # (host parameters for the mixed case)
        extra_b.load_state_dict(extra_a.state_dict())

        def build(mod, extra, init):
            kw = dict(lr=0.01)
            if case == "two_groups":
                params = [{"params": mod[0].parameters(), "weight_decay": 1e-5}, {"params": mod[1].parameters(), "weight_decay": 1e-3}]
            elif case == "generator_group":
                params = [{"params": (p for p in mod.parameters())}]
            elif case == "mixed_devices":
                params = list(mod.parameters()) + list(extra.parameters())
            else:
                params = mod.parameters()
            if case == "capturable":
                kw["capturable"] = True
            if case == "foreach":
                kw["foreach"] = True
            if case == "amsgrad":
                kw["amsgrad"] = True
            opt = torch.optim.Adam.__new__(torch.optim.Adam)
            init(opt, params, **kw)
            return opt

        oa, ob = build(a, extra_a, patched_init), build(b, extra_b, orig)
        fused = [g.get("fused") for g in oa.param_groups]
        if case in ("foreach", "mixed_devices"):
            assert not any(fused)                # an explicit foreach is respected; host parameters: never fused
        else:
            assert all(fused)
        x = torch.randn(32, 8, device=cuda)
        for _ in range(5):
            for mod, extra, opt in ((a, extra_a, oa), (b, extra_b, ob)):
                opt.zero_grad()
                out = mod(x)
                loss = out.pow(2).mean() + (extra(out.detach().cpu()).pow(2).mean() if case == "mixed_devices" else 0.0)
                loss.backward()
                opt.step()
        for pa, pb in zip(a.parameters(), b.parameters()):
            assert float((pa - pb).abs().max()) <= 1e-6
    finally:
        torch.optim.Adam.__init__ = orig
        torch.optim.Adam._sgf_patched = was
