"""End-to-end parity of the drop-in module (sgformer_amd/ours.py) against the oracle restatement of
large/ours.py: logits, loss, every parameter gradient, BatchNorm running statistics, a short
training trajectory, and state_dict interchange.  Tolerance: 1e-4 absolute on fp32 logits
(BASELINE.json north_star), relative on gradients.
"""
import copy

import pytest
import torch

from oracle import sgformer_oracle as O

pytestmark = pytest.mark.gpu

CONFIGS = {
    # large/run.sh:2-5 (ogbn-arxiv recipe; dropout off for parity)
    "arxiv": dict(trans_num_layers=1, trans_num_heads=1, trans_use_bn=True, trans_use_residual=True,
                  trans_use_weight=True, trans_use_act=False, gnn_num_layers=3, gnn_use_bn=True,
                  gnn_use_residual=True, gnn_use_weight=True, gnn_use_init=False, gnn_use_act=True,
                  graph_weight=0.5, aggregate="add"),
    # large/run.sh:15-19 (amazon2m / products recipe): use_init
    "products": dict(trans_num_layers=1, trans_num_heads=1, trans_use_bn=True, trans_use_residual=True,
                     trans_use_weight=True, trans_use_act=False, gnn_num_layers=3, gnn_use_bn=True,
                     gnn_use_residual=True, gnn_use_weight=True, gnn_use_init=True, gnn_use_act=True,
                     graph_weight=0.5, aggregate="add"),
    "heads_cat": dict(trans_num_layers=2, trans_num_heads=2, trans_use_act=True, gnn_num_layers=1,
                      aggregate="cat"),
    "bare": dict(trans_use_weight=False, trans_use_bn=False, trans_use_residual=False,
                 trans_use_act=False, gnn_use_weight=False, gnn_use_bn=False, gnn_use_residual=False,
                 gnn_use_act=False, gnn_num_layers=2),
    "no_graph": dict(use_graph=False, trans_num_layers=2),
    "alpha_100m": dict(alpha=0.3, trans_num_layers=1, gnn_num_layers=2, gnn_use_init=True),
}


def _build(cfg, f, d, c, cuda, seed=0):
    from sgformer_amd.ours import SGFormer
    p = O.init_params(cfg, f, d, c, seed=seed)
    m = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, **cfg)
    sd = m.state_dict()
    for k, v in p.items():
        assert sd[k].shape == v.shape, k
    missing = [k for k in sd if k not in p and "num_batches_tracked" not in k]
    assert not missing, missing
    m.load_state_dict({**sd, **p})
    return m.to(cuda), p


@pytest.mark.parametrize("name", list(CONFIGS.keys()))
@pytest.mark.parametrize("shape", [(700, 24, 64, 7), (1500, 36, 128, 5), (2100, 40, 256, 10)])
def test_forward_backward_parity(cuda, name, shape):
    cfg = CONFIGS[name]
    n, f, d, c = shape
    torch.manual_seed(n)
    x = torch.randn(n, f)
    ei = O.synthetic_graph(n, 7.0, seed=n)
    y = torch.randint(0, c, (n,))
    idx = torch.randperm(n)[: n // 2]
    m, p = _build(cfg, f, d, c, cuda)

    # oracle in fp64 (training mode: batch statistics)
    p64 = {k: v.double().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in p.items()}
    stats = {}
    logits_ref = O.sgformer_forward(p64, x.double(), ei, cfg, training=True, bn_stats=stats)
    loss_ref = O.nll_loss(logits_ref, y, idx)
    loss_ref.backward()

    m.train()
    logits = m(x.to(cuda), ei.to(cuda))
    loss = O.nll_loss(logits, y.to(cuda), idx.to(cuda))
    loss.backward()
    assert float((logits.detach().double().cpu() - logits_ref.detach()).abs().max()) <= 1e-4
    assert abs(float(loss) - float(loss_ref)) <= 1e-5
    gmax = max(float(v.grad.norm()) for v in p64.values() if v.grad is not None)
    for k, prm in m.named_parameters():
        g_ref = p64[k].grad
        if g_ref is None:              # unused W of GraphConvLayer (large/ours.py:20 vs :36-40)
            assert prm.grad is None or float(prm.grad.abs().max()) == 0.0, k
            continue
        num = float((prm.grad.double().cpu() - g_ref).norm())
        den = float(g_ref.norm())
        # floor: a Linear bias in front of BatchNorm has an exactly-zero gradient in exact arithmetic
        assert num <= 2e-4 * den + 1e-6 * gmax, (k, num, den)
    # BatchNorm running statistics after one step (momentum 0.1 from the init_params values)
    if cfg.get("gnn_use_bn", True) and cfg.get("use_graph", True):
        for key, (mu, var_unb) in stats.items():
            rm = 0.9 * p[key + ".running_mean"].double() + 0.1 * mu
            rv = 0.9 * p[key + ".running_var"].double() + 0.1 * var_unb
            sd = m.state_dict()
            assert float((sd[key + ".running_mean"].double().cpu() - rm).abs().max()) <= 1e-5
            assert float((sd[key + ".running_var"].double().cpu() - rv).abs().max()) <= 1e-5

    # eval mode: running statistics, no dropout
    m.eval()
    with torch.no_grad():
        logits_e = m(x.to(cuda), ei.to(cuda))
    pe = {k: v.double().cpu() for k, v in m.state_dict().items()}
    ref_e = O.sgformer_forward(pe, x.double(), ei, cfg, training=False)
    assert float((logits_e.double().cpu() - ref_e).abs().max()) <= 1e-4


def test_training_trajectory(cuda):
    """5 Adam steps with the reference's two parameter groups (large/main.py:114-119): the loss
    curve of the GPU module tracks the fp32 CPU oracle."""
    cfg = CONFIGS["arxiv"]
    n, f, d, c = 1200, 32, 64, 6
    torch.manual_seed(1)
    x = torch.randn(n, f)
    ei = O.synthetic_graph(n, 6.0, seed=9)
    y = torch.randint(0, c, (n,))
    idx = torch.arange(0, n, 2)
    m, p = _build(cfg, f, d, c, cuda)
    pc = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in p.items()}
    opt_g = torch.optim.Adam([{"params": m.params1, "weight_decay": 1e-3},
                              {"params": m.params2, "weight_decay": 5e-4}], lr=0.01)
    names1 = [k for k, _ in m.trans_conv.named_parameters(prefix="trans_conv")]
    names2 = [k for k, _ in m.graph_conv.named_parameters(prefix="graph_conv")] + ["fc.weight", "fc.bias"]
    opt_c = torch.optim.Adam([{"params": [pc[k] for k in names1], "weight_decay": 1e-3},
                              {"params": [pc[k] for k in names2], "weight_decay": 5e-4}], lr=0.01)
    m.train()
    for step in range(5):
        opt_g.zero_grad()
        lg = O.nll_loss(m(x.to(cuda), ei.to(cuda)), y.to(cuda), idx.to(cuda))
        lg.backward()
        opt_g.step()
        opt_c.zero_grad()
        stats = {}
        lc = O.nll_loss(O.sgformer_forward(pc, x, ei, cfg, training=True, bn_stats=stats), y, idx)
        lc.backward()
        opt_c.step()
        with torch.no_grad():
            for key, (mu, vu) in stats.items():
                pc[key + ".running_mean"].mul_(0.9).add_(0.1 * mu)
                pc[key + ".running_var"].mul_(0.9).add_(0.1 * vu)
        assert abs(float(lg) - float(lc)) <= 2e-4 * max(1.0, abs(float(lc))), (step, float(lg), float(lc))


def test_surface_matches_reference_contract(cuda):
    """ctor keywords, state_dict keys / shapes, params1 / params2, reset_parameters."""
    from sgformer_amd.ours import SGFormer
    kw = dict(trans_num_layers=1, trans_num_heads=1, trans_dropout=0.5, trans_use_bn=True,
              trans_use_residual=True, trans_use_weight=True, trans_use_act=False, gnn_num_layers=3,
              gnn_dropout=0.5, gnn_use_weight=True, gnn_use_init=True, gnn_use_bn=True,
              gnn_use_residual=True, gnn_use_act=True, use_graph=True, graph_weight=0.5, aggregate="add")
    m = SGFormer(128, 256, 40, **kw)
    sd = m.state_dict()
    assert len(sd) == 42
    assert sum(p.numel() for p in m.parameters()) == 670760
    assert len(m.params1) == 12 and len(m.params2) == 18
    assert sd["graph_conv.convs.0.W.weight"].shape == (256, 512)
    assert sd["trans_conv.convs.0.Wq.weight"].shape == (256, 256)
    before = copy.deepcopy(sd)
    m.reset_parameters()
    after = m.state_dict()
    assert torch.equal(before["fc.weight"], after["fc.weight"])          # fc is never reset
    assert not torch.equal(before["trans_conv.fcs.0.weight"], after["trans_conv.fcs.0.weight"])
    m = m.to(cuda)
    x = torch.randn(300, 128, device=cuda)
    ei = O.synthetic_graph(300, 5.0, seed=2).to(cuda)
    m.train()
    out = m(x, ei)                                                          # dropout 0.5 active
    assert out.shape == (300, 40) and torch.isfinite(out).all()
    out.sum().backward()
    att = m.get_attentions(x)
    assert att.shape == (1, 300, 300)


@pytest.mark.parametrize("cfg_name,n,f,d", [("products", 3000, 40, 256),     # two-operand Linear + stems (f % 4 == 0)
                                            ("arxiv", 2000, 24, 128),        # single-operand Linear + statistics
                                            ("products", 1700, 100, 64),     # the headline's 100 features: ragged k-step
                                            ("alpha_100m", 1500, 30, 64),    # f % 4 != 0: the stems stay on the library
                                            ("heads_cat", 1200, 16, 64)])    # two heads, 'cat': materialised attention
def test_bf16_activation_mode(cuda, cfg_name, n, f, d):
    """BASELINE.json config 3: bf16 activation storage (fp32 master weights, fp32 accumulation in
    every kernel).  The reference has no bf16 path, so parity is defined
    against the fp64 oracle with a tolerance set from the measured bf16 error: every activation is
    rounded to 8 mantissa bits (2^-9 relative) once per op, ~25 ops deep.  The recipes route the Linear layers
    through every form of the streaming row kernels (csrc/rowgemm.hip) and through the library fallbacks."""
    cfg = CONFIGS[cfg_name]
    c = 10
    torch.manual_seed(3)
    x = torch.randn(n, f)
    ei = O.synthetic_graph(n, 8.0, seed=5)
    y = torch.randint(0, c, (n,))
    idx = torch.randperm(n)[: n // 2]
    m, p = _build(cfg, f, d, c, cuda)
    m.compute_dtype = torch.bfloat16
    m.train()
    logits = m(x.to(cuda), ei.to(cuda))
    assert logits.dtype == torch.float32
    loss = O.nll_loss(logits, y.to(cuda), idx.to(cuda))
    loss.backward()
    p64 = {k: v.double().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in p.items()}
    ref = O.sgformer_forward(p64, x.double(), ei, cfg, training=True)
    lref = O.nll_loss(ref, y, idx)
    lref.backward()
    rel = float((logits.detach().double().cpu() - ref.detach()).norm() / ref.detach().norm())
    assert rel <= 3e-2, rel
    assert abs(float(loss) - float(lref)) <= 3e-2 * abs(float(lref))
    for k, prm in m.named_parameters():
        assert prm.grad is not None and prm.grad.dtype == torch.float32 and torch.isfinite(prm.grad).all(), k
    # the big, well-conditioned gradients agree to bf16 accuracy
    last = max(int(k.split(".")[2]) for k, _ in m.named_parameters() if k.startswith("graph_conv.convs."))
    for k in ["fc.weight", f"graph_conv.convs.{last}.W.weight", "graph_conv.fcs.0.weight"]:
        g, gr = dict(m.named_parameters())[k].grad.double().cpu(), p64[k].grad
        assert float((g - gr).norm() / gr.norm()) <= 0.12, k      # three BatchNorm layers deep in bf16: 5-8 % measured


def test_host_tensors_are_evaluated_on_the_gpu(cuda):
    """large/eval.py:36-65 (`evaluate_large`) moves the model and the graph to the CPU and calls
    model(x, edge_index) under no_grad: the drop-in stages copies to the GPU, runs the HIP path and
    returns host logits (no CPU implementation exists); with autograd enabled it refuses."""
    cfg = CONFIGS["products"]
    n, f, d, c = 900, 24, 64, 7
    torch.manual_seed(1)
    x = torch.randn(n, f)
    ei = O.synthetic_graph(n, 6.0, seed=2)
    m, _ = _build(cfg, f, d, c, cuda)
    m.eval()
    with torch.no_grad():
        ref = m(x.to(cuda), ei.to(cuda)).cpu()
        m.to("cpu")
        out = m(x, ei)
    assert out.device.type == "cpu" and torch.equal(out, ref)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(x, ei)
    m.to(cuda)
    with torch.no_grad():
        assert torch.equal(m(x.to(cuda), ei.to(cuda)).cpu(), ref)


def test_training_with_dropout(cuda):
    """The arxiv recipe trains with dropout 0.5 (large/run.sh:2-5): fused dropout(+residual) kernels in
    both branches; stochastic in train mode, deterministic and dropout-free in eval mode, finite grads,
    and E[output] over many masks close to the dropout-free output scale (sanity of the 1/(1-p) scaling)."""
    cfg = CONFIGS["arxiv"]
    n, f, d, c = 1500, 20, 64, 6
    torch.manual_seed(2)
    x = torch.randn(n, f)
    ei = O.synthetic_graph(n, 7.0, seed=3)
    from sgformer_amd.ours import SGFormer
    m = SGFormer(f, d, c, trans_dropout=0.5, gnn_dropout=0.5, **cfg).to(cuda)
    xg, eig = x.to(cuda), ei.to(cuda)
    m.train()
    a, b = m(xg, eig), m(xg, eig)
    assert float((a - b).abs().max()) > 1e-3                     # different masks
    loss = a.float().logsumexp(1).mean()
    loss.backward()
    for k, prm in m.named_parameters():
        if prm.grad is not None:
            assert bool(torch.isfinite(prm.grad).all()), k
    m.eval()
    with torch.no_grad():
        e1, e2 = m(xg, eig), m(xg, eig)
    assert torch.equal(e1, e2)
    m0 = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, **cfg).to(cuda)
    m0.load_state_dict(m.state_dict())
    m0.eval()
    with torch.no_grad():
        assert torch.equal(m0(xg, eig), e1)


class _Data:
    def __init__(self, x, ei):
        self.graph = {"node_feat": x, "edge_index": ei}


@pytest.mark.parametrize("shape,gcn_layers,cfg", [
    ((2708, 1433, 64, 7), 4, dict(num_layers=1, alpha=0.5, graph_weight=0.8)),        # Cora, medium/run.sh:2-7
    ((900, 50, 128, 5), 2, dict(num_layers=2, num_heads=2, use_weight=False, aggregate="cat")),
])
def test_medium_variant_parity(cuda, shape, gcn_layers, cfg):
    """BASELINE.json config 1: medium/ours.py SGFormer(data) with the GCN backbone (models.GCN over
    GCNConv), Cora shape, fp32: logits within 1e-4 of the fp64 oracle, gradients relative."""
    from sgformer_amd import ours_medium as M
    n, f, d, c = shape
    torch.manual_seed(7)
    gnn = M.GCN(f, d, d, num_layers=gcn_layers, dropout=0.0)
    m = M.SGFormer(f, d, c, dropout=0.0, gnn=gnn, **cfg)
    with torch.no_grad():
        for k, v in m.state_dict().items():
            if k.endswith("bias"):
                v.normal_(0, 0.1)
    x = (torch.rand(n, f) < 0.02).float()                      # bag-of-words like Cora
    x = x / x.sum(1, keepdim=True).clamp_min(1.0)
    ei = O.synthetic_graph(n, 3.9, seed=6)[:, :-n]
    y = torch.randint(0, c, (n,))
    idx = torch.randperm(n)[:140]
    p = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to(cuda).train()
    logits = m(_Data(x.to(cuda), ei.to(cuda)))
    loss = O.nll_loss(logits, y.to(cuda), idx.to(cuda))
    loss.backward()
    p64 = {k: v.double().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in p.items()}
    ref = O.medium_forward(p64, x.double(), ei, cfg, training=True)
    O.nll_loss(ref, y, idx).backward()
    assert float((logits.detach().double().cpu() - ref.detach()).abs().max()) <= 1e-4
    for k, prm in m.named_parameters():
        g = p64[k].grad
        assert prm.grad is not None and g is not None, k
        if float(g.norm()) > 1e-9:
            rel = float((prm.grad.double().cpu() - g).norm() / g.norm())
            assert rel <= 5e-4, (k, rel)
    m.eval()
    pe = {k: v.detach().double().cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        le = m(_Data(x.to(cuda), ei.to(cuda)))
    assert float((le.double().cpu() - O.medium_forward(pe, x.double(), ei, cfg, training=False)).abs().max()) <= 1e-4


# ------------------------------------------------------------------------------------------------
# BASELINE.json configs 2-5 at full hidden width, through size-independent properties
# ------------------------------------------------------------------------------------------------
def _recipe(name):
    from sgformer_amd import synth
    return dict(synth.RECIPES[name])


@pytest.mark.parametrize("name,n,f,d,c,deg,dtype", [
    ("ogbn-arxiv", 40000, 128, 256, 40, 13.7, torch.float32),       # config 2
    ("ogbn-products", 60000, 100, 256, 47, 50.5, torch.bfloat16),   # config 3
    ("pokec", 50000, 65, 256, 2, 27.3, torch.float32),              # config 4 (labels contain -1)
])
def test_baseline_configs_properties(cuda, name, n, f, d, c, deg, dtype):
    """Recipes of large/run.sh at the hidden width BASELINE.json names, on graphs too large for the
    fp64 oracle to be cheap: (1) node-permutation equivariance — relabelling the nodes permutes the
    logits and leaves the loss unchanged (every reduction in the path is over all nodes, so this
    exercises CSR build, SpMM, both attention reductions and BatchNorm statistics at once);
    (2) all gradients finite, fp32 master grads; (3) a first-order check of the whole backward: a
    gradient step sized for a 2 % decrease of the loss decreases it by about 2 %."""
    from sgformer_amd.ours import SGFormer
    cfg = _recipe(name)
    torch.manual_seed(11)
    x = torch.randn(n, f)
    ei = O.synthetic_graph(n, deg, seed=13)
    y = torch.randint(0, c, (n,))
    if name == "pokec":
        y[torch.rand(n) < 0.3] = -1                                  # unlabeled nodes (large/data_utils.py:15-16)
    labeled = (y >= 0).nonzero().view(-1)
    idx = labeled[torch.randperm(labeled.numel())[: labeled.numel() // 2]]
    m = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0,
                 compute_dtype=None if dtype == torch.float32 else dtype, **cfg).to(cuda).train()
    xg, eig, yg, idxg = x.to(cuda), ei.to(cuda), y.to(cuda), idx.to(cuda)
    logits = m(xg, eig)
    loss = O.nll_loss(logits, yg, idxg)
    # (1) permutation equivariance
    perm = torch.randperm(n)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(n)
    xp, eip = x[perm], inv[ei]                                       # new id of old node v is inv[v]
    logits_p = m(xp.to(cuda), eip.to(cuda))
    tol = 2e-4 if dtype == torch.float32 else 6e-2
    scale = float(logits.detach().abs().max())
    assert float((logits_p.detach()[inv.to(cuda)] - logits.detach()).abs().max()) <= tol * max(scale, 1.0)
    # (2) + (3): all gradients finite and fp32; and a directional-derivative check of the WHOLE backward:
    # a plain gradient step sized for a 0.4 % first-order decrease must decrease the loss by about that
    m.zero_grad(set_to_none=True)
    loss.backward()
    ps = [prm for prm in m.parameters() if prm.grad is not None]
    for k, prm in m.named_parameters():
        if prm.grad is not None:
            assert prm.grad.dtype == torch.float32 and bool(torch.isfinite(prm.grad).all()), k
    gnorm2 = float(sum((prm.grad.double() ** 2).sum() for prm in ps))
    l0 = float(loss)
    frac = 0.004                                   # small enough for the first-order model to hold
    eps = frac * l0 / gnorm2
    with torch.no_grad():
        for prm in ps:
            prm.add_(prm.grad, alpha=-eps)
        l1 = float(O.nll_loss(m(xg, eig), yg, idxg))
    lo, hi = (0.75, 1.25) if dtype == torch.float32 else (0.5, 1.5)
    assert lo * frac * l0 <= l0 - l1 <= hi * frac * l0, (l0, l1, frac * l0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg,d", [(dict(num_layers=2, num_heads=1), 64),
                                   (dict(num_layers=1, num_heads=2, use_weight=False, graph_weight=0.3, use_source=True), 32),
                                   (dict(num_layers=2, num_heads=1, alpha=0.2, graph_weight=0.7), 256)])
def test_difformer_parity(cuda, dtype, cfg, d):
    """medium/difformer.py drop-in (row N4): logits and parameter gradients against the fp64 oracle."""
    from sgformer_amd import difformer as M
    n, f, c = 3000, 40, 7
    torch.manual_seed(5)
    m = M.DIFFormer(f, d, c, dropout=0.0, **cfg)
    with torch.no_grad():
        for k, v in m.state_dict().items():
            if k.endswith("bias"):
                v.normal_(0, 0.1)
    x = torch.randn(n, f)
    ei = O.synthetic_graph(n, 8.0, seed=6)
    y = torch.randint(0, c, (n,))
    idx = torch.randperm(n)[: n // 2]
    p = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to(cuda).train()
    logits = m(_Data(x.to(cuda, dtype), ei.to(cuda)))
    O.nll_loss(logits.float(), y.to(cuda), idx.to(cuda)).backward()
    p64 = {k: v.double().requires_grad_(True) for k, v in p.items()}
    ref = O.difformer_forward(p64, x.to(dtype).double(), ei, cfg)
    O.nll_loss(ref, y, idx).backward()
    if dtype == torch.float32:
        assert float((logits.detach().double().cpu() - ref.detach()).abs().max()) <= 1e-4
    else:
        assert float((logits.detach().double().cpu() - ref.detach()).norm() / ref.detach().norm()) <= 3e-2
    gmax = max(float(v.grad.norm()) for v in p64.values() if v.grad is not None)
    for k, prm in m.named_parameters():
        g = p64[k].grad
        if g is None:
            continue
        assert prm.grad is not None and prm.grad.dtype == torch.float32, k
        err = float((prm.grad.double().cpu() - g).norm())
        tol = 1e-3 if dtype == torch.float32 else 8e-2
        assert err <= tol * (float(g.norm()) + 1e-2 * gmax), (k, err, float(g.norm()))


# ------------------------------------------------------------------------------------------------
# property test over the constructor-flag space
# ------------------------------------------------------------------------------------------------
try:
    from hypothesis import HealthCheck, given, settings, strategies as st
    _HAVE_HYP = True
except Exception:  # pragma: no cover
    _HAVE_HYP = False

if _HAVE_HYP:
    _flags = st.fixed_dictionaries(dict(
        trans_num_layers=st.integers(1, 2), trans_num_heads=st.integers(1, 2), trans_use_bn=st.booleans(),
        trans_use_residual=st.booleans(), trans_use_weight=st.booleans(), trans_use_act=st.booleans(),
        gnn_num_layers=st.integers(1, 3), gnn_use_weight=st.booleans(), gnn_use_init=st.booleans(),
        gnn_use_bn=st.booleans(), gnn_use_residual=st.booleans(), gnn_use_act=st.booleans(),
        use_graph=st.booleans(), graph_weight=st.sampled_from([0.2, 0.5, 0.8]),
        aggregate=st.sampled_from(["add", "cat"]), alpha=st.sampled_from([None, 0.3])))

    @settings(max_examples=30, deadline=None, derandomize=True,
              suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
    @given(cfg=_flags, n=st.integers(5, 700), d=st.sampled_from([16, 32, 64, 100]), directed=st.booleans(),
           seed=st.integers(0, 10 ** 6))
    def test_flag_space_parity(cuda, cfg, n, d, directed, seed):
        """Random points of the constructor-flag space x ragged sizes (N not a multiple of any tile,
        directed graphs -> transposed CSR in the backward, isolated nodes): logits within 1e-4 of
        the fp64 oracle, every parameter gradient within 2e-3 relative (pooled norm)."""
        from hypothesis import assume
        # use_graph=False with aggregate='cat' crashes in the reference too (fc expects 2d inputs,
        # large/ours.py:257-259 vs :273-275): not a point of the valid flag space
        assume(cfg["use_graph"] or cfg["aggregate"] == "add")
        f, c = 12, 5
        torch.manual_seed(seed)
        x = torch.randn(n, f)
        ei = O.synthetic_graph(n, 4.0, seed=seed % 1000, directed=directed)
        y = torch.randint(0, c, (n,))
        idx = torch.randperm(n)[: max(n // 2, 1)]
        m, p = _build(cfg, f, d, c, cuda, seed=seed % 97)
        if n < 2 and cfg["gnn_use_bn"]:
            return
        m.train()
        logits = m(x.to(cuda), ei.to(cuda))
        O.nll_loss(logits, y.to(cuda), idx.to(cuda)).backward()
        p64 = {k: v.double().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in p.items()}
        ref = O.sgformer_forward(p64, x.double(), ei, cfg, training=True)
        O.nll_loss(ref, y, idx).backward()
        assert float((logits.detach().double().cpu() - ref.detach()).abs().max()) <= 1e-4
        gmax = max([float(v.grad.norm()) for v in p64.values() if v.grad is not None] + [1e-30])
        for k, prm in m.named_parameters():
            g = p64[k].grad
            if g is None:
                continue
            assert prm.grad is not None, k
            err = float((prm.grad.double().cpu() - g).norm())
            assert err <= 2e-3 * (float(g.norm()) + 1e-3 * gmax), (k, err, float(g.norm()))
