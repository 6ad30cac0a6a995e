"""CPU tests of everything above the C ABI (no GPU, no kernels launched):
  * the C-ABI library loads and exports every symbol include/sgf.h declares, with the arity the
    ctypes binding uses;
  * ops.py / ours.py host logic, driven by the CPU kernel table of tests/cpu_kernels.py, against
    the oracle (forward, backward, BatchNorm bookkeeping, surface);
  * the graph cache rules of SURVEY.md Appendix A.
"""
import ctypes
import os
import re

import pytest
import torch

from oracle import sgformer_oracle as O
from tests.cpu_kernels import CpuKernels

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------------------------
# C ABI
# ------------------------------------------------------------------------------------------------
def _header_decls():
    src = open(os.path.join(ROOT, "include", "sgf.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(?:int|int32_t|size_t|int64_t|const char\*)\s+(sgf_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        decls[m.group(1)] = n
    return decls


def test_header_binding_and_library_agree():
    from sgformer_amd import _lib
    decls = _header_decls()
    assert len(decls) >= 22
    assert set(decls) == set(_lib.SIGNATURES), set(decls) ^ set(_lib.SIGNATURES)
    for name, n in decls.items():
        assert len(_lib.SIGNATURES[name][1]) == n, name
    if not _lib.available():
        pytest.skip("libsgf.so not built (run `make`)")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in decls:
        assert hasattr(lib, name), f"{name} not exported"
    lib.sgf_version.restype = ctypes.c_int
    want = int(re.search(r"#define\s+SGF_VERSION\s+(\d+)", open(os.path.join(ROOT, "include", "sgf.h")).read()).group(1))
    assert lib.sgf_version() == want
    # pure host queries are safe without a GPU
    lib.sgf_attn_stats_len.restype = ctypes.c_int64
    assert lib.sgf_attn_stats_len(2, 64) == 2 * 64 * 64 + 2 * 64 + 2
    lib.sgf_attn_bstats_len.restype = ctypes.c_int64
    assert lib.sgf_attn_bstats_len(1, 256) == 256 * 256 + 256 + 1


def test_argument_errors_are_reported_per_thread():
    """Bad arguments are rejected on the host BEFORE any HIP call (safe without a GPU) with
    SGF_E_INVALID and an errno-style, per-thread message."""
    import threading
    from sgformer_amd import _lib
    if not _lib.available():
        pytest.skip("libsgf.so not built (run `make`)")
    lib = _lib.load()
    assert lib.sgf_spmm(None, None, None, None, 0, 10, None, 0, 10, 256, 0, None) == -1      # null pointers
    main_msg = lib.sgf_last_error()
    assert b"sgf_spmm" in main_msg
    seen = {}

    def other():
        seen["before"] = lib.sgf_last_error()
        seen["rc"] = lib.sgf_axpby(None, 0, 1.0, None, 0, 1.0, 5, 4, 0, None, 0, None)
        seen["after"] = lib.sgf_last_error()

    t = threading.Thread(target=other)
    t.start()
    t.join()
    assert seen["before"] == b"" and seen["rc"] < 0 and b"sgf_axpby" in seen["after"]
    assert lib.sgf_last_error() == main_msg          # the other thread's failure did not overwrite ours


def test_every_entry_point_rejects_null_pointers_on_the_host():
    """All-NULL device pointers with otherwise plausible sizes and a valid dtype: every compute entry
    point must return a negative SGF_E_* code and name itself in sgf_last_error() — validated on the
    host, before any launch (so this runs, and cannot crash, on the GPU-less build host)."""
    from sgformer_amd import _lib
    if not _lib.available():
        pytest.skip("libsgf.so not built (run `make`)")
    lib = _lib.load()
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "sgf.h")).read(), flags=re.S)
    checked = 0
    for name, (_res, argtypes) in sorted(_lib.SIGNATURES.items()):
        if name in ("sgf_version", "sgf_last_error", "sgf_reload_env", "sgf_comm_available", "sgf_comm_destroy") or \
                name.endswith(("_bytes", "_len", "_supported")):
            continue                                  # (queries; sgf_comm_destroy(NULL) is a no-op by contract)
        m = re.search(r"\b" + name + r"\s*\(([^;]*?)\)\s*;", src, flags=re.S)
        pnames = [a.strip().split()[-1].lstrip("*") for a in m.group(1).split(",")]
        assert len(pnames) == len(argtypes), name
        vals = [None if t is ctypes.c_void_p else 0.5 if t in (ctypes.c_float, ctypes.c_double)
                else 0 if "dtype" in pn else 8 for t, pn in zip(argtypes, pnames)]
        rc = getattr(lib, name)(*vals)
        assert rc < 0, (name, rc)
        assert name.encode() in lib.sgf_last_error(), (name, lib.sgf_last_error())
        checked += 1
    assert checked >= 25


def test_missing_library_fails_loudly(monkeypatch):
    from sgformer_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libsgf.so")
    with pytest.raises(_lib.SgfError, match="no CPU"):
        _lib.load()


def test_product_has_no_cpu_path():
    from sgformer_amd import ops
    assert ops.K.name == "hip"
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.attention(torch.zeros(4, 192), None, 1, 64)
    from sgformer_amd.ours import SGFormer
    m = SGFormer(8, 16, 3, trans_dropout=0.0, gnn_dropout=0.0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(5, 8), torch.zeros((2, 0), dtype=torch.int64))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "sgformer_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            assert not re.search(r"^\s*(from|import)\s+(oracle|tests)\b", open(os.path.join(pkg, fn)).read(), re.M), fn


# ------------------------------------------------------------------------------------------------
# host logic with the CPU kernel table
# ------------------------------------------------------------------------------------------------
@pytest.fixture
def cpu_table():
    from sgformer_amd import ops
    prev = ops.set_kernels(CpuKernels())
    yield
    ops.set_kernels(prev)


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))


@pytest.mark.parametrize("n,h,d,shared", [(60, 1, 16, False), (45, 2, 8, False), (45, 3, 8, True)])
def test_attention_function_matches_autograd(cpu_table, n, h, d, shared):
    """The un-normalised two-pass decomposition + hand-derived backward that the
    HIP kernels implement equals autograd through the reference arithmetic."""
    from sgformer_amd import ops
    torch.manual_seed(0)
    q, k = torch.randn(n, h, d, dtype=torch.float64), torch.randn(n, h, d, dtype=torch.float64)
    v = torch.randn(n, 1 if shared else h, d, dtype=torch.float64)
    qd, kd, vd = (t.clone().requires_grad_(True) for t in (q, k, v))
    w = torch.randn(n, d, dtype=torch.float64)
    (O.attention(qd, kd, vd) * w).sum().backward()
    qk = torch.cat([q.reshape(n, -1), k.reshape(n, -1)], 1).float()
    if shared:
        qkv, vx = qk.requires_grad_(True), v.reshape(n, d).float().requires_grad_(True)
    else:
        qkv, vx = torch.cat([qk, v.reshape(n, -1).float()], 1).requires_grad_(True), None
    out = ops.attention(qkv, vx, h, d)
    (out * w.float()).sum().backward()
    hd = h * d
    assert _rel(out, O.attention(q, k, v)) < 1e-6
    assert _rel(qkv.grad[:, :hd], qd.grad.reshape(n, -1)) < 1e-3
    assert _rel(qkv.grad[:, hd:2 * hd], kd.grad.reshape(n, -1)) < 1e-3
    gv = vx.grad if shared else qkv.grad[:, 2 * hd:]
    assert _rel(gv, vd.grad.reshape(n, -1)) < 1e-5


@pytest.mark.parametrize("shared", [False, True])
def test_full_attention_conv_per_head_outputs_are_differentiable(cpu_table, shared):
    """VERDICT r04 missing #5: full_attention_conv returns [N, H, D] for H > 1 UNDER AUTOGRAD (medium/ours.py:14-46) — the
    per-head cotangent reaches Q, K, V as autograd through the reference arithmetic gives it."""
    from sgformer_amd.ours import full_attention_conv
    n, h, d = 70, 3, 8
    g = torch.Generator().manual_seed(9)
    q, k = torch.randn(n, h, d, generator=g, dtype=torch.float64), torch.randn(n, h, d, generator=g, dtype=torch.float64)
    v = torch.randn(n, 1 if shared else h, d, generator=g, dtype=torch.float64)
    cot = torch.randn(n, h, d, generator=g, dtype=torch.float64)
    qd, kd, vd = (t.clone().requires_grad_(True) for t in (q, k, v))
    qn, kn = qd / qd.norm(), kd / kd.norm()
    kvs = torch.einsum("lhm,lhd->hmd", kn, vd.expand(n, h, d))
    ref = (torch.einsum("nhm,hmd->nhd", qn, kvs) + n * vd) / (torch.einsum("nhm,hm->nh", qn, kn.sum(0)).unsqueeze(-1) + n)
    gq, gk, gv = torch.autograd.grad((ref * cot).sum(), (qd, kd, vd))
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    out = full_attention_conv(qf, kf, vf)
    assert out.shape == (n, h, d) and _rel(out, ref) < 1e-6
    dq, dk, dv = torch.autograd.grad((out * cot.float()).sum(), (qf, kf, vf))
    assert dv.shape == v.shape and _rel(dv, gv) < 1e-5
    assert _rel(dq, gq) < 2e-3 and _rel(dk, gk) < 2e-3


CONFIGS = {
    "arxiv": dict(trans_num_layers=1, trans_use_act=False, gnn_num_layers=3, graph_weight=0.5),
    "products": dict(trans_num_layers=1, trans_use_act=False, gnn_num_layers=2, gnn_use_init=True,
                     graph_weight=0.5),
    "heads_cat": dict(trans_num_layers=2, trans_num_heads=2, gnn_num_layers=1, aggregate="cat"),
    "bare": dict(trans_use_weight=False, trans_use_bn=False, trans_use_residual=False, trans_use_act=False,
                 gnn_use_weight=False, gnn_use_bn=False, gnn_use_residual=False, gnn_use_act=False,
                 gnn_num_layers=2),
    "alpha": dict(alpha=0.3, gnn_num_layers=1),
}


@pytest.mark.parametrize("name", list(CONFIGS))
def test_module_host_logic_matches_oracle(cpu_table, name):
    from sgformer_amd.ours import SGFormer
    cfg = CONFIGS[name]
    n, f, d, c = 180, 12, 16, 4
    torch.manual_seed(1)
    x = torch.randn(n, f)
    ei = O.synthetic_graph(n, 5.0, seed=3)
    y = torch.randint(0, c, (n,))
    idx = torch.arange(0, n, 2)
    p = O.init_params(cfg, f, d, c, seed=2)
    m = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, **cfg)
    m.load_state_dict({**m.state_dict(), **p})
    m.train()
    logits = m(x, ei)
    O.nll_loss(logits, y, idx).backward()
    p64 = {k: v.double().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in p.items()}
    stats = {}
    ref = O.sgformer_forward(p64, x.double(), ei, cfg, training=True, bn_stats=stats)
    O.nll_loss(ref, y, idx).backward()
    assert float((logits.detach().double() - ref.detach()).abs().max()) < 2e-5
    gmax = max(float(v.grad.norm()) for v in p64.values() if v.grad is not None)
    for k, prm in m.named_parameters():
        if p64[k].grad is None:
            assert prm.grad is None
            continue
        err = float((prm.grad.double() - p64[k].grad).norm())
        assert err <= 2e-3 * float(p64[k].grad.norm()) + 1e-5 * gmax, k
    for key, (mu, vu) in stats.items():
        sd = m.state_dict()
        assert torch.allclose(sd[key + ".running_mean"].double(), 0.9 * p[key + ".running_mean"].double() + 0.1 * mu, atol=1e-5)
        assert torch.allclose(sd[key + ".running_var"].double(), 0.9 * p[key + ".running_var"].double() + 0.1 * vu, atol=1e-5)
        assert int(sd[key + ".num_batches_tracked"]) == 1
    m.eval()
    with torch.no_grad():
        le = m(x, ei)
    pe = {k: v.double() for k, v in m.state_dict().items()}
    assert float((le.double() - O.sgformer_forward(pe, x.double(), ei, cfg, training=False)).abs().max()) < 2e-5


def test_dropout_and_attention_maps(cpu_table):
    from sgformer_amd.ours import SGFormer
    torch.manual_seed(0)
    m = SGFormer(10, 16, 3, trans_num_layers=2, trans_dropout=0.5, gnn_dropout=0.5, gnn_num_layers=2)
    x, ei = torch.randn(50, 10), O.synthetic_graph(50, 4.0, seed=1)
    m.train()
    a, b = m(x, ei), m(x, ei)
    assert not torch.equal(a, b)                     # dropout active in training mode
    m.eval()
    assert torch.equal(m(x, ei), m(x, ei))           # and off in eval mode
    att = m.get_attentions(x)
    assert att.shape == (2, 50, 50)
    with pytest.raises(TypeError):
        SGFormer(10, 16, 3, trans_dropout=None)(x, ei)   # large/parse.py:95 hazard, surfaced


def test_graph_cache_rules(cpu_table):
    from sgformer_amd import ops
    ei = O.synthetic_graph(40, 4.0, seed=1)
    g1 = ops.graph_cache.get(ei, 40)
    assert ops.graph_cache.get(ei, 40) is g1                     # same tensor -> cached
    assert ops.graph_cache.get(ei[:, :], 40) is g1               # a view of the same memory too
    ei2 = ei.clone()
    assert ops.graph_cache.get(ei2, 40) is not g1                # new tensor (mini-batch trainers)
    ei2[0, 0] = (ei2[0, 0] + 1) % 40                             # in-place edit bumps _version
    g3 = ops.graph_cache.get(ei2, 40)
    assert g3 is not ops.graph_cache._d.get(None) and g3.nnz == ei2.shape[1]
    for s in range(10):                                          # bounded (LRU)
        ops.graph_cache.get(O.synthetic_graph(30, 3.0, seed=s), 30)
    assert len(ops.graph_cache._d) <= ops.graph_cache.capacity
    with pytest.raises(IndexError):
        ops.CSRGraph(torch.tensor([[0, 9], [1, 2]]), 4)
    with pytest.raises(ValueError):
        ops.CSRGraph(torch.zeros((2, 3), dtype=torch.int32), 4)


# ------------------------------------------------------------------------------------------------
# medium variant (medium/ours.py + models.GCN) on the CPU kernel table
# ------------------------------------------------------------------------------------------------
class _Data:
    def __init__(self, x, ei):
        self.graph = {"node_feat": x, "edge_index": ei}


@pytest.mark.parametrize("cfg,gcn_layers", [(dict(num_layers=1, alpha=0.5, graph_weight=0.8), 4),
                                            (dict(num_layers=2, num_heads=2, use_weight=False, aggregate="cat"), 2)])
def test_medium_module_matches_oracle(cpu_table, cfg, gcn_layers):
    from sgformer_amd import ours_medium as M
    n, f, d, c = 170, 21, 16, 5
    torch.manual_seed(2)
    gnn = M.GCN(f, d, d, num_layers=gcn_layers, dropout=0.0)
    m = M.SGFormer(f, d, c, dropout=0.0, gnn=gnn, **cfg)
    with torch.no_grad():
        for k, v in m.state_dict().items():
            if k.endswith("bias"):
                v.normal_(0, 0.1)
    x = torch.randn(n, f)
    ei = O.synthetic_graph(n, 5.0, seed=4)[:, :-n]
    y = torch.randint(0, c, (n,))
    idx = torch.arange(0, n, 2)
    p = {k: v.detach().clone() for k, v in m.state_dict().items()}
    assert len(m.params1) + len(m.params2) == len(list(m.parameters()))
    m.train()
    logits = m(_Data(x, ei))
    loss = O.nll_loss(logits, y, idx)
    loss.backward()
    p64 = {k: v.double().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in p.items()}
    ref = O.medium_forward(p64, x.double(), ei, cfg, training=True)
    O.nll_loss(ref, y, idx).backward()
    assert float((logits.detach().double() - ref.detach()).abs().max()) <= 2e-5
    for k, prm in m.named_parameters():
        g = p64[k].grad
        assert prm.grad is not None and g is not None, k
        if float(g.norm()) > 1e-9:
            assert _rel(prm.grad, g) <= 2e-4, k
    m.eval()
    pe = {k: v.detach().double() for k, v in m.state_dict().items()}
    with torch.no_grad():
        assert float((m(_Data(x, ei)).double() - O.medium_forward(pe, x.double(), ei, cfg, training=False)).abs().max()) <= 2e-5


def test_medium_state_dict_matches_reference_keys():
    """Same parameter names / shapes as medium/ours.py + models.GCN over PyG 1.7.2 GCNConv
    (`gnn.convs.i.weight [in, out]`, `gnn.convs.i.bias`, `gnn.bns.i.*`): checkpoints interchange."""
    from oracle import ref_shim
    if not ref_shim.reference_available():
        pytest.skip("/root/reference not mounted")
    from sgformer_amd import ours_medium as M
    ref = ref_shim.load_reference("medium")
    torch.manual_seed(0)
    a = ref.SGFormer(30, 16, 4, num_layers=1, gnn=ref.models.GCN(30, 16, 16, num_layers=4))
    torch.manual_seed(0)
    b = M.SGFormer(30, 16, 4, num_layers=1, gnn=M.GCN(30, 16, 16, num_layers=4))
    sa, sb = a.state_dict(), b.state_dict()
    assert list(sa.keys()) == list(sb.keys())
    for k in sa:
        assert sa[k].shape == sb[k].shape, k
        assert torch.equal(sa[k], sb[k]), k          # same creation order -> same seeded init
    b.load_state_dict(sa)


# ------------------------------------------------------------------------------------------------
# trainer-side helpers (rows N1 / N4) and the launcher's patches, host logic
# ------------------------------------------------------------------------------------------------
def test_batching_subgraph_semantics(cpu_table):
    """sgformer_amd.batching.subgraph == torch_geometric.utils.subgraph (PyG 1.7.2 semantics as
    restated in tests/standins): index / bool-mask subsets, relabelling, edge attributes."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "_standin_tg_utils", os.path.join(ROOT, "tests", "standins", "torch_geometric", "utils", "__init__.py"))
    tgu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tgu)
    pyg_subgraph = tgu.subgraph
    from sgformer_amd import batching
    g = torch.Generator().manual_seed(0)
    n = 300
    ei = torch.randint(0, n, (2, 4000), generator=g)
    w = torch.randn(4000, generator=g)
    sub = torch.randperm(n, generator=g)[:90]
    for relabel in (False, True):
        a, aw = batching.subgraph(sub, ei, edge_attr=w, relabel_nodes=relabel, num_nodes=n)
        b, bw = pyg_subgraph(sub, ei, edge_attr=w, relabel_nodes=relabel, num_nodes=n)
        assert torch.equal(a, b) and torch.equal(aw, bw)
    mask = torch.zeros(n, dtype=torch.bool)
    mask[sub] = True
    a, none = batching.subgraph(mask, ei, num_nodes=n)
    assert none is None and torch.equal(a, pyg_subgraph(mask, ei, num_nodes=n)[0])


def test_fused_loss_matches_trainer_lines(cpu_table):
    """sgformer_amd.loss.log_softmax_nll == large/main.py:139-141 (value and gradient)."""
    from sgformer_amd.loss import log_softmax_nll
    g = torch.Generator().manual_seed(1)
    out = torch.randn(500, 11, generator=g)
    label = torch.randint(0, 11, (500, 1), generator=g)
    idx = torch.randperm(500, generator=g)[:200]
    a = out.clone().requires_grad_(True)
    ref = torch.nn.NLLLoss()(torch.log_softmax(a, dim=1)[idx], label.squeeze(1)[idx])
    ref.backward()
    b = out.clone().requires_grad_(True)
    loss = log_softmax_nll(b, label, idx)
    loss.backward()
    assert abs(float(loss) - float(ref)) <= 1e-6
    assert _rel(b.grad, a.grad) <= 1e-6


def test_launcher_variants_and_patches(monkeypatch, tmp_path):
    """launch.install registers the right drop-in per variant; the medium / main-batch patches swap
    `models.GCN` and `torch_geometric.utils.subgraph` before the trainer imports them."""
    import sys
    import types
    from sgformer_amd import launch
    monkeypatch.setitem(sys.modules, "ours", None)
    for variant, modname in (("large", "sgformer_amd.ours"), ("100M", "sgformer_amd.ours_100m"),
                             ("medium", "sgformer_amd.ours_medium")):
        mod = launch.install(variant)
        assert mod.__name__ == modname and sys.modules["ours"] is mod
        assert hasattr(mod, "SGFormer") and hasattr(mod, "TransConv") and hasattr(mod, "full_attention_conv")
    with pytest.raises(SystemExit):
        launch.install("nope")
    fake_models = types.ModuleType("models")
    fake_models.GCN = object
    monkeypatch.setitem(sys.modules, "models", fake_models)
    launch.patch_medium_gcn()
    from sgformer_amd import ours_medium
    assert fake_models.GCN is ours_medium.GCN
    tg = types.ModuleType("torch_geometric")
    tgu = types.ModuleType("torch_geometric.utils")
    tgu.subgraph = object
    tg.utils = tgu
    monkeypatch.setitem(sys.modules, "torch_geometric", tg)
    monkeypatch.setitem(sys.modules, "torch_geometric.utils", tgu)
    launch.patch_subgraph()
    from sgformer_amd import batching
    assert tgu.subgraph is batching.subgraph
    # option parsing
    argv = ["--sgf-dtype", "bf16", "x.py", "--sgf-variant=large", "--foo"]
    assert launch._pop_option(argv, "--sgf-dtype") == "bf16" and launch._pop_option(argv, "--sgf-variant") == "large"
    assert argv == ["x.py", "--foo"]


# ------------------------------------------------------------------------------------------------
# DIFFormer drop-in (medium/difformer.py, row N4) on the CPU kernel table
# ------------------------------------------------------------------------------------------------
DIFF_CFGS = [dict(num_layers=2, num_heads=1),
             dict(num_layers=1, num_heads=2, use_weight=False, graph_weight=0.3, use_source=True),
             dict(num_layers=2, num_heads=2, use_graph=False, use_bn=False, use_residual=False),
             dict(num_layers=1, num_heads=1, alpha=0.2, graph_weight=0.7)]


@pytest.mark.parametrize("cfg", DIFF_CFGS)
def test_difformer_module_matches_oracle(cpu_table, cfg):
    from sgformer_amd import difformer as M
    n, f, d, c = 160, 18, 16, 5
    torch.manual_seed(3)
    m = M.DIFFormer(f, d, c, dropout=0.0, **cfg)
    with torch.no_grad():
        for k, v in m.state_dict().items():
            if k.endswith("bias"):
                v.normal_(0, 0.1)
    x = torch.randn(n, f)
    ei = O.synthetic_graph(n, 5.0, seed=4)
    y = torch.randint(0, c, (n,))
    idx = torch.arange(0, n, 2)
    p = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m.train()
    logits = m(_Data(x, ei))
    O.nll_loss(logits, y, idx).backward()
    p64 = {k: v.double().requires_grad_(True) for k, v in p.items()}
    ref = O.difformer_forward(p64, x.double(), ei, cfg)
    O.nll_loss(ref, y, idx).backward()
    assert float((logits.detach().double() - ref.detach()).abs().max()) <= 2e-5
    for k, prm in m.named_parameters():
        g = p64[k].grad
        if g is None:                       # unused parameters (LayerNorms with use_bn=False)
            assert prm.grad is None, k
            continue
        assert prm.grad is not None, k
        if float(g.norm()) > 1e-9:
            assert _rel(prm.grad, g) <= 3e-4, k
    # the dense (materialised) path — what the sigmoid kernel and the attention maps use — agrees too
    with torch.no_grad():
        att = m.get_attentions(x)
    assert att.shape[:3] == (cfg.get("num_layers", 2), n, n)


def test_difformer_state_dict_matches_reference_keys():
    from oracle import ref_shim
    if not ref_shim.reference_available():
        pytest.skip("/root/reference not mounted")
    from sgformer_amd import difformer as M
    ref = ref_shim.load_reference("difformer")
    torch.manual_seed(0)
    a = ref.DIFFormer(30, 16, 4, num_layers=2, num_heads=2)
    torch.manual_seed(0)
    b = M.DIFFormer(30, 16, 4, num_layers=2, num_heads=2)
    sa, sb = a.state_dict(), b.state_dict()
    assert list(sa.keys()) == list(sb.keys())
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


def test_kernel_probe_drives_the_kernel_table(cpu_table):
    """scripts/kernel_probe.py (the per-kernel GB/s loop for the GPU box) calls every table entry with
    the right signature — checked here through the CPU table so it cannot rot between GPU sessions."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("kernel_probe", os.path.join(ROOT, "scripts", "kernel_probe.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for dtype in (torch.float32, torch.bfloat16):
        rows = mod.run(200, 64, dtype, "cpu", reps=1)
        assert len(rows) >= 15 and all(r["ms"] >= 0 and r["GBps"] >= 0 for r in rows)
        unit = 200 * 64 * (4 if dtype == torch.float32 else 2) / 1e9
        assert abs(rows[0]["algorithmic_GB"] - round(2 * unit, 3)) < 1e-9
    assert [r["kernel"] for r in mod.run(200, 64, torch.float32, "cpu", reps=1, only=("apply",))] == \
        ["sgf_attn_h_bwd_apply", "sgf_bn_apply (residual, relu)", "sgf_bn_bwd_apply"]


# ------------------------------------------------------------------------------------------------
# the constructor-flag space on the CPU table (the GPU suite runs the same property on libsgf)
# ------------------------------------------------------------------------------------------------
try:
    from hypothesis import HealthCheck, assume, given, settings, strategies as st
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

    @settings(max_examples=25, deadline=None, derandomize=True,
              suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
    @given(cfg=_flags, n=st.integers(3, 150), d=st.sampled_from([8, 12, 16]), directed=st.booleans(),
           seed=st.integers(0, 10 ** 6))
    def test_flag_space_host_logic(cpu_table, cfg, n, d, directed, seed):
        """Random points of the constructor-flag space (which fast paths the module picks — attention
        from the input vs materialised Q/K/V, fan-out hub, fused BN/LN glue, cat vs add — depends on
        them) x ragged sizes x directed graphs: the module's wiring equals the fp64 oracle."""
        from sgformer_amd.ours import SGFormer
        # use_graph=False with aggregate='cat' crashes in the reference too (large/ours.py:257-259 vs :273-275)
        assume(cfg["use_graph"] or cfg["aggregate"] == "add")
        f, c = 7, 3
        torch.manual_seed(seed)
        x = torch.randn(n, f)
        ei = O.synthetic_graph(n, 4.0, seed=seed % 1000, directed=directed)
        y = torch.randint(0, c, (n,))
        idx = torch.randperm(n)[: max(n // 2, 1)]
        p = O.init_params(cfg, f, d, c, seed=seed % 97)
        m = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, **cfg)
        m.load_state_dict({**m.state_dict(), **p})
        m.train()
        logits = m(x, ei)
        O.nll_loss(logits, y, idx).backward()
        p64 = {k: v.double().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in p.items()}
        ref = O.sgformer_forward(p64, x.double(), ei, cfg, training=True)
        O.nll_loss(ref, y, idx).backward()
        assert float((logits.detach().double() - ref.detach()).abs().max()) <= 1e-4
        gmax = max([float(v.grad.norm()) for v in p64.values() if v.grad is not None] + [1e-30])
        for k, prm in m.named_parameters():
            g = p64[k].grad
            if g is None:
                continue
            assert prm.grad is not None, k
            err = float((prm.grad.double() - g).norm())
            assert err <= 2e-3 * (float(g.norm()) + 1e-3 * gmax), (k, err, float(g.norm()))


@pytest.mark.parametrize("variant", ["large", "100m"])
def test_bf16_streaming_paths_on_the_cpu_table(cpu_table, monkeypatch, variant):
    """bf16 storage routes the square Linear layers (+ BatchNorm sums), the two-operand [A x | x0] Linear and both
    input stems through the streaming row kernels (ops.linear_bn_stats / ops.stem_pair).  With the CPU kernel table
    standing in for libsgf, the model must take those paths and agree with the same model on the unfused route
    (library GEMM + sgf_colstats) to bf16 rounding — forward and every parameter gradient."""
    from sgformer_amd import ops, synth
    from sgformer_amd import ours as large, ours_100m
    monkeypatch.setattr(ops, "_require_cuda", lambda *a, **k: None)
    cls = large.SGFormer if variant == "large" else ours_100m.SGFormer
    cfg = dict(synth.RECIPES["ogbn-products"])
    n, f, d, c = 700, 100, 64, 7
    torch.manual_seed(3)
    x = torch.randn(n, f)
    ei = synth.synthetic_graph(n, 6.0, seed=2)
    w = torch.randn(n, c)
    calls = {"stem": 0, "cat": 0, "stats": 0}
    for name, key in (("stem_pair", "stem"), ("gcn_epilogue_cat", "cat"), ("gcn_epilogue_stats", "stats")):
        orig = getattr(CpuKernels, name)
        monkeypatch.setattr(CpuKernels, name, staticmethod(
            lambda *a, _o=orig, _k=key, **k: (calls.__setitem__(_k, calls[_k] + 1), _o(*a, **k))[1]))

    def run(fused):
        torch.manual_seed(11)
        m = cls(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, compute_dtype=torch.bfloat16, **cfg).train()
        if not fused:
            monkeypatch.setattr(CpuKernels, "gcn_epilogue_supported", staticmethod(lambda *a: False))
            monkeypatch.setattr(CpuKernels, "stem_pair_supported", staticmethod(lambda *a: False))
        out = m(x, ei)
        (out.float() * w).sum().backward()
        return out.detach().float(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}

    out_f, g_f = run(True)
    assert calls["stem"] >= 2 and calls["cat"] >= 3, calls          # sample + full launch; three conv layers
    out_u, g_u = run(False)
    assert _rel(out_f, out_u) <= 2e-2
    assert set(g_f) == set(g_u)
    for k in g_u:
        if k.startswith("graph_conv") and k.endswith("bias") and ("fcs.0" in k or ".W." in k):
            continue      # a bias in front of a BatchNorm: its exact gradient is zero, what is there is rounding noise
        assert _rel(g_f[k], g_u[k]) <= 0.2, k      # bf16 noise through BatchNorm at N = 700; gross agreement only


def test_grad_tap_folds_the_second_gradient(cpu_table, monkeypatch):
    """ops.grad_tap + attention_from_input(tap=...): the residual's gradient of the layer input reaches the attention's
    backward (folded into its last pass) or passes through autograd — the total is the same in either execution
    order of the two backward nodes, and over a retained graph twice."""
    from sgformer_amd import ops
    monkeypatch.setattr(ops, "_require_cuda", lambda *a, **k: None)
    torch.manual_seed(0)
    n, d = 300, 64
    for dtype in (torch.bfloat16, torch.float32):
        x = (torch.rand(n, d) + 0.1).to(dtype).requires_grad_(True)
        ws = [torch.randn(d, d) / 8 for _ in range(3)]
        bs = [torch.randn(d) * 0.1 for _ in range(3)]
        wgt = torch.randn(n, d)

        def total(use_tap, tap_first=True):
            x.grad = None
            holder = {} if use_tap else None
            if use_tap and not tap_first:                    # create the tap's node BEFORE the attention's: it runs last
                r = ops.grad_tap(x, holder)
                h = ops.attention_from_input(x, ws[0], bs[0], ws[1], bs[1], ws[2], bs[2], None, 4.0, tap=holder)
            else:
                h = ops.attention_from_input(x, ws[0], bs[0], ws[1], bs[1], ws[2], bs[2], None, 4.0, tap=holder)
                r = ops.grad_tap(x, holder) if use_tap else x
            y = ((h.float() + 0.5 * r.float()) * wgt).sum()
            y.backward(retain_graph=True)
            g1 = x.grad.clone()
            x.grad = None
            y.backward()
            return g1, x.grad.clone()

        ref, ref2 = total(False)
        for first in (True, False):
            a, b = total(True, first)
            tol = 1e-6 if dtype == torch.float32 else 2e-2
            assert _rel(a.float(), ref.float()) <= tol and _rel(b.float(), ref2.float()) <= tol, (dtype, first)


def test_graft_entry_build_check_matches_the_header():
    """__graft_entry__.build() (the driver's "does it build" hook) compares the library's version with include/sgf.h —
    not with a constant that a version bump would leave behind."""
    src = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert "SGF_VERSION" in src and not re.search(r"sgf_version\(\)\s*==\s*\d", src)


def test_gather_nll_equals_aten_nll_loss():
    """sgformer_amd.loss.gather_nll (what the launcher installs behind F.nll_loss for the unchanged trainers'
    `criterion(out[train_idx], y[train_idx])`, large/main.py:139-141) == torch's nll_loss: value and gradient, with and
    without ignored targets (pokec's labels contain -1 -> rows a trainer may mask with ignore_index)."""
    import torch.nn.functional as F
    from sgformer_amd.loss import gather_nll
    torch.manual_seed(0)
    for ignore in (-100, -1):
        logp = torch.log_softmax(torch.randn(500, 7, dtype=torch.float64), 1).requires_grad_(True)
        tgt = torch.randint(0, 7, (500,))
        tgt[::9] = ignore
        ref = F.nll_loss(logp, tgt, ignore_index=ignore)
        g_ref, = torch.autograd.grad(ref, logp)
        got = gather_nll(logp, tgt, ignore)
        g_got, = torch.autograd.grad(got, logp)
        assert abs(float(got) - float(ref)) <= 1e-12 and float((g_got - g_ref).abs().max()) <= 1e-15


def test_launcher_nll_patch_installs_and_restores():
    import torch.nn.functional as F
    from sgformer_amd import launch
    before = F.nll_loss
    launch.patch_nll_loss()
    try:
        assert F.nll_loss is not before and F.nll_loss._sgf_orig is before
        x = torch.log_softmax(torch.randn(20, 4), 1)
        t = torch.randint(0, 4, (20,))
        assert torch.allclose(torch.nn.NLLLoss()(x, t), before(x, t))      # CPU input: the original is reached
        launch.patch_nll_loss()                                            # idempotent
        assert F.nll_loss._sgf_orig is before
    finally:
        launch.unpatch_nll_loss()
    assert F.nll_loss is before


def test_launcher_runs_the_trainers_three_loss_lines_in_one_pass():
    """large/main.py:139-141 AS WRITTEN under the launcher's patches: F.log_softmax returns a lazy tensor, `out[train_idx]`
    lazy rows, nn.NLLLoss() on them runs sgf_nll_fwd / sgf_nll_bwd (here: the CPU kernel table) — value and gradient equal
    ATen's three lines; every other use of the lazy objects (argmax, arithmetic, slices, boolean masks, a second index,
    class weights, reduction='sum', ignore_index rows) gives exactly what the un-patched functions give."""
    import torch.nn as nn
    import torch.nn.functional as F
    from sgformer_amd import launch, ops
    from sgformer_amd.loss import LazyLogSoftmax
    from tests.cpu_kernels import CpuKernels
    prev = ops.set_kernels(CpuKernels())
    ls0, nll0 = F.log_softmax, F.nll_loss
    g = torch.Generator().manual_seed(0)
    n, c = 500, 7
    logits = torch.randn(n, c, generator=g, requires_grad=True)
    label = torch.randint(0, c, (n, 1), generator=g)
    idx = torch.randperm(n, generator=g)[:200]
    ref = nll0(ls0(logits, dim=1)[idx], label.squeeze(1)[idx])
    g_ref, = torch.autograd.grad(ref, logits)
    launch.patch_nll_loss()
    try:
        calls, real_ls = [], []
        real = ops.K.nll_fwd
        ops.K.nll_fwd = staticmethod(lambda *a: (calls.append(1), real(*a))[1])
        import sgformer_amd.loss as L
        orig_value = L.LazyLogSoftmax._sgf_value
        L.LazyLogSoftmax._sgf_value = lambda self: (real_ls.append(1), orig_value(self))[1]
        criterion = nn.NLLLoss()
        out = F.log_softmax(logits, dim=1)                       # the trainer's three lines
        loss = criterion(out[idx], label.squeeze(1)[idx])
        assert isinstance(out, LazyLogSoftmax) and out.shape == (n, c) and out.dtype == logits.dtype
        assert out.device == logits.device and out.dim() == 2 and out.size(0) == n
        assert calls == [1] and not isinstance(loss, LazyLogSoftmax)
        assert real_ls == [], "the one-pass path must not compute the full log-softmax"
        L.LazyLogSoftmax._sgf_value = orig_value
        g_got, = torch.autograd.grad(loss, logits)
        assert abs(float(loss) - float(ref)) <= 1e-6 and float((g_got - g_ref).abs().max()) <= 1e-7
        # rows whose target is ignore_index leave numerator and divisor
        t2 = label.squeeze(1)[idx].clone()
        t2[::3] = -100
        got = criterion(F.log_softmax(logits, dim=1)[idx], t2)
        assert abs(float(got) - float(nll0(ls0(logits, dim=1)[idx], t2))) <= 1e-6 and len(calls) == 2
        # everything else falls back to the real values
        out = F.log_softmax(logits, dim=1)
        full = ls0(logits, dim=1)
        assert torch.equal(out.argmax(1), full.argmax(1)) and len(calls) == 2
        assert torch.allclose((F.log_softmax(logits, dim=1) * 2.0 + 1.0).detach(), (full * 2.0 + 1.0).detach())
        assert torch.equal(F.log_softmax(logits, dim=1)[3:9].detach(), full[3:9].detach())
        mask = torch.zeros(n, dtype=torch.bool)
        mask[idx] = True
        assert torch.equal(F.log_softmax(logits, dim=1)[mask].detach(), full[mask].detach())
        # the mini-batch trainer's form (large/main-batch.py:146): rows picked by a BOOLEAN mask -> one pass as well
        before = len(calls)
        got_m = criterion(F.log_softmax(logits, dim=1)[mask], label.squeeze(1)[mask])
        gm, = torch.autograd.grad(got_m, logits)
        ref_m = nll0(ls0(logits, dim=1)[mask], label.squeeze(1)[mask])
        grm, = torch.autograd.grad(ref_m, logits)
        assert len(calls) == before + 1 and abs(float(got_m) - float(ref_m)) <= 1e-6 and float((gm - grm).abs().max()) <= 1e-7
        none = torch.zeros(n, dtype=torch.bool)                                              # an empty mask: ATen's nan, not a crash
        assert torch.isnan(criterion(F.log_softmax(logits, dim=1)[none], label.squeeze(1)[none]))
        calls[:] = calls[:2]
        rows = F.log_softmax(logits, dim=1)[idx]
        assert torch.equal(rows[:5].detach(), full[idx][:5].detach())                       # a second index: real rows
        w = torch.rand(c, generator=g)
        assert torch.allclose(F.nll_loss(F.log_softmax(logits, dim=1)[idx], label.squeeze(1)[idx], weight=w),
                              nll0(full[idx], label.squeeze(1)[idx], weight=w))
        assert torch.allclose(F.nll_loss(F.log_softmax(logits, dim=1)[idx], label.squeeze(1)[idx], reduction="sum"),
                              nll0(full[idx], label.squeeze(1)[idx], reduction="sum"))
        assert len(calls) == 2
        lg, = torch.autograd.grad(F.log_softmax(logits, dim=1).sum(), logits)                # autograd through the fallback
        lr, = torch.autograd.grad(full.sum(), logits)
        assert torch.allclose(lg, lr)
        assert not isinstance(F.log_softmax(logits, dim=0), LazyLogSoftmax)                  # other dims: ATen
        # a row index with REPEATED rows must accumulate per occurrence — ATen's path, not the storing kernels
        dup = torch.tensor([1, 3, 3, 7, 3, 1])
        td = label.squeeze(1)[dup]
        ncall = len(calls)
        rows = F.log_softmax(logits, dim=1)[dup]
        got_d = criterion(rows, td)
        gd, = torch.autograd.grad(got_d, logits)
        ref_d = nll0(ls0(logits, dim=1)[dup], td)
        grd, = torch.autograd.grad(ref_d, logits)
        assert len(calls) == ncall and abs(float(got_d) - float(ref_d)) <= 1e-6 and float((gd - grd).abs().max()) <= 1e-7
        neg = torch.tensor([-1, 2, 5])                                                       # wrapping (negative) entries too
        assert torch.allclose(criterion(F.log_softmax(logits, dim=1)[neg], label.squeeze(1)[neg]),
                              nll0(ls0(logits, dim=1)[neg], label.squeeze(1)[neg])) and len(calls) == ncall
        assert F.log_softmax(logits, dim=1).requires_grad and not F.log_softmax(logits.detach(), dim=1).requires_grad
        assert not isinstance(F.log_softmax(torch.randn(4, 100), dim=1), LazyLogSoftmax)     # more than 64 classes: ATen
    finally:
        ops.K.nll_fwd = real
        launch.unpatch_nll_loss()
        ops.set_kernels(prev)
    assert F.log_softmax is ls0 and F.nll_loss is nll0


@pytest.mark.parametrize("sum_v", [False, True])
@pytest.mark.parametrize("d,d_in", [(8, 8), (16, 12)])
def test_packed_attention_algebra_matches_the_term_by_term_form(cpu_table, sum_v, d, d_in):
    """ops._attn_h_small_packed (augmented operands, 12 launches) == ops._attn_h_small (include/sgf.h's formulas term by term),
    values and all eight gradients, in fp64 to 1e-12."""
    import torch
    from sgformer_amd import ops
    g = torch.Generator().manual_seed(d + d_in + int(sum_v))
    h = torch.randn(50, d_in, generator=g, dtype=torch.float64)
    G, s = (h.t() @ h), h.sum(0)
    par = [torch.randn(d, d_in, generator=g, dtype=torch.float64) * 0.3, torch.randn(d, generator=g, dtype=torch.float64) * 0.1,
           torch.randn(d, d_in, generator=g, dtype=torch.float64) * 0.3, torch.randn(d, generator=g, dtype=torch.float64) * 0.1,
           torch.randn(d, d_in, generator=g, dtype=torch.float64) * 0.3, torch.randn(d, generator=g, dtype=torch.float64) * 0.1]
    if d != d_in:
        pytest.skip("M = c wq^T s0 + N wv^T needs d == d_in") if not sum_v else None
    leaves = [t.clone().requires_grad_(True) for t in (G, s, *par)]
    from tests import attn_algebra as A
    ref = A.attn_h_small(leaves[0], leaves[1], 50.0, 70.0, *leaves[2:], sum_v=sum_v)
    cot = [torch.randn(t.shape, generator=g, dtype=torch.float64) for t in ref]
    gref = torch.autograd.grad(ref, leaves, cot)
    old = ops._F32
    ops._F32 = torch.float64
    try:
        packed = [t.requires_grad_(True) for t in ops._attn_h_pack(G, s, 50.0, *par)]
    finally:
        ops._F32 = old
    out = ops._attn_h_small_packed(*packed, 70.0, sum_v=sum_v)
    for a, b in zip(out, ref):
        assert a.shape == b.shape and float((a - b).abs().max()) <= 1e-12 * max(1.0, float(b.abs().max()))
    gGt, gWqk, gWv = torch.autograd.grad(out, packed, cot)
    got = [gGt[:d_in, :d_in], gGt[:d_in, d_in] + gGt[d_in, :d_in], gWqk[:d, :d_in], gWqk[:d, d_in], gWqk[d:, :d_in],
           gWqk[d:, d_in], gWv[:, :d_in], gWv[:, d_in]]
    # dG: the term-by-term form's gradient of the SYMMETRIC G is not symmetric itself; both feed D = dG + dG^T
    assert float(((got[0] + got[0].t()) - (gref[0] + gref[0].t())).abs().max()) <= 1e-11 * max(1.0, float(gref[0].abs().max()))
    for a, b in zip(got[1:], gref[1:]):
        assert float((a - b).abs().max()) <= 1e-11 * max(1.0, float(b.abs().max()))


def test_long_row_bound_uses_what_the_caller_knows():
    """ADVICE r02 / VERDICT r03: the host-side bound on long-row segments sent every small graph with more than 1024 entries
    down the split path.  A caller-supplied bound on the longest row (a sampled batch: its fan-out; any graph: its node count)
    at most LONG_ROW switches the split path off without a device read."""
    from sgformer_amd import ops
    rowptr = torch.arange(0, 5001, 5)                                       # 1000 rows of 5 entries: nnz = 5000 > LONG_ROW
    assert ops.long_row_segments(rowptr, 5000) > 0                           # nothing known: the bound
    assert ops.long_row_segments(rowptr, 5000, max_row_len=15) == 0          # a sampled batch with fan-outs <= 15
    assert ops.long_row_segments(rowptr, 5000, max_row_len=1000) == 0        # a graph of 1000 nodes
    assert ops.long_row_segments(rowptr, 5000, max_row_len=ops.LONG_ROW + 1) > 0


@pytest.mark.parametrize("d", [5, 8, 16])
def test_packed_attention_algebra_hand_written_backward(cpu_table, d):
    """ops._attn_h_packed_fwd / _bwd (what the training step runs: 12 + 20 launches, no autograd graph) == autograd through
    ops._attn_h_small_packed, in fp64 to 1e-12: the four outputs and the gradients of the three packed operands (Gt's through
    the symmetrised form D = dG + dG^T and ds, which is all the caller uses)."""
    from sgformer_amd import ops
    from tests import attn_algebra as A
    g = torch.Generator().manual_seed(d)
    h = torch.randn(40, d, generator=g, dtype=torch.float64)
    par = [torch.randn(d, d, generator=g, dtype=torch.float64) * 0.3 if i % 2 == 0 else
           torch.randn(d, generator=g, dtype=torch.float64) * 0.1 for i in range(6)]
    old = ops._F32
    ops._F32 = torch.float64
    try:
        packed = [t.requires_grad_(True) for t in ops._attn_h_pack(h.t() @ h, h.sum(0), 40.0, *par)]
        ref = ops._attn_h_small_packed(*packed, 55.0, sum_v=False)
        cot = [torch.randn(t.shape, generator=g, dtype=torch.float64) for t in ref]
        gref = torch.autograd.grad(ref, packed, cot)
        with torch.no_grad():
            pk = [t.detach() for t in packed]
            out, saved = A.attn_h_packed_fwd(*pk, 55.0)
            gout = torch.zeros(d + 1, d + 1, dtype=torch.float64)
            gout[:d, :d], gout[d, :d], gout[:d, d], gout[d, d:] = cot[0], cot[1], cot[2], cot[3]
            got = A.attn_h_packed_bwd(*pk, 55.0, saved, gout)
    finally:
        ops._F32 = old
    for a, b in zip((out[:d, :d], out[d, :d], out[:d, d], out[d:, d]), ref):
        assert float((a - b).abs().max()) <= 1e-12 * max(1.0, float(b.abs().max()))
    sym = lambda m: m + m.t()      # noqa: E731
    assert float((sym(got[0]) - sym(gref[0])).abs().max()) <= 1e-12 * max(1.0, float(gref[0].abs().max()))
    for a, b in zip(got[1:], gref[1:]):
        assert float((a - b).abs().max()) <= 1e-12 * max(1.0, float(b.abs().max()))


def test_patch_adam_keeps_generator_valued_param_groups(monkeypatch):
    """Adam([{'params': model.parameters()}]) — a GENERATOR inside a group dict — under launch.patch_adam must
    train every parameter (the patch used to exhaust the generator while looking at it: a group of size 0, silently)."""
    import torch.nn as nn
    from sgformer_amd import launch
    monkeypatch.delenv("SGF_FUSED_ADAM", raising=False)
    orig = torch.optim.Adam.__init__
    was = getattr(torch.optim.Adam, "_sgf_patched", False)
    if was:
        pytest.skip("Adam already patched in this process")
    try:
        launch.patch_adam()
        m = nn.Linear(3, 2)
        opt = torch.optim.Adam([{"params": m.parameters(), "weight_decay": 1e-5}], lr=0.1)
        assert len(opt.param_groups) == 1 and len(opt.param_groups[0]["params"]) == 2
        opt2 = torch.optim.Adam([{"params": m.weight}, {"params": (p for p in [m.bias])}], lr=0.1)
        assert [len(g["params"]) for g in opt2.param_groups] == [1, 1]
        w0 = m.weight.detach().clone()
        m(torch.ones(1, 3)).sum().backward()
        opt.step()
        assert not torch.equal(w0, m.weight.detach())
    finally:
        torch.optim.Adam.__init__ = orig
        torch.optim.Adam._sgf_patched = False


@pytest.mark.parametrize("name,dtype", [("arxiv", None), ("products", torch.bfloat16)])
def test_feature_width_that_is_not_a_multiple_of_4_is_padded_once_at_the_entry(cpu_table, name, dtype):
    """r05 (pokec: f = 65, large/run.sh:22-26): SGFormer.forward zero-pads x to the next multiple of 4 columns in its entry copy
    (K.pad_rows) and the stems run with zero-padded weight columns — same logits as the oracle, gradients in the PARAMETERS'
    own shapes, and a feature tensor seen before is not copied again (the copy is keyed on identity + version)."""
    from sgformer_amd import ops
    from sgformer_amd.ours import SGFormer
    cfg = CONFIGS[name]
    n, f, d, c = 150, 13, 16, 3
    torch.manual_seed(3)
    x = torch.randn(n, f)
    ei = O.synthetic_graph(n, 5.0, seed=4)
    y = torch.randint(0, c, (n,))
    idx = torch.arange(0, n, 2)
    p = O.init_params(cfg, f, d, c, seed=5)
    m = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, compute_dtype=dtype, **cfg)
    m.load_state_dict({**m.state_dict(), **p})
    m.train()
    calls = []
    real = ops.K.pad_rows
    ops.K.pad_rows = staticmethod(lambda *a: (calls.append(a[2]), real(*a))[1])
    try:
        logits = m(x, ei)
        O.nll_loss(logits.float(), y, idx).backward()
        with torch.no_grad():
            m(x, ei)
        assert calls == [16], calls                      # 13 -> 16 columns, ONE copy for both forwards
        x.add_(0.0)                                       # an in-place write bumps the version: the copy is redone
        with torch.no_grad():
            m(x, ei)
        assert calls == [16, 16]
    finally:
        ops.K.pad_rows = real
    p64 = {k: v.double().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in p.items()}
    ref = O.sgformer_forward(p64, x.double(), ei, cfg, training=True)
    O.nll_loss(ref, y, idx).backward()
    tol = 2e-5 if dtype is None else 6e-2
    assert float((logits.detach().double() - ref.detach()).abs().max()) < tol * max(1.0, float(ref.detach().abs().max()))
    for k, prm in m.named_parameters():
        if p64[k].grad is not None:
            assert prm.grad is not None and prm.grad.shape == prm.shape, k
    if dtype is None:
        for k in ("graph_conv.fcs.0.weight", "trans_conv.fcs.0.weight"):
            assert _rel(dict(m.named_parameters())[k].grad, p64[k].grad) < 2e-3, k


def test_captured_step_table_does_not_travel_with_copies_of_the_model():
    """sgformer_amd.graphed keeps a model's captured steps in the module's __dict__; `copy.deepcopy(model)` (100M/nb-sample.py:197
    snapshots its best model that way) and pickling must start without them — a hipGraph can be neither copied nor pickled."""
    import copy
    import pickle
    import threading
    from sgformer_amd import graphed
    from sgformer_amd.ours import SGFormer
    model = SGFormer(8, 16, 3, trans_num_layers=1, gnn_num_layers=1)
    table = graphed._PerModel()
    table[(1, 2)] = threading.Lock()                 # stands in for a captured graph: deepcopy / pickle of it raise
    model.__dict__["_sgf_graphed"] = table
    twin = copy.deepcopy(model)
    assert isinstance(twin.__dict__["_sgf_graphed"], graphed._PerModel) and len(twin.__dict__["_sgf_graphed"]) == 0
    assert len(model.__dict__["_sgf_graphed"]) == 1
    again = pickle.loads(pickle.dumps(table))
    assert isinstance(again, graphed._PerModel) and len(again) == 0
    assert "_sgf_graphed" not in model.state_dict()


def test_model_core_on_fixed_capacity_csr_arrays_equals_the_forward():
    """What sgformer_amd.graphed captures, on the CPU kernel table: SGFormer._core on a StaticCSR (fixed-capacity arrays with a
    stale tail beyond the batch's entries, A^T == A) after _entry_copy_uncached == model(x, edge_index), logits and gradients —
    for two different graphs loaded into the SAME arrays one after the other."""
    from sgformer_amd import graphed, ops, synth
    from sgformer_amd.ours import SGFormer
    from tests.cpu_kernels import CpuKernels
    prev = ops.set_kernels(CpuKernels())
    try:
        n, f, c, d = 300, 10, 5, 16
        torch.manual_seed(0)
        model = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, **synth.RECIPES["ogbn-products"]).train()
        static = None
        for seed in (1, 2):
            ei = synth.synthetic_graph(n, 6.0 + seed, seed=seed)
            x = torch.randn(n, f, generator=torch.Generator().manual_seed(seed))
            g = ops.CSRGraph(ei, n)
            g.transposed()
            assert g.symmetric
            if static is None:
                static = graphed.StaticCSR(n, g.nnz * 2 + 100, torch.device("cpu"), 0)
                static.colind.fill_(n - 1)                       # a stale tail that would be wrong if it were read
                static.val.fill_(7.0)
            static.load(g.rowptr, g.colind, g.val)
            assert static.nnz == g.nnz and static.transposed()[1] is static.colind
            bn = {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}
            want = model(x, ei)
            gw = torch.autograd.grad(want.sum(), list(model.parameters()), allow_unused=True)
            model.load_state_dict(bn, strict=False)              # same BatchNorm buffers for the second evaluation
            got = model._core(model._entry_copy_uncached(x, x.dtype), static, None, None, x.dtype)
            gg = torch.autograd.grad(got.sum(), list(model.parameters()), allow_unused=True)
            assert torch.allclose(got, want, atol=1e-6, rtol=1e-6)
            for a, b in zip(gg, gw):
                assert (a is None) == (b is None)
                if a is not None:
                    assert torch.allclose(a, b, atol=1e-5, rtol=1e-5)
            ops.graph_cache.clear()
    finally:
        ops.set_kernels(prev)
