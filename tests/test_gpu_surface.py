"""Small surface rows of SURVEY.md §8b / VERDICT r1 item 8 on the GPU: edge-weighted GCNConv and
DIFFormer gcn_conv (medium/models.py:55-62, medium/difformer.py:63-79), per-head outputs of
full_attention_conv for H > 1 (medium/ours.py:14-46), GCNConv with a class-count output width
(`--method gcn`, medium/parse.py:19-23)."""
import numpy as np
import pytest
import torch

from oracle import ref_shim
from oracle import sgformer_oracle as O

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


@pytest.mark.parametrize("out_channels", [64, 7])
@pytest.mark.parametrize("weighted", [False, True])
def test_gcnconv_matches_pyg_semantics(cuda, weighted, out_channels):
    """GCNConv(x, edge_index[, edge_weight]) against the fp64 PyG 1.7.2 restatement of oracle/ref_shim.py
    (gcn_norm with add_remaining_self_loops: an existing self-loop keeps its weight), forward and grads."""
    from sgformer_amd import ours_medium as M
    n, f = 900, 40
    g = torch.Generator().manual_seed(out_channels)
    ei = O.synthetic_graph(n, 6.0, seed=3)[:, : -n // 2]          # some nodes keep a self-loop, some do not
    w = (torch.rand(ei.shape[1], generator=g) + 0.2) if weighted else None
    x = torch.randn(n, f, generator=g)
    conv = M.GCNConv(f, out_channels)
    with torch.no_grad():
        conv.bias.normal_(0, 0.1)
    ref = ref_shim._GCNConv(f, out_channels).double()
    ref.load_state_dict({k: v.double() for k, v in conv.state_dict().items()})
    xr = x.double().requires_grad_(True)
    yr = ref(xr, ei, None if w is None else w.double())
    go = torch.randn(n, out_channels, generator=g)
    (yr * go.double()).sum().backward()
    conv = conv.to(cuda)
    xg = x.to(cuda).requires_grad_(True)
    y = conv(xg, ei.to(cuda), None if w is None else w.to(cuda))
    (y * go.to(cuda)).sum().backward()
    assert y.shape == (n, out_channels)
    assert _rel(y, yr.detach()) <= 2e-6
    assert _rel(xg.grad, xr.grad) <= 2e-6
    assert _rel(conv.weight.grad, ref.weight.grad) <= 2e-6 and _rel(conv.bias.grad, ref.bias.grad) <= 2e-6


def test_difformer_gcn_conv_with_edge_weight(cuda):
    """medium/difformer.py:63-79 restated in numpy fp64: value = w * (1/d[col]).sqrt() * (1/d[row]).sqrt(),
    nan_to_num, A[col, row] = value, per head."""
    from sgformer_amd.difformer import gcn_conv
    n, h, d = 700, 2, 32
    g = torch.Generator().manual_seed(1)
    ei = O.synthetic_graph(n, 5.0, seed=8, directed=True)
    w = torch.rand(ei.shape[1], generator=g) + 0.1
    x = torch.randn(n, h, d, generator=g)
    row, col = ei[0].numpy(), ei[1].numpy()
    deg = np.bincount(col, minlength=n).astype(np.float64)
    with np.errstate(divide="ignore"):
        val = w.double().numpy() * np.sqrt(1.0 / deg[col]) * np.sqrt(1.0 / deg[row])
    val = np.nan_to_num(val, nan=0.0, posinf=0.0, neginf=0.0)
    ref = np.zeros((n, h, d))
    np.add.at(ref, col, val[:, None, None] * x.double().numpy()[row])
    xg = x.to(cuda).requires_grad_(True)
    y = gcn_conv(xg, ei.to(cuda), w.to(cuda))
    assert _rel(y, torch.from_numpy(ref)) <= 2e-6
    go = torch.randn(n, h, d, generator=g)
    (y * go.to(cuda)).sum().backward()
    gref = np.zeros((n, h, d))
    np.add.at(gref, row, val[:, None, None] * go.double().numpy()[col])          # A^T
    assert _rel(xg.grad, torch.from_numpy(gref)) <= 2e-6


@pytest.mark.parametrize("v_heads", [2, 1])
def test_full_attention_conv_per_head_outputs(cuda, v_heads):
    """H = 2: [N, H, D] per-head outputs against the restatement of medium/ours.py:14-46 — values, and (r05) the gradients of
    Q, K, V for an arbitrary per-head cotangent through sgf_attn_bwd_reduce_heads / _apply_heads vs fp64 autograd."""
    from sgformer_amd.ours import full_attention_conv
    n, h, d = 1500, 2, 64
    g = torch.Generator().manual_seed(4)
    qs, ks = torch.randn(n, h, d, generator=g), torch.randn(n, h, d, generator=g)
    vs = torch.randn(n, v_heads, d, generator=g)
    with torch.no_grad():
        out = full_attention_conv(qs.to(cuda), ks.to(cuda), vs.to(cuda))
    assert out.shape == (n, h, d)
    q64, k64, v64 = qs.double(), ks.double(), vs.double()
    qn, kn = q64 / q64.norm(), k64 / k64.norm()
    kvs = torch.einsum("lhm,lhd->hmd", kn, v64.expand(n, h, d) if v_heads == 1 else v64)
    num = torch.einsum("nhm,hmd->nhd", qn, kvs) + n * v64
    den = torch.einsum("nhm,hm->nh", qn, kn.sum(0)).unsqueeze(-1) + n
    assert _rel(out, num / den) <= 2e-6
    q64, k64, v64 = (t.double().requires_grad_(True) for t in (qs, ks, vs))
    qn, kn = q64 / q64.norm(), k64 / k64.norm()
    kvs = torch.einsum("lhm,lhd->hmd", kn, v64.expand(n, h, d) if v_heads == 1 else v64)
    ref = (torch.einsum("nhm,hmd->nhd", qn, kvs) + n * v64) / (torch.einsum("nhm,hm->nh", qn, kn.sum(0)).unsqueeze(-1) + n)
    cot = torch.randn(n, h, d, generator=g)
    # a cotangent whose all-pair part matters: SURVEY 0.5 — the self term N V swamps the rest, so weigh the small term too
    gq, gk, gv = torch.autograd.grad((ref * cot.double()).sum(), (q64, k64, v64))
    qd, kd, vd = (t.to(cuda).requires_grad_(True) for t in (qs, ks, vs))
    got = full_attention_conv(qd, kd, vd)
    assert got.shape == (n, h, d) and got.requires_grad
    dq, dk, dv = torch.autograd.grad((got * cot.to(cuda)).sum(), (qd, kd, vd))
    assert dv.shape == vs.shape
    assert _rel(dv.cpu(), gv) <= 1e-5
    assert _rel(dq.cpu(), gq) <= 2e-3 and _rel(dk.cpu(), gk) <= 2e-3
