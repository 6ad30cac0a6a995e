"""CPU restatement of the SGFormer hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (sgformer_amd/) never does and has no CPU path of its own.

Every function cites the reference lines (relative to /root/reference) whose arithmetic it
restates.  The restatement is *functional* (parameters in a dict keyed like the reference's
state_dict) and dtype-generic, so the same code serves as the fp64 oracle for kernel-level parity
(the attention term is tiny next to N*V and must be checked on intermediates, in
fp64, with relative tolerances) and as the fp32 CPU baseline timed by bench.py.

Parity pin: the reference repository has no tests or golden vectors of its own, so
this file is pinned against the reference *itself*: oracle/ref_shim.py imports
/root/reference/large/ours.py unchanged (with stand-ins for the un-vendored torch_sparse /
torch_geometric calls), oracle/make_golden.py dumps its outputs, gradients and intermediates into
tests/golden/, and tests/test_oracle.py checks this restatement against those fixtures (and,
when /root/reference is present, against the live reference).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch

Tensor = torch.Tensor


# ------------------------------------------------------------------------------------------------
# T1  adjacency normalisation + CSR           large/ours.py:26-33 (= 100M/ours.py:72-79)
# ------------------------------------------------------------------------------------------------
def csr_build(edge_index: np.ndarray, n: int):
    """rowptr int64 [n+1], colind int64 [nnz], val fp32 [nnz], deg int64 [n].

    row, col = edge_index; d = degree(col, N) (:28); value = 1 * sqrt(1/d[col]) * sqrt(1/d[row])
    (:29-31) in fp32; nan_to_num -> 0 (:32); SparseTensor(row=col, col=row, value) (:33): entries
    keyed and sorted by (col, row) — torch_sparse's storage order, the key large/data_utils.py:159
    spells out as `col * N + row` — duplicates kept.
    """
    ei = np.asarray(edge_index, dtype=np.int64)
    row, col = ei[0], ei[1]
    deg = np.bincount(col, minlength=n).astype(np.int64)
    d32 = deg.astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        d_in = np.sqrt(np.float32(1.0) / d32[col])
        d_out = np.sqrt(np.float32(1.0) / d32[row])
        value = (np.float32(1.0) * d_in) * d_out
    value = np.nan_to_num(value, nan=0.0, posinf=0.0, neginf=0.0).astype(np.float32)
    perm = np.argsort(col * np.int64(n) + row, kind="stable")
    colind = row[perm]
    val = value[perm]
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(deg, out=rowptr[1:])
    return rowptr, colind, val, deg


def csr_transpose(edge_index: np.ndarray, n: int):
    """CSR of A^T (what torch_sparse's autograd uses for dX = A^T dY) + symmetry flag."""
    ei = np.asarray(edge_index, dtype=np.int64)
    rowptr, colind, val, deg = csr_build(ei, n)
    row, col = ei[0], ei[1]
    d32 = deg.astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        value = (np.float32(1.0) * np.sqrt(np.float32(1.0) / d32[col])) * np.sqrt(np.float32(1.0) / d32[row])
    value = np.nan_to_num(value, nan=0.0, posinf=0.0, neginf=0.0).astype(np.float32)
    perm = np.argsort(row * np.int64(n) + col, kind="stable")
    t_colind = col[perm]
    t_val = value[perm]
    t_rowptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(row, minlength=n), out=t_rowptr[1:])
    sym = bool(np.array_equal(t_rowptr, rowptr) and np.array_equal(t_colind, colind))
    return t_rowptr, t_colind, t_val, sym


# ------------------------------------------------------------------------------------------------
# T2  sum-reduce SpMM                         large/ours.py:34 (torch_sparse.matmul, reduce="sum")
# ------------------------------------------------------------------------------------------------
def spmm(rowptr: np.ndarray, colind: np.ndarray, val: np.ndarray, x: Tensor) -> Tensor:
    """Y[i] = sum_e val[e] * X[colind[e]] — via a torch CSR product in x's dtype (fp64 for parity)."""
    n = len(rowptr) - 1
    a = torch.sparse_csr_tensor(torch.from_numpy(np.ascontiguousarray(rowptr)),
                                torch.from_numpy(np.ascontiguousarray(colind).astype(np.int64)),
                                torch.from_numpy(np.ascontiguousarray(val)).to(x.dtype),
                                size=(n, x.shape[0]))
    return a @ x


def build_adj(edge_index: Tensor, n: int, dtype=torch.float32):
    """The normalised adjacency of :26-33 as a torch sparse-CSR tensor (MKL SpMM on CPU).  Used by
    the timed CPU baseline so that it is not a straw man; built once per graph,
    which is kinder to the CPU than the reference, which re-sorts the edges in every layer."""
    rowptr, colind, val, _ = csr_build(edge_index.cpu().numpy(), n)
    return torch.sparse_csr_tensor(torch.from_numpy(rowptr), torch.from_numpy(colind),
                                   torch.from_numpy(val).to(dtype), size=(n, n))


def gcn_propagate(x: Tensor, edge_index, adj=None) -> Tensor:
    """large/ours.py:26-34 end to end, differentiable (index_add form, any dtype); with a prebuilt
    `adj` (build_adj) the product is one sparse-CSR matmul instead."""
    if adj is not None:
        return torch.sparse.mm(adj, x)
    n = x.shape[0]
    row, col = edge_index[0], edge_index[1]
    d = torch.zeros(n, dtype=torch.float32).index_add_(0, col, torch.ones(col.numel(), dtype=torch.float32))
    value = (1.0 / d[col]).sqrt() * (1.0 / d[row]).sqrt()      # fp32, as the reference
    value = torch.nan_to_num(value, nan=0.0, posinf=0.0, neginf=0.0).to(x.dtype)
    out = torch.zeros_like(x)
    return out.index_add(0, col, value.unsqueeze(1) * x[row])   # A[col,row] = value


# ------------------------------------------------------------------------------------------------
# T3  linear global attention                 large/ours.py:130-157
# ------------------------------------------------------------------------------------------------
def attention(qs: Tensor, ks: Tensor, vs: Tensor, return_parts: bool = False, n_total=None):
    """qs, ks: [N,H,M]; vs: [N,H,D] or [N,1,D].  Returns mean over heads [N,D] (:157).

    `n_total` overrides the N of :133 in num/den (the global node count of a node-sharded run, and
    a handle for tests to make the all-pair term visible next to N*V)."""
    qn = qs / torch.norm(qs, p=2)                                  # :131 global Frobenius norm
    kn = ks / torch.norm(ks, p=2)                                  # :132
    n = qs.shape[0] if n_total is None else n_total
    kvs = torch.einsum("lhm,lhd->hmd", kn, vs.expand(-1, qs.shape[1], -1))   # :136
    num = torch.einsum("nhm,hmd->nhd", qn, kvs) + n * vs           # :137-138
    ks_sum = kn.sum(dim=0)                                         # :141-142
    den = torch.einsum("nhm,hm->nh", qn, ks_sum).unsqueeze(-1) + n  # :143-148
    o = num / den                                                  # :149
    out = o.mean(dim=1)                                            # :157
    if return_parts:
        return out, {"kvs": kvs, "ks_sum": ks_sum, "num": num, "den": den, "o": o,
                     "nq": torch.norm(qs, p=2), "nk": torch.norm(ks, p=2)}
    return out


def attention_raw_stats(qs: Tensor, ks: Tensor, vs: Tensor) -> Tensor:
    """[S0 = K^T V | z0 = sum K | ssq_q | ssq_k]: the un-normalised partials libsgf reduces."""
    h = qs.shape[1]
    s0 = torch.einsum("lhm,lhd->hmd", ks, vs.expand(-1, h, -1))
    z0 = ks.sum(dim=0)
    return torch.cat([s0.reshape(-1), z0.reshape(-1), (qs * qs).sum().reshape(1),
                      (ks * ks).sum().reshape(1)])


# ------------------------------------------------------------------------------------------------
# T4-T7  the module, functionally            large/ours.py:74-94, 121-162, 194-219, 265-276
# ------------------------------------------------------------------------------------------------
DEFAULT_CFG = dict(
    trans_num_layers=1, trans_num_heads=1, trans_use_bn=True, trans_use_residual=True,
    trans_use_weight=True, trans_use_act=True, gnn_num_layers=1, gnn_use_weight=True,
    gnn_use_init=False, gnn_use_bn=True, gnn_use_residual=True, gnn_use_act=True, use_graph=True,
    graph_weight=0.8, aggregate="add", alpha=None)


def _linear(p, key, x):
    return x @ p[key + ".weight"].t() + p[key + ".bias"]


def _layer_norm(p, key, x, eps=1e-5):
    mu = x.mean(dim=1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * p[key + ".weight"] + p[key + ".bias"]


def _batch_norm(p, key, x, training, stats_out, eps=1e-5):
    if training:
        mu = x.mean(dim=0)
        var = ((x - mu) ** 2).mean(dim=0)
        if stats_out is not None:
            n = x.shape[0]
            stats_out[key] = (mu.detach(), (var * n / max(n - 1, 1)).detach())
    else:
        mu, var = p[key + ".running_mean"], p[key + ".running_var"]
    return (x - mu) / torch.sqrt(var + eps) * p[key + ".weight"] + p[key + ".bias"]


def trans_conv(p: Dict[str, Tensor], x: Tensor, cfg: dict, parts: Optional[dict] = None) -> Tensor:
    """TransConv.forward with dropout inactive (large/ours.py:194-219; 100M/ours.py:247-272)."""
    pre = "trans_conv."
    h = cfg["trans_num_heads"]
    x = _linear(p, pre + "fcs.0", x)                               # :198
    if cfg["trans_use_bn"]:
        x = _layer_norm(p, pre + "bns.0", x)                       # :200
    x = torch.relu(x)                                              # :201
    layer_ = [x]
    for i in range(cfg["trans_num_layers"]):
        c = f"{pre}convs.{i}."
        d = p[c + "Wq.weight"].shape[0] // h
        qs = _linear(p, c + "Wq", x).reshape(-1, h, d)             # :123
        ks = _linear(p, c + "Wk", x).reshape(-1, h, d)             # :124
        if cfg["trans_use_weight"]:
            vs = _linear(p, c + "Wv", x).reshape(-1, h, d)         # :126
        else:
            vs = x.reshape(-1, 1, d)                               # :128
        if parts is not None:
            out, pr = attention(qs, ks, vs, return_parts=True)
            parts[f"attn{i}"] = dict(pr, qs=qs, ks=ks, vs=vs, out=out)
        else:
            out = attention(qs, ks, vs)
        x = out
        if cfg["trans_use_residual"]:
            if cfg.get("alpha") is None:
                x = (x + layer_[i]) / 2.0                          # large/ours.py:211
            else:
                a = cfg["alpha"]
                x = a * x + (1 - a) * layer_[i]                    # 100M/ours.py:264
        if cfg["trans_use_bn"]:
            x = _layer_norm(p, f"{pre}bns.{i + 1}", x)             # :213
        if cfg["trans_use_act"]:
            x = torch.relu(x)                                      # :215
        layer_.append(x)
    return x


def graph_conv(p: Dict[str, Tensor], x: Tensor, edge_index: Tensor, cfg: dict, training: bool,
               bn_stats: Optional[dict] = None, adj=None) -> Tensor:
    """GraphConv.forward with dropout inactive (large/ours.py:74-94)."""
    pre = "graph_conv."
    x = _linear(p, pre + "fcs.0", x)                               # :77
    if cfg["gnn_use_bn"]:
        x = _batch_norm(p, pre + "bns.0", x, training, bn_stats)   # :79
    x = torch.relu(x)                                              # :80
    x0 = x                                                         # layer_[0], never extended (:83)
    for i in range(cfg["gnn_num_layers"]):
        y = gcn_propagate(x, edge_index, adj)                      # GraphConvLayer :26-34
        if cfg["gnn_use_init"]:
            y = _linear(p, f"{pre}convs.{i}.W", torch.cat([y, x0], 1))   # :36-38
        elif cfg["gnn_use_weight"]:
            y = _linear(p, f"{pre}convs.{i}.W", y)                 # :39-40
        if cfg["gnn_use_bn"]:
            y = _batch_norm(p, f"{pre}bns.{i + 1}", y, training, bn_stats)   # :88
        if cfg["gnn_use_act"]:
            y = torch.relu(y)                                      # :90
        if cfg["gnn_use_residual"]:
            y = y + x0                                             # :93 (layer_[-1] is layer_[0])
        x = y
    return x


def sgformer_forward(p: Dict[str, Tensor], x: Tensor, edge_index: Tensor, cfg: dict,
                     training: bool = True, parts: Optional[dict] = None,
                     bn_stats: Optional[dict] = None, adj=None) -> Tensor:
    """SGFormer.forward (large/ours.py:265-276) with dropout p = 0."""
    c = dict(DEFAULT_CFG)
    c.update(cfg)
    x1 = trans_conv(p, x, c, parts)
    if c["use_graph"]:
        x2 = graph_conv(p, x, edge_index, c, training, bn_stats, adj)
        if c["aggregate"] == "add":
            gw = c["graph_weight"]
            xx = gw * x2 + (1 - gw) * x1                           # :270
        else:
            xx = torch.cat((x1, x2), dim=1)                        # :272
        if parts is not None:
            parts["x1"], parts["x2"] = x1, x2
    else:
        xx = x1
    return _linear(p, "fc", xx)                                    # :275


# ------------------------------------------------------------------------------------------------
# medium variant (BASELINE.json config 1, Cora): medium/ours.py + its injected `gnn` = models.GCN,
# a stack of PyG GCNConv layers (medium/models.py:14-63; third-party torch_geometric 1.7.2 GCNConv)
# ------------------------------------------------------------------------------------------------
def gcn_norm_edges(edge_index: Tensor, n: int):
    """torch_geometric 1.7.2 gcn_norm on an unweighted graph: add_remaining_self_loops == drop the
    existing self-loops and append one per node (fill 1).  Returns the edge_index GCNConv propagates
    over; its weights deg^-1/2[row] * deg^-1/2[col] (deg = in-degree) are the same symmetric
    normalisation as large/ours.py:28-31, so csr_build / gcn_propagate apply to it unchanged
    (value differences vs PyG's deg.pow(-0.5) are last-bit only)."""
    keep = edge_index[0] != edge_index[1]
    loops = torch.arange(n, dtype=edge_index.dtype)
    return torch.cat([edge_index[:, keep], torch.stack([loops, loops])], dim=1)


def medium_gcn(p: Dict[str, Tensor], x: Tensor, edge_index: Tensor, training: bool, use_bn: bool = True,
               bn_stats: Optional[dict] = None, pre: str = "gnn.") -> Tensor:
    """models.GCN.forward with dropout inactive (medium/models.py:49-63): conv -> BN -> relu for all
    but the last GCNConv; GCNConv = A_norm (x @ weight) + bias (weight [in, out])."""
    ei = gcn_norm_edges(edge_index, x.shape[0])
    row, col = ei[0], ei[1]
    # gcn_norm in x's dtype: deg = scatter_add(1, col); deg^-1/2 (inf -> 0); norm = dis[row] * 1 * dis[col]
    deg = torch.zeros(x.shape[0], dtype=x.dtype).index_add_(0, col, torch.ones(col.numel(), dtype=x.dtype))
    dis = deg.pow(-0.5)
    dis = torch.where(torch.isinf(dis), torch.zeros_like(dis), dis)
    norm = (dis[row] * dis[col]).unsqueeze(1)
    n_conv = len([k for k in p if k.startswith(pre + "convs.") and k.endswith(".weight")])
    for i in range(n_conv):
        xw = x @ p[f"{pre}convs.{i}.weight"]
        x = torch.zeros_like(xw).index_add(0, col, norm * xw[row]) + p[f"{pre}convs.{i}.bias"]
        if i < n_conv - 1:
            if use_bn:
                x = _batch_norm(p, f"{pre}bns.{i}", x, training, bn_stats)
            x = torch.relu(x)
    return x


MEDIUM_DEFAULT_CFG = dict(num_layers=2, num_heads=1, alpha=0.5, use_bn=True, use_residual=True,
                          use_weight=True, use_graph=True, graph_weight=0.8, aggregate="add",
                          gnn_use_bn=True)


def medium_forward(p: Dict[str, Tensor], x: Tensor, edge_index: Tensor, cfg: dict, training: bool = True,
                   bn_stats: Optional[dict] = None) -> Tensor:
    """medium/ours.py SGFormer.forward(data) (:202-213) with gnn = models.GCN.  The TransConv is the
    100M flavour (alpha residual) and NEVER applies the post-layer activation: SGFormer passes its
    arguments positionally and drops use_act (medium/ours.py:183)."""
    c = dict(MEDIUM_DEFAULT_CFG)
    c.update(cfg)
    tc = dict(DEFAULT_CFG, trans_num_layers=c["num_layers"], trans_num_heads=c["num_heads"],
              trans_use_bn=c["use_bn"], trans_use_residual=c["use_residual"],
              trans_use_weight=c["use_weight"], trans_use_act=False, alpha=c["alpha"])
    x1 = trans_conv(p, x, tc)
    if c["use_graph"]:
        x2 = medium_gcn(p, x, edge_index, training, c["gnn_use_bn"], bn_stats)
        if c["aggregate"] == "add":
            xx = c["graph_weight"] * x2 + (1 - c["graph_weight"]) * x1
        else:
            xx = torch.cat((x1, x2), dim=1)
    else:
        xx = x1
    return _linear(p, "fc", xx)


# ------------------------------------------------------------------------------------------------
# DIFFormer (medium/difformer.py): the "simple" kernel + gcn_conv
# ------------------------------------------------------------------------------------------------
def difformer_attention(qs: Tensor, ks: Tensor, vs: Tensor) -> Tensor:
    """medium/difformer.py:18-39: SGFormer's attention with sum_l V_l in place of N * V_n."""
    qn = qs / torch.norm(qs, p=2)
    kn = ks / torch.norm(ks, p=2)
    n = qs.shape[0]
    kvs = torch.einsum("lhm,lhd->hmd", kn, vs.expand(-1, qs.shape[1], -1))
    num = torch.einsum("nhm,hmd->nhd", qn, kvs) + vs.sum(dim=0, keepdim=True)
    den = torch.einsum("nhm,hm->nh", qn, kn.sum(dim=0)).unsqueeze(-1) + n
    return num / den


DIFFORMER_DEFAULT_CFG = dict(num_layers=2, num_heads=1, alpha=0.5, use_bn=True, use_residual=True,
                             use_weight=True, use_graph=True, graph_weight=-1, use_source=False)


def difformer_forward(p: Dict[str, Tensor], x: Tensor, edge_index: Tensor, cfg: dict) -> Tensor:
    """DIFFormer.forward with kernel='simple' and dropout inactive (medium/difformer.py:180-207)."""
    c = dict(DIFFORMER_DEFAULT_CFG)
    c.update(cfg)
    h = c["num_heads"]
    x = _linear(p, "fcs.0", x)
    if c["use_bn"]:
        x = _layer_norm(p, "bns.0", x)
    x = torch.relu(x)
    layer_ = [x]
    for i in range(c["num_layers"]):
        pre = f"convs.{i}."
        d = p[pre + "Wq.weight"].shape[0] // h
        qs = _linear(p, pre + "Wq", x).reshape(-1, h, d)
        ks = _linear(p, pre + "Wk", x).reshape(-1, h, d)
        vs = _linear(p, pre + "Wv", x).reshape(-1, h, d) if c["use_weight"] else x.reshape(-1, 1, d)
        out = difformer_attention(qs, ks, vs)                                   # :118-121
        if c["use_graph"]:                                                      # :124-129
            g = torch.stack([gcn_propagate(vs[:, j], edge_index) for j in range(vs.shape[1])], dim=1)
            gw = c["graph_weight"]
            out = (1 - gw) * out + gw * g if gw > 0 else out + g
        out = out.mean(dim=1)                                                   # :132
        if c["use_source"]:
            out = out + layer_[0]                                               # :134-135 (x_0 = layer_[0], :195)
        if c["use_residual"]:
            out = c["alpha"] * out + (1 - c["alpha"]) * layer_[i]               # :197
        if c["use_bn"]:
            out = _layer_norm(p, f"bns.{i + 1}", out)                           # :199
        x = out
        layer_.append(x)
    return _linear(p, "fcs.1", x)                                               # :205


def nll_loss(logits: Tensor, y: Tensor, idx: Tensor) -> Tensor:
    """large/main.py:139-141: log_softmax + NLLLoss on the training rows."""
    lp = torch.log_softmax(logits, dim=1)
    return torch.nn.functional.nll_loss(lp[idx], y[idx])


# ------------------------------------------------------------------------------------------------
# synthetic inputs shared by tests, smoke() and bench.py
# ------------------------------------------------------------------------------------------------
def synthetic_graph(n, avg_deg, seed=123, directed=False, device="cpu"):
    """The product-side generator (sgformer_amd/synth.py), re-exported for the tests."""
    from sgformer_amd.synth import synthetic_graph as _g
    return _g(n, avg_deg, seed=seed, directed=directed, device=device)


def init_params(cfg: dict, f: int, d: int, c: int, seed: int = 0, dtype=torch.float32) -> Dict[str, Tensor]:
    """Random parameters with the reference's state_dict keys and shapes."""
    cc = dict(DEFAULT_CFG)
    cc.update(cfg)
    g = torch.Generator().manual_seed(seed)
    h = cc["trans_num_heads"]

    def lin(key, out_f, in_f, p):
        b = 1.0 / math.sqrt(in_f)
        p[key + ".weight"] = ((torch.rand(out_f, in_f, generator=g) * 2 - 1) * b).to(dtype)
        p[key + ".bias"] = ((torch.rand(out_f, generator=g) * 2 - 1) * b).to(dtype)

    def norm(key, p, bn):
        p[key + ".weight"] = (1 + 0.1 * torch.randn(d, generator=g)).to(dtype)
        p[key + ".bias"] = (0.1 * torch.randn(d, generator=g)).to(dtype)
        if bn:
            p[key + ".running_mean"] = (0.1 * torch.randn(d, generator=g)).to(dtype)
            p[key + ".running_var"] = (1 + 0.1 * torch.rand(d, generator=g)).to(dtype)

    p: Dict[str, Tensor] = {}
    lin("trans_conv.fcs.0", d, f, p)
    norm("trans_conv.bns.0", p, False)
    for i in range(cc["trans_num_layers"]):
        lin(f"trans_conv.convs.{i}.Wk", d * h, d, p)
        lin(f"trans_conv.convs.{i}.Wq", d * h, d, p)
        if cc["trans_use_weight"]:
            lin(f"trans_conv.convs.{i}.Wv", d * h, d, p)
        norm(f"trans_conv.bns.{i + 1}", p, False)
    lin("graph_conv.fcs.0", d, f, p)
    norm("graph_conv.bns.0", p, True)
    for i in range(cc["gnn_num_layers"]):
        lin(f"graph_conv.convs.{i}.W", d, 2 * d if cc["gnn_use_init"] else d, p)
        norm(f"graph_conv.bns.{i + 1}", p, True)
    lin("fc", c, 2 * d if cc["aggregate"] == "cat" else d, p)
    return p
