"""Drop-in for the reference's `medium/difformer.py` (`--method difformer`): the
DIFFormer "simple" linear attention + normalised-adjacency propagation, on the same libsgf kernels.

DIFFormer's simple kernel (medium/difformer.py:18-39) is SGFormer's attention with `sum_l V_l` in
place of `N * V_n` in the numerator, so it runs on `ops.attention_from_input(..., sum_v=True)` (Gram
of the layer input + d x d algebra + one apply pass; Q / K never materialised); `gcn_conv`
(:63-79) is the same symmetric normalisation and sum-reduce SpMM as large/ours.py:26-34, applied to
the projected V.  Same class names, constructor signatures, parameter names and creation order as
the reference, so `state_dict`s interchange.  The O(N^2) `sigmoid` kernel and `output_attn` maps are
plain PyTorch on the GPU, as in the reference.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .ours import _drop

__all__ = ["DIFFormer", "DIFFormerConv", "full_attention_conv", "gcn_conv"]


def _graph(edge_index, n):
    return ops.graph_cache.get(edge_index, n)


def _difformer_values(ei, w, n):
    """medium/difformer.py:66-74: value = edge_weight * (1/d[col]).sqrt() * (1/d[row]).sqrt() with d the
    UNWEIGHTED in-degree, non-finite -> 0."""
    row, col = ei[0], ei[1]
    d = torch.bincount(col, minlength=n).float()
    v = w.detach().float() * (1.0 / d[col]).sqrt() * (1.0 / d[row]).sqrt()
    return ei, torch.nan_to_num(v, nan=0.0, posinf=0.0, neginf=0.0)


def gcn_conv(x, edge_index, edge_weight=None):
    """medium/difformer.py:63-79: per head, D^-1/2 A D^-1/2 x.  x: [N, H, D] -> [N, H, D]."""
    n, h, d = x.shape
    g = _graph(edge_index, n) if edge_weight is None else ops.weighted_graph(edge_index, edge_weight, n,
                                                                             _difformer_values, "difformer_w")
    x2 = x.reshape(n, h * d)
    return torch.stack([ops.spmm(g, x2[:, i * d:(i + 1) * d]) for i in range(h)], dim=1)


def _dense_attention(qs, ks, vs, kernel, want_attn):
    """The reference arithmetic in plain PyTorch (sigmoid kernel and attention maps: O(N^2))."""
    if kernel == 'simple':
        qs = qs / torch.norm(qs, p=2)
        ks = ks / torch.norm(ks, p=2)
        n = qs.shape[0]
        kvs = torch.einsum("lhm,lhd->hmd", ks, vs)
        num = torch.einsum("nhm,hmd->nhd", qs, kvs) + vs.sum(dim=0, keepdim=True)
        den = torch.einsum("nhm,hm->nh", qs, ks.sum(dim=0)).unsqueeze(-1) + n
        out = num / den
        # [N, L, H] / [N, 1, H]  (the reference divides by [N, H, 1], which only broadcasts for H = 1)
        attn = torch.einsum("nhm,lhm->nlh", qs, ks) / den.transpose(1, 2) if want_attn else None
    elif kernel == 'sigmoid':
        num = torch.sigmoid(torch.einsum("nhm,lhm->nlh", qs, ks))
        attn = num / num.sum(dim=1, keepdim=True)
        out = torch.einsum("nlh,lhd->nhd", attn, vs)
    else:
        raise ValueError(f"unknown kernel {kernel!r}")
    return out, attn


def full_attention_conv(qs, ks, vs, kernel, output_attn=False):
    """medium/difformer.py:10-61 as a free function on materialised Q / K / V (plain PyTorch; the
    module's fast path goes through ops.attention_from_input instead)."""
    out, attn = _dense_attention(qs, ks, vs, kernel, output_attn)
    return (out, attn) if output_attn else out


class DIFFormerConv(nn.Module):
    """one DIFFormer layer (medium/difformer.py:81-141)."""

    def __init__(self, in_channels, out_channels, num_heads, kernel='simple', use_graph=True,
                 use_weight=True, graph_weight=-1, use_source=False):
        super().__init__()
        self.Wk = nn.Linear(in_channels, out_channels * num_heads)
        self.Wq = nn.Linear(in_channels, out_channels * num_heads)
        if use_weight:
            self.Wv = nn.Linear(in_channels, out_channels * num_heads)
        self.out_channels = out_channels
        self.num_heads = num_heads
        self.kernel = kernel
        self.use_graph = use_graph
        self.use_weight = use_weight
        self.graph_weight = graph_weight
        self.use_source = use_source
        self._shard = None

    def reset_parameters(self):
        self.Wk.reset_parameters()
        self.Wq.reset_parameters()
        if self.use_weight:
            self.Wv.reset_parameters()

    def _attention(self, x, value, output_attn):
        h, d = self.num_heads, self.out_channels
        n = x.shape[0]
        # one head (every recipe): Gram-based kernels.  With several heads the Frobenius norms of Q and
        # K span ALL heads (torch.norm without dim, medium/difformer.py:20-21), which the one-head
        # algebra does not model: those calls take the reference arithmetic in plain PyTorch.
        fast = (h == 1 and self.kernel == 'simple' and not output_attn and x.shape[1] == d
                and d % 4 == 0 and d <= 256)
        if fast:
            wv, bv = (self.Wv.weight, self.Wv.bias) if self.use_weight else (None, None)
            out = ops.attention_from_input(x, self.Wq.weight, self.Wq.bias, self.Wk.weight, self.Wk.bias,
                                           wv, bv, self._shard, None, True)
            return out.unsqueeze(1), None
        q = ops.linear(x, self.Wq.weight, self.Wq.bias).reshape(n, h, d)
        k = ops.linear(x, self.Wk.weight, self.Wk.bias).reshape(n, h, d)
        return _dense_attention(q, k, value, self.kernel, output_attn)

    def forward(self, query_input, source_input, edge_index=None, edge_weight=None, x_0=None, output_attn=False):
        ops._require_cuda(query_input, source_input)
        if query_input is not source_input:
            raise NotImplementedError("DIFFormerConv: query and source are the same tensor in every caller")
        h, d = self.num_heads, self.out_channels
        n = source_input.shape[0]
        if self.use_weight:
            need_v = (self.use_graph or h > 1 or self.kernel != 'simple' or output_attn
                      or source_input.shape[1] != d)
            value = ops.linear(source_input, self.Wv.weight, self.Wv.bias).reshape(n, h, d) if need_v else None
        else:
            value = source_input.reshape(n, 1, d)
        attention_output, attn = self._attention(source_input, value, output_attn)     # [N, H, D]
        if self.use_graph:
            g = gcn_conv(value, edge_index, edge_weight)                              # [N, H | 1, D]
            if self.graph_weight > 0:
                final_output = (1 - self.graph_weight) * attention_output + self.graph_weight * g
            else:
                final_output = attention_output + g
        else:
            final_output = attention_output
        final_output = final_output.mean(dim=1)
        if self.use_source:
            final_output = final_output + x_0
        if output_attn:
            return final_output, attn
        return final_output


class DIFFormer(nn.Module):
    """medium/difformer.py:143-228: x [N, D], edge_index [2, E] (inside `data.graph`) -> logits [N, C]."""

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers=2, num_heads=1, kernel='simple',
                 alpha=0.5, dropout=0.5, use_bn=True, use_residual=True, use_weight=True, use_graph=True,
                 graph_weight=-1, use_source=False):
        super().__init__()
        self.convs = nn.ModuleList()
        self.fcs = nn.ModuleList()
        self.fcs.append(nn.Linear(in_channels, hidden_channels))
        self.bns = nn.ModuleList()
        self.bns.append(nn.LayerNorm(hidden_channels))
        for _ in range(num_layers):
            self.convs.append(DIFFormerConv(hidden_channels, hidden_channels, num_heads=num_heads, kernel=kernel,
                                            use_graph=use_graph, use_weight=use_weight,
                                            graph_weight=graph_weight, use_source=use_source))
            self.bns.append(nn.LayerNorm(hidden_channels))
        self.fcs.append(nn.Linear(hidden_channels, out_channels))
        self.dropout = dropout
        self.activation = F.relu
        self.use_bn = use_bn
        self.residual = use_residual
        self.alpha = alpha

    def reset_parameters(self):
        for conv in self.convs:
            conv.reset_parameters()
        for bn in self.bns:
            bn.reset_parameters()
        for fc in self.fcs:
            fc.reset_parameters()

    def _norm(self, ln, x, res, a, b, relu):
        if res is None and not self.use_bn and not relu:
            return x
        gamma, beta = (ln.weight, ln.bias) if self.use_bn else (None, None)
        return ops.ln_res_act(x, res, a, b, gamma, beta, relu, ln.eps)

    def forward(self, data, edge_weight=None):
        x = data.graph['node_feat']
        edge_index = data.graph['edge_index']
        ops._require_cuda(x, edge_index)
        x = ops.linear(x, self.fcs[0].weight, self.fcs[0].bias)
        x = self._norm(self.bns[0], x, None, 1.0, 0.0, True)
        x = _drop(x, self.dropout, self.training)
        layer_ = [x]
        a = float(self.alpha)
        for i, conv in enumerate(self.convs):
            x = conv(x, x, edge_index, edge_weight, layer_[0])
            if self.residual:
                x = self._norm(self.bns[i + 1], x, layer_[i], a, 1.0 - a, False)
            else:
                x = self._norm(self.bns[i + 1], x, None, 1.0, 0.0, False)
            x = _drop(x, self.dropout, self.training)
            layer_.append(x)
        return ops.out_linear(x, self.fcs[-1].weight, self.fcs[-1].bias)

    def get_attentions(self, x):
        layer_, attentions = [], []
        x = self._norm(self.bns[0], ops.linear(x, self.fcs[0].weight, self.fcs[0].bias), None, 1.0, 0.0, True)
        layer_.append(x)
        a = float(self.alpha)
        for i, conv in enumerate(self.convs):
            saved = conv.use_graph, conv.use_source
            conv.use_graph, conv.use_source = False, False     # the reference calls conv(x, x, output_attn=True)
            try:
                x, attn = conv(x, x, output_attn=True)
            finally:
                conv.use_graph, conv.use_source = saved
            attentions.append(attn)
            x = self._norm(self.bns[i + 1], x, layer_[i] if self.residual else None,
                           a if self.residual else 1.0, (1.0 - a) if self.residual else 0.0, False)
            layer_.append(x)
        return torch.stack(attentions, dim=0)
