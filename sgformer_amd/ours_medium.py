"""Drop-in for the reference's `medium/ours.py` (Cora / Citeseer / ... full-graph, BASELINE.json
config 1) plus the GCN branch it is given (`models.GCN`, medium/models.py:14-63 — SURVEY.md row N3).

What differs from the large variant is the surface `medium/parse.py:97-101` and
`medium/main.py:156` bind against:

  * `SGFormer(in, hidden, out, num_layers=2, num_heads=1, alpha=0.5, dropout=0.5, use_bn=True,
    use_residual=True, use_weight=True, use_graph=True, use_act=False, graph_weight=0.8, gnn=None,
    aggregate='add')` (medium/ours.py:180-181) and `forward(data)` reading
    `data.graph['node_feat' | 'edge_index']` (:134-136, :202-205);
  * the GNN branch is an INJECTED module (`gnn`), any callable `gnn(data) -> [N, hidden]`.  With
    the reference's own `models.GCN` that branch runs on PyG; `GCN` / `GCNConv` below are the same
    model on libsgf (Linear first, then the normalised-adjacency SpMM, then bias —
    torch_geometric 1.7.2 GCNConv: gcn_norm with add_remaining_self_loops, weight [in, out]
    glorot, bias zeros), with identical parameter names, so `state_dict`s interchange.
    `sgformer_amd.launch` swaps it in for `models.GCN` when it runs a medium trainer;
  * TransConv: alpha residual, and — as in the reference, which forwards its constructor arguments
    positionally and drops `use_act` (medium/ours.py:183) — never a post-layer activation.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from . import ours as _large
from .ours import full_attention_conv  # noqa: F401  (re-exported)

__all__ = ["TransConvLayer", "TransConv", "SGFormer", "GCN", "GCNConv", "full_attention_conv"]


def _gcn_graph(edge_index: torch.Tensor, n: int):
    """Cached CSR of GCNConv's propagation matrix: self-loops dropped, one added per node
    (add_remaining_self_loops, fill 1), then D^-1/2 A D^-1/2 with D the in-degree — the same
    symmetric normalisation sgf_csr_build implements for large/ours.py:26-33."""
    def build(ei, _n):
        keep = ei[0] != ei[1]
        loops = torch.arange(n, dtype=ei.dtype, device=ei.device)
        return ops.CSRGraph(torch.cat([ei[:, keep], torch.stack([loops, loops])], dim=1), n)
    return ops.graph_cache.get(edge_index, n, factory=build, tag="gcn_norm")


def _gcn_norm_weighted(ei, w, n):
    """torch_geometric 1.7.2 gcn_norm with edge weights: add_remaining_self_loops (an existing self-loop
    keeps its weight, the others get 1), deg = scatter_add(w, target), norm = deg^-1/2[src] w deg^-1/2[tgt]."""
    row, col = ei[0], ei[1]
    w = w.detach().float()
    keep = row != col
    loop_w = torch.ones(n, dtype=w.dtype, device=w.device)
    loop_w[row[~keep]] = w[~keep]
    loops = torch.arange(n, dtype=ei.dtype, device=ei.device)
    row, col = torch.cat([row[keep], loops]), torch.cat([col[keep], loops])
    w = torch.cat([w[keep], loop_w])
    deg = torch.zeros(n, dtype=w.dtype, device=w.device).scatter_add_(0, col, w)
    dis = deg.pow(-0.5)
    dis = torch.where(torch.isinf(dis), torch.zeros_like(dis), dis)
    return torch.stack([row, col]), dis[row] * w * dis[col]


class GCNConv(nn.Module):
    """out = A_norm (x @ weight) + bias   (torch_geometric 1.7.2 GCNConv with its defaults)."""

    def __init__(self, in_channels, out_channels, cached=False, **kwargs):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels))
        self.bias = nn.Parameter(torch.empty(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        a = math.sqrt(6.0 / (self.weight.size(-2) + self.weight.size(-1)))   # PyG glorot
        self.weight.data.uniform_(-a, a)
        self.bias.data.fill_(0)

    def forward(self, x, edge_index, edge_weight=None):
        ops._require_cuda(x, edge_index)
        w_t = self.weight.t()
        pad = -self.out_channels % 4
        if pad:
            # the kernels move 4 features per lane: compute on the width rounded up to a multiple of 4
            # (zero weight rows) and slice — `--method gcn` uses this class with out_channels = the
            # class count (7 on Cora, medium/parse.py:19-23)
            w_t = F.pad(w_t, (0, 0, 0, pad))
        xw = ops.linear(x, w_t, None)
        graph = (_gcn_graph(edge_index, x.shape[0]) if edge_weight is None else
                 ops.weighted_graph(edge_index, edge_weight, x.shape[0], _gcn_norm_weighted, "gcn_norm_w"))
        y = ops.spmm(graph, xw)
        if pad:
            y = y[:, :self.out_channels]
        return y + self.bias.to(y.dtype)


class GCN(nn.Module):
    """medium/models.py:14-63."""

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers=2, dropout=0.5,
                 save_mem=True, use_bn=True):
        super().__init__()
        self.convs = nn.ModuleList()
        self.convs.append(GCNConv(in_channels, hidden_channels, cached=not save_mem))
        self.bns = nn.ModuleList()
        self.bns.append(nn.BatchNorm1d(hidden_channels))
        for _ in range(num_layers - 2):
            self.convs.append(GCNConv(hidden_channels, hidden_channels, cached=not save_mem))
            self.bns.append(nn.BatchNorm1d(hidden_channels))
        self.convs.append(GCNConv(hidden_channels, out_channels, cached=not save_mem))
        self.dropout = dropout
        self.activation = F.relu
        self.use_bn = use_bn
        self._shard = None

    def reset_parameters(self):
        for conv in self.convs:
            conv.reset_parameters()
        for bn in self.bns:
            bn.reset_parameters()

    def forward(self, data):
        x = data.graph['node_feat']
        edge_index = data.graph['edge_index']
        edge_weight = data.graph['edge_weight'] if 'edge_weight' in data.graph else None
        for i, conv in enumerate(self.convs[:-1]):
            x = conv(x, edge_index, edge_weight)
            if self.use_bn:
                x = _large.GraphConv._bn_act_res(self, self.bns[i], x, None, True)   # BN -> relu fused
            else:
                x = torch.relu(x)
            x = _large._drop(x, self.dropout, self.training)
        return self.convs[-1](x, edge_index)     # as the reference: the last conv gets NO edge weights (models.py:62)


class TransConvLayer(_large.TransConvLayer):
    def forward(self, query_input, source_input, edge_index=None, edge_weight=None, output_attn=False, grad_tap=None):
        return super().forward(query_input, source_input, output_attn=output_attn, grad_tap=grad_tap)


class TransConv(_large.TransConv):
    """medium/ours.py:102-176."""

    def __init__(self, in_channels, hidden_channels, num_layers=2, num_heads=1, alpha=0.5, dropout=0.5,
                 use_bn=True, use_residual=True, use_weight=True, use_act=False):
        super().__init__(in_channels, hidden_channels, num_layers, num_heads, dropout, use_bn,
                         use_residual, use_weight, use_act, alpha=alpha, layer_cls=TransConvLayer)
        self.residual = use_residual

    _attn_post_act = False   # this variant's get_attentions has no activation after a layer

    def forward(self, data):
        return super().forward(data.graph['node_feat'])


class SGFormer(nn.Module):
    """medium/ours.py:179-223."""

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers=2, num_heads=1, alpha=0.5,
                 dropout=0.5, use_bn=True, use_residual=True, use_weight=True, use_graph=True,
                 use_act=False, graph_weight=0.8, gnn=None, aggregate='add'):
        super().__init__()
        # positional, and WITHOUT use_act, exactly as medium/ours.py:183
        self.trans_conv = TransConv(in_channels, hidden_channels, num_layers, num_heads, alpha, dropout,
                                    use_bn, use_residual, use_weight)
        self.gnn = gnn
        self.use_graph = use_graph
        self.graph_weight = graph_weight
        self.use_act = use_act
        self.aggregate = aggregate
        if aggregate == 'add':
            self.fc = nn.Linear(hidden_channels, out_channels)
        elif aggregate == 'cat':
            self.fc = nn.Linear(2 * hidden_channels, out_channels)
        else:
            raise ValueError(f'Invalid aggregate type:{aggregate}')
        self.params1 = list(self.trans_conv.parameters())
        self.params2 = list(self.gnn.parameters()) if self.gnn is not None else []
        self.params2.extend(list(self.fc.parameters()))

    def forward(self, data):
        x1 = self.trans_conv(data)
        if self.use_graph:
            x2 = self.gnn(data)
            if self.aggregate == 'add':
                gw = float(self.graph_weight)
                x = ops.axpby(x2, x1, gw, 1.0 - gw) if x1.shape[1] % 4 == 0 else gw * x2 + (1 - gw) * x1
            else:
                x = torch.cat((x1, x2), dim=1)
        else:
            x = x1
        return ops.out_linear(x, self.fc.weight, self.fc.bias)

    def get_attentions(self, x):
        return self.trans_conv.get_attentions(x)

    def reset_parameters(self):
        self.trans_conv.reset_parameters()
        if self.use_graph:
            self.gnn.reset_parameters()
