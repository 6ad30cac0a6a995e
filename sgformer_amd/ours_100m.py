"""Drop-in for the reference's `100M/ours.py` (the ogbn-papers100M neighbour-sampling variant).

Same kernels as `sgformer_amd.ours`; what differs from the large variant is the surface
`100M/parse.py:5-8` and `100M/nb-sample.py:29,41` bind against:

  * `SGFormer(..., alpha=0.5, ...)` with the 100M keyword order (100M/ours.py:293-316);
  * the TransConv residual is `alpha * x + (1 - alpha) * layer_[i]` (100M/ours.py:264) instead of
    the large variant's `(x + layer_[i]) / 2` (large/ours.py:211) — one scalar pair in sgf_ln_fwd;
  * `TransConv.forward(x, edge_index=None)` and `TransConvLayer.forward(q, s, edge_index=None,
    output_attn=False)` accept and ignore an edge_index (100M/ours.py:139,247);
  * `full_attention_conv(qs, ks, vs, output_attn=False)` is a module-level function
    (100M/ours.py:12-53).

Neighbour-sampled mini-batches are DIRECTED (edges point from sampled neighbours to seeds,
100M/nb-sample.py:125-133), so the SpMM backward runs on the transposed CSR that
`ops.CSRGraph.transposed()` builds when its symmetry test fails.  GraphConv / GraphConvLayer are
AST-identical to the large variant's and are re-exported unchanged.
"""
from __future__ import annotations

from . import ours as _large
from .ours import GraphConv, GraphConvLayer, full_attention_conv  # noqa: F401  (re-exported)

__all__ = ["GraphConvLayer", "GraphConv", "TransConvLayer", "TransConv", "SGFormer",
           "full_attention_conv"]


class TransConvLayer(_large.TransConvLayer):
    def forward(self, query_input, source_input, edge_index=None, output_attn=False, grad_tap=None):
        return super().forward(query_input, source_input, output_attn=output_attn, grad_tap=grad_tap)


class TransConv(_large.TransConv):
    """100M/ours.py:198-272: positional order (in, hidden, num_layers, num_heads, alpha, dropout,
    use_bn, use_residual, use_weight, use_act)."""

    def __init__(self, in_channels, hidden_channels, num_layers=2, num_heads=1, alpha=0.5,
                 dropout=0.5, use_bn=True, use_residual=True, use_weight=True, use_act=True):
        super().__init__(in_channels, hidden_channels, num_layers, num_heads, dropout, use_bn,
                         use_residual, use_weight, use_act, alpha=alpha,
                         layer_cls=TransConvLayer)

    _attn_post_act = False   # this variant's get_attentions has no activation after a layer

    def forward(self, x, edge_index=None, stem=None):
        return super().forward(x, stem=stem)


class SGFormer(_large.SGFormer):
    """100M/ours.py:292-380."""

    def __init__(self, in_channels, hidden_channels, out_channels,
                 trans_num_layers=1, trans_num_heads=1, trans_dropout=0.5,
                 gnn_num_layers=1, gnn_dropout=0.5, gnn_use_weight=True, gnn_use_init=False,
                 gnn_use_bn=True, gnn_use_residual=True, gnn_use_act=True,
                 alpha=0.5,
                 trans_use_bn=True, trans_use_residual=True, trans_use_weight=True,
                 trans_use_act=True,
                 use_graph=True, graph_weight=0.8, aggregate="add", compute_dtype="default"):
        super().__init__(in_channels, hidden_channels, out_channels,
                         trans_num_layers=trans_num_layers, trans_num_heads=trans_num_heads,
                         trans_dropout=trans_dropout, trans_use_bn=trans_use_bn,
                         trans_use_residual=trans_use_residual, trans_use_weight=trans_use_weight,
                         trans_use_act=trans_use_act, gnn_num_layers=gnn_num_layers,
                         gnn_dropout=gnn_dropout, gnn_use_weight=gnn_use_weight,
                         gnn_use_init=gnn_use_init, gnn_use_bn=gnn_use_bn,
                         gnn_use_residual=gnn_use_residual, gnn_use_act=gnn_use_act,
                         use_graph=use_graph, graph_weight=graph_weight, aggregate=aggregate,
                         alpha=alpha, compute_dtype=compute_dtype, trans_cls=TransConv)
