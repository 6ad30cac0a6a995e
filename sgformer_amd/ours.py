"""Drop-in for the reference's `large/ours.py` (and, via `variant`, `100M/ours.py`).

Same public surface — class names, constructor keywords, attribute names, `state_dict` keys and
parameter-creation order (hence the same seeded initialisation), `params1` / `params2`,
`reset_parameters`, `get_attentions`, train/eval semantics — so `large/main.py` runs unchanged
once this module is registered as `sys.modules['ours']` (see sgformer_amd/launch.py and
SURVEY.md §8b).  Everything between the parameters and the output is different: the forward and
backward run on the hand-written gfx950 kernels behind include/sgf.h.

    reference                                   here
    ----------------------------------------    ---------------------------------------------------
    large/ours.py:26-33  degree+argsort / layer one cached CSR per edge_index   (ops.CSRGraph)
    large/ours.py:34     torch_sparse.matmul    ops.spmm            (k_spmm_row / k_spmm_seg_bf16x2 / k_spmm_sub)
    large/ours.py:123-157 Wq/Wk/Wv + 2 norms    ops.attention_from_input: Gram of the layer input +
                          + 4 einsums           d x d algebra + one apply pass, Q/K/V never
                                                materialised (one head, query == source); otherwise
                                                one [d -> 3Hd] GEMM + ops.attention (materialised)
    nn.Linear weights / biases under autograd   ops.linear / linear_cat / linear_bn_stats / stem_pair: bf16 square
                                                layers + BatchNorm sums on k_rowgemm_bf16, both stems on k_stem_bf16,
                                                dW / db on sgf_gram
    large/ours.py:198-216 LN / relu / residual  ops.ln_res_act      (k_ln_fwd / k_ln_bwd)
    large/ours.py:77-93  BN / relu / residual   ops.batch_stats + ops.bn_act_res
    large/ours.py:83-93  x0 used 7 times        ops.fan_out         (one fused gradient sum)
    large/ours.py:269-275 weighted add + fc     ops.combine_fc (one kernel, bf16) / ops.axpby + ops.out_linear

GPU only: a CPU tensor raises (no eager fallback by design).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

__all__ = ["GraphConvLayer", "GraphConv", "TransConvLayer", "TransConv", "SGFormer",
           "full_attention_conv"]

# Set by sgformer_amd.dist.shard_model(); None = single GPU.
_NO_SHARD = None

# Activation dtype of SGFormer instances built without an explicit `compute_dtype`.  None = the
# dtype of the input (the reference's fp32 numerics); sgformer_amd.launch --sgf-dtype bf16 sets
# torch.bfloat16 so that an unchanged trainer gets the bf16 mode of BASELINE.json config 3.
DEFAULT_COMPUTE_DTYPE = None

# The attention branch runs on a side HIP stream next to the GCN branch (SGF_OVERLAP=0 turns it off): the two are independent
# until the combine, and autograd replays each backward node on its forward stream, so the backward overlaps too.  r06, MI355X,
# same box back to back, bf16 ogbn-products shape: uniform graph 89.5 -> 86.2 ms per step, community graph 40.5 -> 39.5 ms,
# ogbn-arxiv fp32 9.75 -> 9.32 ms, losses identical to the last bit (every reduction is per-launch deterministic).  It was
# opt-in until r05 because an fp32 run had hung when a LIBRARY GEMM, whose workgroups wait on each other, shared the CUs with
# a persistent one-block-per-CU kernel of the other stream; no library GEMM is left on the path (r05), and no default kernel
# of libsgf waits on another workgroup (the opt-in SGF_GCN_BWD_FUSED rendezvous kernel keeps the single-stream path).
# Full-graph steps of at least OVERLAP_MIN_NODES nodes only; never inside a stream capture (sgformer_amd/graphed.py).
import os as _os
OVERLAP_BRANCHES = _os.environ.get("SGF_OVERLAP", "1") == "1"
OVERLAP_MIN_NODES = 65536
_side_streams = {}


def _side_stream(device):
    s = _side_streams.get(device.index)
    if s is None:
        s = _side_streams[device.index] = torch.cuda.Stream(device=device)
    return s


class _Holder:
    """Where SGFormer.forward keeps the entry copy of a feature tensor when there is no graph view to keep it with."""
    _x_cache = None


import weakref as _weakref
_entry_cache = _weakref.WeakKeyDictionary()


def _stem_w(lin: nn.Linear, x):
    """lin.weight for an input SGFormer.forward zero-padded to whole 16-byte rows (pokec: f = 65 -> 68 / 72; sgf_pad_rows):
    zero weight columns for the padding, differentiable — autograd slices dW back to the parameter's shape."""
    f = lin.weight.shape[1]
    pad = x.shape[1] - f
    # only the entry copy's own padding is accepted (_entry_width): an input with other extra columns raises, as nn.Linear does
    if pad != 0 and x.shape[1] not in ((f + 3) // 4 * 4, (f + 7) // 8 * 8):
        raise RuntimeError(f"input has {x.shape[1]} features, the layer expects {f}")
    return lin.weight if pad == 0 else F.pad(lin.weight, (0, pad))


def _entry_width(f: int, cdt) -> int:
    """Width of the entry copy of f features: rows of whole 16 bytes — 8 bf16 elements (ogbn-products: 100 -> 104, so that
    the stems' weight gradients stream x by LDS-DMA, csrc/gramx.hip) or 4 fp32 elements (pokec: 65 -> 68)."""
    q = 8 if cdt == torch.bfloat16 else 4
    return (f + q - 1) // q * q


def _lin(x, lin: nn.Linear):
    """nn.Linear in the activation dtype: fp32 master weights are cast per call when the model runs
    with bf16 activations (`SGFormer.compute_dtype`).  Forward / dX on the streaming row kernels (bf16 square
    layers, fp32 layers up to 256 wide) or the general matrix-core kernel sgf_gemm, dW / db on sgf_gram (ops.linear)."""
    return ops.linear(x, _stem_w(lin, x), lin.bias)


def _drop(x, p, training, res=None):
    """F.dropout(x, p, training) [+ res]; the fused kernel (no stored mask) when the width allows."""
    if training and p is not None and p > 0.0:
        if x.dim() == 2 and x.shape[1] % 4 == 0:
            return ops.dropout_res(x, res, float(p))
        x = F.dropout(x, p=p, training=True)
        return x if res is None else x + res
    if p is None:
        # large/parse.py:95 leaves --trans_dropout without a default: the reference then crashes
        # inside F.dropout(p=None); surface the same misuse early and clearly.
        raise TypeError("dropout probability is None (pass --trans_dropout / --gnn_dropout)")
    return x if res is None else x + res


def full_attention_conv(qs, ks, vs, output_attn=False, shard=None):
    """medium/ours.py:14-46 / 100M/ours.py:12-53 as a free function: qs, ks [N,H,M], vs [N,H|1,D].

    Returns [N, H, D] like the reference, differentiable for any H: for H > 1 the per-head outputs come from the forward
    kernels' per-head buffer and their gradients enter sgf_attn_bwd_reduce_heads / _apply_heads (TransConvLayer, which takes
    the head mean next as medium/ours.py:98 does, keeps the mean-gradient form).
    """
    n, h, d = qs.shape
    qk = torch.cat([qs.reshape(n, h * d), ks.reshape(n, h * d)], dim=1)
    # H > 1: the per-head outputs [N, H, D], differentiable (sgf_attn_bwd_*_heads take the per-head gradients)
    if vs.shape[1] == h:
        qkv = torch.cat([qk, vs.reshape(n, h * d)], dim=1)
        out = ops.attention(qkv, None, h, d, shard, per_head=h > 1)
    else:
        out = ops.attention(qk, vs.reshape(n, d), h, d, shard, per_head=h > 1)
    out = out.reshape(n, h, d)
    if output_attn:
        return out, _attention_matrix(qs, ks)
    return out


def _attention_matrix(qs, ks):
    """O(N^2) visualisation branch (large/ours.py:152-155); plain PyTorch on purpose."""
    n = qs.shape[0]
    qn = qs / torch.norm(qs, p=2)
    kn = ks / torch.norm(ks, p=2)
    attention = torch.einsum("nhm,lhm->nlh", qn, kn).mean(dim=-1)
    normalizer = (torch.einsum("nhm,hm->nh", qn, kn.sum(dim=0)) + n).mean(dim=-1, keepdim=True)
    return attention / normalizer


def _uses_batch_stats(module: nn.Module, bn: nn.BatchNorm1d) -> bool:
    """nn.BatchNorm1d's rule for normalising with batch (not running) statistics."""
    return module.training or not bn.track_running_stats or bn.running_mean is None


class GraphConvLayer(nn.Module):
    """A_norm X -> [cat x0] -> Linear   (large/ours.py:10-42)."""

    def __init__(self, in_channels, out_channels, use_weight=True, use_init=False):
        super().__init__()
        self.use_init = use_init
        self.use_weight = use_weight
        in_channels_ = 2 * in_channels if use_init else in_channels
        self.W = nn.Linear(in_channels_, out_channels)  # always constructed (kept in state_dict)
        self._shard = _NO_SHARD

    def reset_parameters(self):
        self.W.reset_parameters()

    def propagate(self, x, edge_index):
        """A_norm x (large/ours.py:26-34): the cached CSR of `edge_index` times x."""
        shard = self._shard
        if isinstance(edge_index, ops.CSRGraph):   # SGFormer.forward resolved (and maybe re-ordered) it already
            graph = edge_index
        elif shard is not None and not shard.local_graph:
            graph = shard.graph_for(edge_index)
        else:
            # one GPU — or the batch mode of a sharded run (ShardContext.for_batch): this rank's own induced
            # subgraph, no halo; only the attention / BatchNorm partial sums cross ranks
            graph = ops.graph_cache.get(edge_index, x.shape[0])
        return ops.spmm(graph, x, None if (shard is not None and shard.local_graph) else shard)

    def forward(self, x, edge_index, x0, bn_stats=False):
        """`bn_stats` (not in the reference signature): also return the batch statistics (mean, var, count) of
        the output, which the Linear's streaming pass accumulates for the BatchNorm that follows — (y, stats);
        stats is None when this layer has nothing to fuse them into."""
        shard = self._shard
        y = self.propagate(x, edge_index)
        if self.use_init:
            # W [y | x0] + b without materialising the concatenation
            if bn_stats:
                return ops.linear_bn_stats((y, x0), self.W.weight, self.W.bias, shard)
            y = ops.linear_cat((y, x0), self.W.weight, self.W.bias)
        elif self.use_weight:
            if bn_stats:
                return ops.linear_bn_stats(y, self.W.weight, self.W.bias, shard)
            y = _lin(y, self.W)
        return (y, None) if bn_stats else y


class GraphConv(nn.Module):
    """fc -> BN -> relu -> dropout -> Lg x [conv -> BN -> relu -> dropout -> + layer_[0]]
    (large/ours.py:45-94; the residual always adds the post-stem embedding, :83-93)."""

    def __init__(self, in_channels, hidden_channels, num_layers=2, dropout=0.5, use_bn=True,
                 use_residual=True, use_weight=True, use_init=False, use_act=True):
        super().__init__()
        self.convs = nn.ModuleList()
        self.fcs = nn.ModuleList()
        self.fcs.append(nn.Linear(in_channels, hidden_channels))
        self.bns = nn.ModuleList()
        self.bns.append(nn.BatchNorm1d(hidden_channels))
        for _ in range(num_layers):
            self.convs.append(GraphConvLayer(hidden_channels, hidden_channels, use_weight, use_init))
            self.bns.append(nn.BatchNorm1d(hidden_channels))
        self.dropout = dropout
        self.activation = F.relu
        self.use_bn = use_bn
        self.use_residual = use_residual
        self.use_act = use_act
        self._shard = _NO_SHARD

    def reset_parameters(self):
        for conv in self.convs:
            conv.reset_parameters()
        for bn in self.bns:
            bn.reset_parameters()
        for fc in self.fcs:
            fc.reset_parameters()

    def _bn_act_res(self, bn: nn.BatchNorm1d, x, res, relu, stats=None):
        """[relu](BatchNorm1d(x)) [+ res] with nn.BatchNorm1d's train/eval + running-stat rules.
        `stats`: (mean, var, count) of x if its producer already accumulated them."""
        shard = self._shard
        use_batch = _uses_batch_stats(self, bn)
        if use_batch:
            mean, var, n_tot = stats if stats is not None else ops.batch_stats(x, shard)
            if n_tot <= 1 and self.training:
                raise ValueError("Expected more than 1 value per channel when training")
            if self.training and bn.track_running_stats and bn.running_mean is not None:
                with torch.no_grad():
                    bn.num_batches_tracked += 1
                    m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
                    bn.running_mean.mul_(1.0 - m).add_(mean.to(bn.running_mean.dtype), alpha=m)
                    unbiased = var * (n_tot / max(n_tot - 1.0, 1.0))
                    bn.running_var.mul_(1.0 - m).add_(unbiased.to(bn.running_var.dtype), alpha=m)
        else:
            mean, var, n_tot = bn.running_mean.float(), bn.running_var.float(), float(x.shape[0])
        rstd = torch.rsqrt(var + bn.eps)
        return ops.bn_act_res(x, res, bn.weight, bn.bias, mean, rstd, relu, use_batch, n_tot, shard)

    def _bn_hook(self, bn: nn.BatchNorm1d):
        """nn.BatchNorm1d's bookkeeping as a callback for ops.linear_bn_act_res: hook(None) -> does this call normalise with
        batch statistics?; hook((mean, var, count)) / hook(False) -> (mean, rstd, count, used batch statistics), updating
        the running statistics exactly as _bn_act_res does."""
        def hook(stats):
            use_batch = _uses_batch_stats(self, bn)
            if stats is None:
                return use_batch
            if use_batch and stats[0] == "raw":
                # (sums, shift, count) as the statistics pass left them: mean, rstd and the running update in ONE launch
                _, sums, shift, n_tot = stats
                if n_tot <= 1 and self.training:
                    raise ValueError("Expected more than 1 value per channel when training")
                track = self.training and bn.track_running_stats and bn.running_mean is not None
                if track and bn.momentum is not None and bn.running_mean.dtype == torch.float32:
                    with torch.no_grad():
                        bn.num_batches_tracked += 1
                        mean, rstd = ops.K.bn_finalize(sums, shift, n_tot, bn.eps, float(bn.momentum), bn.running_mean,
                                                       bn.running_var)
                    return mean, rstd, n_tot, True
                if not track:
                    mean, rstd = ops.K.bn_finalize(sums, shift, n_tot, bn.eps, 0.0, None, None)
                    return mean, rstd, n_tot, True
                d = sums.numel() // 2                    # cumulative moving average (momentum None) / non-fp32 buffers
                m1 = sums[:d] / max(n_tot, 1.0)
                stats = ((shift + m1) if shift is not None else m1, (sums[d:] / max(n_tot, 1.0) - m1 * m1).clamp_min_(0.0),
                         n_tot)
            if use_batch:
                mean, var, n_tot = stats
                if n_tot <= 1 and self.training:
                    raise ValueError("Expected more than 1 value per channel when training")
                if self.training and bn.track_running_stats and bn.running_mean is not None:
                    with torch.no_grad():
                        bn.num_batches_tracked += 1
                        m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
                        bn.running_mean.mul_(1.0 - m).add_(mean.to(bn.running_mean.dtype), alpha=m)
                        unbiased = var * (n_tot / max(n_tot - 1.0, 1.0))
                        bn.running_var.mul_(1.0 - m).add_(unbiased.to(bn.running_var.dtype), alpha=m)
            else:
                mean, var, n_tot = bn.running_mean.float(), bn.running_var.float(), 0.0
            return mean, torch.rsqrt(var + bn.eps), n_tot, use_batch
        return hook

    def fused_stem_ok(self, x) -> bool:
        """SGFormer.forward may run this branch's stem Linear + BatchNorm + relu (and TransConv's stem Linear) as ONE
        autograd node (ops.stem_pair_bn) when the layers behind it are the fused ones and no dropout follows the stem."""
        import os
        d = self.fcs[0].out_features
        drop_on = self.training and self.dropout is not None and self.dropout > 0.0
        return (self.use_bn and not drop_on and len(self.convs) > 0 and all(c.use_init for c in self.convs)
                and x.dim() == 2 and x.dtype == torch.bfloat16
                and all(tuple(c.W.weight.shape) == (d, 2 * d) for c in self.convs)
                and hasattr(ops.K, "gcn_bn_bwd_dx_supported") and ops.K.gcn_bn_bwd_dx_supported(d, x.dtype)
                and os.environ.get("SGF_GCN_FUSED", "1") != "0" and os.environ.get("SGF_STEM_FUSED", "1") != "0")

    def _fused_layers(self, x0):
        """The products / pokec / papers100M recipes (use_init, BatchNorm, no active dropout, bf16 rows of 64 / 128 / 256):
        every layer's Linear + BatchNorm + activation + residual is ONE autograd node (ops.linear_bn_act_res)."""
        drop_on = self.training and self.dropout is not None and self.dropout > 0.0
        return (self.use_bn and not drop_on and len(self.convs) > 0 and all(c.use_init for c in self.convs)
                and all(ops.gcn_layer_fused_ok(x0, c.W.weight) for c in self.convs))

    def _stage(self, bn, x, res, relu, stats=None):
        """BN -> act -> dropout -> + res, fused into one pass whenever dropout is inactive."""
        drop_on = self.training and self.dropout is not None and self.dropout > 0.0
        fuse_res = res is not None and not drop_on
        if self.use_bn:
            x = self._bn_act_res(bn, x, res if fuse_res else None, relu, stats)
        else:
            if relu:
                x = torch.relu(x)
            if fuse_res:
                x = x + res
        # dropout active: the residual add of large/ours.py:93 rides along in the dropout kernel
        return _drop(x, self.dropout, self.training, res if (res is not None and not fuse_res) else None)

    def forward(self, x, edge_index, stem=None):
        """edge_index: the int64 [2, nnz] tensor of the reference, or an ops.CSRGraph built from it.
        `stem` (not in the reference signature): (fcs[0](x), its batch statistics or None), when SGFormer.forward
        has computed both branches' first Linear in one pass over x (ops.stem_pair)."""
        ops._require_cuda(x, None if isinstance(edge_index, ops.CSRGraph) else edge_index)
        x_first = None
        if stem is not None and len(stem) == 3:
            # ("bn", x0, x0'): the stem ran as one node with its BatchNorm (ops.stem_pair_bn); x0' feeds the first SpMM
            _, x, x_first = stem
        else:
            if stem is not None:
                x, stats0 = stem
            else:
                x, stats0 = _lin(x, self.fcs[0]), None
            x = self._stage(self.bns[0], x, None, True, stats0)
        if self._fused_layers(x):
            # layer_[0]'s gradient is accumulated inside the layers' backward kernels (ops.GradChain): no fan-out hub
            x0, chain = x, ops.GradChain()
            if x_first is not None:
                x = x_first
            for i, conv in enumerate(self.convs):
                bn = self.bns[i + 1]
                y = conv.propagate(x, edge_index)
                x = ops.linear_bn_act_res(y, x0, conv.W.weight, conv.W.bias, bn.weight, bn.bias, self._bn_hook(bn),
                                          self.use_act, self.use_residual, self._shard, chain, i == 0)
            return x
        # x0 = layer_[0] has up to 2 consumers per layer (the [. | x0] Linear and the residual) plus
        # the first SpMM: hand out aliases through a hub whose backward sums all their gradients in
        # ONE pass (ops.fan_out) instead of autograd's pairwise adds
        uses_init = [c.use_init for c in self.convs]
        k = 1 + sum(uses_init) + (len(self.convs) if self.use_residual else 0)
        hub = list(ops.fan_out(x, k)) if k <= 8 else [x] * k
        x = hub.pop()
        for i, conv in enumerate(self.convs):
            want = self.use_bn and _uses_batch_stats(self, self.bns[i + 1])
            x0 = hub.pop() if uses_init[i] else None
            if want:    # the conv's Linear accumulates the statistics its BatchNorm needs in the same pass
                x, stats = conv(x, edge_index, x0, bn_stats=True)
            else:
                x, stats = conv(x, edge_index, x0), None
            x = self._stage(self.bns[i + 1], x, hub.pop() if self.use_residual else None, self.use_act, stats)
        return x


class TransConvLayer(nn.Module):
    """Q/K/V projections + linear global attention (large/ours.py:96-162)."""

    def __init__(self, in_channels, out_channels, num_heads, use_weight=True):
        super().__init__()
        self.Wk = nn.Linear(in_channels, out_channels * num_heads)
        self.Wq = nn.Linear(in_channels, out_channels * num_heads)
        if use_weight:
            self.Wv = nn.Linear(in_channels, out_channels * num_heads)
        self.out_channels = out_channels
        self.num_heads = num_heads
        self.use_weight = use_weight
        self._shard = _NO_SHARD

    def reset_parameters(self):
        self.Wk.reset_parameters()
        self.Wq.reset_parameters()
        if self.use_weight:
            self.Wv.reset_parameters()

    def _project(self, query_input, source_input):
        """[Q | K (| V)] in ONE buffer: a single [d -> 2Hd or 3Hd] GEMM when query is source."""
        ws = [self.Wq.weight, self.Wk.weight] + ([self.Wv.weight] if self.use_weight else [])
        bs = [self.Wq.bias, self.Wk.bias] + ([self.Wv.bias] if self.use_weight else [])
        if query_input is source_input:
            return ops.linear(query_input, torch.cat(ws, 0), torch.cat(bs, 0))
        q = ops.linear(query_input, ws[0], bs[0])
        kv = ops.linear(source_input, torch.cat(ws[1:], 0), torch.cat(bs[1:], 0))
        return torch.cat([q, kv], 1)

    def forward(self, query_input, source_input, output_attn=False, grad_tap=None):
        """`grad_tap` (not in the reference signature): an ops.grad_tap holder — the caller uses the layer input a
        second time and wants that gradient folded into this layer's last backward kernel."""
        ops._require_cuda(query_input, source_input)
        h, d = self.num_heads, self.out_channels
        if (h == 1 and query_input is source_input and not output_attn and query_input.shape[1] == d
                and d % 4 == 0 and d <= 256):
            # every recipe of the reference: one head, query == source.  Q / K / V are never
            # materialised (ops.attention_from_input; include/sgf.h "attention straight from the
            # un-projected layer input").
            wv, bv = (self.Wv.weight, self.Wv.bias) if self.use_weight else (None, None)
            return ops.attention_from_input(query_input, self.Wq.weight, self.Wq.bias, self.Wk.weight,
                                            self.Wk.bias, wv, bv, self._shard, tap=grad_tap)
        qkv = self._project(query_input, source_input)
        v_ext = None
        if not self.use_weight:
            if source_input.shape[1] != d:
                raise RuntimeError(f"use_weight=False needs in_channels == out_channels ({d})")
            v_ext = source_input
        final_output = ops.attention(qkv, v_ext, h, d, self._shard)
        if output_attn:
            n = qkv.shape[0]
            qs = qkv[:, :h * d].reshape(n, h, d)
            ks = qkv[:, h * d:2 * h * d].reshape(n, h, d)
            return final_output, _attention_matrix(qs, ks)
        return final_output


class TransConv(nn.Module):
    """fc -> LN -> relu -> dropout -> Lt x [attn -> mix residual -> LN -> act -> dropout]
    (large/ours.py:165-219; `bns` are LayerNorms, :174,178).  `alpha` = None is the large
    variant's (x + res)/2; a float is the 100M / medium variant's alpha*x + (1-alpha)*res."""

    def __init__(self, in_channels, hidden_channels, num_layers=2, num_heads=1, dropout=0.5,
                 use_bn=True, use_residual=True, use_weight=True, use_act=True, alpha=None,
                 layer_cls=None):
        super().__init__()
        layer_cls = layer_cls or TransConvLayer
        self.convs = nn.ModuleList()
        self.fcs = nn.ModuleList()
        self.fcs.append(nn.Linear(in_channels, hidden_channels))
        self.bns = nn.ModuleList()
        self.bns.append(nn.LayerNorm(hidden_channels))
        for _ in range(num_layers):
            self.convs.append(layer_cls(hidden_channels, hidden_channels, num_heads=num_heads,
                                        use_weight=use_weight))
            self.bns.append(nn.LayerNorm(hidden_channels))
        self.dropout = dropout
        self.activation = F.relu
        self.use_bn = use_bn
        self.use_residual = use_residual
        self.use_act = use_act
        self.alpha = alpha

    # get_attentions applies the post-layer activation in the large variant only (large/ours.py:234-235);
    # 100M/ours.py:273-290 and medium/ours.py:162-176 never do, whatever use_act says
    _attn_post_act = True

    def reset_parameters(self):
        for conv in self.convs:
            conv.reset_parameters()
        for bn in self.bns:
            bn.reset_parameters()
        for fc in self.fcs:
            fc.reset_parameters()

    def _mix(self):
        return (0.5, 0.5) if self.alpha is None else (float(self.alpha), 1.0 - float(self.alpha))

    def _ln(self, ln: nn.LayerNorm, x, res, a, b, relu):
        if res is None and not self.use_bn and not relu:
            return x
        gamma, beta = (ln.weight, ln.bias) if self.use_bn else (None, None)
        return ops.ln_res_act(x, res, a, b, gamma, beta, relu, ln.eps)

    def _embed(self, x, training, stem=None):
        if isinstance(stem, tuple) and stem[0] == "ln":      # Linear + LayerNorm + relu already applied (ops.stem_pair_bn)
            return _drop(stem[1], self.dropout, training)
        x = _lin(x, self.fcs[0]) if stem is None else stem
        x = self._ln(self.bns[0], x, None, 1.0, 0.0, True)
        return _drop(x, self.dropout, training)

    def forward(self, x, stem=None):
        """`stem` (not in the reference signature): fcs[0](x) when SGFormer.forward already has it (ops.stem_pair)."""
        ops._require_cuda(x)
        x = self._embed(x, self.training, stem)
        layer_ = [x]
        a, b = self._mix()
        for i, conv in enumerate(self.convs):
            if self.use_residual and layer_[i] is x:
                # x feeds the attention AND the residual: the residual's gradient is handed to the attention's last
                # backward kernel instead of being added by autograd in a separate pass (ops.grad_tap)
                holder = {}
                h = conv(x, x, grad_tap=holder)
                x = self._ln(self.bns[i + 1], h, ops.grad_tap(x, holder), a, b, self.use_act)
            elif self.use_residual:
                h = conv(x, x)
                x = self._ln(self.bns[i + 1], h, layer_[i], a, b, self.use_act)
            else:
                h = conv(x, x)
                x = self._ln(self.bns[i + 1], h, None, 1.0, 0.0, self.use_act)
            x = _drop(x, self.dropout, self.training)
            layer_.append(x)
        return x

    def get_attentions(self, x):
        """[Lt, N, N] attention maps (large/ours.py:221-239): no dropout, and — as there — no
        activation after the first layer unless use_act."""
        x = self._embed(x, False)
        layer_, attentions = [x], []
        a, b = self._mix()
        for i, conv in enumerate(self.convs):
            h, attn = conv(x, x, output_attn=True)
            attentions.append(attn)
            res = layer_[i] if self.use_residual else None
            x = self._ln(self.bns[i + 1], h, res, a if res is not None else 1.0,
                         b if res is not None else 0.0, self.use_act and self._attn_post_act)
            layer_.append(x)
        return torch.stack(attentions, dim=0)


class SGFormer(nn.Module):
    """Two branches -> weighted add / concat -> fc   (large/ours.py:242-286)."""

    def __init__(self, in_channels, hidden_channels, out_channels,
                 trans_num_layers=1, trans_num_heads=1, trans_dropout=0.5, trans_use_bn=True,
                 trans_use_residual=True, trans_use_weight=True, trans_use_act=True,
                 gnn_num_layers=1, gnn_dropout=0.5, gnn_use_weight=True, gnn_use_init=False,
                 gnn_use_bn=True, gnn_use_residual=True, gnn_use_act=True,
                 use_graph=True, graph_weight=0.8, aggregate='add', alpha=None, compute_dtype='default',
                 trans_cls=None):
        super().__init__()
        # None: activations in the dtype of the input / parameters (fp32 = the reference's numerics).
        # torch.bfloat16: bf16 activation storage, fp32 master weights and fp32 accumulation in every
        # kernel and GEMM (BASELINE.json config 3); logits are returned in fp32 either way.
        self.compute_dtype = DEFAULT_COMPUTE_DTYPE if isinstance(compute_dtype, str) else compute_dtype
        # dtype of the returned logits; None = the input's.  The fused head computes them in fp32: a caller that stores
        # its features in bf16 and takes the loss in fp32 (bench.py) sets torch.float32 here instead of paying a round trip
        # through bf16 — under sgformer_amd.launch the trainer's features are fp32 and the logits already are.
        self.logits_dtype = None
        self.trans_conv = (trans_cls or TransConv)(
            in_channels, hidden_channels, num_layers=trans_num_layers, num_heads=trans_num_heads,
            dropout=trans_dropout, use_bn=trans_use_bn, use_residual=trans_use_residual,
            use_weight=trans_use_weight, use_act=trans_use_act, alpha=alpha)
        self.graph_conv = GraphConv(in_channels, hidden_channels, gnn_num_layers, gnn_dropout,
                                    gnn_use_bn, gnn_use_residual, gnn_use_weight, gnn_use_init,
                                    gnn_use_act)
        self.use_graph = use_graph
        self.graph_weight = graph_weight
        self.aggregate = aggregate
        self.overlap_branches = OVERLAP_BRANCHES
        if aggregate == 'add':
            self.fc = nn.Linear(hidden_channels, out_channels)
        elif aggregate == 'cat':
            self.fc = nn.Linear(2 * hidden_channels, out_channels)
        else:
            raise ValueError(f'Invalid aggregate type:{aggregate}')
        self.params1 = list(self.trans_conv.parameters())
        self.params2 = list(self.graph_conv.parameters()) if self.graph_conv is not None else []
        self.params2.extend(list(self.fc.parameters()))

    def _forward_on_gpu_from_host(self, x, edge_index):
        """Host tensors + host parameters, no autograd: the reference's CPU evaluation
        (`evaluate_large`, large/eval.py:36-65, moves the model and the whole graph to the CPU and
        large/main-batch.py:157 calls it every eval_step).  There is no CPU implementation here, so
        the same HIP path is run on copies staged to the current GPU and the logits are returned
        on the host — the trainer stays unchanged and nothing is computed on the CPU."""
        if not torch.cuda.is_available():
            ops._require_cuda(x)  # raises the standard no-CPU-path error
        if torch.is_grad_enabled():
            raise RuntimeError("sgformer_amd: CPU tensors with autograd enabled — there is no CPU fallback "
                               "and training runs on MI355X only; move the model and inputs to the GPU")
        dev = torch.device("cuda", torch.cuda.current_device())
        state = {k: v.to(dev) for k, v in list(self.named_parameters()) + list(self.named_buffers())}
        out = torch.func.functional_call(self, state, (x.to(dev), edge_index.to(dev)))
        for k, v in self.named_buffers():  # train-mode BatchNorm under no_grad still tracks stats
            v.copy_(state[k])
        return out.to(x.device)

    def forward(self, x, edge_index):
        if not x.is_cuda and ops.K.name == "hip" and not next(self.parameters()).is_cuda:
            return self._forward_on_gpu_from_host(x, edge_index)
        ops._require_cuda(x, edge_index)
        out_dtype = x.dtype if getattr(self, 'logits_dtype', None) is None else self.logits_dtype
        cdt = self.compute_dtype if self.compute_dtype is not None else x.dtype
        if getattr(edge_index, "_sgf_csr", None) is not None:
            # a mini-batch whose CSR came with its edge list (batching.subgraph): batches of one node count replay ONE captured
            # hipGraph per direction instead of ~150 launches (sgformer_amd/graphed.py; SGF_BATCH_GRAPH=0 turns it off)
            from . import graphed
            out = graphed.maybe_step(self, x, edge_index, cdt, out_dtype)
            if out is not None:
                return out
        # The graph is resolved once per forward.  If its cached view carries a locality-restoring node
        # order (ops.GraphView), the rows of x are permuted here — fused with the storage cast — and the
        # logits un-permuted on the way out; every op in between is permutation-equivariant.
        view = None
        if self.use_graph and self.graph_conv._shard is None and ops.K.name == "hip":
            view = ops.graph_cache.get(edge_index, x.shape[0]).view()
            edge_index = view.graph
        # Node-sharded run on a replicated edge list: partition in sgf_reorder order ACROSS ranks when that makes the
        # cut small enough for the halo exchange (dist.Repartition) — features enter and logits leave through one
        # all-to-all each, the caller keeps its own numbering and partition.
        repart = None
        shard = self.graph_conv._shard if self.use_graph else None
        if shard is not None and not shard.local_edges and not shard.local_graph:
            repart = shard.repartition_for(edge_index)
            if repart is not None:
                x, edge_index = repart.to_new(x), repart.edge_index
        # Entry copy of the features: row permutation (re-ordered graph), storage cast, and zero-padding of a width that is
        # not a multiple of 4 (pokec: 65 -> 68) — ONE pass (sgf_gather_rows / sgf_pad_rows).  Full-graph training hands in
        # the SAME feature tensor every step: the copy is kept, keyed on the tensor's identity and version (the key tensor is
        # pinned so that a recycled data_ptr cannot alias it).
        perm = view.perm if view is not None else None
        f = x.shape[1]
        fp = _entry_width(f, cdt)
        pad = fp != f and not x.requires_grad and hasattr(ops.K, "pad_rows")
        if x.requires_grad:
            if perm is not None:
                x = ops.permute_rows(x, view.perm, view.inv, cdt)
            elif x.dtype != cdt:
                x = x.to(cdt)
        elif perm is not None or pad or x.dtype != cdt:
            # (kept with the graph view, or — without one — in a weak table keyed on the module: NOT an attribute of the
            # module, which `copy.deepcopy(model)` of 100M/nb-sample.py:197 would duplicate together with the features)
            holder = view if view is not None else _entry_cache.setdefault(self, _Holder())
            key = (x.data_ptr(), x._version, tuple(x.shape), x.dtype, cdt, bool(pad), perm is not None)
            hit = getattr(holder, "_x_cache", None)
            if hit is None or hit[0] != key:
                if pad:
                    xe = ops.K.pad_rows(x, perm, fp, cdt)
                elif perm is not None:
                    xe = ops.permute_rows(x, view.perm, view.inv, cdt)
                else:
                    xe = x.to(cdt)
                hit = (key, xe, x)
                holder._x_cache = hit
            x = hit[1]
        return self._core(x, edge_index, view, repart, out_dtype)

    def _entry_copy_uncached(self, x, cdt):
        """The entry copy for ONE use of x (a mini-batch's features): zero-padding to a multiple of 4 columns and the
        storage cast in one pass — the captured step of sgformer_amd.graphed calls this inside its graph."""
        f = x.shape[1]
        fp = _entry_width(f, cdt)
        if fp != f and hasattr(ops.K, "pad_rows"):
            return ops.K.pad_rows(x, None, fp, cdt)
        return x if x.dtype == cdt else x.to(cdt)

    def _core(self, x, edge_index, view, repart, out_dtype):
        """Everything between the entry copy and the logits: stems, both branches, combine + head (large/ours.py:265-276).
        `edge_index`: the reference's tensor, or an object with the CSR arrays (ops.CSRGraph, graphed.StaticCSR)."""
        # K10: the first Linear of both branches reads the same x — one pass, two outputs, the GCN stem's BatchNorm
        # sums on the way (bf16 storage, <= 128 input features)
        stem_t = stem_g = None
        gc, tc = (self.graph_conv if self.use_graph else None), self.trans_conv
        both = gc is not None and hasattr(tc, "fcs") and hasattr(gc, "fcs") and hasattr(gc, "bns")
        wg, wt = (_stem_w(gc.fcs[0], x), _stem_w(tc.fcs[0], x)) if both else (None, None)
        if both and gc.fused_stem_ok(x) and ops.stem_pair_bn_supported(x, wg, wt):
            bn0 = gc.bns[0]
            ln = None
            if getattr(tc, "use_bn", False) and hasattr(tc, "bns") and ops.stem_ln_supported(x, wt):
                # TransConv's stem too: Linear -> LayerNorm -> relu in the same node (its dW comes from sgf_gram_ln_bwd)
                ln = (tc.bns[0].weight, tc.bns[0].bias, tc.bns[0].eps, True)
            x0a, x0b, yt = ops.stem_pair_bn(x, wg, gc.fcs[0].bias, wt, tc.fcs[0].bias,
                                            bn0.weight, bn0.bias, gc._bn_hook(bn0), gc._shard, ln)
            stem_t, stem_g = (("ln", yt) if ln is not None else yt), ("bn", x0a, x0b)
        elif both and ops.stem_pair_supported(x, wg, wt):
            want = gc.use_bn and _uses_batch_stats(gc, gc.bns[0])
            (yg, yt), st = ops.stem_pair(x, wg, gc.fcs[0].bias, wt, tc.fcs[0].bias,
                                         want_stats0=want, shard=gc._shard)
            stem_t, stem_g = yt, (yg, st)
        if (self.use_graph and self.overlap_branches and ops.K.name == "hip" and self.graph_conv._shard is None
                and x.shape[0] >= OVERLAP_MIN_NODES and not torch.cuda.is_current_stream_capturing() and not ops._fused_bwd()):
            # The two branches are independent until the combine: run the attention branch on a side
            # HIP stream so that its latency-bound kernels fill the gaps of the GCN branch (autograd
            # replays each backward node on its forward stream, so the backward overlaps too).
            cur = torch.cuda.current_stream(x.device)
            side = _side_stream(x.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                x1 = self.trans_conv(x) if stem_t is None else self.trans_conv(x, stem=stem_t)
            x.record_stream(side)
            for t in (stem_t if isinstance(stem_t, (tuple, list)) else (stem_t,)):
                if torch.is_tensor(t):
                    t.record_stream(side)
            x2 = self.graph_conv(x, edge_index) if stem_g is None else self.graph_conv(x, edge_index, stem=stem_g)
            cur.wait_stream(side)
            x1.record_stream(cur)
        else:
            x1 = self.trans_conv(x) if stem_t is None else self.trans_conv(x, stem=stem_t)
            if self.use_graph:
                x2 = self.graph_conv(x, edge_index) if stem_g is None else self.graph_conv(x, edge_index, stem=stem_g)
            else:
                x2 = None
        if self.use_graph and self.aggregate == 'add' and ops.combine_fc_supported(x1, self.fc.out_features):
            # gw * x2 + (1 - gw) * x1 -> fc in ONE kernel (large/ours.py:269-270,275): the combined
            # activations are never written, the logits come out in fp32
            gw = float(self.graph_weight)
            if (view is not None and view.perm is not None and repart is None
                    and ops.combine_fc_mapped_supported(x1, self.fc.out_features)):
                # re-ordered graph: the head's stores put the logits back in the caller's row order (and its backward
                # reads the gradient through the same map) — no [N, C] gather pass on either side
                out = ops.combine_fc(x2, x1, self.fc.weight, self.fc.bias, gw, 1.0 - gw, view.perm).to(out_dtype)
                view = None
            else:
                out = ops.combine_fc(x2, x1, self.fc.weight, self.fc.bias, gw, 1.0 - gw).to(out_dtype)
        else:
            if self.use_graph and self.aggregate != 'add':
                # 'cat': fc([x1 | x2]) operand by operand — the [N, 2 d] concatenation is never written
                out = ops.out_linear_cat((x1, x2), self.fc.weight, self.fc.bias).to(out_dtype)
            else:
                if self.use_graph:
                    gw = float(self.graph_weight)
                    x = ops.axpby(x2, x1, gw, 1.0 - gw)
                else:
                    x = x1
                out = ops.out_linear(x, self.fc.weight, self.fc.bias).to(out_dtype)
        if view is not None and view.perm is not None:
            out = ops.permute_rows(out, view.inv, view.perm)       # back to the caller's node order
        if repart is not None:
            out = repart.to_old(out)                               # back to the caller's partition
        return out

    def get_attentions(self, x):
        return self.trans_conv.get_attentions(x)

    def reset_parameters(self):
        # as in the reference (large/ours.py:283-286): `fc` is never reset
        self.trans_conv.reset_parameters()
        if self.use_graph:
            self.graph_conv.reset_parameters()
