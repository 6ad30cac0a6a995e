"""Row N4 of SURVEY.md §8f: the loss of the reference's trainers on the GPU in one pass.

    from sgformer_amd.loss import log_softmax_nll
    loss = log_softmax_nll(out, dataset.label, train_idx)

is the arithmetic of large/main.py:139-141
(`out = F.log_softmax(out, dim=1); loss = criterion(out[train_idx], label.squeeze(1)[train_idx])`
with `criterion = nn.NLLLoss()`) on sgf_nll_fwd / sgf_nll_bwd.  Optional: the unchanged trainers keep
their own three lines (5 ATen kernels; 4.2 ms of `nll_loss` kernels per step at ogbn-products
scale); a maintainer who edits those lines gets the fused form.  `bench.py` times the step with it.
"""
from . import ops


def log_softmax_nll(out, label, train_idx, denom=None):
    """`label` may be [N] or [N, 1] (the trainers keep [N, 1], large/main.py:48-50)."""
    return ops.nll_loss_rows(out, label, train_idx, denom)


def gather_nll(input, target, ignore_index=-100):
    """mean_j -input[j, target[j]] over the rows whose target is not `ignore_index` — F.nll_loss(input, target) for
    2-D log-probabilities, class-index targets, no class weights, reduction 'mean', written as a gather + a masked sum.
    ATen's nll_loss kernels reduce in one block: 2.5 ms forward + 1.7 ms backward for the 1.2 M training rows of an
    ogbn-products step, against ~0.3 ms here; autograd differentiates it (gather -> scatter).  sgformer_amd.launch
    installs it behind torch.nn.functional.nll_loss for the unchanged trainers (large/main.py:139-141 keeps its three
    lines); everything this fast path does not cover goes to the original."""
    import torch
    valid = target != ignore_index
    safe = torch.where(valid, target, torch.zeros_like(target))
    picked = input.gather(1, safe.unsqueeze(1)).squeeze(1)
    w = valid.to(picked.dtype)
    return -(picked * w).sum() / w.sum()


# ------------------------------------------------------------------------------------------------
# The trainers' three loss lines AS WRITTEN on the one-pass kernels (installed by sgformer_amd.launch)
#     out = F.log_softmax(out, dim=1)
#     loss = criterion(out[train_idx], label.squeeze(1)[train_idx])          # criterion = nn.NLLLoss()
# F.log_softmax returns a LAZY tensor; indexing it with a 1-D index tensor gives lazy rows; F.nll_loss on lazy rows runs
# sgf_nll_fwd / sgf_nll_bwd on the logits.  Every OTHER use of either object computes the real log-softmax first (ATen) and
# goes on with an ordinary tensor — results are the reference's in every case, only the common case is one pass.
# ------------------------------------------------------------------------------------------------
import torch as _torch


def _materialise(x):
    if isinstance(x, LazyLogSoftmax):
        return x._sgf_value()
    if isinstance(x, (list, tuple)):
        return type(x)(_materialise(v) for v in x)
    if isinstance(x, dict):
        return {k: _materialise(v) for k, v in x.items()}
    return x


def _meta_funcs():
    t = _torch.Tensor
    out = {t.size, t.dim, t.numel, t.is_floating_point, t.stride, t.element_size, t.nelement, t.ndimension, t.is_contiguous}
    for name in ("shape", "dtype", "device", "ndim", "is_cuda", "requires_grad", "layout", "is_sparse", "is_quantized",
                 "is_meta", "names", "is_leaf"):
        prop = getattr(t, name, None)
        if prop is not None and hasattr(prop, "__get__"):
            out.add(prop.__get__)
    return out


_META = _meta_funcs()


_UNIQUE_IDX = {}


def _idx_is_unique(idx) -> bool:
    """Does the 1-D row index hold every row at most once (and no negative, wrapping entries)?  The one-pass kernels STORE a
    row's gradient (sgf_nll_bwd) and keep one label per node, so `out[idx]` with a repeated row must take ATen's path, which
    accumulates (idx = [1, 3, 3, 7] gave the right loss and a wrong gradient).  One device reduction + host read
    per index TENSOR, cached on its identity and version — a full-graph trainer indexes with the same train_idx every step."""
    key = (idx.data_ptr(), idx._version, idx.numel(), str(idx.device))
    hit = _UNIQUE_IDX.get(key)
    if hit is None:
        n = idx.numel()
        ok = n == 0 or (int(idx.min()) >= 0 and int(_torch.unique(idx).numel()) == n)
        if len(_UNIQUE_IDX) >= 16:
            _UNIQUE_IDX.pop(next(iter(_UNIQUE_IDX)))
        _UNIQUE_IDX[key] = hit = (bool(ok), idx)          # (the tensor is pinned: a recycled data_ptr cannot alias the key)
    return hit[0]


class LazyLogSoftmax(_torch.Tensor):
    """log_softmax(logits, dim=1) that has not been computed yet (or the rows `idx` of it)."""

    @staticmethod
    def __new__(cls, logits, idx=None, orig=None):
        shape = logits.shape if idx is None else (idx.shape[0], logits.shape[1])
        r = _torch.Tensor._make_wrapper_subclass(cls, shape, dtype=logits.dtype, device=logits.device,
                                                 requires_grad=bool(logits.requires_grad))
        r._sgf_logits, r._sgf_idx, r._sgf_orig, r._sgf_cache = logits, idx, orig, None
        return r

    def _sgf_value(self):
        if self._sgf_cache is None:
            full = (self._sgf_orig or _torch.nn.functional.log_softmax)(self._sgf_logits, dim=1)
            self._sgf_cache = full if self._sgf_idx is None else full[self._sgf_idx]
        return self._sgf_cache

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if (func is _torch.Tensor.__getitem__ and len(args) == 2 and isinstance(args[0], LazyLogSoftmax)
                and args[0]._sgf_idx is None and args[0]._sgf_cache is None and _torch.is_tensor(args[1])
                and not isinstance(args[1], LazyLogSoftmax) and args[1].dim() == 1 and args[1].dtype == _torch.long
                and args[1].device == args[0]._sgf_logits.device and _idx_is_unique(args[1])):
            return LazyLogSoftmax(args[0]._sgf_logits, args[1], args[0]._sgf_orig)
        if (func is _torch.Tensor.__getitem__ and len(args) == 2 and isinstance(args[0], LazyLogSoftmax)
                and args[0]._sgf_idx is None and args[0]._sgf_cache is None and _torch.is_tensor(args[1])
                and not isinstance(args[1], LazyLogSoftmax) and args[1].dim() == 1 and args[1].dtype == _torch.bool
                and args[1].shape[0] == args[0].shape[0]):
            # `out_i[train_mask_i]` of the mini-batch trainer (large/main-batch.py:146; the mask lives on the HOST there): the
            # rows of a boolean mask are its nonzero positions — ascending, each once — so the lazy rows need no uniqueness
            # check.  nonzero() runs where the mask is (ATen's own indexing does the same), only the positions cross over.
            idx = args[1].nonzero().view(-1)
            if idx.numel() > 0:                 # (an empty selection: ATen's path and its nan)
                from .staging import h2d        # (a host mask's positions cross over on the prep stream: no wait for the forward)
                return LazyLogSoftmax(args[0]._sgf_logits, h2d(idx, args[0]._sgf_logits.device), args[0]._sgf_orig)
        with _torch._C.DisableTorchFunctionSubclass():
            if func in _META:                  # shape / dtype / device ... live on the wrapper: nothing is computed for them
                return func(*args, **kwargs)
            return func(*_materialise(args), **_materialise(kwargs))


    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):        # (whatever reaches the dispatcher: real values)
        return func(*_materialise(args), **_materialise(kwargs or {}))


def lazy_rows_nll(rows: "LazyLogSoftmax", target, ignore_index=-100):
    """F.nll_loss(rows, target) (mean over the targets != ignore_index) for lazy rows: one pass over the logits' rows
    `idx` (sgf_nll_fwd), gradient written for all N rows (sgf_nll_bwd).  The divisor stays on the device."""
    logits, idx = rows._sgf_logits, rows._sgf_idx
    n = logits.shape[0]
    labels = _torch.full((n,), -1, dtype=_torch.long, device=logits.device)
    labels[idx] = target                    # the kernels index labels by NODE id; ignore_index (< 0) adds nothing there
    denom = (target != ignore_index).sum().to(_torch.float32)          # (all targets ignored: 0 / 0 = nan, as ATen)
    return ops.nll_loss_rows(logits, labels, idx, 1.0) / denom
