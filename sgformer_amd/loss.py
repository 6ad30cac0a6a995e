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
