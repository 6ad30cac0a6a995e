"""Host -> device staging of a mini-batch OFF the compute stream (row N1 of SURVEY.md §8f; large/main-batch.py:134-146).

The unchanged mini-batch trainer prepares batch i + 1 with lines that block the host on the CURRENT stream:

    x_i = x[idx_i].to(device)                     # idx_i is a HOST tensor: a pageable H2D copy of the index, then the gather
    edge_index_i, _ = subgraph(idx_i, edge_index, ...)        # one device -> host read (the edge count)
    y_i = true_label[idx_i].to(device)            # host gather, pageable H2D copy
    loss = criterion(out_i[train_mask_i], y_i.squeeze(1)[train_mask_i])   # host boolean masks: H2D + nonzero

A blocking copy is ordered behind everything already queued on its stream — batch i's backward and optimizer step — so the
host waits for the GPU before it can issue anything of batch i + 1, and the GPU then waits for the host: their times ADD
(profiles/r05_minibatch_sections.json: 2.0 ms of host issue in a 5.8 ms batch).  None of these copies depends on batch i's
compute.  Here they run on a per-device PREP stream: the host blocks only until the few microseconds of prep work are done,
the compute stream picks the results up through an event, and batch i + 1's preparation overlaps batch i's backward.

Installed by sgformer_amd.launch for `main-batch.py` (`patch_resident_features`; `SGF_PREP_STREAM=0` keeps every copy on
the current stream).  Two thin tensor wrappers carry it — every operation they do not cover returns a plain tensor and
behaves exactly as before:

    ResidentRows  a DEVICE tensor (the resident features, a batch's labels): `t[host index or host boolean mask]` gathers its
                  rows on the prep stream
    StagedHost    a HOST tensor (the dataset's labels, stays on the host for everything the trainer does with it there):
                  rows picked from it keep the type, and `.to(cuda device)` of such rows copies on the prep stream
"""
from __future__ import annotations

import os

import torch

_streams = {}


def enabled() -> bool:
    return os.environ.get("SGF_PREP_STREAM", "1") != "0" and torch.cuda.is_available()


def prep_stream(device) -> "torch.cuda.Stream":
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    s = _streams.get(idx)
    if s is None:
        s = _streams[idx] = torch.cuda.Stream(device=idx)
    return s


class on_prep:
    """`with on_prep(device) as hand_over: ...` runs the block on the prep stream; on exit the stream that was current waits
    for it (an event, no host wait).  `hand_over(t, ...)` marks tensors allocated inside as used by that stream."""

    def __init__(self, device):
        self.device = torch.device(device)

    def __enter__(self):
        self.cur = torch.cuda.current_stream(self.device)
        self.prep = prep_stream(self.device)
        self.ctx = torch.cuda.stream(self.prep)
        self.ctx.__enter__()
        return self.hand_over

    def hand_over(self, *tensors):
        for t in tensors:
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(self.cur)

    def __exit__(self, *exc):
        self.ctx.__exit__(*exc)
        self.cur.wait_stream(self.prep)
        return False


def h2d(t: torch.Tensor, device) -> torch.Tensor:
    """`t.to(device)` for a host tensor, copied on the prep stream (the host waits for THAT stream only)."""
    if t.is_cuda or not enabled() or torch.cuda.is_current_stream_capturing():
        return t.to(device)
    with on_prep(device) as hand_over:
        out = t.to(device)
        hand_over(out)
    return out


def _plain(t):
    return t.as_subclass(torch.Tensor) if isinstance(t, (ResidentRows, StagedHost)) else t


def _rows_via_prep(x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """x[idx] for a device x and a HOST 1-D index / boolean mask (what ATen computes: index_select along dim 0)."""
    if idx.dtype == torch.bool:
        idx = idx.nonzero().view(-1)                     # on the host, where the mask lives
    with on_prep(x.device) as hand_over:
        out = _plain(x).index_select(0, idx.to(x.device))
        hand_over(out)
    return out


_KEEP = None


def _keep_funcs():
    """operations whose result is the same rows seen differently: the wrapper type survives them"""
    global _KEEP
    if _KEEP is None:
        t = torch.Tensor
        _KEEP = {t.unsqueeze, t.squeeze, t.view, t.reshape, t.contiguous, t.detach, t.flatten}
    return _KEEP


class ResidentRows(torch.Tensor):
    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if (func is torch.Tensor.__getitem__ and len(args) == 2 and isinstance(args[0], ResidentRows) and args[0].is_cuda
                and torch.is_tensor(args[1]) and not args[1].is_cuda and args[1].dim() == 1
                and args[1].dtype in (torch.long, torch.bool) and enabled() and not torch.cuda.is_current_stream_capturing()
                and (args[1].dtype == torch.long or args[1].shape[0] == args[0].shape[0])):
            return _rows_via_prep(args[0], _plain(args[1]))
        with torch._C.DisableTorchFunctionSubclass():
            out = func(*args, **kwargs)
        if func in _keep_funcs() and torch.is_tensor(out) and out.is_cuda:
            return out.as_subclass(ResidentRows)
        return _plain(out) if torch.is_tensor(out) else out


def _cuda_target(args, kwargs):
    """the CUDA device a Tensor.to(...) call moves to, when that is ALL it does (no dtype / layout change)"""
    if any(k in kwargs for k in ("dtype", "memory_format", "other")):
        return None
    dev = kwargs.get("device")
    for a in args[1:]:
        if isinstance(a, (torch.dtype, torch.Tensor)):
            return None
        if isinstance(a, (torch.device, str, int)) and not isinstance(a, bool):
            dev = a
    if dev is None:
        return None
    try:
        dev = torch.device("cuda", dev) if isinstance(dev, int) else torch.device(dev)
    except (RuntimeError, TypeError):
        return None
    return dev if dev.type == "cuda" else None


class StagedHost(torch.Tensor):
    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if (func is torch.Tensor.to and args and isinstance(args[0], StagedHost) and not args[0].is_cuda and enabled()
                and not torch.cuda.is_current_stream_capturing()):
            dev = _cuda_target(args, kwargs)
            if dev is not None:
                return h2d(_plain(args[0]), dev).as_subclass(ResidentRows)
        with torch._C.DisableTorchFunctionSubclass():
            out = func(*args, **kwargs)
        if (func is torch.Tensor.__getitem__ or func in _keep_funcs()) and torch.is_tensor(out) and not out.is_cuda:
            return out.as_subclass(StagedHost)
        return _plain(out) if torch.is_tensor(out) else out


def resident(t: torch.Tensor) -> torch.Tensor:
    """The device tensor t as ResidentRows (same storage)."""
    return t.as_subclass(ResidentRows) if t.is_cuda else t


def staged(t: torch.Tensor) -> torch.Tensor:
    """The host tensor t as StagedHost (same storage)."""
    return t if t.is_cuda else t.as_subclass(StagedHost)
