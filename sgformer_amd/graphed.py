"""Mini-batch steps as hipGraph replays (row N1 of SURVEY.md §8f).

The reference's mini-batch trainer (large/main-batch.py:129-151) calls `model(x_i, edge_index_i)` on ~100 k-node induced
subgraphs.  At that size a training step is ~150 kernel launches of 5-80 us each: the host's launch rate and the launch gaps
on the device, not bytes, bound it (profiles/r05_minibatch_sections.json: 1.0 + 1.4 ms of host time to ISSUE forward and
backward next to 1.1 + 2.4 ms of device time).  Batches of one node count have identical launch sequences — only the CSR
arrays and the features differ — so the step is captured ONCE per node count with HIP stream capture
(`torch.cuda.make_graphed_callables`: one graph for the forward, one for the backward, both over the kernels of libsgf.so
exactly as the eager path issues them) and replayed for every later batch of that size:

  * the batch's CSR (built with its edge list by batching.subgraph / sgf_subgraph_csr_*) is copied into FIXED-CAPACITY device
    arrays (`StaticCSR`) the captured kernels read; the SpMM kernels bound their work by rowptr, so the stale tail of
    colind / val beyond the batch's nnz is never touched;
  * same arithmetic, same kernels, same order as the eager step: results are bit-identical (tests/test_gpu_graphed.py);
  * engaged only where a replay is known to be equivalent: training mode with autograd, no active dropout (its Philox
    offset is a launch argument), one GPU, a symmetric batch graph (batching.subgraph's promise for a symmetric parent: the
    backward multiplies with the same arrays), features that do not require a gradient.  Everything else — evaluation, the
    first batch of a size (it warms the caches the capture must not contain), odd shapes — runs the eager path;
  * a replay reads its batch from a PRIVATE static input (a copy of the capture batch's features, never the caller's
    tensor), and runs only while no parameter carries a gradient (a replayed backward hands out static gradient buffers that
    autograd adopts as `p.grad`: gradient accumulation over several batches stays on the eager path);
  * a replay writes into the graphs' static buffers: the logits are handed out as a copy, and while a replayed forward's
    backward is still outstanding (its logits alive, no gradient yet — two forwards before one backward) the next call of
    that size runs eager; at most 4 batch sizes per model stay captured (least recently used ones give their pools back);
  * the captured kernels hold the PARAMETERS' device addresses: `evaluate_large` moves the model to the host and back every
    `eval_step` epochs (large/eval.py:41, large/main-batch.py:131), so every replay first compares the parameters' addresses
    with the captured ones and re-captures when they moved.

SGF_BATCH_GRAPH=0 turns the whole mechanism off.  Nothing here runs on the full-graph path.
"""
from __future__ import annotations

import gc
import os
import weakref
from collections import OrderedDict

import torch
import torch.nn as nn

from . import ops

_F32 = torch.float32


# Captured steps that are no longer used — a released entry (re-capture, least-recently-used size), a model that went away.
# Their hipGraphs are NOT destroyed where Python happens to drop the last reference: on ROCm, destroying a graph (and freeing
# its private pool) while a stream is capturing, or while another captured step is still executing, aborts the process, and
# the cyclic collector runs at arbitrary allocations (seen as `Fatal Python error: Aborted` inside the NEXT batch's
# `subset.to(device)`).  They wait here and are destroyed at a safe point: `collect()` — device idle, nothing capturing.
_graveyard = []


def collect():
    """Destroy the captured steps nobody uses any more, with the device idle (called before every capture; a trainer that
    wants the memory back earlier may call it between epochs)."""
    if _graveyard and not torch.cuda.is_current_stream_capturing():
        torch.cuda.synchronize()
        _graveyard.clear()
        gc.collect()


class _PerModel(OrderedDict):
    """{(n, f, x dtype, compute dtype, logits dtype, device): _Entry} of ONE model, kept in the module's __dict__ (so that it dies with
    the model: an entry holds the model through its captured callable, a cycle the garbage collector resolves — a global table
    keyed on the model would pin it).  Copies and pickles of the model start without captures (100M/nb-sample.py:197
    deep-copies its best model; hipGraphs can be neither copied nor pickled)."""

    def __deepcopy__(self, memo):
        return _PerModel()

    def __del__(self):                       # the model went away: its captures wait in the graveyard (see above)
        try:
            for e in self.values():
                e.release()
        except Exception:
            pass

    def __reduce__(self):
        return (_PerModel, ())


_MIN_NODES = 4096          # below this a step is too small to matter; keep the eager path
_SEEN_BEFORE_CAPTURE = 1   # eager batches of a size before its capture (the first one warms caches)
_MAX_CAPTURED = 4          # captured batch sizes kept per model (each holds the activations of one step in its own pool)
counters = {"captures": 0, "replays": 0}      # process-wide, for tests and the bench line


def enabled() -> bool:
    return os.environ.get("SGF_BATCH_GRAPH", "1") != "0" and ops.K.name == "hip" and torch.cuda.is_available()


class StaticCSR(ops.CSRGraph):
    """Fixed-capacity CSR arrays behind ops.CSRGraph's interface (what GraphConv.propagate / ops.spmm_on read): the captured
    SpMM launches hold THESE addresses; `load` puts a batch's arrays there.  A^T == A (the caller checked)."""

    def __init__(self, n: int, cap: int, device, long_segments: int):          # (not CSRGraph.__init__: nothing is built)
        self.n, self.nnz, self.cap, self.device = n, 0, cap, device
        self.edge_index = None
        self.rowptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
        self.colind = torch.zeros(cap, dtype=torch.int32, device=device)
        self.val = torch.zeros(cap, dtype=_F32, device=device)
        self.deg = None
        self.long_segments = self.t_long_segments = int(long_segments)
        self.symmetric = True
        self._t = (self.rowptr, self.colind, self.val)

    def load(self, rowptr, colind, val):
        nnz = self.nnz = int(colind.numel())
        self.rowptr.copy_(rowptr)
        self.colind[:nnz].copy_(colind)
        self.val[:nnz].copy_(val)

    def view(self, now: bool = False):
        return ops.GraphView(self, None, None)


class _Core(nn.Module):
    """x -> logits of `model` on the static graph: what gets captured."""

    def __init__(self, model, graph, cdt, out_dtype):
        super().__init__()
        self.model, self.cdt, self.out_dtype = model, cdt, out_dtype
        self._graph = graph

    def forward(self, x):
        m = self.model
        return m._core(m._entry_copy_uncached(x, self.cdt), self._graph, None, None, self.out_dtype)


class _Entry:
    def __init__(self):
        self.seen = 0
        self.core = None
        self.graph = None
        self.params = ()
        self.param_ptrs = None
        self.failed = False
        self.scratch = None        # the workspaces of the captured launches (kernels.begin_capture_scope)
        self.pending = None        # weakref to the last replay's backward hook until that backward has run

    def busy(self) -> bool:
        """A replayed forward whose backward has not run yet while its autograd graph is still alive: its saved activations
        live in the graphs' static buffers, which another replay would overwrite (two forwards before one backward)."""
        return self.pending is not None and self.pending() is not None

    def release(self):
        if self.core is not None:
            _graveyard.append((self.core, self.graph, self.scratch))           # destroyed by collect(), not here
        self.core = self.graph = self.pending = self.scratch = None
        self.params, self.param_ptrs = (), None


def _eligible(model, x, edge_index) -> bool:
    if not (enabled() and model.training and torch.is_grad_enabled() and x.is_cuda and not x.requires_grad):
        return False
    if not getattr(edge_index, "_sgf_symmetric", False) or x.shape[0] < _MIN_NODES:
        return False
    csr = edge_index._sgf_csr
    if csr[0].numel() != x.shape[0] + 1 or csr[1].numel() != edge_index.shape[1]:
        return False          # (not the CSR of THIS call's graph: the eager path builds its own, as ops.CSRGraph decides too)
    if not model.use_graph or model.graph_conv._shard is not None:
        return False          # (SGFormer._core keeps ONE stream while capturing: two-stream branches inside a capture do not
                              #  survive hipStreamEndCapture, r05)
    for branch in (model.trans_conv, model.graph_conv):
        p = getattr(branch, "dropout", 1.0)        # (a branch without the attribute: unknown, stay eager)
        if p is not None and p > 0.0:
            return False
    return True


def _long_bound(edge_index, cap: int) -> int:
    """`long_segments` of the captured SpMM launches.  The long-row path is an optimisation (the row kernel is correct for any
    row length): a capture without it stays right for a later batch that does have a long row, a capture with it sizes its
    queue for the worst case at capacity (the bound of ops.long_row_segments)."""
    hint = getattr(edge_index, "_sgf_max_in_degree", None)
    if hint is not None and int(hint) <= ops.LONG_ROW:
        return 0
    return cap // ops._SEGMENT + cap // (ops.LONG_ROW + 1) + 1


def _capture(model, entry: _Entry, x, edge_index, cdt, out_dtype):
    n, dev = x.shape[0], x.device
    nnz = int(edge_index.shape[1])
    cap = max(1024, int(nnz * 1.25) + 4096)
    graph = StaticCSR(n, cap, dev, _long_bound(edge_index, cap))
    graph.load(*edge_index._sgf_csr[:3])
    core = _Core(model, graph, cdt, out_dtype)
    core.train(model.training)
    names, params = zip(*core.named_parameters())
    # The capture differentiates with respect to fresh leaf ALIASES of the parameters (same storage), created here: autograd
    # ties a leaf's gradient accumulator to the stream it was first used on, and the trainer's previous eager step (its
    # `out` / `loss` are still alive when the next forward runs) keeps the parameters' own accumulators on the DEFAULT
    # stream — differentiating through those inside the capture makes the engine order the default stream after the
    # capturing one, which HIP cannot end a capture with (hipStreamEndCapture crashes).  For the same reason no warm-up
    # iterations on a side stream: the eager step(s) before this call were the warm-up, on these very shapes.
    aliases = tuple(p.detach().requires_grad_(p.requires_grad) for p in params)

    def step(x, *ps):
        return torch.func.functional_call(core, dict(zip(names, ps)), (x,))

    saved = [(b, b.detach().clone()) for b in model.buffers()]
    # Dead captures (a released entry, a model that went away) are reference cycles: the cyclic collector destroys their
    # hipGraphs at some later allocation — inside THIS capture, if it gets the chance, and destroying a graph / freeing its pool
    # while a stream is capturing crashes.  Collect them now, and keep the collector off until the capture has ended.
    collect()
    gc.collect()
    gc_was_on = gc.isenabled()
    gc.disable()
    from . import kernels as _kernels
    entry.scratch = _kernels.begin_capture_scope()       # the captured launches' workspaces live (and die) with this entry
    try:
        # the sample input becomes the graphs' STATIC input, into which every later replay copies its batch: a private
        # buffer, not the caller's tensor (a trainer that keeps same-sized device batches across epochs would otherwise find
        # its capture batch overwritten by the last batch's features)
        x_static = x.detach().clone()
        fn = torch.cuda.make_graphed_callables(step, (x_static,) + aliases, num_warmup_iters=0, allow_unused_input=True)
    finally:
        _kernels.end_capture_scope()
        if gc_was_on:
            gc.enable()
        with torch.no_grad():           # (captured launches do not execute; kept in case a torch version warms up anyway)
            for b, v in saved:
                b.copy_(v)
    entry.core, entry.graph, entry.params = fn, graph, params
    entry.param_ptrs = tuple((p.data_ptr(), p.requires_grad) for p in params)
    counters["captures"] += 1


def maybe_step(model, x, edge_index, cdt, out_dtype):
    """The logits of model(x, edge_index) from a captured step, or None (the caller runs the eager path)."""
    if not _eligible(model, x, edge_index):
        return None
    per_model = model.__dict__.get("_sgf_graphed")
    if per_model is None:
        per_model = model.__dict__["_sgf_graphed"] = _PerModel()
    key = (int(x.shape[0]), int(x.shape[1]), x.dtype, cdt, out_dtype, x.device)
    entry = per_model.get(key)
    if entry is None:
        entry = per_model[key] = _Entry()
    per_model.move_to_end(key)
    if entry.failed:
        return None
    entry.seen += 1
    if entry.seen <= _SEEN_BEFORE_CAPTURE or entry.busy():
        return None
    # A replayed backward hands out the graphs' STATIC gradient buffers, and autograd's AccumulateGrad adopts a gradient it
    # solely owns as `p.grad`: with a gradient still in place (accumulation over several batches, zero_grad(set_to_none=False))
    # the next replay would overwrite it before adding to it.  The reference trainers clear their gradients every batch
    # (large/main-batch.py:142); anything else keeps the eager path.
    if any(p.grad is not None for p in model.parameters()):
        return None
    csr = edge_index._sgf_csr
    nnz = int(csr[1].numel())
    # captured launches hold the parameters' addresses and which of them get a gradient: moved (the trainer's evaluation
    # round trip), replaced or (un)frozen parameters mean a new capture
    stale = entry.core is not None and (nnz > entry.graph.cap or any(a is not b for a, b in zip(entry.params, model.parameters()))
                                        or entry.param_ptrs != tuple((p.data_ptr(), p.requires_grad) for p in entry.params))
    if entry.core is None or stale:
        entry.release()                          # (frees the old graphs' pool before the new capture)
        held = [e for e in per_model.values() if e.core is not None]
        for e in held[:max(0, len(held) - (_MAX_CAPTURED - 1))]:
            if not e.busy():
                e.release()                      # least recently used sizes give their pools back; they re-capture if they return
                e.seen = 0
        try:
            _capture(model, entry, x, edge_index, cdt, out_dtype)
        except Exception as exc:                 # capture is an optimisation: never let it break a training run
            entry.failed = True
            import warnings
            warnings.warn(f"sgformer_amd.graphed: capture failed, keeping the eager path for batches of {key[0]} nodes: {exc!r}")
            return None
    entry.graph.load(*csr[:3])
    counters["replays"] += 1
    # the graphs' output is a STATIC buffer the next replay overwrites: hand out a copy (19 MB at 100 k x 47: ~10 us), so that
    # logits a caller keeps across batches stay what they were
    out = entry.core(x, *entry.params).clone()
    if not out.requires_grad:               # (every parameter frozen: no backward will come)
        return out
    done = _Done(entry)
    out.register_hook(done)                 # (kept by the autograd node of `out`: lives exactly as long as a backward through
    entry.pending = weakref.ref(done)       #  this replay is still possible, also after `out` itself is gone)
    return out


class _Done:
    """Tensor hook on the replayed logits: their gradient arrived, so this backward call runs the captured backward."""

    def __init__(self, entry):
        self.entry = weakref.ref(entry)

    def __call__(self, grad):
        e = self.entry()
        if e is not None:
            e.pending = None
        return None
