"""Run one of the reference's trainers UNCHANGED on top of this package.

    python -m sgformer_amd.launch /path/to/SGFormer/large/main.py --dataset ogbn-arxiv --method sgformer ...
    python -m sgformer_amd.launch /path/to/SGFormer/large/main-batch.py ...
    python -m sgformer_amd.launch /path/to/SGFormer/100M/nb-sample.py ...

Why a launcher: the reference has no plugin interface; its model is whatever the name `ours`
resolves to (`from ours import *`, large/parse.py:2, 100M/parse.py:1).  CPython puts the trainer's
own directory first on sys.path, so PYTHONPATH cannot displace large/ours.py — but `import` looks
in sys.modules before sys.path.  This launcher registers the drop-in module as sys.modules['ours']
and then executes the trainer file byte-for-byte with runpy.

The variant (which constructor signature the trainer's parse.py expects) is taken from the
trainer's directory name — `large` -> sgformer_amd.ours, `100M` -> sgformer_amd.ours_100m,
`medium` -> sgformer_amd.ours_medium (which also swaps `models.GCN`, the injected GNN branch, for
the libsgf one) — or from --sgf-variant.  `--sgf-dtype bf16` switches every SGFormer the trainer builds to bf16
activation storage (fp32 master weights / accumulation); the default is the reference's fp32.
For `main-batch.py` the per-batch `torch_geometric.utils.subgraph` call is served by the GPU
implementation in sgformer_amd.batching (`--sgf-host-subgraph 1` keeps PyG's host version).  The trainers' own
`nn.NLLLoss()` runs as a gather + masked sum instead of ATen's one-block reduction (`--sgf-aten-loss 1` keeps ATen's).

What the launcher rewires besides `ours`, and how to turn each off (every patch reaches the original for any call it does not
cover; tests/test_launch_patches.py drives each with callers that are not the reference's trainers):

    F.log_softmax / F.nll_loss          lazy log-softmax + one-pass loss on the training rows   --sgf-aten-loss 1
    torch.optim.Adam.__init__           fused=True for all-CUDA float parameters                SGF_FUSED_ADAM=0
    torch_geometric.utils.subgraph      device implementation (main-batch.py only)              --sgf-host-subgraph 1
    torch_geometric.utils.to_undirected / remove_self_loops / add_self_loops                    --sgf-host-prologue 1
    torch_geometric.loader.NeighborLoader (100M only)                                           --sgf-host-sampler 1
    dataset.load_dataset (node features resident on the GPU, main-batch.py only)                --sgf-host-features 1
    torch.set_num_threads(4)  (main-batch.py only)                                              OMP_NUM_THREADS=...

    --sgf-patches minimal  (or SGF_PATCHES=minimal)   none of the above: the module drop-in only.
"""
from __future__ import annotations

import importlib
import os
import runpy
import sys

VARIANTS = {"large": "sgformer_amd.ours", "100M": "sgformer_amd.ours_100m",
            "100m": "sgformer_amd.ours_100m", "medium": "sgformer_amd.ours_medium"}


def install(variant: str = "large", dtype: str | None = None):
    """Register the drop-in as sys.modules['ours']; returns the module."""
    if variant not in VARIANTS:
        raise SystemExit(f"sgformer_amd.launch: unknown variant {variant!r} (choose from {sorted(set(VARIANTS))})")
    mod = importlib.import_module(VARIANTS[variant])
    if dtype in ("bf16", "bfloat16") and variant == "medium":
        # ours_medium / difformer never read DEFAULT_COMPUTE_DTYPE: refuse rather than silently run fp32
        raise SystemExit("sgformer_amd.launch: --sgf-dtype bf16 is implemented for the large and 100M variants "
                         "only (BASELINE.json config 3); the medium trainers run in fp32")
    if dtype in ("bf16", "bfloat16"):
        import torch
        importlib.import_module("sgformer_amd.ours").DEFAULT_COMPUTE_DTYPE = torch.bfloat16
    elif dtype not in (None, "f32", "fp32", "float32"):
        raise SystemExit(f"sgformer_amd.launch: unknown --sgf-dtype {dtype!r}")
    sys.modules["ours"] = mod
    if variant == "medium":   # medium/parse.py:4 `from difformer import *` (--method difformer)
        sys.modules["difformer"] = importlib.import_module("sgformer_amd.difformer")
    return mod


def patch_medium_gcn():
    """medium/parse.py:99 builds the GNN branch from the reference's own `models.GCN` (PyG GCNConv).
    Import the trainer's `models` module (it must be importable: the trainer directory is on
    sys.path by now) and point its `GCN` at the libsgf one, so `from models import *` in parse.py
    picks it up — same class name, constructor and state_dict keys."""
    import importlib as _il
    models = _il.import_module("models")
    models.GCN = _il.import_module("sgformer_amd.ours_medium").GCN
    return models


def patch_100m_data_utils():
    """100M/nb-sample.py:12 imports `load_fixed_splits` from 100M/data_utils.py, which does not define it — the trainer
    cannot be imported as shipped.  Import the trainer's own data_utils (its directory is on sys.path by now) and add the
    missing name as a stub; the papers100M path (nb-sample.py:96-97) never calls it."""
    import importlib as _il
    try:
        du = _il.import_module("data_utils")
    except ModuleNotFoundError:          # a trainer without a data_utils module has nothing to repair
        return None
    if not hasattr(du, "load_fixed_splits"):
        def load_fixed_splits(*args, **kwargs):
            raise NotImplementedError("100M/data_utils.py does not define load_fixed_splits (only ogbn-papers100M's own "
                                      "fixed split is supported by this trainer)")
        du.load_fixed_splits = load_fixed_splits
    return du


def patch_subgraph():
    """Point torch_geometric.utils.subgraph at the GPU implementation (sgformer_amd.batching) before
    the trainer imports it (large/main-batch.py:8, large/eval.py:4): the per-batch induced subgraph
    then runs on the MI355X instead of as an O(E) host pass per batch."""
    import importlib as _il
    tgu = _il.import_module("torch_geometric.utils")
    tgu.subgraph = _il.import_module("sgformer_amd.batching").subgraph
    return tgu


def patch_prologue(keep_input_device: bool = False):
    """Serve the trainer prologue — to_undirected / remove_self_loops / add_self_loops at
    large/main.py:75-79 and 100M/nb-sample.py:79-80 — from the GPU (sgformer_amd.batching,
    sgf_graph_prologue_*).  For large/main*.py the edge_index the trainer then moves
    `.to(device)` is already there.  keep_input_device=True (the 100M trainer): the result goes back to the
    device of the tensor that came in — 100M/nb-sample.py:81-133 puts edge_index into a HOST `Data` object and
    hands it to NeighborLoader workers, which must not receive a CUDA tensor (nor should 52 GB of
    papers100M edges stay in HBM for the whole run)."""
    import importlib as _il
    tgu = _il.import_module("torch_geometric.utils")
    b = _il.import_module("sgformer_amd.batching")
    if not keep_input_device:
        tgu.to_undirected, tgu.remove_self_loops, tgu.add_self_loops = b.to_undirected, b.remove_self_loops, b.add_self_loops
        return tgu

    def _back(fn):
        def wrapped(edge_index, *args, **kwargs):
            out = fn(edge_index, *args, **kwargs)
            dev = edge_index.device
            if isinstance(out, tuple):
                return tuple(o.to(dev) if hasattr(o, "to") else o for o in out)
            return out.to(dev)
        wrapped.__name__ = fn.__name__
        return wrapped

    tgu.to_undirected, tgu.remove_self_loops, tgu.add_self_loops = (_back(b.to_undirected), _back(b.remove_self_loops),
                                                                    _back(b.add_self_loops))
    return tgu


def patch_neighbor_loader():
    """100M/nb-sample.py:11 `from torch_geometric.loader import NeighborLoader` -> sgformer_amd.sampling.NeighborLoader:
    the graph, features and labels stay in HBM and every batch is sampled / relabelled / gathered on the device
    (`--sgf-host-sampler 1` keeps PyG's host workers).  Creates the `torch_geometric.loader` module
    when the installed torch_geometric predates it."""
    import importlib as _il
    import types
    samp = _il.import_module("sgformer_amd.sampling")
    try:
        loader = _il.import_module("torch_geometric.loader")
    except ImportError:
        loader = types.ModuleType("torch_geometric.loader")
        sys.modules["torch_geometric.loader"] = loader
        tg = sys.modules.get("torch_geometric")
        if tg is not None:
            tg.loader = loader
    loader.NeighborLoader = samp.NeighborLoader
    return loader


def patch_resident_features():
    """Keep the node features of the mini-batch trainer on the GPU: wrap the trainer's own
    `dataset.load_dataset` so that `dataset.graph['node_feat']` is a device tensor.  The per-batch
    `x[idx_i].to(device)` of large/main-batch.py:138 is then a device gather + a no-op instead of an 88 ms
    host gather + H2D copy per 100 k-node batch (profiles/r01_minibatch_probe.md: 2.1 s of a 6.3 s epoch).
    Labels and splits stay on the host (the trainer indexes host masks with them)."""
    import importlib as _il
    import torch
    ds_mod = _il.import_module("dataset")
    orig = ds_mod.load_dataset

    def load_dataset(*args, **kwargs):
        ds = orig(*args, **kwargs)
        if torch.cuda.is_available() and torch.is_tensor(ds.graph.get("node_feat")):
            from . import staging
            # ... as ResidentRows: `x[idx_i]` with the trainer's HOST index gathers on the prep stream, and the labels as
            # StagedHost (still a host tensor): `true_label[idx_i].to(device)` copies there too — neither waits for the previous
            # batch's backward (sgformer_amd/staging.py; SGF_PREP_STREAM=0: plain tensors, every copy on the current stream)
            ds.graph["node_feat"] = staging.resident(ds.graph["node_feat"].to(torch.device("cuda", torch.cuda.current_device())))
            if staging.enabled() and torch.is_tensor(getattr(ds, "label", None)):
                ds.label = staging.staged(ds.label)
        return ds

    ds_mod.load_dataset = load_dataset
    return ds_mod


def patch_nll_loss():
    """The trainers keep their own loss lines (large/main.py:139-141: log_softmax, row indexing, nn.NLLLoss).  ATen's
    nll_loss kernels take 4.2 ms of an ogbn-products step (one-block reductions); the same arithmetic as a gather + a
    masked sum (sgformer_amd.loss.gather_nll) takes ~0.3 ms.  Installed behind torch.nn.functional.nll_loss — which
    nn.NLLLoss.forward looks up at call time — for exactly the case the trainers use (2-D CUDA input, 1-D int64 targets,
    no class weights, reduction 'mean'); anything else reaches the original.  `--sgf-aten-loss 1` keeps ATen's."""
    import torch
    import torch.nn.functional as F
    from . import ops
    from .loss import LazyLogSoftmax, gather_nll, lazy_rows_nll
    orig = getattr(F.nll_loss, "_sgf_orig", F.nll_loss)
    orig_ls = getattr(F.log_softmax, "_sgf_orig", F.log_softmax)
    lazy_on = os.environ.get("SGF_LAZY_LOG_SOFTMAX", "1") != "0"

    def on_device(t):                       # (the tests drive this with the CPU kernel table)
        return t.is_cuda if ops.K.name == "hip" else True

    def log_softmax(input, dim=None, _stacklevel=3, dtype=None):
        # r04: the FIRST of the three lines returns a lazy tensor; `out[train_idx]` + nn.NLLLoss then run as ONE pass over
        # the training rows (sgf_nll_fwd / _bwd) instead of log_softmax over all N rows, an index kernel and their backward
        # (1.6 -> 0.3 ms per ogbn-products step); any other use of the result computes the real log-softmax first
        if (lazy_on and torch.is_tensor(input) and not isinstance(input, LazyLogSoftmax) and input.dim() == 2
                and dim in (1, -1) and dtype is None and input.dtype in (torch.float32, torch.bfloat16) and on_device(input)
                and input.shape[0] > 0 and input.shape[1] <= 64 and input.stride(1) == 1):
            return LazyLogSoftmax(input, None, orig_ls)
        return orig_ls(input, dim=dim, _stacklevel=_stacklevel, dtype=dtype)

    def nll_loss(input, target, weight=None, size_average=None, ignore_index=-100, reduce=None, reduction="mean"):
        if isinstance(input, LazyLogSoftmax):
            if (input._sgf_idx is not None and input._sgf_cache is None and torch.is_tensor(target) and target.dim() == 1
                    and target.dtype == torch.long and target.shape[0] == input.shape[0] and weight is None
                    and size_average is None and reduce is None and reduction == "mean" and ignore_index < 0
                    and target.device == input.device):
                return lazy_rows_nll(input, target, ignore_index)
            input = input._sgf_value()
        if (torch.is_tensor(input) and torch.is_tensor(target) and input.is_cuda and input.dim() == 2 and target.dim() == 1
                and input.shape[0] == target.shape[0] and input.shape[0] > 0 and weight is None and size_average is None
                and reduce is None and reduction == "mean" and input.is_floating_point() and target.dtype == torch.long):
            return gather_nll(input, target, ignore_index)
        return orig(input, target, weight=weight, size_average=size_average, ignore_index=ignore_index, reduce=reduce,
                    reduction=reduction)

    nll_loss._sgf_orig = orig
    F.nll_loss = nll_loss
    log_softmax._sgf_orig = orig_ls
    F.log_softmax = log_softmax
    return orig


def patch_adam():
    """torch.optim.Adam as the trainers construct it (large/main.py:114-119: two parameter groups, no `fused` / `foreach`
    argument) runs its for-each form: ~12 multi-tensor launches per group and step.  For CUDA parameters torch's own
    single-kernel form (`fused=True`, same arithmetic) becomes the default here; an explicit `fused` / `foreach` argument of
    the trainer is respected.  SGF_FUSED_ADAM=0 leaves torch.optim.Adam alone."""
    import torch
    if os.environ.get("SGF_FUSED_ADAM", "1") == "0" or getattr(torch.optim.Adam, "_sgf_patched", False):
        return
    orig_init = torch.optim.Adam.__init__

    def __init__(self, params, *args, **kwargs):
        params = list(params)
        # a group's 'params' may be a GENERATOR (Adam([{'params': model.parameters()}])): materialise it on a shallow copy of
        # the group before looking at it, or the optimizer proper would find it exhausted and train nothing
        for i, grp in enumerate(params):
            if isinstance(grp, dict) and "params" in grp:
                grp = dict(grp)
                grp["params"] = [grp["params"]] if torch.is_tensor(grp["params"]) else list(grp["params"])
                params[i] = grp
        if "fused" not in kwargs and "foreach" not in kwargs and len(args) < 6:
            flat = [p for grp in params for p in (grp["params"] if isinstance(grp, dict) else [grp])]
            if flat and all(torch.is_tensor(p) and p.is_cuda and p.is_floating_point() for p in flat):
                kwargs["fused"] = True
        orig_init(self, params, *args, **kwargs)

    torch.optim.Adam.__init__ = __init__
    torch.optim.Adam._sgf_patched = True


def unpatch_nll_loss():
    import torch.nn.functional as F
    orig = getattr(F.nll_loss, "_sgf_orig", None)
    if orig is not None:
        F.nll_loss = orig
    orig_ls = getattr(F.log_softmax, "_sgf_orig", None)
    if orig_ls is not None:
        F.log_softmax = orig_ls


def limit_host_threads():
    """The mini-batch trainer's host lines (large/main-batch.py:134-146) index 100 k-element masks and labels per batch:
    torch's CPU kernels split that over its OpenMP team, and on the 256-thread hosts of MI355X boxes every parallel region
    then waits for its slowest thread, on a shared host for milliseconds (r01: 75 ms stalls with the default 128+ threads;
    r05, epoch of 25 batches, same box, back to back: 16 threads 11.1 / 11.6, 1 thread 15.6 / 15.7, 4 threads 17.4 / 17.7
    M nodes/s — profiles/r05_minibatch_threads.md).  4 threads; OMP_NUM_THREADS set by the user is respected."""
    import torch
    if "OMP_NUM_THREADS" not in os.environ:
        torch.set_num_threads(max(1, min(HOST_THREADS, os.cpu_count() or 1)))


HOST_THREADS = 4


def _select_device(argv):
    """The trainers take `--device N` (large/parse.py, default 0) and build torch.device('cuda:N'); the launcher's
    patches place data on the CURRENT device, so make N current before the trainer runs."""
    import torch
    for i, a in enumerate(argv):
        v = argv[i + 1] if a == "--device" and i + 1 < len(argv) else (a.split("=", 1)[1] if a.startswith("--device=") else None)
        if v is not None and v.isdigit() and torch.cuda.is_available() and int(v) < torch.cuda.device_count():
            torch.cuda.set_device(int(v))
            return int(v)
    return None


def _pop_option(argv, name):
    for i, a in enumerate(argv):
        if a == name and i + 1 < len(argv):
            v = argv[i + 1]
            del argv[i:i + 2]
            return v
        if a.startswith(name + "="):
            del argv[i]
            return a.split("=", 1)[1]
    return None


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    variant = _pop_option(argv, "--sgf-variant")
    dtype = _pop_option(argv, "--sgf-dtype")
    host_subgraph = _pop_option(argv, "--sgf-host-subgraph")   # any value: keep PyG's host subgraph
    host_prologue = _pop_option(argv, "--sgf-host-prologue")   # any value: keep PyG's host to_undirected & co.
    host_features = _pop_option(argv, "--sgf-host-features")   # any value: keep node features on the host
    aten_loss = _pop_option(argv, "--sgf-aten-loss")           # any value: keep ATen's nll_loss kernels
    host_sampler = _pop_option(argv, "--sgf-host-sampler")     # any value: keep PyG's host NeighborLoader (100M)
    patches = _pop_option(argv, "--sgf-patches") or os.environ.get("SGF_PATCHES", "all")
    if patches not in ("all", "minimal"):
        raise SystemExit(f"sgformer_amd.launch: --sgf-patches {patches!r} (choose 'all' or 'minimal')")
    if not argv or argv[0] in ("-h", "--help"):
        raise SystemExit(__doc__)
    trainer = os.path.abspath(argv[0])
    if not os.path.isfile(trainer):
        raise SystemExit(f"sgformer_amd.launch: trainer script not found: {trainer}")
    tdir = os.path.dirname(trainer)
    if variant is None:
        variant = os.path.basename(tdir)
        if variant not in VARIANTS:
            raise SystemExit(f"sgformer_amd.launch: cannot infer the variant from directory {variant!r}; "
                             f"pass --sgf-variant large|100M|medium")
    install(variant, dtype)
    # what `python trainer.py args...` would have set up
    sys.argv = [trainer] + argv[1:]
    if tdir in sys.path:
        sys.path.remove(tdir)
    sys.path.insert(0, tdir)
    if variant == "medium":
        patch_medium_gcn()
    if variant in ("100M", "100m"):
        patch_100m_data_utils()
    _select_device(argv)
    if patches == "minimal":
        # the module drop-in ONLY: `ours` (and, for the medium / 100M trainers, the two repairs above, which touch the
        # TRAINER's own modules) — nothing of torch, torch_geometric or the host thread pool is rewired: F.log_softmax,
        # F.nll_loss, torch.optim.Adam, torch_geometric.utils.* and NeighborLoader stay what the environment provides
        runpy.run_path(trainer, run_name="__main__")
        return
    if host_subgraph is None and os.path.basename(trainer) == "main-batch.py":
        patch_subgraph()
    # --sgf-host-subgraph implies the host prologue: PyG's host subgraph() indexes a CPU mask with edge_index[0],
    # which must then be a host tensor too
    if host_prologue is None and host_subgraph is None and variant != "medium":
        patch_prologue(keep_input_device=(variant == "100M"))
    if variant == "100M" and host_sampler is None:
        patch_neighbor_loader()
    if aten_loss is None:
        patch_nll_loss()
    patch_adam()
    if os.path.basename(trainer) == "main-batch.py":
        limit_host_threads()
        if host_features is None:
            patch_resident_features()
    runpy.run_path(trainer, run_name="__main__")


if __name__ == "__main__":
    main()
