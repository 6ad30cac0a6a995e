"""Generate tests/golden/*.npz from the LIVE reference (run in the build container only).

    python oracle/make_golden.py

Imports /root/reference/{large,100M,medium}/ours.py (and medium/models.py) unchanged through oracle/ref_shim.py, runs one
training-mode forward + loss + backward and one eval-mode forward in float64 (the reference's own
arithmetic, `torch.set_default_dtype(float64)` because large/ours.py:141 creates `all_ones` in the
default dtype), and records

  * inputs (x, edge_index, y, train_idx) and the state_dict,
  * logits (train / eval), loss, every parameter gradient, BatchNorm running stats after the step,
  * the attention intermediates of every TransConvLayer, captured by wrapping torch.einsum while the
    reference runs (kvs "lhm,lhd->hmd", ks_sum "lhm,l->hm", q.kvs "nhm,hmd->nhd", q.ks_sum "nhm,hm->nh"),
  * the sorted COO the reference hands to its SpMM (row, col, value of SparseTensor), i.e. the CSR
    arrays, in the reference's own fp32 value arithmetic.

These fixtures are what pins oracle/sgformer_oracle.py on machines without /root/reference
(tests/test_oracle.py) — the reference ships no tests or golden vectors of its own.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_shim  # noqa: E402
from sgformer_amd.synth import synthetic_graph  # noqa: E402

CASES = {
    # name: (variant, N, f, d, C, avg_deg, directed, ctor kwargs)
    "arxiv_recipe": ("large", 257, 20, 16, 5, 6.0, False,
                     dict(trans_num_layers=1, trans_num_heads=1, trans_use_bn=True, trans_use_residual=True,
                          trans_use_weight=True, trans_use_act=False, gnn_num_layers=3, gnn_use_bn=True,
                          gnn_use_residual=True, gnn_use_weight=True, gnn_use_init=False, gnn_use_act=True,
                          graph_weight=0.5, aggregate="add")),
    "products_recipe": ("large", 300, 12, 32, 7, 8.0, False,
                        dict(trans_num_layers=1, trans_num_heads=1, trans_use_bn=True, trans_use_residual=True,
                             trans_use_weight=True, trans_use_act=False, gnn_num_layers=3, gnn_use_bn=True,
                             gnn_use_residual=True, gnn_use_weight=True, gnn_use_init=True, gnn_use_act=True,
                             graph_weight=0.5, aggregate="add")),
    "heads2_cat_directed": ("large", 199, 10, 16, 4, 5.0, True,
                            dict(trans_num_layers=2, trans_num_heads=2, trans_use_act=True, gnn_num_layers=1,
                                 aggregate="cat")),
    "no_weights": ("large", 150, 16, 16, 3, 4.0, False,
                   dict(trans_use_weight=False, trans_num_heads=2, trans_use_bn=False, gnn_use_weight=False,
                        gnn_use_bn=False, gnn_use_residual=False, gnn_num_layers=2)),
    "alpha_100M": ("100M", 220, 14, 16, 6, 6.0, True,
                   dict(trans_num_layers=1, trans_num_heads=1, alpha=0.3, trans_use_bn=True,
                        trans_use_residual=True, trans_use_weight=True, trans_use_act=False,
                        gnn_num_layers=2, gnn_use_bn=True, gnn_use_residual=True, gnn_use_weight=True,
                        gnn_use_init=True, gnn_use_act=True, graph_weight=0.8, aggregate="add")),
}


# r05 (VERDICT r04 "parity reach"): the recipes at their PRODUCTION widths, so that the fixture test drives the kernel
# instantiations the headline runs (k_rowgemm2_bf16<256>, k_hrow_bf16<256>, k_linear_f32<256>, k_stem_bf16 with f = 100; the
# 100M recipe's d = 128, C = 172) against numbers the reference itself produced.  Parameters, features and gradients are stored
# as fp32 (the reference runs in fp64 ON fp32-representable inputs), which keeps the two files at a few MB.
PRODUCTION_CASES = {
    "products_d256": ("large", 4096, 100, 256, 47, 12.0, False, dict(CASES["products_recipe"][7])),
    "papers_d128": ("100M", 2048, 128, 128, 172, 10.0, True,
                    dict(trans_num_layers=1, trans_num_heads=1, alpha=0.5, trans_use_bn=True, trans_use_residual=True,
                         trans_use_weight=True, trans_use_act=False, gnn_num_layers=3, gnn_use_bn=True,
                         gnn_use_residual=True, gnn_use_weight=True, gnn_use_init=True, gnn_use_act=True, graph_weight=0.8,
                         aggregate="add")),
}


def hub_graph(n, avg_deg, seed, hubs=(1100, 1700, 2600)):
    """The uniform graph of synth.synthetic_graph plus a few HUB nodes joined to 1100 / 1700 / 2600 others (both directions):
    rows longer than the kernels' long-row threshold (1024 stored entries) next to ordinary ones — the path R-MAT / power-law
    graphs take (k_spmm_long_seg), at a size the reference runs in seconds.  Same prologue as the trainer: symmetric, coalesced,
    one self-loop per node (large/main.py:75-79)."""
    ei = synthetic_graph(n, avg_deg, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    src, dst = [ei[0]], [ei[1]]
    for h, k in enumerate(hubs):
        nb = torch.randperm(n, generator=g)[:k]
        nb = nb[nb != h]
        src += [torch.full_like(nb, h), nb]
        dst += [nb, torch.full_like(nb, h)]
    key = torch.unique(torch.cat(src) * n + torch.cat(dst))          # coalesce (the uniform part already holds the loops)
    return torch.stack([key // n, key % n])


def run_case(name, variant, n, f, d, c, avg_deg, directed, kw, store32=False, graph=None):
    ref = ref_shim.load_reference(variant)
    torch.set_default_dtype(torch.float64)
    try:
        torch.manual_seed(1234)
        model = ref.SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, **kw).double()
        # non-trivial norm parameters / running stats so that every term is exercised
        with torch.no_grad():
            for k_, v_ in model.state_dict().items():
                if k_.endswith("running_mean"):
                    v_.normal_(0, 0.1)
                elif k_.endswith("running_var"):
                    v_.uniform_(0.8, 1.3)
                elif ".bns." in k_ and k_.endswith("weight"):
                    v_.normal_(1.0, 0.1)
                elif ".bns." in k_ and k_.endswith("bias"):
                    v_.normal_(0.0, 0.1)
        x = torch.randn(n, f)
        if store32:     # fp32-representable parameters and features: what is stored is exactly what the reference ran on
            x = x.float().double()
            with torch.no_grad():
                for v_ in model.state_dict().values():
                    if v_.is_floating_point():
                        v_.copy_(v_.float().double())
        ei = graph(n, avg_deg, len(name)) if graph is not None else synthetic_graph(n, avg_deg, seed=len(name), directed=directed)
        y = torch.randint(0, c, (n,))
        idx = torch.randperm(n)[: n // 2]
        sd0 = {k_: v_.detach().clone() for k_, v_ in model.state_dict().items()}

        rec = {"einsum": [], "coo": []}
        real_einsum = torch.einsum

        def spy_einsum(eq, *ops_):
            out = real_einsum(eq, *ops_)
            rec["einsum"].append((eq, out.detach().clone()))
            return out

        ts = sys.modules["torch_sparse"]
        real_st = ts.SparseTensor

        class SpyST(real_st):
            def __init__(self, *a, **k):
                super().__init__(*a, **k)
                rec["coo"].append((self.row.clone(), self.col.clone(), self.value.detach().clone()))

        torch.einsum, ref.SparseTensor = spy_einsum, SpyST
        try:
            model.train()
            logits = model(x, ei)
        finally:
            torch.einsum, ref.SparseTensor = real_einsum, real_st
        loss = torch.nn.functional.nll_loss(torch.log_softmax(logits, dim=1)[idx], y[idx])
        loss.backward()
        grads = {k_: (p.grad.detach().clone() if p.grad is not None else None)
                 for k_, p in model.named_parameters()}
        sd1 = {k_: v_.detach().clone() for k_, v_ in model.state_dict().items()}
        model.eval()
        with torch.no_grad():
            logits_eval = model(x, ei)
    finally:
        torch.set_default_dtype(torch.float32)

    out = {"x": x.numpy(), "edge_index": ei.numpy(), "y": y.numpy(), "train_idx": idx.numpy(),
           "logits_train": logits.detach().numpy(), "loss": np.array(float(loss)),
           "logits_eval": logits_eval.numpy()}
    for k_, v_ in sd0.items():
        out["param/" + k_] = v_.numpy()
    for k_, v_ in sd1.items():
        if "running" in k_:
            out["after/" + k_] = v_.numpy()
    for k_, g in grads.items():
        if g is not None:
            out["grad/" + k_] = g.numpy()
    layer = -1
    for eq, t in rec["einsum"]:
        if eq == "lhm,lhd->hmd":
            layer += 1
        tag = {"lhm,lhd->hmd": "kvs", "nhm,hmd->nhd": "q_kvs", "lhm,l->hm": "ks_sum", "nhm,hm->nh": "q_ks_sum"}[eq]
        out[f"attn{layer}/{tag}"] = t.numpy()
    if rec["coo"]:
        r, c_, v_ = rec["coo"][0]
        out["coo/row"], out["coo/col"], out["coo/value"] = r.numpy(), c_.numpy(), v_.numpy().astype(np.float32)
    meta = dict(name=name, variant=variant, n=n, f=f, d=d, c=c, avg_deg=avg_deg, directed=directed, cfg=kw,
                production=bool(store32))
    out["meta"] = np.array(json.dumps(meta))
    if store32:
        for k_ in list(out):
            if k_.endswith(("/q_kvs", "/q_ks_sum")):      # per-node intermediates ([N, H, d] fp64): the d x d ones pin the layer
                del out[k_]
            elif (k_.startswith(("param/", "grad/", "logits_")) or k_ == "x") and out[k_].dtype == np.float64:
                out[k_] = out[k_].astype(np.float32)
    return out


MEDIUM_CASES = {
    # medium/run.sh:2-7 (Cora recipe): 1 attention layer, GCN backbone with num_layers 4, gw 0.8, alpha 0.5
    "cora_medium": dict(n=180, f=24, d=16, c=7, avg_deg=4.0, gcn_layers=4,
                        cfg=dict(num_layers=1, num_heads=1, alpha=0.5, use_bn=True, use_residual=True,
                                 use_weight=True, use_graph=True, graph_weight=0.8, aggregate="add")),
}


class _Data:
    """The `data` object medium/ours.py:134-136 and medium/models.py:50-52 read."""

    def __init__(self, x, ei):
        self.graph = {"node_feat": x, "edge_index": ei, "num_nodes": x.shape[0]}


def run_medium_case(name, n, f, d, c, avg_deg, gcn_layers, cfg):
    ref = ref_shim.load_reference("medium")
    torch.set_default_dtype(torch.float64)
    try:
        torch.manual_seed(4321)
        gnn = ref.models.GCN(f, d, d, num_layers=gcn_layers, dropout=0.0, use_bn=True)
        model = ref.SGFormer(f, d, c, dropout=0.0, gnn=gnn, **cfg).double()
        with torch.no_grad():
            for k_, v_ in model.state_dict().items():
                if k_.endswith("running_mean"):
                    v_.normal_(0, 0.1)
                elif k_.endswith("running_var"):
                    v_.uniform_(0.8, 1.3)
                elif k_.endswith("bias"):
                    v_.normal_(0.0, 0.1)          # GCNConv biases start at zero: make them count
        x = torch.randn(n, f)
        ei = synthetic_graph(n, avg_deg, seed=11)[:, :-n]   # medium/main.py:94 symmetrises, adds no self-loops
        y = torch.randint(0, c, (n,))
        idx = torch.randperm(n)[: n // 2]
        sd0 = {k_: v_.detach().clone() for k_, v_ in model.state_dict().items()}
        data = _Data(x, ei)
        model.train()
        logits = model(data)
        loss = torch.nn.functional.nll_loss(torch.log_softmax(logits, dim=1)[idx], y[idx])
        loss.backward()
        grads = {k_: p.grad.detach().clone() for k_, p in model.named_parameters() if p.grad is not None}
        sd1 = {k_: v_.detach().clone() for k_, v_ in model.state_dict().items()}
        model.eval()
        with torch.no_grad():
            logits_eval = model(data)
    finally:
        torch.set_default_dtype(torch.float32)
    out = {"x": x.numpy(), "edge_index": ei.numpy(), "y": y.numpy(), "train_idx": idx.numpy(),
           "logits_train": logits.detach().numpy(), "loss": np.array(float(loss)),
           "logits_eval": logits_eval.numpy()}
    for k_, v_ in sd0.items():
        out["param/" + k_] = v_.numpy()
    for k_, v_ in sd1.items():
        if "running" in k_:
            out["after/" + k_] = v_.numpy()
    for k_, g in grads.items():
        out["grad/" + k_] = g.numpy()
    out["meta"] = np.array(json.dumps(dict(name=name, variant="medium", n=n, f=f, d=d, c=c, avg_deg=avg_deg,
                                           gcn_layers=gcn_layers, cfg=cfg)))
    return out


# r06: the products recipe at production width on a graph with HUB rows — the long-row path of the SpMM
# (rows beyond 1024 stored entries are reduced by whole workgroups) against numbers of the live reference.
HUB_CASES = {
    "products_d256_hub": ("large", 4096, 100, 256, 47, 10.0, False, dict(CASES["products_recipe"][7])),
}


def main():
    if not ref_shim.reference_available():
        raise SystemExit("reference not mounted: golden vectors can only be generated in the build container")
    dst = os.path.join(ROOT, "tests", "golden")
    os.makedirs(dst, exist_ok=True)
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None     # regenerate ONE fixture
    for name, spec in HUB_CASES.items():
        if only in (None, name):
            out = run_case(name, *spec, store32=True, graph=hub_graph)
            path = os.path.join(dst, name + ".npz")
            np.savez_compressed(path, **out)
            print(f"{path}: {os.path.getsize(path) / 1024:.0f} KiB, {len(out)} arrays")
    if only in HUB_CASES:
        return
    for name, spec in CASES.items():
        out = run_case(name, *spec)
        path = os.path.join(dst, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{path}: {os.path.getsize(path) / 1024:.0f} KiB, {len(out)} arrays")
    for name, spec in PRODUCTION_CASES.items():
        out = run_case(name, *spec, store32=True)
        path = os.path.join(dst, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{path}: {os.path.getsize(path) / 1024:.0f} KiB, {len(out)} arrays")
    for name, spec in MEDIUM_CASES.items():
        out = run_medium_case(name, **spec)
        path = os.path.join(dst, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{path}: {os.path.getsize(path) / 1024:.0f} KiB, {len(out)} arrays")


if __name__ == "__main__":
    main()
