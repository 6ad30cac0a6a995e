"""The HIP path against the REFERENCE's own numbers, in one step ("parity reach").

tests/golden/*.npz were dumped from the live reference (`/root/reference/{large,100M,medium}/ours.py`
executed unchanged in fp64 by oracle/make_golden.py).  Here every fixture's stored inputs and
state_dict are loaded into the drop-in module on the GPU and logits (train / eval), loss, every
parameter gradient, BatchNorm running statistics and the attention intermediates are compared with
the stored values — no oracle in between.  Tolerances are BASELINE.json's: 1e-4 absolute on fp32
logits, relative on gradients; CSR arrays bit-exact.

Second half: BASELINE.json config 2 AS WRITTEN — the full ogbn-arxiv shape (N = 169 343,
nnz ~ 2.48 M, f = 128, d = 256, C = 40, large/run.sh:2-5 recipe, fp32, dropout 0) against the fp64
oracle restatement (large/ours.py:265-276 end to end).
"""
import glob
import json
import os

import numpy as np
import pytest
import torch

from oracle import sgformer_oracle as O

pytestmark = pytest.mark.gpu

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


class _Data:
    def __init__(self, x, ei):
        self.graph = {"node_feat": x, "edge_index": ei, "num_nodes": x.shape[0]}


def _module(meta, z):
    f, d, c, cfg = meta["f"], meta["d"], meta["c"], meta["cfg"]
    if meta["variant"] == "medium":
        from sgformer_amd import ours_medium as M
        gnn = M.GCN(f, d, d, num_layers=meta["gcn_layers"], dropout=0.0, use_bn=True)
        m = M.SGFormer(f, d, c, dropout=0.0, gnn=gnn, **cfg)
    elif meta["variant"] == "100M":
        from sgformer_amd.ours_100m import SGFormer
        m = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, **cfg)
    else:
        from sgformer_amd.ours import SGFormer
        m = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, **cfg)
    sd = m.state_dict()
    params = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param/")}
    assert set(params) == set(sd), (set(params) ^ set(sd))          # the state_dict contract of §8b
    m.load_state_dict({k: v.to(sd[k].dtype) for k, v in params.items()})
    return m


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_hip_module_matches_reference_fixture(cuda, path):
    from sgformer_amd import ops
    z = np.load(path, allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    medium = meta["variant"] == "medium"
    m = _module(meta, z).to(cuda).train()
    x = torch.from_numpy(z["x"]).float().to(cuda)
    ei = torch.from_numpy(z["edge_index"]).to(cuda)
    y = torch.from_numpy(z["y"]).to(cuda)
    idx = torch.from_numpy(z["train_idx"]).to(cuda)

    # capture the input of every attention layer (to rebuild the reference's einsum intermediates)
    layer_inputs = []
    hooks = [conv.register_forward_pre_hook(lambda mod, args: layer_inputs.append(args[0].detach()))
             for conv in m.trans_conv.convs]
    logits = m(_Data(x, ei)) if medium else m(x, ei)
    for h in hooks:
        h.remove()
    loss = torch.nn.functional.nll_loss(torch.log_softmax(logits, dim=1)[idx], y[idx])
    loss.backward()

    ref_logits = z["logits_train"]
    assert np.abs(logits.detach().double().cpu().numpy() - ref_logits).max() <= 1e-4
    assert abs(float(loss) - float(z["loss"])) <= 1e-5
    gmax = max(float(np.linalg.norm(z[k])) for k in z.files if k.startswith("grad/"))
    n_grad = 0
    for k, prm in m.named_parameters():
        if "grad/" + k not in z.files:           # unused GraphConvLayer.W (large/ours.py:20 vs :36-40)
            assert prm.grad is None or float(prm.grad.abs().max()) == 0.0, k
            continue
        g_ref = z["grad/" + k]
        num = float(np.linalg.norm(prm.grad.double().cpu().numpy() - g_ref))
        assert num <= 2e-4 * float(np.linalg.norm(g_ref)) + 1e-6 * gmax, (k, num)
        n_grad += 1
    assert n_grad >= 10
    sd = m.state_dict()
    n_after = 0
    for k in z.files:
        if k.startswith("after/"):
            assert np.abs(sd[k[6:]].double().cpu().numpy() - z[k]).max() <= 1e-5, k
            n_after += 1
    assert n_after > 0 or not meta["cfg"].get("gnn_use_bn", True)

    # attention intermediates: the reference's einsum outputs kvs = k^T v / |K|, ks_sum = sum k / |K|
    # (large/ours.py:136,141-142) from the libsgf partials [S0 | z0 | |Q|^2 | |K|^2] on the same inputs
    for i, conv in enumerate(m.trans_conv.convs):
        if f"attn{i}/kvs" not in z.files:
            continue
        h_in = layer_inputs[i]
        n, heads, d = h_in.shape[0], conv.num_heads, conv.out_channels
        q = torch.nn.functional.linear(h_in, conv.Wq.weight, conv.Wq.bias).reshape(n, heads, d)
        k_ = torch.nn.functional.linear(h_in, conv.Wk.weight, conv.Wk.bias).reshape(n, heads, d)
        v = (torch.nn.functional.linear(h_in, conv.Wv.weight, conv.Wv.bias).reshape(n, heads, d)
             if conv.use_weight else h_in.reshape(n, 1, d))
        raw = ops.attention_stats(q.detach(), k_.detach(), v.detach()).double().cpu()
        vh = v.shape[1]
        s0 = raw[: heads * d * d].reshape(heads, d, d) / torch.sqrt(raw[-1])
        z0 = raw[heads * d * d: heads * d * d + heads * d].reshape(heads, d) / torch.sqrt(raw[-1])
        kvs_ref, ks_ref = z[f"attn{i}/kvs"], z[f"attn{i}/ks_sum"]
        if vh == 1 and heads > 1:                   # V shared by the heads: reference broadcasts it
            kvs_ref = kvs_ref.reshape(s0.shape)
        assert np.abs(s0.numpy() - kvs_ref).max() <= 2e-5 * max(1.0, np.abs(kvs_ref).max()), i
        assert np.abs(z0.numpy() - ks_ref).max() <= 2e-5 * max(1.0, np.abs(ks_ref).max()), i

    # CSR arrays: the reference's sorted COO, bit for bit (large/ours.py:26-33)
    if "coo/row" in z.files:
        g = ops.CSRGraph(ei, x.shape[0])
        rowptr = g.rowptr.cpu().numpy()
        assert np.array_equal(np.repeat(np.arange(x.shape[0]), np.diff(rowptr)), z["coo/row"])
        assert np.array_equal(g.colind.cpu().numpy(), z["coo/col"])
        assert np.array_equal(g.val.cpu().numpy().view(np.uint32), z["coo/value"].view(np.uint32))

    # eval mode on the running statistics the training step left behind
    m.eval()
    with torch.no_grad():
        le = m(_Data(x, ei)) if medium else m(x, ei)
    assert np.abs(le.double().cpu().numpy() - z["logits_eval"]).max() <= 1e-4


PRODUCTION = [p for p in GOLDEN if os.path.basename(p) in ("products_d256.npz", "papers_d128.npz", "products_d256_hub.npz")]


@pytest.mark.parametrize("path", PRODUCTION, ids=[os.path.basename(p)[:-4] for p in PRODUCTION])
def test_bf16_module_matches_reference_fixture_at_production_width(cuda, path):
    """VERDICT r04 "parity reach" (a): the bf16 kernels of the headline — k_stem_bf16 (f = 100), k_rowgemm2_bf16<256>,
    k_dx2acc_bf16<256>, k_hrow_bf16<256>, k_reduce_bf16<256, ...>, k_head_*_bf16 (and the 100M recipe's d = 128, C = 172) —
    against numbers the REFERENCE produced in fp64 (oracle/make_golden.py PRODUCTION_CASES), through the module in bf16
    mode: logits to 1e-2 of their scale, loss to 1e-3, every parameter gradient to a Frobenius-relative bound that is the
    bf16 storage error of a 3-layer network, not a kernel property (fp32 mode, same fixtures: the 1e-4 test above)."""
    z = np.load(path, allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    m = _module(meta, z)
    m.compute_dtype = torch.bfloat16
    m = m.to(cuda).train()
    x = torch.from_numpy(z["x"]).float().to(cuda)
    ei = torch.from_numpy(z["edge_index"]).to(cuda)
    y = torch.from_numpy(z["y"]).to(cuda)
    idx = torch.from_numpy(z["train_idx"]).to(cuda)
    logits = m(x, ei).float()
    loss = torch.nn.functional.nll_loss(torch.log_softmax(logits, dim=1)[idx], y[idx])
    loss.backward()
    ref = z["logits_train"].astype(np.float64)
    scale = max(1.0, float(np.abs(ref).max()))
    err = float(np.abs(logits.detach().double().cpu().numpy() - ref).max())
    report = {"logits_max_abs_err": err, "logits_scale": scale, "loss_err": abs(float(loss) - float(z["loss"]))}
    worst = 0.0
    gmax = max(float(np.linalg.norm(z[k])) for k in z.files if k.startswith("grad/"))
    for k, prm in m.named_parameters():
        if "grad/" + k not in z.files:
            continue
        g_ref = z["grad/" + k].astype(np.float64)
        # (a Linear bias in front of a BatchNorm has an exactly zero gradient — the reference's 1e-17 is rounding noise, the
        # bf16 run's is bf16 noise of ~2e-4 of the largest gradient — so every tensor gets an absolute allowance of 1e-3 of
        # the largest gradient's norm on top of the relative bound)
        err_k = float(np.linalg.norm(prm.grad.double().cpu().numpy() - g_ref))
        rel = max(0.0, err_k - 1e-3 * gmax) / max(float(np.linalg.norm(g_ref)), 1e-300)
        report["grad/" + k] = err_k / max(float(np.linalg.norm(g_ref)), 1e-3 * gmax)
        worst = max(worst, min(rel, err_k / max(float(np.linalg.norm(g_ref)), 1e-300)))
    print("bf16 production fixture:", meta["name"], json.dumps(report))
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):           # measured errors kept next to the run (merged back from the GPU box)
        with open(os.path.join(out_dir, "golden_bf16_reports.jsonl"), "a") as fh:
            fh.write(json.dumps({"fixture": meta["name"], **report}) + "\n")
    assert err <= 1e-2 * scale, report
    assert report["loss_err"] <= 1e-3, report
    assert worst <= 0.2, report
    # ... and per tensor at 2 x what this test measured on MI355X (tests/golden/bf16_bounds.json, scripts/make_bf16_bounds.py):
    # the blanket 20 % above would not notice a mis-tiled dW of ONE layer
    bounds = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bf16_bounds.json"))).get(meta["name"])
    assert bounds is not None, f"no recorded bounds for fixture {meta['name']}"
    over = {k: (report[k], b) for k, b in bounds.items() if k in report and report[k] > b}
    assert not over, over
    assert set(k for k in report if k.startswith("grad/")) <= set(bounds), "a gradient without a recorded bound"


def test_full_ogbn_arxiv_shape_vs_fp64_oracle(cuda):
    """BASELINE.json config 2 at its full size: N = 169 343, 13.7 undirected neighbours per node
    (nnz ~ 2.48 M with self-loops), f = 128, hidden 256, 40 classes, fp32, recipe large/run.sh:2-5,
    dropout 0.  Logits within 1e-4 (absolute) of the fp64 oracle, loss within 1e-5, every parameter
    gradient within 5e-4 relative (Frobenius)."""
    from sgformer_amd import synth
    from sgformer_amd.ours import SGFormer
    n, deg, f, c, d = synth.SHAPES["ogbn-arxiv"]
    cfg = dict(synth.RECIPES["ogbn-arxiv"])
    ei = synth.synthetic_graph(n, deg, seed=123)
    x, y, idx = synth.synthetic_task(n, f, c, seed=123)
    assert n == 169343 and 2.3e6 < ei.shape[1] < 2.7e6
    p = O.init_params(cfg, f, d, c, seed=0)
    m = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, **cfg)
    m.load_state_dict({**m.state_dict(), **p})
    m = m.to(cuda).train()
    logits = m(x.to(cuda), ei.to(cuda))
    loss = O.nll_loss(logits, y.to(cuda), idx.to(cuda))
    loss.backward()
    torch.cuda.synchronize()

    torch.set_num_threads(min(32, os.cpu_count() or 1))
    p64 = {k: v.double().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in p.items()}
    ref = O.sgformer_forward(p64, x.double(), ei, cfg, training=True)
    loss_ref = O.nll_loss(ref, y, idx)
    loss_ref.backward()
    err = float((logits.detach().double().cpu() - ref.detach()).abs().max())
    # gradients: Frobenius-relative 5e-4, OR no worse than 4x what the fp32 CPU oracle itself loses against
    # fp64 on this input.  (At N = 169 343 the weight gradients in front of a BatchNorm are ill-conditioned in
    # fp32 — dZ has an exactly zero column sum that fp32 only approximates, and that residual multiplies the
    # large common mean of the SpMM output — so ANY fp32 implementation, the reference's included, is at ~1e-3
    # there; the per-element absolute error stays far below BASELINE's 1e-4.)
    p32 = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in p.items()}
    O.nll_loss(O.sgformer_forward(p32, x, ei, cfg, training=True), y, idx).backward()
    gmax = max(float(v.grad.norm()) for v in p64.values() if v.grad is not None)
    report = {"logits_max_abs_err": err, "loss_err": abs(float(loss.detach()) - float(loss_ref.detach())),
              "logits_scale": float(ref.detach().abs().max())}
    bad = []
    for k, prm in m.named_parameters():
        g = p64[k].grad
        if g is None:
            continue
        num = float((prm.grad.double().cpu() - g).norm())
        num32 = float((p32[k].grad.double() - g).norm())
        amax = float((prm.grad.double().cpu() - g).abs().max())
        report["grad/" + k] = {"rel": num / max(float(g.norm()), 1e-300), "cpu_fp32_rel": num32 / max(float(g.norm()), 1e-300),
                               "max_abs": amax}
        if amax > 1e-4 or num > max(5e-4 * float(g.norm()) + 1e-6 * gmax, 4.0 * num32):
            bad.append(k)
    print("arxiv-full parity:", json.dumps(report))
    assert err <= 1e-4, report
    assert report["loss_err"] <= 1e-5, report
    assert not bad, (bad, report)
