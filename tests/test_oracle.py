"""Pins for the oracle (CPU, no GPU): oracle/sgformer_oracle.py against
  (a) the committed golden vectors dumped from the LIVE reference by oracle/make_golden.py,
  (b) the live reference itself when /root/reference is mounted (build container only).
Everything is float64: agreement is to rounding (1e-12), not to a model tolerance.
"""
import glob
import json
import os

import numpy as np
import pytest
import torch

from oracle import ref_shim, sgformer_oracle as O

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def _load(path):
    z = np.load(path, allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    return z, meta


def _medium_golden(z, meta):
    """medium/ours.py + models.GCN fixture (BASELINE.json config 1 lineage)."""
    cfg = meta["cfg"]
    x, ei = torch.from_numpy(z["x"]), torch.from_numpy(z["edge_index"])
    y, idx = torch.from_numpy(z["y"]), torch.from_numpy(z["train_idx"])
    p = {k[6:]: torch.from_numpy(z[k]).clone() for k in z.files if k.startswith("param/")}
    for k, v in p.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    stats = {}
    logits = O.medium_forward(p, x, ei, cfg, training=True, bn_stats=stats)
    assert np.abs(logits.detach().numpy() - z["logits_train"]).max() <= 1e-12
    loss = O.nll_loss(logits, y, idx)
    assert abs(float(loss) - float(z["loss"])) <= 1e-12
    loss.backward()
    n_grad = 0
    for k in z.files:
        if k.startswith("grad/"):
            g = p[k[5:]].grad
            assert g is not None, k
            assert np.abs(g.numpy() - z[k]).max() <= 1e-12 + 1e-9 * np.abs(z[k]).max(), k
            n_grad += 1
    assert n_grad >= 20
    for key, (mu, var_unb) in stats.items():
        assert np.abs((0.9 * p[key + ".running_mean"] + 0.1 * mu).numpy() - z["after/" + key + ".running_mean"]).max() <= 1e-12
        assert np.abs((0.9 * p[key + ".running_var"] + 0.1 * var_unb).numpy() - z["after/" + key + ".running_var"]).max() <= 1e-12
    pe = {k: v.detach() for k, v in p.items()}
    for k in z.files:
        if k.startswith("after/"):
            pe[k[6:]] = torch.from_numpy(z[k])
    assert np.abs(O.medium_forward(pe, x, ei, cfg, training=False).numpy() - z["logits_eval"]).max() <= 1e-12


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_matches_golden(path):
    z, meta = _load(path)
    if meta["variant"] == "medium":
        return _medium_golden(z, meta)
    cfg = meta["cfg"]
    # production-width fixtures (r05) store parameters / features / logits / gradients as fp32 (the reference ran in fp64 on
    # exactly those fp32-representable inputs): run the restatement in fp64 on them, compare to the stored precision
    prod = bool(meta.get("production"))
    stored = 2e-7 if prod else 0.0
    x = torch.from_numpy(z["x"]).double()
    ei = torch.from_numpy(z["edge_index"])
    y = torch.from_numpy(z["y"])
    idx = torch.from_numpy(z["train_idx"])
    p = {k[6:]: torch.from_numpy(z[k]).clone() for k in z.files if k.startswith("param/")}
    p = {k: (v.double() if v.is_floating_point() else v) for k, v in p.items()}
    for k, v in p.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    parts, stats = {}, {}
    logits = O.sgformer_forward(p, x, ei, cfg, training=True, parts=parts, bn_stats=stats)
    assert np.abs(logits.detach().numpy() - z["logits_train"]).max() <= 1e-12 + stored * np.abs(z["logits_train"]).max()
    loss = O.nll_loss(logits, y, idx)
    assert abs(float(loss) - float(z["loss"])) <= 1e-12
    loss.backward()
    for k in z.files:
        if k.startswith("grad/"):
            g = p[k[5:]].grad
            assert g is not None, k
            assert np.abs(g.numpy() - z[k]).max() <= 1e-12 + (1e-9 + stored) * np.abs(z[k]).max(), k
    # attention intermediates of the reference's einsum calls (large/ours.py:136-143)
    for i in range(cfg.get("trans_num_layers", 1)):
        pr = parts[f"attn{i}"]
        assert np.abs(pr["kvs"].detach().numpy() - z[f"attn{i}/kvs"]).max() <= 1e-14
        assert np.abs(pr["ks_sum"].detach().numpy() - z[f"attn{i}/ks_sum"]).max() <= 1e-14
        n = x.shape[0]
        if f"attn{i}/q_kvs" in z.files:              # (per-node intermediates are not stored at production size)
            q_kvs = (pr["num"] - n * pr["vs"]).detach().numpy()
            assert np.abs(q_kvs - z[f"attn{i}/q_kvs"]).max() <= 1e-10
            assert np.abs((pr["den"].squeeze(-1) - n).detach().numpy() - z[f"attn{i}/q_ks_sum"]).max() <= 1e-10
        # the un-normalised partials libsgf reduces, rescaled, are the reference's kvs / ks_sum
        raw = O.attention_raw_stats(pr["qs"], pr["ks"], pr["vs"]).detach()
        h, d = pr["qs"].shape[1], pr["qs"].shape[2]
        s0 = raw[: h * d * d].reshape(h, d, d) / torch.sqrt(raw[-1])
        assert np.abs(s0.numpy() - z[f"attn{i}/kvs"]).max() <= 1e-13
    # BatchNorm running statistics after the step (momentum 0.1)
    for key, (mu, var_unb) in stats.items():
        rm = 0.9 * p[key + ".running_mean"] + 0.1 * mu
        rv = 0.9 * p[key + ".running_var"] + 0.1 * var_unb
        assert np.abs(rm.numpy() - z["after/" + key + ".running_mean"]).max() <= 1e-12
        assert np.abs(rv.numpy() - z["after/" + key + ".running_var"]).max() <= 1e-12
    # eval mode on the updated running statistics
    pe = {k: v.detach() for k, v in p.items()}
    for k in z.files:
        if k.startswith("after/"):
            pe[k[6:]] = torch.from_numpy(z[k])
    le = O.sgformer_forward(pe, x, ei, cfg, training=False)
    assert np.abs(le.numpy() - z["logits_eval"]).max() <= 1e-12 + stored * np.abs(z["logits_eval"]).max()
    # CSR arrays: the reference's sorted COO (target, source, fp32 value) bit for bit
    if "coo/row" in z.files:
        rowptr, colind, val, deg = O.csr_build(z["edge_index"], x.shape[0])
        assert np.array_equal(np.repeat(np.arange(x.shape[0]), np.diff(rowptr)), z["coo/row"])
        assert np.array_equal(colind, z["coo/col"])
        assert np.array_equal(val.view(np.uint32), z["coo/value"].view(np.uint32))


@pytest.mark.skipif(not ref_shim.reference_available(), reason="/root/reference not mounted")
@pytest.mark.parametrize("cfg", [
    dict(trans_num_layers=1, gnn_num_layers=2, gnn_use_init=True, graph_weight=0.5, trans_use_act=False),
    dict(trans_num_layers=2, trans_num_heads=2, gnn_num_layers=1, aggregate="cat"),
    dict(trans_use_weight=False, trans_num_heads=2, gnn_use_weight=False, gnn_use_bn=False, trans_use_bn=False),
    dict(use_graph=False),
])
def test_oracle_matches_live_reference(cfg):
    ref = ref_shim.load_reference("large")
    torch.set_default_dtype(torch.float64)
    try:
        torch.manual_seed(3)
        n, f, d, c = 211, 18, 16, 5
        m = ref.SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, **cfg).double()
        x = torch.randn(n, f)
        ei = O.synthetic_graph(n, 6.0, seed=8)
        p = {k: v.detach().clone() for k, v in m.state_dict().items()}
        m.train()
        a = m(x, ei)
        b = O.sgformer_forward(p, x, ei, cfg, training=True)
        assert float((a - b).abs().max()) <= 1e-12
        m.eval()
        p = {k: v.detach().clone() for k, v in m.state_dict().items()}
        assert float((m(x, ei) - O.sgformer_forward(p, x, ei, cfg, training=False)).abs().max()) <= 1e-12
    finally:
        torch.set_default_dtype(torch.float32)


@pytest.mark.skipif(not ref_shim.reference_available(), reason="/root/reference not mounted")
@pytest.mark.parametrize("cfg,gcn_layers", [(dict(num_layers=1, alpha=0.5, graph_weight=0.8), 4),
                                            (dict(num_layers=2, num_heads=2, use_weight=False, aggregate="cat"), 2),
                                            (dict(num_layers=1, use_bn=False, use_residual=False), 3)])
def test_medium_oracle_matches_live_reference(cfg, gcn_layers):
    """medium/ours.py and medium/models.py executed unchanged (GCNConv = the restatement in
    oracle/ref_shim.py: torch_geometric is not installed, SURVEY.md App. D vii)."""
    ref = ref_shim.load_reference("medium")
    torch.set_default_dtype(torch.float64)
    try:
        torch.manual_seed(5)
        n, f, d, c = 160, 22, 16, 6
        gnn = ref.models.GCN(f, d, d, num_layers=gcn_layers, dropout=0.0)
        m = ref.SGFormer(f, d, c, dropout=0.0, gnn=gnn, **cfg).double()
        with torch.no_grad():
            for k, v in m.state_dict().items():
                if k.endswith("bias"):
                    v.normal_(0, 0.1)

        class Data:
            pass
        data = Data()
        x = torch.randn(n, f)
        ei = O.synthetic_graph(n, 5.0, seed=9)[:, :-n]
        data.graph = {"node_feat": x, "edge_index": ei}
        for training in (True, False):
            m.train(training)
            p = {k: v.detach().clone() for k, v in m.state_dict().items()}
            with torch.no_grad():
                a = m(data)
            assert float((a - O.medium_forward(p, x, ei, cfg, training=training)).abs().max()) <= 1e-12
    finally:
        torch.set_default_dtype(torch.float32)


@pytest.mark.skipif(not ref_shim.reference_available(), reason="/root/reference not mounted")
@pytest.mark.parametrize("cfg", [dict(num_layers=2, num_heads=1),
                                 dict(num_layers=1, num_heads=2, use_weight=False, graph_weight=0.3, use_source=True),
                                 dict(num_layers=2, num_heads=2, use_graph=False, use_bn=False, use_residual=False)])
def test_difformer_oracle_matches_live_reference(cfg):
    """medium/difformer.py executed unchanged (kernel='simple') against oracle.difformer_forward."""
    ref = ref_shim.load_reference("difformer")
    torch.set_default_dtype(torch.float64)
    try:
        torch.manual_seed(0)
        n, f, d, c = 150, 20, 16, 5
        m = ref.DIFFormer(f, d, c, dropout=0.0, **cfg).double()
        with torch.no_grad():
            for k, v in m.state_dict().items():
                if k.endswith("bias"):
                    v.normal_(0, 0.1)

        class Data:
            pass
        data = Data()
        x = torch.randn(n, f)
        ei = O.synthetic_graph(n, 5.0, seed=3)
        data.graph = {"node_feat": x, "edge_index": ei}
        m.train()
        p = {k: v.detach().clone() for k, v in m.state_dict().items()}
        with torch.no_grad():
            assert float((m(data) - O.difformer_forward(p, x, ei, cfg)).abs().max()) <= 1e-12
    finally:
        torch.set_default_dtype(torch.float32)


def test_csr_oracle_edge_cases():
    # zero in-degree source -> inf -> 0 (large/ours.py:32); duplicates kept; isolated nodes
    ei = np.array([[0, 0, 2, 2, 3], [1, 1, 1, 2, 1]])
    rowptr, colind, val, deg = O.csr_build(ei, 5)
    assert rowptr.tolist() == [0, 0, 4, 5, 5, 5]
    assert colind.tolist() == [0, 0, 2, 3, 2]
    assert deg.tolist() == [0, 4, 1, 0, 0]
    assert val[0] == 0.0 and val[1] == 0.0 and val[3] == 0.0          # sources 0 and 3 have in-degree 0
    assert val[2] == np.float32(np.sqrt(np.float32(1) / np.float32(4))) * np.float32(1.0)
    t_rowptr, t_colind, t_val, sym = O.csr_transpose(ei, 5)
    assert not sym and t_rowptr.tolist() == [0, 2, 2, 4, 5, 5]


def test_tile_pack_oracle_round_trip():
    """oracle/graph_oracle.py tile_pack / tile_unpack (the packed A tiles of csrc/spmm_pack.hip): unpack(pack(tiles)) is
    the plan's dense fragments bit for bit, sparse and dense groups both occur, entries are in image order, and the
    product evaluated through the unpacked tiles is the fp64 SpMM of the CSR (large/ours.py:34)."""
    from oracle import graph_oracle as G
    n = 3000
    ei = O.synthetic_graph(n, 10.0, seed=5).numpy()
    # two planted communities with dense blocks: one beyond the sparse limit, one below it
    rng = np.random.default_rng(0)
    a = np.stack(np.nonzero(rng.random((40, 40)) < 0.9)) + 0
    bq = np.stack(np.nonzero(rng.random((100, 100)) < 0.2)) + 64
    ei = np.concatenate([ei, a, a[::-1], bq, bq[::-1]], axis=1)
    rowptr, colind, val, _ = O.csr_build(ei, n)
    blk = G.tile_blocks(None, n, 128)
    sh_ptr, sh_cols, tile_ptr, tiles, rem_rowptr, rem_col, rem_val, st = G.tile_plan(rowptr, colind, val, n, blk, 256, 2, 1 << 40)
    nfrag = int(tile_ptr[-1])
    grp, pool = G.tile_pack(blk, tile_ptr, tiles)
    ng = nfrag // 2
    assert (grp[:ng, 1] == -1).any() and (grp[:ng, 1] >= 0).any() and (grp[:ng, 1] <= G.TILE_SPARSE_MAX).all()
    sizes = np.where(grp[:ng, 1] < 0, 256, (grp[:ng, 1] + 1) // 2)
    assert np.array_equal(grp[:ng, 0], np.concatenate([[0], np.cumsum(sizes)[:-1]])) and pool.size == sizes.sum() * 16
    g = int(np.nonzero(grp[:ng, 1] > 1)[0][0])
    ent = pool[grp[g, 0] * 16: grp[g, 0] * 16 + grp[g, 1] * 8].view(np.uint32).reshape(-1, 2)
    order = (ent[:, 0] // 2048) * 2048 + ent[:, 0] % 1024                # k-step, then lane / element
    assert (np.diff(order.astype(np.int64)) > 0).all() and ((ent[:, 0] % 2048) < 1024).all()
    back = G.tile_unpack(grp, pool, nfrag, blk, tile_ptr)
    assert np.array_equal(back, np.asarray(tiles, dtype=np.uint16))
    x = rng.standard_normal((n, 8))
    y = G.spmm_tile(blk, sh_ptr, sh_cols, tile_ptr, back, rem_rowptr, rem_col, rem_val, x)
    ref = np.zeros_like(x)
    np.add.at(ref, np.repeat(np.arange(n), np.diff(rowptr)), val.astype(np.float64)[:, None] * x[colind])
    assert np.abs(y - ref).max() <= 1e-4 * np.abs(ref).max()
