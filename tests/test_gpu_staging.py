"""sgformer_amd/staging.py: the mini-batch trainer's host -> device lines (large/main-batch.py:134-146) off the compute stream.
Same values as the plain lines, the host does not wait for work queued on the compute stream, and an epoch of the trainer's
loop gives bit-identical parameters with and without the prep stream."""
import time

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def test_wrappers_return_what_the_plain_lines_return(cuda, monkeypatch):
    from sgformer_amd import staging
    monkeypatch.setenv("SGF_PREP_STREAM", "1")
    g = torch.Generator().manual_seed(0)
    n = 5000
    x = torch.randn(n, 100, generator=g).to(cuda)
    lab = torch.randint(0, 47, (n,), generator=g)
    idx = torch.randperm(n, generator=g)[:1200]                       # HOST index, as the trainer's
    mask = torch.zeros(1200, dtype=torch.bool)
    mask[::3] = True
    xr = staging.resident(x)
    ls = staging.staged(lab).unsqueeze(1)                               # large/main-batch.py:45-46
    assert isinstance(xr, staging.ResidentRows) and isinstance(ls, staging.StagedHost) and not ls.is_cuda
    x_i = xr[idx].to(cuda)                                              # :138
    assert type(x_i) is torch.Tensor and torch.equal(x_i, x[idx.to(cuda)])
    y_i = ls[idx].to(cuda)                                              # :141
    assert y_i.is_cuda and torch.equal(y_i.as_subclass(torch.Tensor), lab.unsqueeze(1)[idx].to(cuda))
    picked = y_i.squeeze(1)[mask]                                       # :149 — a HOST boolean mask on a device tensor
    assert type(picked) is torch.Tensor and torch.equal(picked, lab[idx][mask].to(cuda))
    # everything else is the plain tensor's behaviour
    assert type(xr + 1) is torch.Tensor and type(xr[idx.to(cuda)]) is torch.Tensor and type(xr[3:9]) is torch.Tensor
    assert int(ls.max().item()) == int(lab.max()) and type(ls.float()) is torch.Tensor
    assert torch.equal(F.one_hot(ls.squeeze(1), 47), F.one_hot(lab, 47))
    assert ls.to(torch.float64).dtype == torch.float64 and not ls.to("cpu").is_cuda
    monkeypatch.setenv("SGF_PREP_STREAM", "0")                         # switched off: the same values on the current stream
    assert torch.equal(xr[idx], x[idx.to(cuda)]) and torch.equal(ls[idx].to(cuda).as_subclass(torch.Tensor), y_i.as_subclass(torch.Tensor))


def test_the_host_does_not_wait_for_the_compute_stream(cuda, monkeypatch):
    """~100 ms of matrix products queued on the current stream, then the trainer's gather lines: with the prep stream the host
    is through them long before the compute stream has drained, and what the compute stream then reads is right."""
    from sgformer_amd import batching, staging, synth
    monkeypatch.setenv("SGF_PREP_STREAM", "1")
    g = torch.Generator().manual_seed(1)
    n = 200000
    x = staging.resident(torch.randn(n, 100, generator=g).to(cuda))
    lab = staging.staged(torch.randint(0, 47, (n, 1), generator=g))
    ei = synth.synthetic_graph(n, 10.0, seed=2)
    idx = torch.randperm(n, generator=g)[:50000]
    batching._parents.clear()
    batching.subgraph(idx, ei, num_nodes=n, relabel_nodes=True)        # (the parent CSR is built once)
    a = torch.randn(8192, 8192, device=cuda)
    torch.cuda.synchronize()

    def busy():
        b = a
        for _ in range(12):
            b = (b @ a) * 1e-4
        return b

    t0 = time.perf_counter()
    busy()
    torch.cuda.synchronize()
    t_busy = time.perf_counter() - t0
    busy()
    t0 = time.perf_counter()
    x_i = x[idx].to(cuda)
    y_i = lab[idx].to(cuda)
    ei_i, _ = batching.subgraph(idx, ei, num_nodes=n, relabel_nodes=True)
    t_host = time.perf_counter() - t0
    out = (x_i.sum(1) + y_i.squeeze(1).float())                         # consumed on the compute stream, behind the products
    torch.cuda.synchronize()
    assert t_busy > 0.03, t_busy
    assert t_host < 0.5 * t_busy, (t_host, t_busy)
    ref = x.as_subclass(torch.Tensor)[idx.to(cuda)].sum(1) + lab.as_subclass(torch.Tensor)[idx].to(cuda).squeeze(1).float()
    assert torch.equal(out, ref)
    monkeypatch.setenv("SGF_PREP_STREAM", "0")
    ei_ref, _ = batching.subgraph(idx, ei, num_nodes=n, relabel_nodes=True)
    assert torch.equal(ei_i, ei_ref) and all(torch.equal(p, q) for p, q in zip(ei_i._sgf_csr, ei_ref._sgf_csr))
    batching._parents.clear()


@pytest.mark.parametrize("graphs", [False, True], ids=["eager", "replayed"])
def test_epoch_is_bit_identical_with_and_without_the_prep_stream(cuda, monkeypatch, graphs):
    """two epochs of large/main-batch.py:129-151's loop lines (host index, host masks, resident features, NLLLoss on masked
    rows under the launcher's loss patch): parameters and per-batch losses equal bit for bit, prep stream on and off."""
    from sgformer_amd import batching, launch, ops, staging, synth
    from sgformer_amd.ours import SGFormer
    monkeypatch.setenv("SGF_BATCH_GRAPH", "1" if graphs else "0")
    n, f, c, d, bs = 60000, 100, 47, 64, 8192
    ei = synth.synthetic_graph(n, 12.0, seed=5)
    x0, y0, train_idx = synth.synthetic_task(n, f, c, seed=5)
    train_mask = torch.zeros(n, dtype=torch.bool)
    train_mask[train_idx] = True

    def run(prep: bool):
        monkeypatch.setenv("SGF_PREP_STREAM", "1" if prep else "0")
        batching._parents.clear()
        x = staging.resident(x0.to(cuda))
        true_label = staging.staged(y0.clone()).unsqueeze(1)
        torch.manual_seed(2)
        model = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, compute_dtype=torch.bfloat16,
                         **synth.RECIPES["ogbn-products"]).to(cuda)
        opt = torch.optim.Adam(model.parameters(), weight_decay=1e-5, lr=0.01)
        criterion = nn.NLLLoss()
        gen = torch.Generator().manual_seed(7)
        losses = []
        launch.patch_nll_loss()
        try:
            for _ in range(2):
                model.train()
                idx = torch.randperm(n, generator=gen)
                for i in range(n // bs + 1):
                    idx_i = idx[i * bs:(i + 1) * bs]
                    train_mask_i = train_mask[idx_i]
                    x_i = x[idx_i].to(cuda)
                    ei_i, _ = batching.subgraph(idx_i, ei, num_nodes=n, relabel_nodes=True)
                    ei_i = ei_i.to(cuda)
                    y_i = true_label[idx_i].to(cuda)
                    opt.zero_grad()
                    out_i = F.log_softmax(model(x_i, ei_i), dim=1)
                    loss = criterion(out_i[train_mask_i], y_i.squeeze(1)[train_mask_i])
                    loss.backward()
                    opt.step()
                    losses.append(loss.detach().clone())
        finally:
            launch.unpatch_nll_loss()
        torch.cuda.synchronize()
        state = {k: v.detach().clone() for k, v in model.state_dict().items()}
        ops.graph_cache.clear()
        return torch.stack(losses), state

    l0, s0 = run(False)
    l1, s1 = run(True)
    assert torch.equal(l0, l1), float((l0 - l1).abs().max())
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k
