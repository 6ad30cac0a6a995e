"""The drop-in chain on the GPU box: `python -m sgformer_amd.launch <trainer>` with the HIP kernel table.

tests/test_trainer_unchanged.py runs the REAL reference trainers, but only where /root/reference is mounted
(the build container, CPU kernel table).  Here a committed stand-in trainer that binds to the model the way
large/main.py does (tests/standins/trainers/large/main_standin.py) is driven through launch.main() on the
MI355X: sys.modules['ours'] registration, runpy, the patched graph prologue (N2, on the device), the HIP
kernels, Adam, train / eval switching.  Its printed loss curve is compared with the fp32 CPU oracle run from
the same initial state_dict and the same inputs (large/main.py:125-143 semantics)."""
import json
import os
import subprocess
import sys

import pytest
import torch

from oracle import graph_oracle as G
from oracle import sgformer_oracle as O

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
TRAINER = os.path.join(HERE, "standins", "trainers", "large", "main_standin.py")


def test_launcher_drives_a_trainer_on_the_hip_path(cuda, tmp_path):
    dump = str(tmp_path / "init.pt")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(HERE, "standins"), ROOT]))
    p = subprocess.run([sys.executable, "-m", "sgformer_amd.launch", TRAINER, "--epochs", "4", "--dump", dump],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "STANDIN_PROLOGUE_DEVICE cuda" in p.stdout              # the prologue ran on the device (launch.patch_prologue)
    log = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("STANDIN_LOG ")][0][12:])
    assert len(log) == 4

    init = torch.load(dump)
    ei, x, y, split = init["edge_index"], init["x"], init["y"].squeeze(1), init["split"]
    n = x.shape[0]
    # N2: the device prologue produced what the host PyG calls would have
    g = torch.Generator().manual_seed(123)
    raw = torch.randint(0, n, (2, 6 * n), generator=g)
    assert torch.equal(ei, torch.from_numpy(G.graph_prologue(raw.numpy(), n, undirected=True)))
    cfg = dict(trans_num_layers=1, trans_num_heads=1, trans_use_bn=True, trans_use_residual=True,
               trans_use_weight=True, trans_use_act=False, gnn_num_layers=3, gnn_use_bn=True,
               gnn_use_residual=True, gnn_use_weight=True, gnn_use_init=False, gnn_use_act=True,
               graph_weight=0.5, aggregate="add")
    pc = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in init["state"].items()
          if "num_batches_tracked" not in k}
    names1 = [k for k in pc if k.startswith("trans_conv.")]
    names2 = [k for k in pc if not k.startswith("trans_conv.") and pc[k].requires_grad]
    opt = torch.optim.Adam([{"params": [pc[k] for k in names1]}, {"params": [pc[k] for k in names2]}], lr=0.01)
    idx = split["train"]
    for step in range(4):
        opt.zero_grad()
        stats = {}
        loss = O.nll_loss(O.sgformer_forward(pc, x, ei, cfg, training=True, bn_stats=stats), y, idx)
        loss.backward()
        opt.step()
        with torch.no_grad():
            for key, (mu, vu) in stats.items():
                pc[key + ".running_mean"].mul_(0.9).add_(0.1 * mu)
                pc[key + ".running_var"].mul_(0.9).add_(0.1 * vu)
        assert abs(float(loss) - log[step]["loss"]) <= 5e-4 * max(1.0, abs(float(loss))), (step, float(loss), log[step])
    with torch.no_grad():
        pred = O.sgformer_forward({k: v.detach() for k, v in pc.items()}, x, ei, cfg, training=False).argmax(1)
    acc = float((pred[split["test"]] == y[split["test"]]).float().mean()) * 100
    assert abs(acc - log[-1]["test"]) <= 1.5, (acc, log[-1])


def test_launcher_bf16_and_loss_switches(cuda, tmp_path):
    """`--sgf-dtype bf16` (bf16 activation storage: the streaming row kernels, stems, fused head) and `--sgf-aten-loss 1`
    (ATen's nll_loss instead of the gather form) through the same stand-in trainer: both runs finish, the bf16 loss curve
    follows the fp32 one to bf16 accuracy, the fp32 curve does not depend on which nll_loss served it."""
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(HERE, "standins"), ROOT]))

    def run(*extra):
        p = subprocess.run([sys.executable, "-m", "sgformer_amd.launch", TRAINER, "--epochs", "3", "--dump",
                            str(tmp_path / "i.pt"), *extra], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
        return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("STANDIN_LOG ")][0][12:])

    f32 = run()
    aten = run("--sgf-aten-loss", "1")
    bf16 = run("--sgf-dtype", "bf16")
    for a, b in zip(f32, aten):
        assert abs(a["loss"] - b["loss"]) <= 2e-5 * max(1.0, abs(a["loss"])), (a, b)
    for a, b in zip(f32, bf16):
        assert abs(a["loss"] - b["loss"]) <= 3e-2 * max(1.0, abs(a["loss"])), (a, b)


def test_launcher_host_fallbacks_and_100m_prologue(cuda, tmp_path):
    """`--sgf-host-subgraph 1` must leave the prologue on the host too (PyG's host subgraph() indexes a CPU
    mask with edge_index), and for the 100M trainer the device prologue hands its result back on the INPUT tensor's
    device (100M/nb-sample.py:81-133 builds a host Data object for NeighborLoader workers).  Same loss curve either way."""
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(HERE, "standins"), ROOT]))

    def run(*extra):
        p = subprocess.run([sys.executable, "-m", "sgformer_amd.launch", *extra, TRAINER, "--epochs", "2"], cwd=ROOT,
                           env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
        dev = [ln for ln in p.stdout.splitlines() if ln.startswith("STANDIN_PROLOGUE_DEVICE ")][0].split()[1]
        return dev, json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("STANDIN_LOG ")][0][12:])

    d0, base = run()
    d1, host = run("--sgf-host-subgraph", "1")
    d2, m100 = run("--sgf-variant", "100M")
    assert (d0, d1, d2) == ("cuda", "cpu", "cpu")
    for a, b in zip(base, host):
        assert abs(a["loss"] - b["loss"]) <= 2e-5 * max(1.0, abs(a["loss"])), (a, b)
    assert all(torch.isfinite(torch.tensor(e["loss"])) for e in m100)
