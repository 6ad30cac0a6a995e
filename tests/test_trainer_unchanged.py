"""The drop-in boundary, end to end: the reference's trainer scripts run byte-for-byte unchanged
on top of sgformer_amd, and print the same losses / accuracies as with the
reference's own `ours.py`.

Needs /root/reference (skipped on the GPU box, where it is not mounted).  No GPU here, so the
drop-in runs on the CPU kernel table of tests/cpu_kernels.py: what is under test is the module
surface the trainers bind against — constructor keywords from parse.py, params1/params2,
reset_parameters, train/eval switching, state moving between devices, indexing of the outputs.
"""
import os
import re
import subprocess
import sys

import pytest

from oracle import ref_shim

HERE = os.path.dirname(os.path.abspath(__file__))
REF = ref_shim.REFERENCE_ROOT
pytestmark = pytest.mark.skipif(not ref_shim.reference_available(), reason="/root/reference not mounted")

ARXIV = ("--method sgformer --dataset ogbn-arxiv --metric acc --lr 0.001 --hidden_channels 32 --use_graph "
         "--graph_weight 0.5 --gnn_num_layers 3 --gnn_dropout 0. --gnn_weight_decay 0. --gnn_use_residual "
         "--gnn_use_weight --gnn_use_bn --gnn_use_act --trans_num_layers 1 --trans_dropout 0. "
         "--trans_weight_decay 0. --trans_use_residual --trans_use_weight --trans_use_bn --seed 123 --runs 1 "
         "--epochs 4 --eval_step 1 --display_step 1 --cpu").split()
BATCH = ("--method sgformer --dataset ogbn-arxiv --metric acc --lr 0.01 --hidden_channels 32 "
         "--gnn_num_layers 2 --gnn_dropout 0. --gnn_weight_decay 0. --gnn_use_residual --gnn_use_weight "
         "--gnn_use_bn --gnn_use_init --gnn_use_act --trans_num_layers 1 --trans_dropout 0. "
         "--trans_weight_decay 0. --trans_use_residual --trans_use_weight --trans_use_bn --use_graph "
         "--graph_weight 0.5 --batch_size 250 --seed 123 --runs 1 --epochs 3 --eval_step 1 --display_step 1 "
         "--cpu").split()


def _run(mode, script, args, tmp_path, folder="large", env_extra=None, pattern=None):
    env = {**os.environ, "SGF_FAKE_OGB": "600,6,24,5", "PYTHONHASHSEED": "0", **(env_extra or {})}
    p = subprocess.run([sys.executable, os.path.join(HERE, "run_trainer.py"), mode,
                        os.path.join(REF, folder, script)] + args,
                       cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    if pattern is not None:
        rows = re.findall(pattern, p.stdout)
        assert rows, p.stdout[-2000:]
        return rows, p.stdout
    rows = re.findall(r"Epoch: (\d+), Loss: ([\d.]+), Train: ([\d.]+)%, Valid: ([\d.]+)%, Test: ([\d.]+)%",
                      p.stdout)
    assert rows, p.stdout[-2000:]
    return [(int(e), float(l), float(a), float(b), float(c)) for e, l, a, b, c in rows], p.stdout


@pytest.mark.parametrize("script,args", [("main.py", ARXIV), ("main-batch.py", BATCH)])
def test_reference_trainer_runs_unchanged(script, args, tmp_path):
    ref, ref_out = _run("reference", script, args, tmp_path)
    ours, out = _run("ours", script, args, tmp_path)
    assert "SGF_OURS_MODULE sgformer_amd.ours" in out                    # the drop-in served `ours`
    assert f"SGF_OURS_MODULE {os.path.join(REF, 'large', 'ours.py')}" in ref_out
    assert len(ref) == len(ours) and len(ref) >= 3
    for (e0, l0, *acc0), (e1, l1, *acc1) in zip(ref, ours):
        assert e0 == e1
        assert abs(l0 - l1) <= 2e-3, (ref, ours)            # printed with 4 decimals
        assert max(abs(a - b) for a, b in zip(acc0, acc1)) <= 1.0, (ref, ours)   # <= a few nodes of 150-300


PAPERS = ("--dataset ogbn-papers100M --method ours --lr 0.001 --num_layers 3 --hidden_channels 32 --dropout 0. "
          "--weight_decay 1e-5 --use_residual --use_weight --use_bn --use_init --use_act --ours_layers 1 --ours_dropout 0. "
          "--ours_use_residual --ours_use_weight --ours_use_bn --use_graph --graph_weight 0.8 --batch_size 60 --seed 123 "
          "--runs 1 --epochs 3 --display_step 1 --device 0").split()


def test_100m_trainer_runs_unchanged(tmp_path):
    """100M/nb-sample.py (the papers100M recipe of 100M/run.sh:2-6 at hidden 32 on a synthetic graph) byte-for-byte
    unchanged under sgformer_amd.launch — variant `100M`: sgformer_amd.ours_100m serves `from ours import *`
    (100M/parse.py:1,4-9: the alpha keyword, its own argument order), the prologue patch keeps edge_index on the Data's
    device, CrossEntropyLoss on the seed rows, three NeighborLoaders over one Data — next to the reference's own
    100M/ours.py on the same batches (both runs sample with the host stand-in loader, tests/standins/torch_geometric/
    loader.py: `--sgf-host-sampler 1`; the device sampler has its own tests, tests/test_gpu_sampler.py).  The trainer
    hard-codes a CUDA device; SGF_CUDA_AS_CPU=1 (tests/run_trainer.py) maps that name to the CPU in this GPU-less container."""
    pat = r"Epoch: (\d+) Loss: ([\d.]+) Valid acc: ([\d.]+)% Test acc: ([\d.]+)%"
    env = {"SGF_CUDA_AS_CPU": "1", "SGF_FAKE_OGB": "500,8,24,5"}
    ref, ref_out = _run("reference", "nb-sample.py", PAPERS, tmp_path, "100M", env, pat)
    ours, out = _run("ours", "nb-sample.py", ["--sgf-host-sampler", "1"] + PAPERS, tmp_path, "100M", env, pat)
    assert "SGF_OURS_MODULE sgformer_amd.ours_100m" in out
    assert f"SGF_OURS_MODULE {os.path.join(REF, '100M', 'ours.py')}" in ref_out
    assert len(ref) == len(ours) == 3
    for (e0, l0, v0, t0), (e1, l1, v1, t1) in zip(ref, ours):
        assert e0 == e1
        assert abs(float(l0) - float(l1)) <= 5e-3 * max(1.0, float(l0)), (ref, ours)   # summed over the epoch's batches
        assert abs(float(v0) - float(v1)) <= 1.0 and abs(float(t0) - float(t1)) <= 1.0, (ref, ours)


CORA = ("--backbone gcn --dataset cora --lr 0.01 --num_layers 2 --hidden_channels 32 --weight_decay 5e-4 --dropout 0. "
        "--method ours --ours_layers 1 --use_graph --graph_weight 0.8 --ours_dropout 0. --use_residual --alpha 0.5 "
        "--ours_weight_decay 0.001 --rand_split_class --valid_num 100 --test_num 150 --no_feat_norm --seed 123 --runs 1 "
        "--epochs 4 --display_step 1 --cpu").split()


@pytest.mark.parametrize("method", ["ours", "difformer"])
def test_medium_trainer_runs_unchanged(method, tmp_path):
    """medium/main.py (the Cora recipe of medium/run.sh:2-7 at hidden 32, BASELINE.json config 1's trainer) byte-for-byte
    unchanged under sgformer_amd.launch — variant `medium`: sgformer_amd.ours_medium serves `from ours import *`
    (medium/parse.py:2,97-101: SGFormer(..., gnn=...)), `models.GCN` (the injected GNN branch, medium/parse.py:99 ->
    medium/models.py:14-63) is the libsgf one, `--method difformer` resolves to sgformer_amd.difformer — next to the
    reference's own modules on the same synthetic Planetoid-shaped graph (tests/standins/torch_geometric/datasets.py)."""
    args = [a if a != "ours" or i == 0 else a for i, a in enumerate(CORA)]
    args[args.index("--method") + 1] = method
    pat = r"Epoch: (\d+), Loss: ([\d.]+), Train: ([\d.]+)%, Valid: ([\d.]+)%, Test: ([\d.]+)%"
    ref, ref_out = _run("reference", "main.py", args + ["--data_dir", str(tmp_path) + "/"], tmp_path, "medium", None, pat)
    ours, out = _run("ours", "main.py", args + ["--data_dir", str(tmp_path) + "/"], tmp_path, "medium", None, pat)
    assert "SGF_OURS_MODULE sgformer_amd.ours_medium" in out
    assert f"SGF_OURS_MODULE {os.path.join(REF, 'medium', 'ours.py')}" in ref_out
    assert len(ref) == len(ours) >= 4
    for (e0, l0, *acc0), (e1, l1, *acc1) in zip(ref, ours):
        assert e0 == e1
        assert abs(float(l0) - float(l1)) <= 2e-3, (ref, ours)
        assert max(abs(float(a) - float(b)) for a, b in zip(acc0, acc1)) <= 1.0, (ref, ours)
