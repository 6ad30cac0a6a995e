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


def _run(mode, script, args, tmp_path):
    env = dict(os.environ, SGF_FAKE_OGB="600,6,24,5", PYTHONHASHSEED="0")
    p = subprocess.run([sys.executable, os.path.join(HERE, "run_trainer.py"), mode,
                        os.path.join(REF, "large", script)] + args,
                       cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
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
