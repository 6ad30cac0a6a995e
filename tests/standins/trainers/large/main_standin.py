"""A stand-in for the reference's full-graph trainer (large/main.py) — TEST INFRASTRUCTURE.

/root/reference is not mounted on the GPU box, so tests/test_gpu_launch.py cannot run the real trainer
there.  This script binds to the model EXACTLY the way large/main.py does — `from ours import *` resolved
by import (large/parse.py:2), the constructor keywords of large/parse.py:36-39, the third-party prologue
calls of large/main.py:75-79, the two Adam parameter groups of :115-118, log_softmax + NLLLoss on the
training rows (:139-141), eval-mode accuracy every epoch — on a synthetic graph, and prints the trainer's
log line.  `python -m sgformer_amd.launch <this file> ...` therefore exercises the whole drop-in chain
(sys.modules['ours'] + runpy + patched prologue + HIP kernels) on a GPU box.  It is not a copy of the
reference file: data loading, logging, argument parsing and the baselines are absent.
"""
import argparse
import json

import torch
import torch.nn.functional as F
from torch_geometric.utils import to_undirected, remove_self_loops, add_self_loops   # patched by the launcher

from ours import *   # noqa: F401,F403  (the name the reference resolves its model through)

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=3000)
ap.add_argument("--hidden_channels", type=int, default=64)
ap.add_argument("--epochs", type=int, default=4)
ap.add_argument("--seed", type=int, default=123)
ap.add_argument("--dump", default="")
args = ap.parse_args()

device = torch.device("cuda:0")
torch.manual_seed(args.seed)
n, f, c = args.nodes, 20, 5
g = torch.Generator().manual_seed(args.seed)
raw = torch.randint(0, n, (2, 6 * n), generator=g)                 # directed pairs with duplicates and self-loops
x = torch.randn(n, f, generator=g)
y = torch.randint(0, c, (n, 1), generator=g)
perm = torch.randperm(n, generator=g)
split = {"train": perm[: n // 2], "valid": perm[n // 2: 3 * n // 4], "test": perm[3 * n // 4:]}

edge_index = to_undirected(raw)
edge_index, _ = remove_self_loops(edge_index)
edge_index, _ = add_self_loops(edge_index, num_nodes=n)
print("STANDIN_PROLOGUE_DEVICE", edge_index.device.type)   # where the (possibly patched) prologue left its result
edge_index, x = edge_index.to(device), x.to(device)

model = SGFormer(f, args.hidden_channels, c, trans_num_layers=1, trans_num_heads=1, trans_dropout=0.0,   # noqa: F405
                 trans_use_bn=True, trans_use_residual=True, trans_use_weight=True, trans_use_act=False,
                 gnn_num_layers=3, gnn_dropout=0.0, gnn_use_weight=True, gnn_use_init=False, gnn_use_bn=True,
                 gnn_use_residual=True, gnn_use_act=True, use_graph=True, graph_weight=0.5,
                 aggregate="add").to(device)
model.reset_parameters()
if args.dump:
    torch.save({"state": {k: v.cpu() for k, v in model.state_dict().items()}, "edge_index": edge_index.cpu(),
                "x": x.cpu(), "y": y, "split": split}, args.dump)
optimizer = torch.optim.Adam([{"params": model.params1, "weight_decay": 0.0},
                              {"params": model.params2, "weight_decay": 0.0}], lr=0.01)
criterion = torch.nn.NLLLoss()
train_idx = split["train"].to(device)
log = []
for epoch in range(args.epochs):
    model.train()
    optimizer.zero_grad()
    out = F.log_softmax(model(x, edge_index), dim=1)
    loss = criterion(out[train_idx], y.squeeze(1).to(device)[train_idx])
    loss.backward()
    optimizer.step()
    model.eval()
    with torch.no_grad():
        pred = model(x, edge_index).argmax(dim=-1, keepdim=True).cpu()
    acc = {k: float((pred[v] == y[v]).float().mean()) * 100 for k, v in split.items()}
    log.append({"epoch": epoch, "loss": float(loss), **acc})
    print(f"Epoch: {epoch:02d}, Loss: {float(loss):.4f}, Train: {acc['train']:.2f}%, Valid: {acc['valid']:.2f}%, "
          f"Test: {acc['test']:.2f}%")
print("STANDIN_LOG " + json.dumps(log))
print("STANDIN_EDGE_DEVICE", edge_index.device.type)
