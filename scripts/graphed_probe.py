#!/usr/bin/env python
"""Host / device time of a replayed mini-batch forward and backward, step by step (sgformer_amd/graphed.py)."""
import os, sys, time
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgformer_amd import batching, graphed, launch, ops, synth
from sgformer_amd.ours import SGFormer

dev = torch.device("cuda:0")
n, f, c, d, m = 400000, 100, 47, 256, 100000
ei = synth.synthetic_graph(n, 51.5, seed=1, device=dev)
x, y, _ = synth.synthetic_task(n, f, c, seed=1)
x, y = x.to(dev), y.to(dev)
torch.manual_seed(0)
model = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, compute_dtype=torch.bfloat16, **synth.RECIPES["ogbn-products"]).to(dev)
model.logits_dtype = torch.float32
launch.patch_adam()
opt = torch.optim.Adam(model.parameters(), lr=0.01)
gen = torch.Generator().manual_seed(3)
for step in range(8):
    idx = torch.randperm(n, generator=gen)[:m].to(dev)
    ei_i, _ = batching.subgraph(idx, ei, num_nodes=n, relabel_nodes=True)
    xi = x[idx]
    yi = y[idx]
    torch.cuda.synchronize()
    c0 = dict(graphed.counters)
    t0 = time.perf_counter()
    opt.zero_grad()
    out = model(xi, ei_i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    loss = F.nll_loss(F.log_softmax(out.float(), dim=1), yi)
    loss.backward()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    opt.step()
    torch.cuda.synchronize()
    print(f"step {step}: fwd issue {1e3*(t1-t0):.2f} ms, fwd done {1e3*(t2-t0):.2f}; bwd issue {1e3*(t3-t2):.2f}, done {1e3*(t4-t2):.2f}; "
          f"counters +{ {k: graphed.counters[k]-c0[k] for k in c0} } loss {float(loss):.5f}", flush=True)
