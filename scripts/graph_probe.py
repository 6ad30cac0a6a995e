#!/usr/bin/env python
"""Which part of the step does not survive HIP stream capture?  Captures forward and backward of one piece of the model in
torch.cuda.graph and replays them.    python scripts/graph_probe.py --piece {full,trans,gcn,linear,spmm,combine}
    python scripts/graph_probe.py --mgc after-eager            # the capture of sgformer_amd.graphed after an eager step: fine
    python scripts/graph_probe.py --mgc after-eager --naive    # the same through the module's own parameters: segfault (r05)"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgformer_amd import _lib, batching, graphed, ops, synth  # noqa: E402
from sgformer_amd.ours import SGFormer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--piece", default="full")
    ap.add_argument("--log", action="store_true")
    ap.add_argument("--fwd-only", action="store_true")
    ap.add_argument("--mgc", choices=["direct", "after-eager", "after-eager-step"], default=None,
                    help="capture through sgformer_amd.graphed._capture, optionally after an eager training step")
    ap.add_argument("--naive", action="store_true",
                    help="with --mgc: torch.cuda.make_graphed_callables on the module itself (its parameters' own gradient "
                         "accumulators, warm-up on a side stream) — after an eager step this crashes in hipStreamEndCapture")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    n, f, c, d, m = 30000, 100, 47, 64, 6144
    ei = synth.synthetic_graph(n, 14.0, seed=11)
    x, _, _ = synth.synthetic_task(n, f, c, seed=11)
    x = x.to(dev)
    torch.manual_seed(5)
    model = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, compute_dtype=torch.bfloat16,
                     **synth.RECIPES["ogbn-products"]).to(dev).train()
    idx = torch.randperm(n)[:m]
    ei_i, _ = batching.subgraph(idx, ei, num_nodes=n, relabel_nodes=True)
    xi = x[idx.to(dev)]
    nnz = int(ei_i.shape[1])
    graph = graphed.StaticCSR(m, nnz + 1024, dev, 0)
    graph.load(*ei_i._sgf_csr[:3])
    xc = model._entry_copy_uncached(xi, torch.bfloat16)
    h = torch.randn(m, d, device=dev).bfloat16().requires_grad_(True)
    h2 = torch.randn(m, d, device=dev).bfloat16().requires_grad_(True)
    params = [p for p in model.parameters()]
    pieces = {
        "full": (lambda: model._core(model._entry_copy_uncached(xi, torch.bfloat16), graph, None, None, torch.float32), params),
        "trans": (lambda: model.trans_conv(xc), list(model.trans_conv.parameters())),
        "gcn": (lambda: model.graph_conv(xc, graph), list(model.graph_conv.parameters())),
        "linear": (lambda: ops.linear(h, model.trans_conv.fcs[0].weight.new_zeros(d, d) + 0.01, None), [h]),
        "spmm": (lambda: ops.spmm(graph, h), [h]),
        "combine": (lambda: ops.combine_fc(h, h2, model.fc.weight, model.fc.bias, 0.5, 0.5), [h, h2, model.fc.weight, model.fc.bias]),
    }
    fn, wrt = pieces[a.piece]
    if a.mgc:
        if a.mgc != "direct":
            out = model(xi, ei_i)
            out.float().sum().backward()
            if a.mgc == "after-eager-step":
                opt = torch.optim.Adam(model.parameters(), lr=0.01)
                opt.step()
                opt.zero_grad()
            torch.cuda.synchronize()
            print("eager step ok", flush=True)
        if a.naive:
            core = graphed._Core(model, graph, torch.bfloat16, torch.float32).train()
            fn = torch.cuda.make_graphed_callables(core, (xi.detach(),), allow_unused_input=True)
            print("make_graphed_callables ok", flush=True)
            out = fn(xi)
        else:
            entry = graphed._Entry()
            graphed._capture(model, entry, xi, ei_i, torch.bfloat16, torch.float32)
            print("make_graphed_callables ok", flush=True)
            out = entry.core(xi, *entry.params)
        out.float().sum().backward()
        torch.cuda.synchronize()
        print("replayed through autograd:", float(out.detach().float().abs().sum()), flush=True)
        return
    if a.log:
        orig = _lib.call

        def logged(name, *args):
            print("   call", name, "capturing" if torch.cuda.is_current_stream_capturing() else "", flush=True)
            return orig(name, *args)
        _lib.call = logged
        import sgformer_amd.kernels as kmod
        kmod._lib.call = logged
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(2):
            out = fn()
            torch.autograd.grad(out, wrt, torch.ones_like(out), allow_unused=True)
    torch.cuda.synchronize()
    print("warm ok", flush=True)
    g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        out = fn()
    print("forward captured", flush=True)
    go = torch.ones_like(out)
    if not a.fwd_only:
        with torch.cuda.graph(g2, pool=g1.pool()):
            grads = torch.autograd.grad(out, wrt, go, allow_unused=True)
        print("backward captured", flush=True)
    g1.replay()
    if not a.fwd_only:
        g2.replay()
    torch.cuda.synchronize()
    print("replayed:", a.piece, float(out.float().abs().sum()), flush=True)


if __name__ == "__main__":
    main()
