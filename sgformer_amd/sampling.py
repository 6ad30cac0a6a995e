"""Device neighbour sampling for the 100M recipe (second half).

`NeighborLoader(data, input_nodes=..., num_neighbors=[15, 10, 5], batch_size=..., shuffle=...)` of
100M/nb-sample.py:125-151 runs PyG's sampler on 12 host workers and ships every batch over PCIe; here the graph
(CSR over target nodes, built once by sgf_csr_build), the node features and the labels stay in HBM (288 GB: the
papers100M features are 57 GB in fp32, 28 GB in bf16) and a batch is sampled, relabelled and gathered by
sgf_neighbor_sample_* / sgf_gather_rows on the GPU.  `sgformer_amd.launch` installs this class as
`torch_geometric.loader.NeighborLoader` for the 100M trainer.

Semantics kept from PyG (replace=False, directed=True): seeds first in the batch's node list, then nodes in order
of first appearance hop by hop; every frontier node receives min(in-degree, fanout) sampled in-neighbours; edges
point neighbour -> node (so the batch graph is DIRECTED and ops.CSRGraph builds a separate transpose for the
backward).  The random stream is this library's own (a counter-based hash of seed / batch / hop / node), not
std::mt19937's: parity is structural (tests/test_gpu_sampler.py, oracle/graph_oracle.py::neighbor_sample).
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence

import torch

from . import _lib, ops


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class SampledBatch:
    """What the trainer reads off a NeighborLoader batch (100M/nb-sample.py:172-175,27-45): x, edge_index, y,
    batch_size, n_id; `.to(device)` is a no-op for tensors that are already there."""

    def __init__(self, x, edge_index, y, batch_size, n_id):
        self.x, self.edge_index, self.y, self.batch_size, self.n_id = x, edge_index, y, int(batch_size), n_id
        self.num_nodes = int(n_id.numel())

    def to(self, device, *args, **kwargs):
        dev = torch.device(device)
        mv = lambda t: t if (t is None or t.device == dev) else t.to(dev)   # noqa: E731
        out = SampledBatch(mv(self.x), mv(self.edge_index), mv(self.y), self.batch_size, mv(self.n_id))
        if out.edge_index is not self.edge_index:
            for attr in ("_sgf_trusted", "_sgf_max_in_degree"):       # what ops.CSRGraph reads off a sampled edge list
                if hasattr(self.edge_index, attr):
                    setattr(out.edge_index, attr, getattr(self.edge_index, attr))
        return out


class _DeviceGraph:
    """What every loader over one Data object shares on the device: the in-neighbour CSR (rowptr / colind only — the
    sampler never reads the normalisation values), the node -> local-id table, the largest in-degree.  100M/nb-sample.py:
    125-151 builds THREE loaders (train / valid / test) from the same Data: at papers100M scale one copy of the CSR is
    13 GB and the H2D copy + sort of the edge list 52 GB of transients — once, not three times."""

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, dev):
        g = ops.CSRGraph(edge_index.to(dev), int(num_nodes))
        self.rowptr, self.colind = g.rowptr, g.colind            # in-neighbours of every node
        self.n = int(num_nodes)
        self.max_deg = int((self.rowptr[1:] - self.rowptr[:-1]).max()) if self.n > 0 else 0
        self.local_of = torch.full((self.n,), torch.iinfo(torch.int32).min, dtype=torch.int32, device=dev)   # not in a batch
        del g                                                    # frees val / deg


# (tensor identity, version, shape, device) -> shared device copies; each entry pins its key tensors so that a recycled
# data_ptr can never alias it.  Small and bounded: a trainer has one Data object.
_shared: "dict[tuple, tuple]" = {}
_SHARED_MAX = 4


def _shared_get(kind: str, key_tensors, extra, build):
    key = (kind, tuple((t.data_ptr(), t._version, tuple(t.shape), str(t.device), t.dtype) for t in key_tensors), extra)
    hit = _shared.get(key)
    if hit is None:
        while len(_shared) >= _SHARED_MAX:
            _shared.pop(next(iter(_shared)))
        hit = _shared[key] = (build(), key_tensors)
    return hit[0]


# above this worst-case entry count a batch is sampled hop by hop (buffers sized from the real frontier) instead of in one call
_ONE_CALL_MAX_ENTRIES = 64 << 20


class NeighborSampler:
    """sample(seeds) -> (n_id int64 [nodes], edge_index int64 [2, edges] in local ids, batch_size).

    Samplers over the same graph share ONE node -> local-id table and one workspace (`_shared_get`): they must be drawn from
    one batch at a time on one stream — which is how 100M/nb-sample.py uses its train / valid / test loaders."""

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, num_neighbors: Sequence[int], seed: int = 0,
                 device: Optional[torch.device] = None):
        dev = torch.device(device) if device is not None else (
            edge_index.device if edge_index.is_cuda else torch.device("cuda", torch.cuda.current_device()))
        dg = _shared_get("graph", (edge_index,), (int(num_nodes), str(dev)), lambda: _DeviceGraph(edge_index, num_nodes, dev))
        self.rowptr, self.colind, self.local_of, self.max_deg = dg.rowptr, dg.colind, dg.local_of, dg.max_deg
        self.n, self.device = int(num_nodes), dev
        self.fanouts = [int(k) for k in num_neighbors]
        self.seed = int(seed) & (2 ** 64 - 1)
        self.batches = 0
        # non-negative fan-outs: the whole batch is one sgf_neighbor_sample_batch call with its hop bookkeeping on the
        # device and ONE host read (the sizes of the views); -1 (all neighbours) has no a-priori capacity and keeps the
        # hop-by-hop path with its read per hop
        self._fan_dev = (torch.tensor(self.fanouts, dtype=torch.int32, device=dev)
                         if self.fanouts and all(k >= 0 for k in self.fanouts) else None)
        self._fan_host = (ctypes.c_int32 * len(self.fanouts))(*self.fanouts)
        self.host_reads = 0

    def sample(self, seeds: torch.Tensor, batch_id: Optional[int] = None):
        dev = self.device
        seeds32 = seeds.to(dev).to(torch.int32).contiguous()
        bs = int(seeds32.numel())
        batch_id = self.batches if batch_id is None else int(batch_id)
        self.batches += 1
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        if self._fan_dev is not None and self._batch_capacity(bs) <= _ONE_CALL_MAX_ENTRIES:
            return self._sample_batch(seeds32, bs, batch_id, st)
        # deep / wide fan-outs: the one-call form sizes every buffer to the worst case bs * (k1 + k1 k2 + ...) BEFORE any
        # de-duplication (GBs per batch for e.g. [25, 20, 15, 10] at 1024 seeds); hop by hop the buffers follow the
        # actual frontier
        nodes = [seeds32]
        srcs, dsts = [], []
        with torch.cuda.device(dev):
            _lib.call("sgf_neighbor_sample_mark", _ptr(self.local_of), _ptr(seeds32), bs, 0, st)
            frontier, local0, n_known = seeds32, 0, bs
            try:
                self._hops(st, nodes, srcs, dsts, frontier, local0, n_known, batch_id)
            except BaseException:
                # a hop failed (workspace, allocation, fan-out overflow): its kernel may have marked nodes that never made
                # it into `nodes` — stale marks would hand out-of-range local ids to the next batch.  Reset the whole table.
                self.local_of.fill_(torch.iinfo(torch.int32).min)
                raise
            # back to "not in a batch" for every id of this batch
            n_id32 = torch.cat(nodes)
            _lib.call("sgf_neighbor_sample_mark", _ptr(self.local_of), _ptr(n_id32), int(n_id32.numel()), -1, st)
        ei = torch.stack([torch.cat(srcs), torch.cat(dsts)]).long() if srcs else torch.zeros(2, 0, dtype=torch.int64, device=dev)
        ei._sgf_trusted = True          # local ids are in range by construction: ops.CSRGraph skips its host check
        return n_id32.long(), ei, bs

    def _batch_capacity(self, bs: int) -> int:
        """Worst-case number of sampled entries of one batch: bs * (k1 + k1 k2 + ...)."""
        total, width = 0, bs
        for k in self.fanouts:
            width *= max(k, 0)
            total += width
        return total

    def _sample_batch(self, seeds32, bs, batch_id, st):
        dev = self.device
        lib = _lib.load()
        ncap, ecap = ctypes.c_int64(0), ctypes.c_int64(0)
        nbytes = lib.sgf_neighbor_sample_batch_workspace_bytes(bs, self._fan_host, len(self.fanouts), ctypes.byref(ncap), ctypes.byref(ecap))
        if nbytes == 0:
            raise RuntimeError("NeighborSampler: batch too large (more than 2^31 sampled entries)")
        ncap, ecap = ncap.value, ecap.value
        nodes = torch.empty(max(ncap, 1), dtype=torch.int32, device=dev)
        e_src = torch.empty(max(ecap, 1), dtype=torch.int32, device=dev)
        e_dst = torch.empty(max(ecap, 1), dtype=torch.int32, device=dev)
        counts = torch.empty(2 + 2 * len(self.fanouts), dtype=torch.int64, device=dev)
        ws = ops._workspace(dev, "nbr_sample", nbytes)
        with torch.cuda.device(dev):
            try:
                _lib.call("sgf_neighbor_sample_batch", _ptr(self.rowptr), _ptr(self.colind), _ptr(seeds32), bs, self._fan_host,
                          len(self.fanouts), ctypes.c_uint64(self.seed), ctypes.c_uint64(batch_id), _ptr(self.local_of),
                          _ptr(nodes), ncap, _ptr(e_src), _ptr(e_dst), ecap, _ptr(counts), _ptr(ws), ws.numel(), st)
            except BaseException:
                self.local_of.fill_(torch.iinfo(torch.int32).min)
                raise
        nn, ne = (int(v) for v in counts[:2].tolist())          # the one host read of the batch
        self.host_reads += 1
        ei = torch.stack([e_src[:ne], e_dst[:ne]]).long()
        ei._sgf_trusted = True
        ei._sgf_max_in_degree = max(self.fanouts)       # every node is a frontier node once: at most fan-out in-edges
        return nodes[:nn].long(), ei, bs

    def _hops(self, st, nodes, srcs, dsts, frontier, local0, n_known, batch_id):
        """The hop-by-hop path (a fan-out of -1 has no a-priori capacity): one sgf_neighbor_sample_hop per hop, its edge and
        node counts read back to size the next hop's buffers."""
        dev = self.device
        for hop, k in enumerate(self.fanouts):
            m = int(frontier.numel())
            if m == 0:
                break
            cap = m * (k if k >= 0 else max(self.max_deg, 1))
            e_src = torch.empty(cap, dtype=torch.int32, device=dev)
            e_dst = torch.empty(cap, dtype=torch.int32, device=dev)
            s_glob = torch.empty(cap, dtype=torch.int32, device=dev)
            new = torch.empty(cap, dtype=torch.int32, device=dev)
            counts = torch.zeros(2, dtype=torch.int64, device=dev)
            nbytes = _lib.load().sgf_neighbor_sample_workspace_bytes(m, cap)
            ws = ops._workspace(dev, "nbr_sample", nbytes)
            _lib.call("sgf_neighbor_sample_hop", _ptr(self.rowptr), _ptr(self.colind), _ptr(frontier), m, local0, k,
                      ctypes.c_uint64(self.seed), ctypes.c_uint64(batch_id), hop, _ptr(self.local_of), n_known, cap,
                      _ptr(e_src), _ptr(e_dst), _ptr(s_glob), _ptr(new), _ptr(counts), _ptr(ws), ws.numel(), st)
            ne, nn = (int(v) for v in counts.tolist())
            self.host_reads += 2        # this one and the edge count inside sgf_neighbor_sample_hop
            srcs.append(e_src[:ne])
            dsts.append(e_dst[:ne])
            frontier, local0 = new[:nn].contiguous(), n_known
            n_known += nn
            nodes.append(frontier)


class NeighborLoader:
    """The part of torch_geometric.loader.NeighborLoader's surface 100M/nb-sample.py:125-151 uses: construction from
    a Data-like object (x, edge_index, y), input_nodes, num_neighbors, batch_size, shuffle; iteration yields batches
    with x / edge_index / y / batch_size (seed rows first); len() = number of batches.  num_workers /
    persistent_workers are accepted and ignored (there are no host workers)."""

    def __init__(self, data, input_nodes=None, num_neighbors=(15, 10, 5), batch_size: int = 1, shuffle: bool = False,
                 num_workers: int = 0, persistent_workers: bool = False, seed: Optional[int] = None,
                 feature_dtype=None, **_ignored):
        dev = torch.device("cuda", torch.cuda.current_device())
        x, y, ei = data.x, data.y, data.edge_index
        n = int(x.shape[0])
        self.sampler = NeighborSampler(ei, n, list(num_neighbors), seed=torch.initial_seed() if seed is None else seed,
                                       device=dev)
        # the device copies of the features and labels are shared by every loader over the same Data (train / valid / test)
        self.x = _shared_get("x", (x,), (str(dev), feature_dtype),
                             lambda: x.to(dev) if feature_dtype is None else x.to(dev).to(feature_dtype))
        self.y = _shared_get("y", (y,), (str(dev),), lambda: y.to(dev)) if y is not None else None
        if input_nodes is None:
            input_nodes = torch.arange(n)
        elif input_nodes.dtype == torch.bool:
            input_nodes = torch.nonzero(input_nodes).squeeze(1)
        self.input_nodes = input_nodes.to(dev).long()
        self.batch_size, self.shuffle = int(batch_size), bool(shuffle)
        self._gen = torch.Generator(device="cpu").manual_seed(int(self.sampler.seed % (2 ** 63)))

    def __len__(self):
        return (int(self.input_nodes.numel()) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        ids = self.input_nodes
        if self.shuffle:
            ids = ids[torch.randperm(int(ids.numel()), generator=self._gen).to(ids.device)]
        for b in range(len(self)):
            seeds = ids[b * self.batch_size:(b + 1) * self.batch_size]
            n_id, ei, bs = self.sampler.sample(seeds)
            x = ops.gather_rows(self.x, n_id)
            y = self.y[n_id] if self.y is not None else None
            yield SampledBatch(x, ei, y, bs, n_id)
