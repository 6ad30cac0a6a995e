"""torch_geometric.loader.NeighborLoader — the HOST neighbour sampler of 100M/nb-sample.py:125-151, restated for the
tests (../README.md): `replace=False, directed=True` semantics of PyG 2.x — the seeds lead the batch's node list, new
nodes follow in order of first appearance hop by hop, every frontier node receives min(in-degree, fanout) distinct
in-neighbours, edges point neighbour -> node in batch-local ids; batches carry x, edge_index, y, batch_size, n_id.
num_workers / persistent_workers are accepted and ignored.  The draws come from a numpy generator seeded from torch's
seed, so two runs of one trainer see the same batches."""
import numpy as np
import torch

from .data import Data


class NeighborLoader:
    def __init__(self, data, input_nodes=None, num_neighbors=(15, 10, 5), batch_size=1, shuffle=False, num_workers=0,
                 persistent_workers=False, **_ignored):
        self.data = data
        n = data.num_nodes
        ei = data.edge_index.cpu().numpy()
        order = np.lexsort((ei[0], ei[1]))                       # by (target, source)
        self.colind = ei[0][order]
        self.rowptr = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(np.bincount(ei[1], minlength=n), out=self.rowptr[1:])
        if input_nodes is None:
            input_nodes = torch.arange(n)
        elif input_nodes.dtype == torch.bool:
            input_nodes = torch.nonzero(input_nodes).squeeze(1)
        self.input_nodes = input_nodes.cpu().numpy().astype(np.int64)
        self.fanouts, self.batch_size, self.shuffle = [int(k) for k in num_neighbors], int(batch_size), bool(shuffle)
        self.rng = np.random.default_rng(int(torch.initial_seed()) % (2 ** 32) + len(self.input_nodes))

    def __len__(self):
        return (len(self.input_nodes) + self.batch_size - 1) // self.batch_size

    def _sample(self, seeds):
        n_id = [int(v) for v in seeds]
        local = {g: i for i, g in enumerate(n_id)}
        frontier, local0 = list(n_id), 0
        src, dst = [], []
        for k in self.fanouts:
            new = []
            for i, f in enumerate(frontier):
                nb = self.colind[self.rowptr[f]:self.rowptr[f + 1]]
                if 0 <= k < len(nb):
                    nb = self.rng.choice(nb, size=k, replace=False)
                for g in nb:
                    g = int(g)
                    if g not in local:
                        local[g] = len(n_id) + len(new)
                        new.append(g)
                    src.append(local[g])
                    dst.append(local0 + i)
            frontier, local0 = new, len(n_id)
            n_id += new
            if not frontier:
                break
        return np.asarray(n_id, dtype=np.int64), np.asarray([src, dst], dtype=np.int64).reshape(2, -1)

    def __iter__(self):
        ids = self.input_nodes
        if self.shuffle:
            ids = ids[self.rng.permutation(len(ids))]
        for b in range(len(self)):
            seeds = ids[b * self.batch_size:(b + 1) * self.batch_size]
            n_id, ei = self._sample(seeds)
            n_t = torch.from_numpy(n_id)
            out = Data(x=self.data.x[n_t], edge_index=torch.from_numpy(ei), y=self.data.y[n_t])
            out.batch_size, out.n_id = len(seeds), n_t
            yield out
