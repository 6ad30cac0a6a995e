"""ogb.nodeproppred stand-in: a seeded synthetic dataset with the interface
large/dataset.py:354-368 (load_ogb_dataset) consumes: .graph{edge_index,node_feat,num_nodes,
edge_feat,node_year} as numpy, .labels [N,1], .get_idx_split().  Sizes from SGF_FAKE_OGB
("N,avg_deg,features,classes"; default a small ogbn-arxiv-like graph)."""
import os

import numpy as np


class NodePropPredDataset:
    def __init__(self, name, root=None):
        n, deg, f, c = (float(v) for v in os.environ.get("SGF_FAKE_OGB", "600,6,24,5").split(","))
        n, f, c = int(n), int(f), int(c)
        rng = np.random.default_rng(1234)
        m = int(n * deg / 2)
        self.graph = {
            "edge_index": rng.integers(0, n, size=(2, m)).astype(np.int64),   # directed, as ogbn-arxiv
            "node_feat": rng.standard_normal((n, f)).astype(np.float32),
            "edge_feat": None, "num_nodes": n,
        }
        self.labels = rng.integers(0, c, size=(n, 1)).astype(np.int64)
        perm = rng.permutation(n)
        self._split = {"train": perm[: n // 2], "valid": perm[n // 2: 3 * n // 4], "test": perm[3 * n // 4:]}

    def get_idx_split(self):
        return self._split


class _PygData:
    """What `PygNodePropPredDataset(...)[0]` gives 100M/dataset.py:81-95: edge_index / x / y tensors and num_nodes."""


class PygNodePropPredDataset(NodePropPredDataset):
    def __getitem__(self, idx):
        import torch
        assert idx == 0
        d = _PygData()
        d.edge_index = torch.from_numpy(self.graph["edge_index"])
        d.x = torch.from_numpy(self.graph["node_feat"])
        d.y = torch.from_numpy(self.labels.astype(np.float32))      # papers100M stores float labels (NaN = unlabeled)
        d.num_nodes = int(self.graph["num_nodes"])
        return d

    def get_idx_split(self):
        import torch
        return {k: torch.from_numpy(v) for k, v in self._split.items()}
