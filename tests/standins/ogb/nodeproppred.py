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


class PygNodePropPredDataset(NodePropPredDataset):
    pass
