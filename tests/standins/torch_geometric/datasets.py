"""torch_geometric.datasets stand-in (tests only).  Planetoid: a seeded synthetic citation-style graph with the attributes
medium/dataset.py:124-151 reads (x, y, edge_index, num_nodes, train / val / test masks); sizes from SGF_FAKE_PLANETOID
("N,avg_deg,features,classes").  Amazon / Coauthor need the network."""
import os

import numpy as np
import torch


def _nope(name):
    class _D:
        def __init__(self, *a, **k):
            raise NotImplementedError(f"stand-in: dataset {name} needs the network")
    _D.__name__ = name
    return _D


class _Data:
    pass


class Planetoid:
    def __init__(self, root=None, name="cora", transform=None, **_):
        n, deg, f, c = (float(v) for v in os.environ.get("SGF_FAKE_PLANETOID", "400,4,32,5").split(","))
        n, f, c = int(n), int(f), int(c)
        rng = np.random.default_rng(4321)
        m = int(n * deg / 2)
        d = _Data()
        d.edge_index = torch.from_numpy(rng.integers(0, n, size=(2, m)).astype(np.int64))
        d.x = torch.from_numpy(rng.random((n, f)).astype(np.float32))
        d.y = torch.from_numpy(rng.integers(0, c, size=(n,)).astype(np.int64))
        d.num_nodes = n
        perm = rng.permutation(n)
        for key, sel in (("train_mask", perm[: n // 4]), ("val_mask", perm[n // 4: n // 2]), ("test_mask", perm[n // 2:])):
            mask = torch.zeros(n, dtype=torch.bool)
            mask[torch.from_numpy(sel)] = True
            setattr(d, key, mask)
        self._data = transform(d) if transform is not None else d

    def __getitem__(self, idx):
        assert idx == 0
        return self._data


Amazon, Coauthor = (_nope(n) for n in ("Amazon", "Coauthor"))
