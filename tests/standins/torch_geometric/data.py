"""torch_geometric.data.Data — the attribute container 100M/nb-sample.py:81 builds (x, edge_index, y) and moves with
`.to(device)`.  Stand-in, tests only (../README.md)."""
import torch


class Data:
    def __init__(self, x=None, edge_index=None, y=None, **kwargs):
        self.x, self.edge_index, self.y = x, edge_index, y
        for k, v in kwargs.items():
            setattr(self, k, v)

    @property
    def num_nodes(self):
        if self.x is not None:
            return int(self.x.shape[0])
        return int(self.edge_index.max()) + 1 if self.edge_index is not None and self.edge_index.numel() else 0

    def to(self, device, *args, **kwargs):
        out = Data()
        for k, v in self.__dict__.items():
            setattr(out, k, v.to(device) if torch.is_tensor(v) else v)
        return out
