"""torch_geometric.transforms stand-in (tests only): the names large/dataset.py and medium/dataset.py import."""


class NormalizeFeatures:
    def __call__(self, data):
        s = data.x.sum(dim=-1, keepdim=True).clamp(min=1.0)
        data.x = data.x / s
        return data


class ToUndirected:
    def __call__(self, data):
        from .utils import to_undirected
        data.edge_index = to_undirected(data.edge_index)
        return data
