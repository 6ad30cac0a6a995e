"""Names large/gnns.py imports at module level (baseline GNN zoo; never run on the sgformer path)."""
import torch.nn as nn


class MessagePassing(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()


def _placeholder(name):
    class _P(nn.Module):
        def __init__(self, *a, **k):
            raise NotImplementedError(f"stand-in: torch_geometric.nn.{name} is not on the sgformer path")
    _P.__name__ = name
    return _P


GCNConv, SGConv, GATConv, JumpingKnowledge, APPNP = (
    _placeholder(n) for n in ("GCNConv", "SGConv", "GATConv", "JumpingKnowledge", "APPNP"))
