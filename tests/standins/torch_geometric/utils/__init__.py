"""torch_geometric.utils — the six functions the reference's trainers call."""
import torch


def degree(index, num_nodes=None, dtype=None):
    n = int(index.max()) + 1 if num_nodes is None else num_nodes
    out = torch.zeros((n,), dtype=dtype if dtype is not None else torch.get_default_dtype(),
                      device=index.device)
    return out.scatter_add_(0, index, torch.ones((index.numel(),), dtype=out.dtype, device=index.device))


def _num_nodes(edge_index, num_nodes):
    if num_nodes is not None:
        return int(num_nodes)
    return int(edge_index.max()) + 1 if edge_index.numel() else 0


def to_undirected(edge_index, num_nodes=None):
    """concat both directions, then coalesce: sort by (row, col), drop duplicates."""
    n = _num_nodes(edge_index, num_nodes)
    row = torch.cat([edge_index[0], edge_index[1]])
    col = torch.cat([edge_index[1], edge_index[0]])
    key = torch.unique(row * n + col)
    return torch.stack([key // n, key % n])


def remove_self_loops(edge_index, edge_attr=None):
    mask = edge_index[0] != edge_index[1]
    return edge_index[:, mask], (None if edge_attr is None else edge_attr[mask])


def add_self_loops(edge_index, edge_weight=None, fill_value=1.0, num_nodes=None):
    n = _num_nodes(edge_index, num_nodes)
    loops = torch.arange(n, dtype=edge_index.dtype, device=edge_index.device)
    out = torch.cat([edge_index, torch.stack([loops, loops])], dim=1)  # appended at the END
    if edge_weight is not None:
        edge_weight = torch.cat([edge_weight, edge_weight.new_full((n,), fill_value)])
    return out, edge_weight


def subgraph(subset, edge_index, edge_attr=None, relabel_nodes=False, num_nodes=None):
    """keep edges with both endpoints in `subset`; relabel node subset[j] -> j."""
    n = _num_nodes(edge_index, num_nodes)
    device = edge_index.device
    if subset.dtype == torch.bool:
        node_mask = subset
        subset = node_mask.nonzero().view(-1)
    else:
        node_mask = torch.zeros(n, dtype=torch.bool, device=device)
        node_mask[subset] = True
    mask = node_mask[edge_index[0]] & node_mask[edge_index[1]]
    ei = edge_index[:, mask]
    if relabel_nodes:
        idx = torch.zeros(n, dtype=torch.long, device=device)
        idx[subset] = torch.arange(subset.numel(), device=device)
        ei = idx[ei]
    return ei, (None if edge_attr is None else edge_attr[mask])


def k_hop_subgraph(*args, **kwargs):
    raise NotImplementedError("stand-in: k_hop_subgraph is not on the sgformer path")


def to_dense_adj(edge_index, max_num_nodes=None):
    """[1, N, N] dense adjacency (medium/dataset.py:18 imports it for the graphormer preprocessing only)."""
    n = _num_nodes(edge_index, max_num_nodes)
    adj = torch.zeros((1, n, n), dtype=torch.float32, device=edge_index.device)
    adj[0, edge_index[0], edge_index[1]] = 1.0
    return adj
