"""Synthetic inputs with the shapes of the reference's datasets.

There is no network, so tests, smoke() and bench.py run on synthetic graphs.  The generator applies
the trainer prologue of large/main.py:75-79 (to_undirected = symmetrise + coalesce, remove
self-loops, add one self-loop per node), so `edge_index` looks exactly like what the reference
hands to `model(x, edge_index)`.
"""
from __future__ import annotations

import torch

# name -> (N, average degree of the symmetrised graph before self-loops, features, classes, hidden)
SHAPES = {
    "cora": (2708, 3.9, 1433, 7, 64),
    "ogbn-arxiv": (169343, 13.7, 128, 40, 256),
    "ogbn-products": (2449029, 50.5, 100, 47, 256),
    "pokec": (1632803, 27.3, 65, 2, 256),
    # BASELINE.json config 5 (ogbn-papers100M-shaped: 111,059,956 nodes, ~30 stored entries per row,
    # 128 features, 172 classes, hidden 128) cut to the share ONE of 8 GPUs holds
    "papers100M-shard8": (13882494, 29.0, 128, 172, 128),
    # the same, as a WEAK-scaling workload: N is the node count PER RANK, the graph spans world * N nodes and
    # every rank generates only its own rows (synthetic_graph_shard): 8 ranks = the full 111 M-node shape
    "papers100M-weak": (13882494, 29.0, 128, 172, 128),
}

# constructor keywords of the large/run.sh recipes (dropout overridden by the caller)
RECIPES = {
    # large/run.sh:2-5
    "ogbn-arxiv": dict(trans_num_layers=1, trans_num_heads=1, trans_use_bn=True, trans_use_residual=True,
                       trans_use_weight=True, trans_use_act=False, gnn_num_layers=3, gnn_use_bn=True,
                       gnn_use_residual=True, gnn_use_weight=True, gnn_use_init=False, gnn_use_act=True,
                       use_graph=True, graph_weight=0.5, aggregate="add"),
    # large/run.sh:15-19 (amazon2m = the ogbn-products graph)
    "ogbn-products": dict(trans_num_layers=1, trans_num_heads=1, trans_use_bn=True, trans_use_residual=True,
                          trans_use_weight=True, trans_use_act=False, gnn_num_layers=3, gnn_use_bn=True,
                          gnn_use_residual=True, gnn_use_weight=True, gnn_use_init=True, gnn_use_act=True,
                          use_graph=True, graph_weight=0.5, aggregate="add"),
    # 100M/run.sh:3-7 (alpha residual; dropout overridden)
    "papers100M-shard8": dict(trans_num_layers=1, trans_num_heads=1, trans_use_bn=True, trans_use_residual=True,
                              trans_use_weight=True, trans_use_act=False, gnn_num_layers=3, gnn_use_bn=True,
                              gnn_use_residual=True, gnn_use_weight=True, gnn_use_init=True, gnn_use_act=True,
                              use_graph=True, graph_weight=0.8, aggregate="add", alpha=0.5),
    # large/run.sh:22-26
    "pokec": dict(trans_num_layers=1, trans_num_heads=1, trans_use_bn=True, trans_use_residual=True,
                  trans_use_weight=True, trans_use_act=False, gnn_num_layers=2, gnn_use_bn=True,
                  gnn_use_residual=True, gnn_use_weight=True, gnn_use_init=True, gnn_use_act=True,
                  use_graph=True, graph_weight=0.5, aggregate="add"),
}


RECIPES["papers100M-weak"] = RECIPES["papers100M-shard8"]

# (trans_dropout, gnn_dropout) of the same recipes: large/run.sh:2-5 (arxiv: 0.5 / 0.5), :15-19 and :22-26 (amazon2m / pokec:
# 0 / 0), 100M/run.sh:3-7 (0.5 / 0.2)
RECIPE_DROPOUT = {"cora": (0.2, 0.5), "ogbn-arxiv": (0.5, 0.5), "ogbn-products": (0.0, 0.0), "pokec": (0.0, 0.0),
                  "papers100M-shard8": (0.5, 0.2), "papers100M-weak": (0.5, 0.2)}


def synthetic_graph(n: int, avg_deg: float, seed: int = 123, directed: bool = False,
                    device="cpu") -> torch.Tensor:
    """Uniform random graph + trainer prologue.  int64 [2, nnz], coalesced edges then N self-loops."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    m = int(n * avg_deg / 2)
    src = torch.randint(0, n, (m,), generator=g).to(device)
    dst = torch.randint(0, n, (m,), generator=g).to(device)
    if not directed:
        src, dst = torch.cat([src, dst]), torch.cat([dst, src])
    keep = src != dst
    key = torch.unique(src[keep] * n + dst[keep])          # coalesce: sorted, duplicates dropped
    del src, dst, keep
    loops = torch.arange(n, device=device)
    return torch.stack([torch.cat([key // n, loops]), torch.cat([key % n, loops])])


def synthetic_graph_local(n: int, avg_deg: float, locality: float = 0.9, window: int = 4096,
                          seed: int = 123, device="cpu") -> torch.Tensor:
    """Same sizes and prologue as `synthetic_graph`, but a fraction `locality` of the undirected
    pairs joins nodes whose ids differ by ~N(0, window): the banded / community structure a real
    co-purchase or social graph has after any locality-preserving node ordering (the uniform
    generator is the zero-locality worst case: it is an expander, no ordering can help it)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    m = int(n * avg_deg / 2)
    src = torch.randint(0, n, (m,), generator=g).to(device)
    far = torch.randint(0, n, (m,), generator=g).to(device)
    off = (torch.randn(m, generator=g) * float(window)).round().long().to(device)
    is_local = (torch.rand(m, generator=g) < locality).to(device)
    dst = torch.where(is_local, (src + off).clamp_(0, n - 1), far)
    del far, off, is_local
    src, dst = torch.cat([src, dst]), torch.cat([dst, src])
    keep = src != dst
    key = torch.unique(src[keep] * n + dst[keep])
    del src, dst, keep
    loops = torch.arange(n, device=device)
    return torch.stack([torch.cat([key // n, loops]), torch.cat([key % n, loops])])


def synthetic_graph_community(n: int, avg_deg: float, seed: int = 123, comm_size=(64, 256), comms_per_super: int = 64,
                              p_comm: float = 0.80, p_super: float = 0.15, shuffle_ids: bool = True,
                              device="cpu", return_labels: bool = False) -> torch.Tensor:
    """Same sizes and prologue as `synthetic_graph`, but with the two-level community structure of a
    co-purchase / social graph, and — `shuffle_ids` — with node ids that carry NO trace of it:

      * nodes are split into communities of uniform random size in `comm_size`, and `comms_per_super`
        consecutive communities form a super-community;
      * an undirected pair starts at a uniform random node and ends, with probability `p_comm`, at a
        uniform node of the same community; with `p_super`, of the same super-community; else anywhere;
      * finally all node ids are renamed by a seeded random permutation.

    This is SURVEY.md §8d input class (b) with the locality HIDDEN: whatever reuse an SpMM gets out of
    this graph it must first recover from the edge list (sgf_reorder), as it would have to on ogbn ids.
    The uniform generator stays the zero-structure worst case (an expander: nothing to recover)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    mean = (comm_size[0] + comm_size[1]) / 2
    nc = int(n / mean * 1.3) + 8
    sizes = torch.randint(comm_size[0], comm_size[1] + 1, (nc,), generator=g)
    bounds = torch.cumsum(sizes, 0)
    nc = int((bounds < n).sum()) + 1
    starts = torch.cat([torch.zeros(1, dtype=torch.long), bounds[: nc - 1]])
    ends = torch.cat([bounds[: nc - 1], torch.tensor([n])])
    comm = torch.repeat_interleave(torch.arange(nc), ends - starts)
    sup_of_comm = torch.arange(nc) // comms_per_super
    sup_start = starts[torch.arange(0, nc, comms_per_super)]
    sup_end = torch.cat([sup_start[1:], torch.tensor([n])])
    m = int(n * avg_deg / 2)
    src = torch.randint(0, n, (m,), generator=g)
    u = torch.rand(m, generator=g)
    r = torch.rand(m, generator=g, dtype=torch.float64)
    c = comm[src]
    sc = sup_of_comm[c]
    d1 = starts[c] + (r * (ends[c] - starts[c])).long()
    d2 = sup_start[sc] + (r * (sup_end[sc] - sup_start[sc])).long()
    d3 = (r * n).long().clamp_(max=n - 1)
    dst = torch.where(u < p_comm, d1, torch.where(u < p_comm + p_super, d2, d3))
    del u, r, c, sc, d1, d2, d3
    labels = comm
    if shuffle_ids:
        perm = torch.randperm(n, generator=g)
        src, dst = perm[src], perm[dst]
        labels = torch.empty_like(comm)
        labels[perm] = comm
    src, dst = src.to(device), dst.to(device)
    src, dst = torch.cat([src, dst]), torch.cat([dst, src])
    keep = src != dst
    key = torch.unique(src[keep] * n + dst[keep])
    del src, dst, keep
    loops = torch.arange(n, device=device)
    ei = torch.stack([torch.cat([key // n, loops]), torch.cat([key % n, loops])])
    # return_labels: the planted community of every node (probes only: what a perfect sgf_reorder would recover)
    return (ei, labels.to(device)) if return_labels else ei


def synthetic_graph_community_powerlaw(n: int, avg_deg: float, seed: int = 123, size_range=(16, 4096), alpha: float = 1.5,
                                       comms_per_super: int = 32, p_comm: float = 0.75, p_super: float = 0.15,
                                       p_hub: float = 0.03, hub_gamma: float = 3.0, shuffle_ids: bool = True,
                                       device="cpu") -> torch.Tensor:
    """`synthetic_graph_community` with the skew of a real co-purchase / social graph (SURVEY.md §8d input class (b)):

      * community sizes follow a truncated power law (density ~ size^-(alpha+1) on `size_range`): many small
        communities, a few of thousands of nodes (longer than any row block);
      * inside a community the far endpoint is drawn with a heavy tail (index = size * u^2): local hubs;
      * a fraction `p_hub` of the pairs ends at a GLOBAL hub (id = n * u^hub_gamma): a handful of rows collect
        tens of thousands of entries — the long-row path (ogbn-products' largest row has ~17 k entries);
      * ids renamed by a random permutation, as in `synthetic_graph_community`."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    lo, hi = size_range
    u = torch.rand(int(n / lo) + 8, generator=g, dtype=torch.float64)
    # inverse CDF of the truncated Pareto on [lo, hi]
    sizes = (lo * (1.0 - u * (1.0 - (lo / hi) ** alpha)) ** (-1.0 / alpha)).long().clamp_(lo, hi)
    bounds = torch.cumsum(sizes, 0)
    nc = int((bounds < n).sum()) + 1
    starts = torch.cat([torch.zeros(1, dtype=torch.long), bounds[: nc - 1]])
    ends = torch.cat([bounds[: nc - 1], torch.tensor([n])])
    comm = torch.repeat_interleave(torch.arange(nc), ends - starts)
    sup_of_comm = torch.arange(nc) // comms_per_super
    sup_start = starts[torch.arange(0, nc, comms_per_super)]
    sup_end = torch.cat([sup_start[1:], torch.tensor([n])])
    m = int(n * avg_deg / 2)
    src = torch.randint(0, n, (m,), generator=g)
    sel = torch.rand(m, generator=g)
    r = torch.rand(m, generator=g, dtype=torch.float64)
    c = comm[src]
    sc = sup_of_comm[c]
    d1 = starts[c] + (r * r * (ends[c] - starts[c])).long()
    d2 = sup_start[sc] + (r * (sup_end[sc] - sup_start[sc])).long()
    d3 = (r * n).long().clamp_(max=n - 1)
    d4 = (r ** hub_gamma * n).long().clamp_(max=n - 1)
    dst = torch.where(sel < p_comm, d1, torch.where(sel < p_comm + p_super, d2,
                                                   torch.where(sel < 1.0 - p_hub, d3, d4)))
    del sel, r, c, sc, d1, d2, d3, d4
    if shuffle_ids:
        perm = torch.randperm(n, generator=g)
        src, dst = perm[src], perm[dst]
    src, dst = src.to(device), dst.to(device)
    src, dst = torch.cat([src, dst]), torch.cat([dst, src])
    keep = src != dst
    key = torch.unique(src[keep] * n + dst[keep])
    del src, dst, keep
    loops = torch.arange(n, device=device)
    return torch.stack([torch.cat([key // n, loops]), torch.cat([key % n, loops])])


def synthetic_graph_rmat(n: int, avg_deg: float, seed: int = 123, abc=(0.57, 0.19, 0.19), shuffle_ids: bool = True,
                         device="cpu") -> torch.Tensor:
    """R-MAT (Chakrabarti, Zhan, Faloutsos 2004) with the Graph500 parameters a, b, c = 0.57, 0.19, 0.19 (d = 0.05) — the
    STANDARD skewed generator SURVEY.md §8d(b) names, so that one structured number does not come from a generator tuned in
    this repository.  Every undirected pair picks, bit by bit over ceil(log2 n) levels, one quadrant of the adjacency
    matrix (a: both ids get a 0 bit, b: the target a 1, c: the source a 1, d: both); pairs with an id >= n are dropped
    (n is not a power of two: the kept pairs are the generator restricted to the n x n corner) and the draw is sized so
    that ~n * avg_deg / 2 pairs remain before coalescing; ids are renamed by a seeded random permutation (as Graph500
    does); then the trainer prologue of `synthetic_graph` (symmetrise + coalesce, no self pairs, one self-loop per node).
    Heavy-tailed degrees (the largest rows hold 10^4-10^5 entries at products size) and almost no community structure."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    a, b, c = abc
    scale = max(1, (n - 1).bit_length())
    # P(id < n) for the source / target marginals: bit = 1 with probability c + d / b + d, most significant bit first
    def keep_prob(p_one):
        # probability that a scale-bit number with iid bits (P(bit = 1) = p_one) is < n
        prob, pref = 0.0, 1.0
        for level in range(scale - 1, -1, -1):
            bit = (n >> level) & 1
            if bit:
                prob += pref * (1.0 - p_one)       # this bit 0 while n has 1: everything below is free
                pref *= p_one
            else:
                pref *= (1.0 - p_one)
        return prob
    p_keep = max(keep_prob(1.0 - a - b) * keep_prob(1.0 - a - c), 1e-3)     # (independent marginals: a sizing estimate only)
    want = int(n * avg_deg / 2)
    m = int(want / p_keep * 1.05) + 16
    src = torch.zeros(m, dtype=torch.long, device=device)
    dst = torch.zeros(m, dtype=torch.long, device=device)
    for level in range(scale):
        r = torch.rand(m, generator=g).to(device)
        src = (src << 1) | (r >= a + b).long()                               # quadrants c, d: source bit 1
        dst = (dst << 1) | (((r >= a) & (r < a + b)) | (r >= a + b + c)).long()   # quadrants b, d: target bit 1
        del r
    ok = (src < n) & (dst < n)
    src, dst = src[ok][:want], dst[ok][:want]
    del ok
    if shuffle_ids:
        perm = torch.randperm(n, generator=g).to(device)
        src, dst = perm[src], perm[dst]
    src, dst = torch.cat([src, dst]), torch.cat([dst, src])
    keep = src != dst
    key = torch.unique(src[keep] * n + dst[keep])
    del src, dst, keep
    loops = torch.arange(n, device=device)
    return torch.stack([torch.cat([key // n, loops]), torch.cat([key % n, loops])])


def synthetic_graph_skewed(n: int, avg_deg: float, gamma: float = 2.0, seed: int = 123, device="cpu") -> torch.Tensor:
    """Same prologue as `synthetic_graph`, but one endpoint of every pair is drawn from a heavy-tailed
    distribution (id = floor(n * u^gamma)): a few hub nodes collect a large share of the edges — node 0
    about n^(-1/gamma) of them — the long-row regime of power-law graphs (ogbn-products' largest row has
    ~17 k entries) that a uniform random graph never exercises (input class (b))."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    m = int(n * avg_deg / 2)
    src = torch.randint(0, n, (m,), generator=g).to(device)
    dst = (torch.rand(m, generator=g, dtype=torch.float64) ** gamma * n).long().clamp_(0, n - 1).to(device)
    src, dst = torch.cat([src, dst]), torch.cat([dst, src])
    keep = src != dst
    key = torch.unique(src[keep] * n + dst[keep])
    del src, dst, keep
    loops = torch.arange(n, device=device)
    return torch.stack([torch.cat([key // n, loops]), torch.cat([key % n, loops])])


def synthetic_graph_shard(n_per_rank: int, avg_deg: float, rank: int, world: int, seed: int = 123,
                          device="cpu") -> torch.Tensor:
    """Rank `rank`'s share of a uniform random graph over world * n_per_rank nodes, generated WITHOUT
    ever materialising (or sorting) the global edge list — the weak-scaling workload of BASELINE.json
    config 5 (papers100M-shaped: 111 M nodes over 8 GPUs).

    Returns int64 [2, nnz_local] with GLOBAL node ids: exactly the edges whose target
    (edge_index[1], the row of the normalised adjacency, large/ours.py:25-33) lies in
    [rank * n_per_rank, (rank + 1) * n_per_rank), after the trainer prologue (symmetrise + coalesce,
    no self pairs, one self-loop per node).  The undirected pairs between the node ranges of ranks
    a <= b are drawn from a generator seeded by (seed, a, b), so ranks a and b produce the SAME
    pairs independently and the union over ranks is a symmetric graph; every pair of ranges gets
    its uniform share of the n_total * avg_deg / 2 pairs.  (Use a generator of the same device
    type on every rank — the two sides of a block must see the same stream.)"""
    n_total = n_per_rank * world
    m_block = n_total * avg_deg / 2 / (world * world)      # pairs per ORDERED pair of ranges
    src_parts, dst_parts = [], []
    base = rank * n_per_rank
    for peer in range(world):
        a, b = min(rank, peer), max(rank, peer)
        g = torch.Generator(device=device).manual_seed(seed * 1000003 + a * 1009 + b)
        m = int(round(m_block if a == b else 2 * m_block))
        u = torch.randint(0, n_per_rank, (m,), generator=g, device=device) + a * n_per_rank
        v = torch.randint(0, n_per_rank, (m,), generator=g, device=device) + b * n_per_rank
        if a == b:                       # both ends mine: both directions
            src_parts += [u, v]
            dst_parts += [v, u]
        elif rank == a:                  # u is mine: the stored entry is (source v -> target u)
            src_parts.append(v)
            dst_parts.append(u)
        else:
            src_parts.append(u)
            dst_parts.append(v)
    src, dst = torch.cat(src_parts), torch.cat(dst_parts)
    del src_parts, dst_parts
    keep = src != dst
    key = torch.unique(dst[keep] * n_total + src[keep])     # coalesce
    del src, dst, keep
    loops = torch.arange(base, base + n_per_rank, device=device)
    return torch.stack([torch.cat([key % n_total, loops]), torch.cat([key // n_total, loops])])


def synthetic_task(n: int, f: int, c: int, seed: int = 123, device="cpu", dtype=torch.float32):
    """randn features, uniform labels, first half of a seeded permutation as the training split
    (rand_train_test_idx, large/data_utils.py:13-37, with train_prop = 0.5)."""
    g = torch.Generator(device="cpu").manual_seed(seed + 1)
    x = torch.randn(n, f, generator=g).to(device=device, dtype=dtype)
    y = torch.randint(0, c, (n,), generator=g).to(device)
    train_idx = torch.randperm(n, generator=g)[: n // 2].to(device)
    return x, y, train_idx
