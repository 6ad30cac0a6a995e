"""torch_geometric.nn stand-in (tests only): GCNConv / MessagePassing are the single restatement in
oracle/ref_shim.py; the other names large/gnns.py and medium/models.py import at module level are
placeholders (baseline GNN zoo, never run on the sgformer path)."""
from oracle.ref_shim import _GCNConv as GCNConv, _MessagePassing as MessagePassing, _placeholder  # noqa: F401

SGConv, GATConv, JumpingKnowledge, APPNP = (_placeholder(n) for n in ("SGConv", "GATConv", "JumpingKnowledge", "APPNP"))
