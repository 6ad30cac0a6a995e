"""Stand-in for torch_geometric 1.7.2 (tests only; see ../README.md)."""
from . import data, loader, seed, utils  # noqa: F401
