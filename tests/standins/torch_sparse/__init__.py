"""torch_sparse 0.6.10 stand-in: the semantics live in oracle/ref_shim.py (one restatement)."""
from oracle.ref_shim import _SparseTensor as SparseTensor, _matmul as matmul  # noqa: F401
