"""sgformer_amd — MI355X-native SGFormer forward/backward behind the reference's `ours` module surface.

    sgformer_amd.ours        drop-in for large/ours.py (SGFormer, TransConv, GraphConv, ...)
    sgformer_amd.ops         autograd operators over the C ABI (include/sgf.h -> lib/libsgf.so)
    sgformer_amd.dist        node-sharded multi-GPU execution (RCCL via torch.distributed)
    sgformer_amd.launch      run the reference's trainers unchanged on top of this package
"""
__version__ = "0.1.0"
