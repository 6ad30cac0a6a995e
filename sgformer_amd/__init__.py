"""sgformer_amd — MI355X-native SGFormer forward/backward behind the reference's `ours` module surface.

    sgformer_amd.ours          drop-in for large/ours.py (SGFormer, TransConv, GraphConv, ...)
    sgformer_amd.ours_100m     drop-in for 100M/ours.py (alpha residual, neighbour-sampled batches)
    sgformer_amd.ours_medium   drop-in for medium/ours.py + the models.GCN backbone (GCNConv on libsgf)
    sgformer_amd.difformer     drop-in for medium/difformer.py (DIFFormer, simple kernel)
    sgformer_amd.ops           autograd operators over the C ABI (include/sgf.h -> lib/libsgf.so)
    sgformer_amd.dist          node-sharded multi-GPU execution (RCCL via torch.distributed)
    sgformer_amd.batching      GPU induced subgraph for the mini-batch trainer (PyG-compatible `subgraph`)
    sgformer_amd.loss          fused log_softmax + NLL on the training rows
    sgformer_amd.launch        run the reference's trainers unchanged on top of this package
    sgformer_amd.synth         synthetic graphs / tasks with the shapes of the reference's datasets
"""
__version__ = "0.1.0"
