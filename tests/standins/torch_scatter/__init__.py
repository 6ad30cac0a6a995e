"""torch_scatter stand-in: `scatter` is imported by large/main.py:9 but never called on this path."""


def scatter(*args, **kwargs):
    raise NotImplementedError("stand-in: torch_scatter.scatter is not on the sgformer path")
