"""performer_pytorch stand-in (tests only): medium/graphgps.py:6 imports SelfAttention at module level for a competitor
model that is never built on the sgformer path."""


class SelfAttention:
    def __init__(self, *a, **k):
        raise NotImplementedError("stand-in: performer_pytorch is not installed")
