def _nope(name):
    class _D:
        def __init__(self, *a, **k):
            raise NotImplementedError(f"stand-in: dataset {name} needs the network")
    _D.__name__ = name
    return _D


Planetoid, Amazon, Coauthor = (_nope(n) for n in ("Planetoid", "Amazon", "Coauthor"))
