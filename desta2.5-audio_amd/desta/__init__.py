"""MI355X-native DeSTA2.5-Audio hot path (drop-in import name of the reference package).

`from desta import DeSTA25AudioModel` works as in the reference (desta/__init__.py:1).
"""
__all__ = ["DeSTA25AudioModel", "DeSTA25Config"]


def __getattr__(name):
    if name in __all__:
        from .models import modeling_desta25 as _m
        return getattr(_m, name)
    raise AttributeError(name)
