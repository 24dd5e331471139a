"""Name -> class registry shared by the two layer families.

Contract of the reference's registries (moe_model/model/moe/register.py:1-21, moe_pretrain_model/layers/moe/register.py):
a decorator that files a class under one or more names, a lookup that raises ValueError for an unknown name, and an
AssertionError when a name is claimed by a second, different class.  Unlike upstream the decorator hands the class back, so
the decorated name stays bound to it."""
from __future__ import annotations

from typing import Callable, Dict, Type


class Registry:
    def __init__(self, what: str):
        self.what = what
        self.classes: Dict[str, Type] = {}

    def register(self, *names: str) -> Callable[[Type], Type]:
        def file_under(cls: Type) -> Type:
            for name in names:
                holder = self.classes.setdefault(name, cls)
                if holder is not cls:
                    raise AssertionError(f"{self.what} '{name}' is already registered to {holder.__qualname__}; "
                                         f"{cls.__qualname__} needs another name")
            return cls
        return file_under

    def get(self, name: str) -> Type:
        cls = self.classes.get(name)
        if cls is None:
            raise ValueError(f"no {self.what} is registered as '{name}' (known: {sorted(self.classes)})")
        return cls
