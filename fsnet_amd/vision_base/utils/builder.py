"""The plugin boundary: build(name, *args, **kwargs) instantiates any importable dotted path.
Mirrors vision_base/utils/builder.py:5-72 of the reference (same names, same call semantics), so a
reference config runs with only its `name=` strings repointed into fsnet_amd."""
import numpy as np

from .utils import find_object


def build(name, *args, **kwargs):
    return find_object(name)(*args, **kwargs)


class _Combinator(object):
    def __init__(self, cfg_list, **common_keywords):
        self.children = []
        for item in cfg_list:
            merged = dict(common_keywords)
            merged.update(item)
            self.children.append(build(**merged))

    @staticmethod
    def _chain(children, args, kwargs):
        result = None
        for i, child in enumerate(children):
            if i == 0:
                result = child(*args, **kwargs)
            elif isinstance(result, tuple):
                result = child(*result)
            else:
                result = child(result)
        return result


class Sequential(_Combinator):
    """children run in order, each fed the previous result (tuples are splatted)."""

    def __call__(self, *args, **kwargs):
        return self._chain(self.children, args, kwargs)


class Parallel(_Combinator):
    """every child sees the same inputs; results are returned as a list."""

    def __call__(self, *args, **kwargs):
        return [child(*args, **kwargs) for child in self.children]


class Shuffle(_Combinator):
    """children run chained in a fresh random order on every call (np.random.permutation)."""

    def __call__(self, *args, **kwargs):
        order = np.random.permutation(len(self.children))
        return self._chain([self.children[i] for i in order], args, kwargs)
