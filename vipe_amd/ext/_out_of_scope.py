"""Submodules of `vipe_ext` that are off the SLAM hot path: they must exist as
attributes (vipe/ext/__init__.py:41-42) but raise when called."""


class _Absent:
    def __init__(self, name, fns):
        self._name = name
        for f in fns:
            setattr(self, f, self._raise(f))

    def _raise(self, f):
        def fn(*a, **k):
            raise NotImplementedError(f"{self._name}.{f} is outside the MI355X hot-path scope (SURVEY.md 2.1)")
        return fn


grounding_dino_ext = _Absent("grounding_dino_ext", ["ms_deform_attn_forward", "ms_deform_attn_backward"])
