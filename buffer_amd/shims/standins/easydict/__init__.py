"""Stand-in for `easydict` (ThreeDMatch/config.py:2, ThreeDMatch/dataloader.py:10): a dict whose keys are also
attributes; nested dicts (also inside lists/tuples) are converted on assignment, like the original."""


class EasyDict(dict):
    def __init__(self, d=None, **kwargs):
        super().__init__()
        items = dict(d or {}, **kwargs)
        for k, v in items.items():
            setattr(self, k, v)

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, cls):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(x) for x in v)
        return v

    def __setattr__(self, name, value):
        value = self._wrap(value)
        super().__setattr__(name, value)
        super().__setitem__(name, value)

    __setitem__ = __setattr__

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None

    def update(self, e=None, **f):
        for k, v in dict(e or {}, **f).items():
            setattr(self, k, v)

    def pop(self, k, *d):
        if hasattr(self, k):
            super().__delattr__(k)
        return super().pop(k, *d)
