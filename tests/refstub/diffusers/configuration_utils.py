"""`ConfigMixin` / `register_to_config` as the reference uses them (models/unet_3d_condition_mask.py:22,86):
ctor arguments (with defaults) become `self.config.<name>`."""
import functools
import inspect
from types import SimpleNamespace


class _Config(SimpleNamespace):
    def __getitem__(self, k):
        return getattr(self, k)

    def get(self, k, default=None):
        return getattr(self, k, default)

    def keys(self):
        return vars(self).keys()


class ConfigMixin:
    config_name = "config.json"

    @property
    def config(self):
        return self._internal_dict

    def register_to_config(self, **kwargs):
        self._internal_dict = _Config(**kwargs)


def register_to_config(init):
    sig = inspect.signature(init)

    @functools.wraps(init)
    def wrapper(self, *args, **kwargs):
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = {k: v for k, v in bound.arguments.items() if k != "self"}
        ConfigMixin.register_to_config(self, **cfg)
        init(self, *args, **kwargs)

    return wrapper
