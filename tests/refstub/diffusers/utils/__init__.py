import logging as _pylogging
from collections import OrderedDict
from dataclasses import fields


class BaseOutput(OrderedDict):
    """Dataclass-style output container (diffusers.utils.BaseOutput): attribute and key access."""

    def __post_init__(self):
        for f in fields(self):
            v = getattr(self, f.name)
            if v is not None:
                self[f.name] = v

    def to_tuple(self):
        return tuple(self[k] for k in self.keys())


class logging:  # noqa: N801  (module-like namespace: `from diffusers.utils import logging`)
    @staticmethod
    def get_logger(name):
        return _pylogging.getLogger(name)
