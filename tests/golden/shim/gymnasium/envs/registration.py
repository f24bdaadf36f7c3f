"""Stand-in for gymnasium.envs.registration (container-only; see the gymnasium shim): `register` records the spec, `make` resolves the
entry point "module:attr" and constructs it with the keyword arguments - what the tests of the gymnasium-present branch need."""
import importlib

registry = {}


def register(id, entry_point=None, **kwargs):      # noqa: A002 - gymnasium's own parameter name
    registry[id] = {"id": id, "entry_point": entry_point, "kwargs": kwargs}
    return None


def make(id, **kwargs):                            # noqa: A002
    spec = registry[id]
    mod, attr = spec["entry_point"].split(":")
    return getattr(importlib.import_module(mod), attr)(**kwargs)
