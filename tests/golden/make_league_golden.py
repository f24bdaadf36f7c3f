#!/usr/bin/env python3
"""Cut the golden vectors of the league slot mapping from the REFERENCE's own mapping function
(train/callbk/league_based_self_play_callback.py:1286-1344).  Runs only in the build container
(needs /root/reference); `ray` is absent there, so every `ray.*` import resolves to an empty stand-in
module - the mapping function itself touches none of it.

    python tests/golden/make_league_golden.py      -> tests/golden/league_mapping.json
"""
import importlib.abc
import importlib.machinery
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "shim"))
sys.path.insert(0, "/root/reference")


class _Anything:
    def __init__(self, *a, **k):
        pass

    def __call__(self, f=None, *a, **k):
        return f


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        if name.isupper():
            return name.lower()
        t = type(name, (_Anything,), {})
        setattr(self, name, t)
        return t


class _RayFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name == "ray" or name.startswith("ray."):
            return importlib.machinery.ModuleSpec(name, self, is_package=True)

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


for _k in [k for k in sys.modules if k == "ray" or k.startswith("ray.")]:
    del sys.modules[_k]
sys.meta_path.insert(0, _RayFinder())
from gym_continuousDoubleAuction.train.callbk.league_based_self_play_callback import SelfPlayCallback  # noqa: E402


class _Episode:
    def __init__(self, i):
        self.id_ = i


def main():
    cases = []
    for (A, k, nrand, ow, cw, champions) in [(4, 2, 2, 1.0, 9.0, ["champion_1"]), (4, 1, 3, 1.0, 1.0, []),
                                             (8, 3, 5, 2.0, 5.0, ["champion_1", "champion_2", "champion_7"]),
                                             (5, 0, 5, 1.0, 3.0, ["champion_3"]), (4, 4, 0, 1.0, 1.0, [])]:
        cb = SelfPlayCallback(num_trainable_policies=k, num_random_policies=nrand, original_opponent_weight=ow, champion_weight=cw)
        for c in champions:
            cb.available_modules.append(c)
        fn = SelfPlayCallback.get_mapping_fn(cb)
        ids = [f"episode_{i}" for i in range(40)] + ["a1b2c3d4e5f6", 12345, "0", "market-17/ep-3"] + [f"{i:032x}" for i in range(977, 977 + 20)]
        table = [[fn(f"agent_{a}", _Episode(e)) for a in range(A)] for e in ids]
        cases.append({"num_agents": A, "num_trainable": k, "num_fixed": nrand, "original_opponent_weight": ow, "champion_weight": cw,
                      "champions": champions, "available_modules": list(cb.available_modules), "episode_ids": ids, "assignment": table})
    out = os.path.join(HERE, "league_mapping.json")
    with open(out, "w") as fh:
        json.dump(cases, fh, indent=0)
    print("wrote", out, sum(len(c["episode_ids"]) for c in cases), "episodes")


if __name__ == "__main__":
    main()
