#!/usr/bin/env python3
"""Build-container tool (needs /root/reference): beyond the committed goldens, pit the CPU oracle against the REAL reference
on many fresh random episodes - random agent counts, balances, size ranges, price ranges, history depths, coefficients, action
laws, agent subsets, dict key orders that change every step, books of thousands of orders - comparing every recorded field bit for bit (the same comparison tests/test_oracle_golden.py runs).
Nothing is written; the point is the count of episodes that agree.

    PYTHONPATH=tests/golden/shim:/root/reference:tests:. python tests/golden/crosscheck_oracle.py [n_episodes] [seed]
"""
import contextlib
import io
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "shim"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_goldens as MG  # noqa: E402
import golden_util as G  # noqa: E402
import oracle_lib as O  # noqa: E402


def random_case(rng, i):
    from fuzz_cases import random_config, random_order
    cfg, law, present_p = random_config(rng)
    T = int(rng.integers(40, 140)) if law != "trend" else int(rng.integers(200, 700))      # (the trend law needs time to outgrow a tile)
    return f"x{i}", cfg, int(rng.integers(0, 2 ** 63)), T, int(rng.integers(0, 2 ** 31)), law, present_p, random_order(rng)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
    steps = 0
    for i in range(n):
        name, cfg, seed, T, aseed, law, present_p, order = random_case(rng, i)
        with contextlib.redirect_stdout(io.StringIO()):
            rec = MG.run_trace(name, cfg, seed, T, aseed, law=law, present_p=present_p, dict_order=order)
        rec = {k: (v if isinstance(v, np.ndarray) else np.asarray(v)) for k, v in rec.items()}
        rec["config"] = json.loads(str(rec["config"]))
        rec["name"] = name
        full_cfg = dict(rec["config"])
        env = O.OracleEnv(full_cfg, n_markets=1)
        try:
            steps += G.run_group(env, [rec], state_every=1, trace_getter=lambda: env.trace)
        except AssertionError as e:
            print(f"MISMATCH in episode {i}: config={cfg} seed={seed} law={law} present_p={present_p} order={order}\n  {e}")
            return 1
        finally:
            env.close()
        if (i + 1) % 50 == 0:
            print(f"{i + 1} episodes, {steps} steps: all fields identical")
    print(f"oracle == reference on {n} random episodes ({steps} steps)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
