"""MI355X-native vectorised continuous-double-auction environment.

Public surface:
    CDAVecEnv  - N markets stepped in lockstep on one GPU (torch tensors in / out)
    CDAEnv     - single-market facade with the reference's dict-shaped MultiAgentEnv API
    CDAVecMultiAgentEnv - N markets as N sub-envs of one object (tensor views or lists of per-market dicts)
The compute path is hand-written HIP for gfx950 behind the C-ABI of include/cda.h; importing this
package never touches the GPU, constructing an env does and fails loudly without one.
"""
from . import _capi  # noqa: F401

__all__ = ["CDAVecEnv", "CDAEnv", "CDAVecMultiAgentEnv", "run_random"]


def __getattr__(name):
    if name == "CDAVecEnv":
        from .vec_env import CDAVecEnv
        return CDAVecEnv
    if name == "CDAEnv":
        from .env import CDAEnv
        return CDAEnv
    if name == "CDAVecMultiAgentEnv":
        from .env import CDAVecMultiAgentEnv
        return CDAVecMultiAgentEnv
    if name == "run_random":
        from .cda_rand import run_random
        return run_random
    raise AttributeError(name)
