from typing import TypeVar

ObsType = TypeVar("ObsType")

try:  # gym is optional: only its Env base class / VideoRecorder are used
    from gym import Env as GymEnv
except Exception:  # pragma: no cover - gym absent
    class GymEnv:  # minimal stand-in so that isinstance checks by callers still work
        metadata = {}
