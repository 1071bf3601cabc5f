from .irp import IRPEnv  # noqa: F401
from .tsp import TSPEnv  # noqa: F401
from .vrp import VRPEnv  # noqa: F401
