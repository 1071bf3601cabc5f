"""VRPEnv — TSP with a re-visitable depot (reference: gym_vrp/envs/vrp.py:13-37).
Only the mask rule differs; it is selected inside the HIP kernels by `kind`."""
from .tsp import TSPEnv


class VRPEnv(TSPEnv):
    KIND = 1  # VRP_KIND_VRP
