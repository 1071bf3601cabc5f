"""MI355X-native drop-in for the `agents` package of kevin-schumann/VRP-GYM."""
from .graph_decoder import GraphDecoder  # noqa: F401
from .graph_encoder import GraphDemandEncoder, GraphEncoder  # noqa: F401
from .graph_irp_agent import IRPAgent  # noqa: F401
from .graph_tsp_agent import TSPAgent  # noqa: F401
from .graph_vrp_agent import VRPAgent  # noqa: F401
from .random_agent import RandomAgent  # noqa: F401
