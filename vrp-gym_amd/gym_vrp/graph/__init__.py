from .instances import draw_instances  # noqa: F401
from .vrp_graph import VRPGraph  # noqa: F401
from .vrp_network import VRPNetwork  # noqa: F401
