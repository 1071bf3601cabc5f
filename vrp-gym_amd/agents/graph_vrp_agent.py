"""VRPModel / VRPAgent (reference: agents/graph_vrp_agent.py)."""
from .graph_encoder import GraphDemandEncoder
from .graph_tsp_agent import TSPAgent, TSPModel


class VRPModel(TSPModel):
    ENV_KINDS = (1,)  # VRP_KIND_VRP

    def __init__(self, depot_dim, node_dim, emb_dim, hidden_dim, num_attention_layers, num_heads):
        # the base constructor's encoder is built first and then replaced — this is what
        # the reference does and it fixes the RNG position of every later parameter
        super().__init__(node_dim=node_dim, emb_dim=emb_dim, hidden_dim=hidden_dim,
                         num_attention_layers=num_attention_layers, num_heads=num_heads)
        self.encoder = GraphDemandEncoder(depot_input_dim=depot_dim, node_input_dim=node_dim,
                                          embedding_dim=emb_dim, hidden_dim=hidden_dim,
                                          num_attention_layers=num_attention_layers,
                                          num_heads=num_heads)


class VRPAgent(TSPAgent):
    _MODEL = VRPModel

    def __init__(self, depot_dim: int = 2, node_dim: int = 2, emb_dim: int = 128,
                 hidden_dim: int = 512, num_attention_layers: int = 3, num_heads: int = 8,
                 lr: float = 1e-4, csv_path: str = "loss_log.csv", seed=69):
        # the TSP constructor runs first (two throw-away TSPModels advance the torch
        # stream), then the real model pair is built (graph_vrp_agent.py:118-148)
        super().__init__(node_dim=node_dim, emb_dim=emb_dim, hidden_dim=hidden_dim,
                         num_attention_layers=num_attention_layers, num_heads=num_heads,
                         lr=lr, csv_path=csv_path, seed=seed)
        arch = dict(node_dim=node_dim, emb_dim=emb_dim, hidden_dim=hidden_dim,
                    num_attention_layers=num_attention_layers, num_heads=num_heads)
        self._build(arch, dict(depot_dim=depot_dim), lr)
