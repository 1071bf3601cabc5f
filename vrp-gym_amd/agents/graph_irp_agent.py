"""IRPModel / IRPAgent (reference: agents/graph_irp_agent.py): node features are
(x, y, demand) and the decoder context carries the vehicle load."""
from .graph_tsp_agent import TSPAgent
from .graph_vrp_agent import VRPModel


class IRPModel(VRPModel):
    ENV_KINDS = (2,)  # VRP_KIND_IRP


class IRPAgent(TSPAgent):
    _MODEL = IRPModel

    def __init__(self, depot_dim: int = 2, node_dim: int = 3, emb_dim: int = 128,
                 hidden_dim: int = 512, num_attention_layers: int = 3, num_heads: int = 8,
                 lr: float = 1e-4, csv_path: str = "loss_log.csv", seed: int = 69):
        super().__init__(node_dim=node_dim, emb_dim=emb_dim, hidden_dim=hidden_dim,
                         num_attention_layers=num_attention_layers, num_heads=num_heads,
                         lr=lr, csv_path=csv_path, seed=seed)
        arch = dict(node_dim=node_dim, emb_dim=emb_dim, hidden_dim=hidden_dim,
                    num_attention_layers=num_attention_layers, num_heads=num_heads)
        self._build(arch, dict(depot_dim=depot_dim), lr)
