from .graph import Graph, get_k_adjacency  # noqa: F401
from .partition_strategy import GraphPartitionStrategy  # noqa: F401
