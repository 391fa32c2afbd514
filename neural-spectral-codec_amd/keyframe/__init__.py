"""Mirror of the reference's ``keyframe`` package for the hot path: only the graph container and
the offline graph builder that feed the GNN (src/keyframe/graph_manager.py:471-606)."""
from .graph_manager import Data, build_graph_from_keyframes_batch, build_chain_graph

__all__ = ["Data", "build_graph_from_keyframes_batch", "build_chain_graph"]
