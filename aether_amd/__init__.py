"""aether_amd: the Aether state2state hot path on MI355X (gfx950).

Public surface mirrors the reference for this path only:
``aether_amd.nn.state2state.aether.Aether`` (drop-in nn.Module) and the host helpers
``aether_amd.edges`` / ``aether_amd.synthetic``.
"""
from .edges import get_edges, prepare_edge_attr  # noqa: F401

__all__ = ["get_edges", "prepare_edge_attr"]
