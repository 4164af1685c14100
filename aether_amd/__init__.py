"""aether_amd: the Aether hot paths on MI355X (gfx950), behind the reference's module surface.

Headline path: ``aether_amd.nn.state2state.aether.Aether`` (drop-in nn.Module: forward, HIP backward, device rollout)
with the host helpers ``aether_amd.edges`` / ``aether_amd.synthetic`` / ``aether_amd.parallel`` / ``aether_amd.training``.
Widened rows of SURVEY.md 8f (each mirrors the reference module of the same name):

* ``aether_amd.nn.state2state.dynamic_field_aether.DynamicFieldAether``
* ``aether_amd.nn.seq2seq.{aether.Aether, dynamic_field_aether.DynamicFieldAether}`` (+ ``encoder``, ``decoder``, ``field``,
  ``localizer``)
* ``aether_amd.nn.dynamicvars.{aether_dynamicvars.AetherDynamicVars, encoder.Encoder, decoder.Decoder}``
* data side: ``aether_amd.knn``, ``aether_amd.sim``, ``aether_amd.data``, ``aether_amd.evaluate``, ``aether_amd.rollout``

Everything computes in ``libaether_hip.so`` (``include/aether_hip.h``); there is no CPU fallback.
"""
from .edges import get_edges, prepare_edge_attr  # noqa: F401

__all__ = ["get_edges", "prepare_edge_attr"]
