"""the dynamic-field variant (attention-pooled graph summary + FiLM field net) vs the golden
vectors captured from the imported reference DynamicFieldAether and vs the oracle on fresh batches."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, scale_rel_err
from aether_amd import _lib
from aether_amd.edges import get_edges
from aether_amd.nn.state2state.dynamic_field_aether import DynamicFieldAether
from aether_amd.synthetic import make_batch
from oracle import aether_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _load(D):
    d = np.load(os.path.join(GOLDEN, f"dynfield_D{D}.npz"))
    sd = {str(k): torch.from_numpy(d["sd." + str(k)]) for k in d["keys"]}
    m = DynamicFieldAether(2 * D, 64, 0.0, D, device="cuda")
    assert list(m.state_dict().keys()) == list(sd.keys())               # reference key order
    m.load_state_dict(sd)
    return d, sd, m


@pytest.mark.parametrize("flags", [0, _lib.FLAG_FORCE_STREAMED])
@pytest.mark.parametrize("D", [2, 3])
def test_dynamic_field_matches_reference(D, flags):
    d, sd, m = _load(D)
    m.flags = flags
    for name in ("small", "cfg"):
        B, N = int(d[f"{name}.B"]), int(d[f"{name}.N"])
        t = lambda k: torch.from_numpy(d[f"{name}.in.{k}"]).cuda()
        edges = get_edges(B, N, device="cuda")
        out = m(None, t("x"), edges, t("vel"), t("edge_attr"), t("charges"), N)
        assert scale_rel_err(m.last_field.cpu(), torch.from_numpy(d[f"{name}.ref.field"])) <= TOL, name
        assert scale_rel_err(out.cpu(), torch.from_numpy(d[f"{name}.ref.out"])) <= TOL, name


@pytest.mark.parametrize("D", [2, 3])
def test_dynamic_field_fresh_batches_vs_oracle(D):
    d, sd, m = _load(D)
    for (B, N, seed) in [(1, 2, 1), (128, 20, 2), (3, 300, 3)]:          # 300 nodes: streamed path, several passes per thread
        inp = make_batch(B, N, D, seed=seed)
        want = O.dynamic_field_aether_forward(sd, inp["x"], inp["vel"], inp["edges"], inp["edge_attr"], inp["charges"], N)
        out = m(None, inp["x"].cuda(), [e.cuda() for e in inp["edges"]], inp["vel"].cuda(), inp["edge_attr"].cuda(),
                inp["charges"].cuda(), N)
        assert scale_rel_err(out.cpu(), want) <= TOL, (B, N)
        out2 = m(None, inp["x"].cuda(), [e.cuda() for e in inp["edges"]], inp["vel"].cuda(), inp["edge_attr"].cuda(),
                 inp["charges"].cuda(), N)
        assert torch.equal(out, out2)                                   # deterministic, workspace reuse
