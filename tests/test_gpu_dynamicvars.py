"""Variable-N decoder step (second half): the HIP path vs the reference's own outputs and vs the
oracle on fresh scenes with graphs built on the device."""
import pytest
import torch

from conftest import load_dyn_decoder, scale_rel_err
from aether_amd import _lib
from aether_amd.knn import get_knn_graph_info
from aether_amd.nn.dynamicvars.decoder import Decoder
from oracle import dynamicvars_oracle as DO

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.mark.parametrize("name", ["full8", "tail6", "gaps", "knn20", "empty"])
def test_decoder_step_matches_reference(name):
    c, dec, params = load_dyn_decoder(name)
    dec = dec.cuda()
    gi = tuple(c[k].cuda() for k in ("send", "recv", "e2n")) if "send" in c else None
    pred, hid = dec(c["inputs"].cuda(), c["hidden"].cuda(), c["edges"].cuda(), c["masks"].unsqueeze(0).cuda(), gi,
                    c["field"].cuda())
    assert pred.shape == c["ref.pred"].shape and hid.shape == c["ref.hidden"].shape
    assert scale_rel_err(pred.cpu(), c["ref.pred"]) <= TOL and scale_rel_err(hid.cpu(), c["ref.hidden"]) <= TOL
    absent = c["masks"] == 0
    assert (pred.cpu()[0, absent] == 0).all() and torch.equal(hid.cpu()[0, absent], c["hidden"][0, absent])


@pytest.mark.parametrize("Nmax,p,K,skip,posrep,H", [(40, 0.8, 4, True, "cart", 256), (64, 0.5, 2, False, "polar", 128),
                                                     (3, 1.0, 1, False, "cart", 128)])
def test_decoder_step_vs_oracle_device_graph(Nmax, p, K, skip, posrep, H):
    """inD-like sizes (4 edge types, the first skipped, hidden 256), graphs from aether_knn_edges; one-hot edge
    types as the sampler produces them."""
    params = {"input_size": 4, "gpu": True, "decoder_hidden": H, "num_edge_types": K, "skip_first": skip,
              "decoder_dropout": 0.0, "pos_representation": posrep}
    torch.manual_seed(17)
    dec = Decoder(params, device="cuda").eval()
    sd = {k: v.detach().cpu() for k, v in dec.state_dict().items()}
    g = torch.Generator().manual_seed(Nmax)
    inputs = torch.randn(1, Nmax, 4, generator=g)
    hidden = torch.randn(1, Nmax, H, generator=g) * 0.3
    field = torch.randn(1, Nmax, 2, generator=g) * 0.3
    masks = (torch.rand(Nmax, generator=g) < p).float()
    masks[:2] = 1.0
    nv = int(masks.sum())
    send, recv = get_knn_graph_info(inputs[0].cuda(), masks.cuda(), nv)
    k = min(10, nv - 1)
    # the reference's edge2node_inds: edge ids sorted by `recv`, k per row (single_ind_data.py:213-215)
    e2n = torch.argsort(recv, stable=True).view(-1, k)
    types = torch.randint(0, K, (send.numel(),), generator=g)
    edges = torch.nn.functional.one_hot(types, K).float().unsqueeze(0)
    want_p, want_h = DO.decoder_step(sd, inputs, hidden, edges, masks, (send.cpu(), recv.cpu(), e2n.cpu()), field, skip, posrep)
    got_p, got_h = dec(inputs.cuda(), hidden.cuda(), edges.cuda(), masks.cuda(), (send, recv, e2n), field.cuda())
    assert scale_rel_err(got_p.cpu(), want_p) <= TOL and scale_rel_err(got_h.cpu(), want_h) <= TOL


def test_decoder_errors():
    params = {"input_size": 4, "gpu": True, "decoder_hidden": 128, "num_edge_types": 2, "skip_first": False,
              "decoder_dropout": 0.0, "pos_representation": "cart"}
    dec = Decoder(params, device="cuda")
    x, h, f = torch.randn(1, 4, 4), torch.zeros(1, 4, 128), torch.zeros(1, 4, 2)
    with pytest.raises(_lib.AetherHipError):
        dec(x, h, None, torch.ones(4), None, f)                                       # CPU tensors: no fallback
    one = torch.tensor([0., 1., 0., 0.]).cuda()
    with pytest.raises(_lib.AetherHipError):
        dec(x.cuda(), h.cuda(), None, one, None, f.cuda())                            # single object: as the reference
    with pytest.raises(ValueError):
        dec(torch.randn(2, 4, 4).cuda(), h.cuda(), None, one, None, f.cuda())
    with pytest.raises(ValueError):
        Decoder(dict(params, decoder_hidden=96), device="cuda")
