"""CPU-side checks: edge index (bit-exact), state_dict surface, C-ABI symbols."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import GOLDEN, REPO, load_state_dict
from aether_amd import _lib
from aether_amd.edges import fully_connected_edges, get_edges, prepare_edge_attr
from aether_amd.nn.state2state.aether import Aether


@pytest.mark.parametrize("B,N", [(1, 5), (3, 5), (128, 20), (2, 2), (1, 3)])
def test_edge_index_bit_exact(B, N):
    """dataset4newton.py:84-94 -- int64 equality against tensors captured from the reference."""
    d = np.load(os.path.join(GOLDEN, "edges.npz"))
    send, recv = get_edges(B, N)
    assert send.dtype == torch.int64 and recv.dtype == torch.int64
    assert np.array_equal(send.numpy(), d[f"send_B{B}_N{N}"])
    assert np.array_equal(recv.numpy(), d[f"recv_B{B}_N{N}"])


def test_edge_order_is_where_not_eye():
    """Same order as torch.where(~torch.eye(N)) (nn/utils/augmented_global_to_local.py:31-32)."""
    for n in (2, 4, 7):
        s, r = torch.where(~torch.eye(n, dtype=torch.bool))
        rows, cols = fully_connected_edges(n)
        assert rows == s.tolist() and cols == r.tolist()
        a, b = get_edges(1, n)
        assert torch.equal(a, s) and torch.equal(b, r)


def test_prepare_edge_attr():
    x = torch.tensor([[0.0, 0.0], [3.0, 4.0], [0.0, 1.0]])
    edges = get_edges(1, 3)
    q = torch.tensor([[1.0], [-1.0], [1.0]])
    ea = prepare_edge_attr(x, edges, q[edges[0]] * q[edges[1]])
    assert ea.shape == (6, 2)
    assert torch.allclose(ea[0], torch.tensor([-1.0, 5.0]))


@pytest.mark.parametrize("D", [2, 3])
def test_state_dict_surface_and_default_init(D):
    """Keys, shapes, order and (under the runner's seed 1) values equal the reference's."""
    torch.manual_seed(1)
    m = Aether(2 * D, 64, 0.0, D, device="cpu")
    ref = load_state_dict(D)
    sd = m.state_dict()
    assert list(sd.keys()) == list(ref.keys())
    assert len(sd) == 47
    for k in ref:
        assert sd[k].shape == ref[k].shape, k
        assert torch.equal(sd[k], ref[k]), k
    assert int(m.params) == (131892 if D == 2 else 132822)
    m2 = Aether(2 * D, 64, 0.0, D, device="cpu")
    m2.load_state_dict(ref, strict=True)


def test_ctor_rejects_unsupported():
    with pytest.raises(ValueError):
        Aether(4, 0, 0.0, 2, device="cpu")               # no width
    from aether_amd.nn.state2state.dynamic_field_aether import DynamicFieldAether
    for cls in (Aether, DynamicFieldAether):             # any width the reference accepts (experiments/lorentz/main.py:42-43)
        for H in (20, 64, 96, 128):
            m = cls(4, H, 0.0, 2, device="cpu")
            sd = m.state_dict()
            assert sd["gnn.layer_3.message_fn.0.weight"].shape == (H, 3 * H) and sd["gnn.out_mlp.6.weight"].shape == (2, H)
            assert m._kw == (64 if H <= 64 else 128)
    with pytest.raises(ValueError):
        Aether(4, 6, 0.0, 2, device="cpu")               # hidden_size == 3 D: the reference drops layer_1.res there
    with pytest.raises(ValueError):
        Aether(4, 64, 1.0, 2, device="cpu")              # dropout_prob in [0, 1)
    m = Aether(4, 64, 0.1, 2, device="cpu")              # p > 0: same parameters / state_dict (nn.Dropout has none) ...
    assert list(m.state_dict().keys()) == list(Aether(4, 64, 0.0, 2, device="cpu").state_dict().keys())


def test_narrow_hidden_size_keeps_the_reference_parameter_shapes():
    """hidden_size < 64 (--nf of experiments/lorentz/main.py:42-43): parameters and state_dict have the narrow model's own
    shapes (the 64-wide engine it runs on is not part of them), and constructing it leaves the RNG stream where the
    reference's constructor would."""
    torch.manual_seed(3)
    m = Aether(4, 32, 0.0, 2, device="cpu")
    after = torch.rand(1)
    sd = m.state_dict()
    assert sd["gnn.layer_2.message_fn.0.weight"].shape == (32, 96)
    assert sd["gnn.layer_1.update_fn.0.weight"].shape == (64, 32) and sd["gnn.out_mlp.6.weight"].shape == (2, 32)
    assert not any("engine" in k for k in sd) and len(sd) == 47
    torch.manual_seed(3)
    from aether_amd.nn.state2state.aether import _GNN, _FieldNetwork
    _GNN(4, 32, 0.0, 2, additional_features=2); _FieldNetwork(2, 32, 16)
    assert torch.equal(after, torch.rand(1))


def test_cpu_tensor_fails_loudly():
    m = Aether(4, 64, 0.0, 2, device="cpu")
    e = get_edges(1, 3)
    with pytest.raises(_lib.AetherHipError):
        m(torch.zeros(3, 1), torch.zeros(3, 2), e, torch.ones(3, 2), torch.zeros(6, 2), torch.ones(3, 1))


def test_loss_and_optimizer_fail_loudly_on_the_cpu():
    """aether_amd.optim has no CPU path either: CPU tensors raise instead of computing something."""
    from aether_amd.optim import FusedAdamW, mse_loss_grad
    with pytest.raises(_lib.AetherHipError):
        mse_loss_grad(torch.zeros(4, 2), torch.zeros(4, 2))
    p = torch.nn.Parameter(torch.zeros(8))
    p.grad = torch.ones(8)
    opt = FusedAdamW([p], lr=1e-3)
    with pytest.raises(_lib.AetherHipError):
        opt.step()
    assert torch.equal(p.detach(), torch.zeros(8))
    with pytest.raises(ValueError):
        FusedAdamW([p], lr=1e-3, betas=(1.0, 0.999))
    assert ctypes.sizeof(_lib.AetherAdamWTensor) == 40          # include/aether_hip.h: four pointers and an int64


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads and exports what include/aether_hip.h declares."""
    hdr = open(os.path.join(REPO, "include", "aether_hip.h")).read()
    declared = set(re.findall(r"\b(aether_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.aether_version()
    # size queries are host-only arithmetic
    assert lib.aether_workspace_bytes(2560, 48640, 2, 0) > 48640 * 64 * 4 * 4
    assert lib.aether_workspace_bytes(10, 10, 4, 0) == 0


def test_params_struct_layout_matches_header():
    hdr = open(os.path.join(REPO, "include", "aether_hip.h")).read()
    body = hdr[hdr.index("typedef struct AetherParams"):hdr.index("} AetherParams;")]
    names = re.findall(r"float\*\s+([a-z0-9_]+)(?:\[3\])?;", body)
    assert names == [n for n, _ in _lib.PARAM_FIELDS]
    n_ptr = sum(3 if "{}" in k else 1 for _, k in _lib.PARAM_FIELDS)
    assert ctypes.sizeof(_lib.AetherParams) == 8 * n_ptr == 8 * 47


def test_public_header_is_valid_c99_and_cpp(tmp_path):
    """include/aether_hip.h is the drop-in boundary: it must compile on its own as C99 (cgo / ctypes-style bindings)
    and as C++."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    src = tmp_path / "hdr_check.c"
    src.write_text('#include "%s"\nint main(void) { return (int)sizeof(AetherParams) == 0; }\n'
                   % os.path.join(REPO, "include", "aether_hip.h"))
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-c", str(src), "-o", str(tmp_path / "a.o")],
                   check=True)
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-x", "c++", "-c", str(src), "-o", str(tmp_path / "b.o")], check=True)


def test_built_library_passes_the_isa_check():
    """tools/isa_check.py on the built code object: no packed fp32 instruction with an op_sel bit on
    src0 / src1 anywhere in the library -- the form that returned wrong low halves in lanes 48-63 next to bf16 MFMAs
    and corrupted ~3 tiles per million in a round-2 build of k_edge_layer1.  The same scanner must flag the culprit
    instruction when it is handed to it."""
    import sys
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import isa_check
    _lib.load()                                                   # builds the library if it is stale
    txt = isa_check.disassemble(_lib.LIB_PATH)
    kernels = list(isa_check.split_kernels(txt))
    assert len(kernels) > 300, len(kernels)
    bad = {name: res["r3"][:2] for name, body in kernels for res in [isa_check.check_kernel(body)] if res["r3"]}
    assert not bad, bad
    # the scanner itself: the instruction of the corrupting build, harmless neighbours, and the accumulator patterns
    res = isa_check.check_kernel([
        "v_pk_fma_f32 v[6:7], v[20:21], v[18:19], 0 op_sel_hi:[1,0,0]",
        "v_pk_fma_f32 v[6:7], v[2:3], v[18:19], v[6:7] op_sel:[0,1,0]",
        "v_pk_mul_f32 v[8:9], v[2:3], v[4:5] op_sel:[1,0]",
        "v_pk_fma_f32 v[6:7], v[2:3], v[18:19], v[6:7] op_sel:[0,0,1] op_sel_hi:[1,1,0]",
        "ds_read_b128 a[4:7], v70 offset:36992",
        "v_mfma_f32_16x16x32_bf16 a[2:5], v[2:5], v[10:13], a[4:7]",
    ])
    assert len(res["r3"]) == 2 and res["bf16"] and len(res["r1"]) == 1 and len(res["r2"]) == 1, res
    # rule R4: the by-value-struct kernels of a captured variable-N step consume no hidden kernel
    # arguments -- round 3's k_s2s_filter_split_types read gridDim.x and faulted as a graph node replayed back to back
    assert isa_check.check_hidden_args(_lib.LIB_PATH) == []
    notes = isa_check.kernel_notes(_lib.LIB_PATH)
    split_types = [v for k, v in notes.items() if "k_s2s_filter_split_types" in k]
    assert split_types and all(seg <= 256 and not hidden for seg, hidden, _priv in split_types), split_types
    # rule R6 (round 4): the inference k_fused keeps nothing in scratch memory, the streaming split GEMMs copy no
    # accumulator between AGPRs and VGPRs (both were found by their cost: + 1 MB of writes per launch, + 0.4 ms per step)
    assert isa_check.check_scratch(_lib.LIB_PATH) == []
    for name, body in kernels:
        if any(k in name for k in isa_check.R6_NO_AGPR_COPIES):
            assert not [ins for ins in body if "v_accvgpr_" in ins], name


def test_host_side_size_functions_of_the_round_3_entries():
    """Pure host arithmetic of the C ABI (no GPU): the dropout masks lie inside the training workspace, 256-byte aligned;
    aether_dyn_step's workspace covers its four stages and refuses the sizes the step would refuse; the rollout's covers
    its largest step."""
    import ctypes as C
    from aether_amd import _lib
    from aether_amd.nn.dynamicvars.aether_dynamicvars import _DynStepConfig
    lib = _lib.load()
    for (nn_, ee, D) in [(2560, 48640, 2), (100, 0, 3), (32768, 33521664, 2)]:
        off = lib.aether_dropout_mask_offset(nn_, ee, D)
        total = lib.aether_workspace_bytes(nn_, ee, D, 1)
        assert off % 256 == 0 and off + 2 * nn_ * 64 * 4 <= total
        assert off >= lib.aether_workspace_bytes(nn_, ee, D, 0) or ee == 0          # behind the inference part
    assert lib.aether_dropout_mask_offset(0, 0, 2) == 0 and lib.aether_dropout_mask_offset(5, 5, 4) == 0
    cfg = _DynStepConfig(256, 256, 64, 3, 128, 4, 256, 1, 0, 0, 10, 0.5)
    need = lib.aether_dyn_step_workspace_bytes(C.byref(cfg), 40, 24, 240)
    parts = (lib.aether_dyn_field_workspace_bytes(24, 256) + lib.aether_knn_workspace_bytes(1, 40, 10)
             + lib.aether_dyn_prior_workspace_bytes(256, 64, 128, 24, 240) + lib.aether_dyn_decoder_workspace_bytes(256, 24, 240))
    assert need > parts
    assert lib.aether_dyn_step_workspace_bytes(C.byref(cfg), 40, 24, 239) == 0          # not n * min(k, n - 1) edges
    assert lib.aether_dyn_step_workspace_bytes(C.byref(cfg), 40, 1, 0) == 0             # a step needs two present objects
    assert lib.aether_dyn_step_workspace_bytes(C.byref(cfg), 40, 41, 410) == 0          # more present than rows
    bad = _DynStepConfig(256, 200, 64, 3, 128, 4, 256, 1, 0, 0, 10, 0.5)                # encoder hidden not a multiple of 128
    assert lib.aether_dyn_step_workspace_bytes(C.byref(bad), 40, 24, 240) == 0
    counts = (C.c_int64 * 4)(24, 0, 3, 40)
    edges = (C.c_int64 * 4)(240, 0, 6, 400)
    roll = lib.aether_dyn_rollout_workspace_bytes(C.byref(cfg), 40, 4, counts, edges)
    assert roll >= lib.aether_dyn_step_workspace_bytes(C.byref(cfg), 40, 40, 400)
    assert roll >= need
    # a step whose graph has another edge count than the encoder's kNN graph (n * min(k, n - 1)) is refused up front
    # (the rollout used to derive E itself and read only the first n * 10 edges of a longer graph)
    edges_bad = (C.c_int64 * 4)(240, 0, 6, 40 * 39)
    assert lib.aether_dyn_rollout_workspace_bytes(C.byref(cfg), 40, 4, counts, edges_bad) == 0
