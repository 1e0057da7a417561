"""CPU: host-side logic of the DiT mirror -- state-dict inventory equals the reference's (through
oracle.dit_param_shapes, itself asserted against the reference module in make_golden.py), config
attributes, timestep-row extraction, cfg_skip wrapper, loud failure without a GPU."""
import pytest
import torch

from oracle import cases as C
from oracle import dit as O


def tiny_model():
    from flexam_amd.wan_transformer3d_FlexAM import Wan2_2Transformer3DModel_FlexAM
    cfg = dict(O.DIT_TINY)
    cfg.pop("eps")
    return Wan2_2Transformer3DModel_FlexAM(**cfg), dict(O.DIT_TINY)


def test_state_dict_matches_reference_inventory():
    m, cfg = tiny_model()
    sd = m.state_dict()
    shapes = O.dit_param_shapes(cfg)
    assert set(sd) == set(shapes), sorted(set(sd) ^ set(shapes))
    for k, v in shapes.items():
        assert tuple(sd[k].shape) == tuple(v), k
    m.load_state_dict(C.dit_weights(cfg, 7), strict=True)


def test_5b_inventory_on_meta_device():
    from flexam_amd.wan_transformer3d_FlexAM import Wan2_2Transformer3DModel_FlexAM
    cfg = dict(O.DIT_5B)
    cfg.pop("eps")
    with torch.device("meta"):
        m = Wan2_2Transformer3DModel_FlexAM(**cfg)
    n = sum(p.numel() for p in m.parameters())
    assert abs(n - 5.032e9) < 0.01e9, n                       # 5.032 B parameters
    assert m.state_dict()["patch_embedding.weight"].shape == (3072, 148, 1, 2, 2)
    assert m.state_dict()["time_projection.1.weight"].shape == (18432, 3072)
    assert m.state_dict()["blocks.29.ffn.0.weight"].shape == (14336, 3072)


def test_config_attributes_and_switches():
    m, _ = tiny_model()
    assert m.config.patch_size == (1, 2, 2) and m.config.get("add_ref_conv") is True and m.config.in_channels == 16
    assert len(m.blocks) == 2 and m.freqs.shape == (1024, 64)
    m.enable_cfg_skip(0.25, 8)
    assert m.cfg_skip_ratio == 0.25 and m.num_inference_steps == 8
    m.disable_cfg_skip()
    m.enable_riflex()
    assert not torch.equal(m.freqs, O.rope_angles(1024, 128))
    m.disable_riflex()
    torch.testing.assert_close(m.freqs, O.rope_angles(1024, 128))
    m.enable_teacache([1.0, 0.0], 4, 0.1)
    assert m.teacache.num_steps == 4
    m.disable_teacache()
    # reference zero-init (FX.py:1172-1188) is reproduced, randomize_zero_init redraws it
    assert float(m.head.head.weight.abs().max()) == 0 and float(m.density_projection[1].weight.abs().max()) == 0
    m.randomize_zero_init()
    assert float(m.head.head.weight.abs().max()) > 0


def test_timestep_rows_two_values():
    from flexam_amd.wan_transformer3d_FlexAM import WanTransformer3DModel_FlexAM as M
    lvid, ref = 12, 4
    t = torch.full((2, lvid), 731.5)
    t[:, :4] = 0.0
    rows, index, U = M._timestep_rows(t, 2, lvid + ref, ref)
    assert U == 2 and rows.tolist() == [0.0, 731.5, 0.0, 731.5]
    full = rows[index.long()].view(2, lvid + ref)
    assert full[0, :ref].tolist() == [731.5] * ref                 # ref tokens take the last token's t (FX.py:900-904)
    assert full[0, ref:ref + 4].tolist() == [0.0] * 4 and full[1, -1].item() == 731.5


def test_cfg_skip_wrapper_halves_and_duplicates():
    from flexam_amd.cfg_optimization import cfg_skip

    class Dummy:
        cfg_skip_ratio, current_steps, num_inference_steps = 0.5, 3, 4
        calls = []

        @cfg_skip()
        def forward(self, x, t, context=None, scalar=3):
            self.calls.append((x.shape[0], t.shape[0], len(context), scalar))
            return x * 2
    d = Dummy()
    out = d.forward(torch.arange(4.0).view(2, 2), torch.tensor([1.0, 2.0]), context=["u", "c"])
    assert d.calls == [(1, 1, 1, 3)] and out.tolist() == [[4.0, 6.0], [4.0, 6.0]]
    d.current_steps = 0
    d.forward(torch.zeros(2, 2), torch.zeros(2), context=["u", "c"])
    assert d.calls[-1] == (2, 2, 2, 3)


def test_forward_fails_loudly_on_cpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m, cfg = tiny_model()
    case = C.dit_case(cfg, 41)
    with pytest.raises(RuntimeError, match="GPU|cuda|libflexam"):
        m(**case)


def test_riflex_table_matches_reference_golden(golden):
    """G11: cos/sin of the REFERENCE model's complex rope table after enable_riflex() (defaults k=6, L_test=66,
    L_test_scale=4.886) and after disable_riflex(); the product stores the angles (FX.py:57-113, 774-795)."""
    fx = golden("g11_riflex")
    m, _ = tiny_model()
    m.enable_riflex()
    torch.testing.assert_close(torch.cos(m.freqs).float(), fx["riflex_cos"], rtol=0, atol=2e-6)
    torch.testing.assert_close(torch.sin(m.freqs).float(), fx["riflex_sin"], rtol=0, atol=2e-6)
    m.disable_riflex()
    torch.testing.assert_close(torch.cos(m.freqs).float(), fx["base_cos"], rtol=0, atol=2e-6)
