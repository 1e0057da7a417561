"""Build-container only: oracle vs the live reference modules on cases the committed fixtures do
not cover (other seeds / shapes).  Skipped where /root/reference is absent (GPU box)."""
import warnings

import pytest
import torch

from oracle import cases as C
from oracle import dit as O
from oracle import ref_import
from oracle import vae as OV

pytestmark = pytest.mark.needs_reference


@torch.no_grad()
def test_dit_matches_reference_other_shape():
    warnings.filterwarnings("ignore")
    ref = ref_import.load_reference()
    cfg = dict(O.DIT_TINY, num_layers=1)
    sd = C.dit_weights(cfg, 99)
    m = ref.dit.Wan2_2Transformer3DModel_FlexAM(
        model_type="ti2v", in_dim=148, dim=256, ffn_dim=512, num_heads=2, num_layers=1, out_dim=48, text_dim=64, text_len=16,
        add_ref_conv=True, in_dim_ref_conv=48, add_cnn_block=True, in_dim_cnn_block=288, out_dim_cnn_block=48).eval()
    m.load_state_dict(sd)
    case = C.dit_case(cfg, 5, frames=2, h=12, w=20, batch=1, t_value=133.0)
    torch.testing.assert_close(O.dit_forward(sd, cfg, **case), m(**case), rtol=2e-4, atol=2e-5)


@torch.no_grad()
def test_vae_decode_matches_reference_four_chunks():
    warnings.filterwarnings("ignore")
    ref = ref_import.load_reference()
    V = ref.vae.AutoencoderKLWan2_2_(dim=32, dec_dim=16, z_dim=48, temperal_downsample=[False, True, True]).eval()
    vsd = C.vae_weights(C.VAE_SMALL, seed=77, prefix="")
    V.load_state_dict(vsd, strict=False)
    z = C.vae_case(seed=78, frames=4, h=2, w=4)
    mean, std = torch.tensor(OV.LATENT_MEAN), torch.tensor(OV.LATENT_STD)
    want = V.decode(z, [mean, 1.0 / std]).clamp(-1, 1)
    got = OV.vae_decode(vsd, z, C.VAE_SMALL["temporal_up"], OV.LATENT_MEAN, OV.LATENT_STD, prefix="")
    assert got.shape == (1, 3, 13, 32, 64)
    torch.testing.assert_close(got, want, rtol=2e-4, atol=2e-5)
