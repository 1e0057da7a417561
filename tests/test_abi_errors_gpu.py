"""GPU: the C ABI rejects bad arguments with a FLEXAM_E_* code and a message (flexam_last_error) instead of launching --
the host mirror surfaces them as RuntimeError.  Error behaviour is part of the drop-in contract: no silent fallback."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF, F32 = torch.bfloat16, torch.float32


def dev():
    return torch.device("cuda:0")


def test_gemm_argument_errors():
    from flexam_amd import hip as H
    a = torch.zeros(64, 96, dtype=BF, device=dev())                 # K = 96 is not a multiple of 64
    w = torch.zeros(32, 96, dtype=BF, device=dev())
    with pytest.raises(RuntimeError, match="multiple of 64"):
        H.gemm(a, w)
    a = torch.zeros(64, 64, dtype=BF, device=dev())
    with pytest.raises(RuntimeError, match="K mismatch"):
        H.gemm(a, torch.zeros(32, 128, dtype=BF, device=dev()))
    with pytest.raises(RuntimeError, match="multiples of 4"):
        H.gemm(a, torch.zeros(30, 64, dtype=BF, device=dev()))     # N = 30
    with pytest.raises(RuntimeError, match="out shape"):
        H.gemm(a, torch.zeros(32, 64, dtype=BF, device=dev()), out=torch.zeros(64, 16, dtype=BF, device=dev()))
    with pytest.raises(RuntimeError, match="f32 output"):
        H.gemm(a, torch.zeros(32, 64, dtype=BF, device=dev()), epilogue=H.EPI_GELU_TANH, out_dtype=F32)
    with pytest.raises(RuntimeError):
        H.gemm(a.float(), torch.zeros(32, 64, dtype=BF, device=dev()))    # wrong dtype is caught on the host side


def test_attention_argument_errors():
    from flexam_amd import hip as H
    q = torch.zeros(1, 64, 2, 64, dtype=BF, device=dev())           # head_dim 64
    with pytest.raises(RuntimeError, match="head_dim"):
        H.attn_fwd(q, q, q)
    q = torch.zeros(1, 64, 2, 128, dtype=BF, device=dev())
    with pytest.raises(RuntimeError, match="packed"):
        H.attn_fwd(q.transpose(1, 2).contiguous().transpose(1, 2), q, q)      # heads not packed along the row
    with pytest.raises(RuntimeError, match="key splits"):
        H.attn_fwd(q, q, q, kv_splits=4)                            # one key tile cannot be cut in four


def test_row_kernel_and_sampler_argument_errors():
    from flexam_amd import hip as H
    x = torch.zeros(8, 100, device=dev())                           # C = 100 is not a multiple of 8
    with pytest.raises(RuntimeError, match="C%8"):
        H.ln_modulate(x, out=torch.zeros(8, 100, dtype=BF, device=dev()))
    lat = torch.zeros(48, 2, 5, 6, device=dev())                    # odd H
    tok = torch.zeros(100, 192, device=dev())
    with pytest.raises(RuntimeError, match="even"):
        H.cfg_euler_blend(tok, None, 0, 1.0, -0.1, lat)
    t = torch.zeros(4, 8, device=dev())
    with pytest.raises(RuntimeError, match="terms"):
        H.lincomb(t, [(1.0, t)] * 9)
    with pytest.raises(RuntimeError, match="shape"):
        H.lincomb(t, [(1.0, torch.zeros(4, 4, device=dev()))])


def test_models_refuse_to_run_on_cpu():
    """No CPU or eager fallback anywhere: the drop-in classes raise when their parameters are not on a GPU."""
    from flexam_amd import AutoencoderKLWan3_8, WanT5EncoderModel, Wan2_2Transformer3DModel_FlexAM
    from oracle import dit as O
    from oracle import t5 as OT
    kw = dict(O.DIT_TINY)
    kw.pop("eps")
    with pytest.raises(RuntimeError, match="GPU"):
        Wan2_2Transformer3DModel_FlexAM(**kw).engine()
    with pytest.raises(RuntimeError, match="GPU"):
        AutoencoderKLWan3_8(c_dim=16, dec_dim=16).decode(torch.zeros(1, 48, 1, 2, 2))
    with pytest.raises(RuntimeError, match="GPU"):
        WanT5EncoderModel(**OT.T5_TINY)(torch.zeros(1, 8, dtype=torch.long), torch.ones(1, 8, dtype=torch.long))
