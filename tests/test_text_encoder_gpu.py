"""GPU: the umT5 text encoder on the HIP path against golden G12 (outputs of the REFERENCE
WanT5EncoderModel on seeded weights: per-layer relative-position bias, right-padded prompts of 24 and 9 tokens) and
the `prompt=` entry of the sampler with a stand-in tokenizer.  Tolerance: bf16 GEMM operands, fp32 residual stream
and softmax -> rel-RMS <= 2e-2, PSNR >= 40 dB over the VALID token rows (the pipeline slices them, PIPE.py:232)."""
import pytest
import torch

from oracle import cases as C
from oracle import t5 as OT

pytestmark = pytest.mark.gpu


def build(cfg, seed=5):
    from flexam_amd import WanT5EncoderModel
    m = WanT5EncoderModel(**cfg)
    sd = OT.seeded_t5_weights(cfg, seed)
    m.load_state_dict(sd, strict=True)
    return m.to("cuda:0"), sd


def check(got, want, what):
    got, want = got.float().cpu(), want.float()
    rel = ((got - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item()
    p = C.psnr(got, want)
    print(f"{what}: rel-rms {rel:.3e}, psnr {p:.1f} dB")
    assert rel <= 2e-2 and p >= 40.0


def test_t5_matches_reference_golden(golden):
    fx = golden("g12_t5")
    cfg = dict(OT.T5_TINY)
    m, sd = build(cfg)
    ids, mask = OT.t5_case(cfg)
    out = m(ids.cuda(), mask.cuda())[0]
    assert out.shape == (2, 24, 128)
    for b, n in enumerate((24, 9)):
        check(out[b, :n], fx["out"][b, :n], f"t5 g12 prompt {b} ({n} tokens)")


def test_t5_shared_bias_wider_config_vs_oracle():
    cfg = dict(OT.T5_TINY, shared_pos=True, num_heads=4, dim=256, dim_attn=256, dim_ffn=384, num_layers=3)
    m, sd = build(cfg, seed=9)
    ids, mask = OT.t5_case(cfg, seed=78, batch=2, length=40, lens=(33, 40))
    want = OT.t5_encode(sd, cfg, ids, mask)
    out = m(ids.cuda(), mask.cuda())[0]
    for b, n in enumerate((33, 40)):
        check(out[b, :n], want[b, :n], f"t5 shared-pos prompt {b}")
    with pytest.raises(ValueError):
        m(ids[:, :39].cuda(), mask[:, :39].cuda())


class FakeTokenizer:
    """Stand-in for the umT5 tokenizer: deterministic ids from characters, right padding to max_length."""

    def __call__(self, prompt, padding=None, max_length=None, truncation=None, add_special_tokens=True, return_tensors="pt"):
        prompt = [prompt] if isinstance(prompt, str) else prompt
        rows = [[1 + (ord(ch) % 97) for ch in p][: (max_length or 10 ** 9) - 1] + [1] for p in prompt]
        width = max_length if padding == "max_length" else max(len(r) for r in rows)
        ids = torch.zeros(len(rows), width, dtype=torch.long)
        mask = torch.zeros(len(rows), width, dtype=torch.long)
        for i, r in enumerate(rows):
            ids[i, :len(r)] = torch.tensor(r)
            mask[i, :len(r)] = 1
        return type("Enc", (), dict(input_ids=ids, attention_mask=mask))()


def test_pipeline_prompt_strings_use_the_text_encoder():
    from flexam_amd import Wan2_2FunControlPipeline_FlexAM
    cfg = dict(OT.T5_TINY)
    enc, sd = build(cfg)
    pipe = Wan2_2FunControlPipeline_FlexAM(tokenizer=FakeTokenizer(), text_encoder=enc, transformer=enc)   # transformer only supplies .device here
    ctx_c, ctx_u = pipe.encode_prompt("a red car turning left", "blurry", True, max_sequence_length=32, device=torch.device("cuda:0"))
    tok = FakeTokenizer()
    for text, got in (("a red car turning left", ctx_c[0]), ("blurry", ctx_u[0])):
        e = tok(text, padding="max_length", max_length=32)
        want = OT.t5_encode(sd, cfg, e.input_ids, e.attention_mask)[0, : int(e.attention_mask.sum())]
        assert got.shape == want.shape
        check(got, want, f"encode_prompt({text!r})")
