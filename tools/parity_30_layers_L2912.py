"""Dev tool (round-4 verdict item 1c): ONE forward of the full 30-layer, d = 3072 model on a [2,48,25,16,28] latent (L = 2912 tokens: the
per-rank token count at 8 GPUs, 46 key tiles per row; CFG-style pair with two prompt lengths, per-token timesteps) -- HIP through the
drop-in class vs the fp32 oracle, and the oracle with the reference's bf16 roundings (oracle.dit.bf16_emulation) as the yardstick; once with
the seeded weights (logits ~N(0,1)) and once with self_attn.norm_q / norm_k scaled to logit std 6.  Minutes of host time: a record under
profiles/, not part of the suite.  usage: parity_30_layers_L2912.py [layers]"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import cases as C, dit as O

torch.set_num_threads(min(32, torch.get_num_threads()))
nl = int(sys.argv[1]) if len(sys.argv) > 1 else 30
cfg = dict(O.DIT_5B, num_layers=nl)
t0 = time.time()
sd = C.dit_weights_threaded(cfg, 101)
print(f"{sum(v.numel() for v in sd.values()) / 1e9:.2f} B parameters drawn in {time.time() - t0:.0f} s", flush=True)
from flexam_amd.wan_transformer3d_FlexAM import Wan2_2Transformer3DModel_FlexAM
kw = dict(cfg); kw.pop("eps")
with torch.device("cuda:0"):
    m = Wan2_2Transformer3DModel_FlexAM(**kw)
m.load_state_dict(sd, strict=True)
case = C.dit_case(cfg, 16, frames=25, h=16, w=28, batch=2, text_lens=(77, 126))
dcase = {k: ([u.cuda() for u in v] if isinstance(v, list) else (v.cuda() if torch.is_tensor(v) else v)) for k, v in case.items()}
rel = lambda a, b: ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()
base = {k: sd[k].clone() for k in sd if ".self_attn.norm_q.weight" in k or ".self_attn.norm_k.weight" in k}
for S in (1.0, 6.0):
    with torch.no_grad():
        params = dict(m.named_parameters())
        for k, v in base.items():
            sd[k].copy_(v * math.sqrt(S)); params[k].copy_(sd[k])
    st = C.self_attention_row_stats(sd, cfg, case); over = st.pop("over_first_tile")
    print(f"logit std {S:g}: block-0 rows (exp2 units) std {st['std']:.2f}, max - mean {st['max_minus_mean']:.1f}, effective keys {st['n_eff']:.1f} of {st['keys']}, "
          f"rows > 8 above their first tile's maximum {float((over > 8).float().mean()):.2f}", flush=True)
    out = m(**dcase).float().cpu()
    assert m.engine().cond["L"] == 2912
    os.environ["VIDEOX_ATTENTION_TYPE"] = "SAGE_ATTENTION"
    try:
        out8 = m(**dcase).float().cpu()
    finally:
        os.environ.pop("VIDEOX_ATTENTION_TYPE")
    t0 = time.time()
    with torch.no_grad():
        want = O.dit_forward(sd, cfg, **case)
        t1 = time.time()
        with O.bf16_emulation():
            emu = O.dit_forward(sd, cfg, **case)
    print(f"  fp32 oracle {t1 - t0:.0f} s, bf16-emulated oracle {time.time() - t1:.0f} s on {torch.get_num_threads()} host threads")
    print(f"  {nl} layers, L = 2912, logit std {S:g}: HIP bf16 vs fp32 oracle rel-rms {rel(out, want):.3e} psnr {C.psnr(out, want):.1f} dB | "
          f"bf16-emulated oracle vs fp32 oracle rel-rms {rel(emu, want):.3e} psnr {C.psnr(emu, want):.1f} dB | "
          f"HIP with MXFP8 self-attention rel-rms {rel(out8, want):.3e} psnr {C.psnr(out8, want):.1f} dB", flush=True)
