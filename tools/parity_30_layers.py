"""Dev tool (round-4 verdict item 1c, round-5 item 7): ONE forward of the full 30-layer, d = 3072 model (5.03 B seeded parameters) --
HIP through the drop-in class against the fp32 oracle, with the oracle carrying the reference's bf16 roundings (oracle.dit.bf16_emulation)
as the yardstick; once with the seeded weights (logits ~N(0,1)) and once with self_attn.norm_q / norm_k scaled to logit std 6.

  parity_30_layers.py [--layers 30] [--h 16 --w 28] [--batch 2] [--host-oracle seeded|all|none]

  --h 16 --w 28 (default)  latent [B,48,25,16,28], L = 2912 tokens: the per-rank token count at 8 GPUs (r5 record)
  --h 32 --w 56 --batch 1  latent [1,48,25,32,56], L = 11648: BASELINE configs[1]'s token count, one sample (r6 record)

Where the oracle runs.  It is the same fp32 torch code either way (oracle/dit.py, nothing of flexam_amd in it).  At L = 11648 one
30-layer forward of it takes ~6 min on 32 host threads; the four oracle forwards of this record (fp32 and bf16-emulated, two logit
scales) are therefore run by torch ON THE GPU in fp32 (`with torch.device("cuda")`: rocBLAS / hipBLASLt fp32 GEMMs, no TF32, no bf16 --
seconds each), and `--host-oracle seeded` (default) runs the seeded fp32 forward on the host cores as well and prints how far the two
executions of the oracle are apart (fp32 summation order only).  Minutes of box time: a record under profiles/, not part of the suite."""
import argparse, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import cases as C, dit as O

ap = argparse.ArgumentParser()
ap.add_argument("--layers", type=int, default=30)
ap.add_argument("--h", type=int, default=16)
ap.add_argument("--w", type=int, default=28)
ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--host-oracle", choices=["seeded", "all", "none"], default="seeded")
ap.add_argument("--scales", type=float, nargs="*", default=[1.0, 6.0])
ap.add_argument("--fp8", action="store_true", help="also run the fp8 QKV / FFN variant (BASELINE configs[4]) and fp8 + MXFP8 attention")
args = ap.parse_args()

torch.set_num_threads(min(32, torch.get_num_threads()))
torch.backends.cuda.matmul.allow_tf32 = False
nl = args.layers
cfg = dict(O.DIT_5B, num_layers=nl)
t0 = time.time()
sd = C.dit_weights_threaded(cfg, 101)
print(f"{sum(v.numel() for v in sd.values()) / 1e9:.2f} B parameters drawn in {time.time() - t0:.0f} s", flush=True)
from flexam_amd.wan_transformer3d_FlexAM import Wan2_2Transformer3DModel_FlexAM
kw = dict(cfg); kw.pop("eps")
with torch.device("cuda:0"):
    m = Wan2_2Transformer3DModel_FlexAM(**kw)
m.load_state_dict(sd, strict=True)
case = C.dit_case(cfg, 16, frames=25, h=args.h, w=args.w, batch=args.batch, text_lens=(77, 126))
L = 25 * (args.h // 2) * (args.w // 2) + (args.h // 2) * (args.w // 2)
to_dev = lambda d: {k: ([u.cuda() for u in v] if isinstance(v, list) else (v.cuda() if torch.is_tensor(v) else v)) for k, v in d.items()}
dcase = to_dev(case)
rel = lambda a, b: ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()
base = {k: sd[k].clone() for k in sd if ".self_attn.norm_q.weight" in k or ".self_attn.norm_k.weight" in k}


def oracle_on_gpu(emulate: bool):
    sd_dev = {k: v.cuda() for k, v in sd.items()}
    with torch.no_grad(), torch.device("cuda:0"):
        if emulate:
            with O.bf16_emulation():
                out = O.dit_forward(sd_dev, cfg, **dcase)
        else:
            out = O.dit_forward(sd_dev, cfg, **dcase)
    torch.cuda.synchronize()
    out = out.float().cpu()
    del sd_dev
    torch.cuda.empty_cache()
    return out


for S in args.scales:
    with torch.no_grad():
        params = dict(m.named_parameters())
        for k, v in base.items():
            sd[k].copy_(v * math.sqrt(S)); params[k].copy_(sd[k])
    if args.batch * L <= 8192:                     # the row statistics materialise block 0's scores on the host: the small shape only
        st = C.self_attention_row_stats(sd, cfg, case); over = st.pop("over_first_tile")
        print(f"logit std {S:g}: block-0 rows (exp2 units) std {st['std']:.2f}, max - mean {st['max_minus_mean']:.1f}, effective keys {st['n_eff']:.1f} of {st['keys']}, "
              f"rows > 8 above their first tile's maximum {float((over > 8).float().mean()):.2f}", flush=True)
    out = m(**dcase).float().cpu()
    assert m.engine().cond["L"] == L
    os.environ["VIDEOX_ATTENTION_TYPE"] = "SAGE_ATTENTION"
    try:
        out8 = m(**dcase).float().cpu()
    finally:
        os.environ.pop("VIDEOX_ATTENTION_TYPE")
    out_f8 = out_f8s = None
    if args.fp8:
        m.enable_fp8_gemm(True)
        try:
            out_f8 = m(**dcase).float().cpu()
            os.environ["VIDEOX_ATTENTION_TYPE"] = "SAGE_ATTENTION"
            try:
                out_f8s = m(**dcase).float().cpu()
            finally:
                os.environ.pop("VIDEOX_ATTENTION_TYPE")
        finally:
            m.enable_fp8_gemm(False)
    t0 = time.time()
    want = oracle_on_gpu(False)
    t1 = time.time()
    emu = oracle_on_gpu(True)
    print(f"  fp32 oracle {t1 - t0:.0f} s, bf16-emulated oracle {time.time() - t1:.0f} s (torch fp32 on the GPU)")
    if args.host_oracle == "all" or (args.host_oracle == "seeded" and S == 1.0):
        th = time.time()
        with torch.no_grad():
            host = O.dit_forward(sd, cfg, **case)
        print(f"  the same fp32 oracle on {torch.get_num_threads()} host threads: {time.time() - th:.0f} s; GPU-run oracle vs host-run oracle rel-rms {rel(want, host):.3e} "
              f"psnr {C.psnr(want, host):.1f} dB; HIP bf16 vs the HOST-run oracle rel-rms {rel(out, host):.3e} psnr {C.psnr(out, host):.1f} dB", flush=True)
    print(f"  {nl} layers, L = {L}, batch {args.batch}, logit std {S:g}: HIP bf16 vs fp32 oracle rel-rms {rel(out, want):.3e} psnr {C.psnr(out, want):.1f} dB | "
          f"bf16-emulated oracle vs fp32 oracle rel-rms {rel(emu, want):.3e} psnr {C.psnr(emu, want):.1f} dB | "
          f"HIP with MXFP8 self-attention rel-rms {rel(out8, want):.3e} psnr {C.psnr(out8, want):.1f} dB"
          + (f" | fp8 QKV / FFN rel-rms {rel(out_f8, want):.3e} psnr {C.psnr(out_f8, want):.1f} dB | fp8 + MXFP8 attention rel-rms {rel(out_f8s, want):.3e} "
             f"psnr {C.psnr(out_f8s, want):.1f} dB" if out_f8 is not None else ""), flush=True)
