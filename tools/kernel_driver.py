"""Dev tool: launches each hot kernel of one DiT block (+ the sampler's fused step) a few times at the BASELINE config-2 shapes
(97x512x896: L = 11648, B = 2, d = 3072, ffn 14336, text 512) -- the process rocprofv3 wraps for per-kernel traces and PMC
passes (tools/pmc_pass.sh).  usage: kernel_driver.py [reps] [name ...]   names: see KERNELS below (default: all)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H
if os.environ.get("FLEXAM_DRIVER_LIB"):          # a diagnostic build of the library (tools/build_attn_variants.py) instead of the tree's
    H.load_library(os.environ["FLEXAM_DRIVER_LIB"])

dev = torch.device("cuda:0"); BF = torch.bfloat16
L, B, d, f, T, nh = 11648, 2, 3072, 14336, 512, 24
M = B * L
g = torch.Generator().manual_seed(0)


def rnd(*s, scale=0.5):
    return (torch.randn(*s, generator=g) * scale).to(BF).to(dev)


def build_kernels():
    """{name: launch closure} of every hot kernel of one DiT block at the config-2 shapes (tools/power_model.py loops them too)."""
    x = torch.randn(M, d, device=dev)                                  # fp32 residual stream
    hb = torch.empty(M, d, dtype=BF, device=dev)
    tab = torch.randn(4, 6, d, device=dev) * 0.1
    rows = ((torch.arange(M) % L >= 448 + 448).int() + 2 * (torch.arange(M) // L).int()).to(torch.int32).to(dev)   # the sampler's two rows per sample
    qkv = rnd(M, 3 * d)
    ao = torch.empty(M, d, dtype=BF, device=dev)
    ffn = rnd(M, f)
    wqkv, wo, w1, w2 = rnd(3 * d, d, scale=0.02), rnd(d, d, scale=0.02), rnd(f, d, scale=0.02), rnd(d, f, scale=0.02)
    bq, bo, b1 = torch.randn(3 * d, device=dev), torch.randn(d, device=dev), torch.randn(f, device=dev)
    nw = torch.ones(d, device=dev)
    cos, sin = torch.ones(L, 64, device=dev), torch.zeros(L, 64, device=dev)
    q4 = qkv.view(B, L, 3 * d)[:, :, 0:d].unflatten(2, (nh, 128))
    k4 = qkv.view(B, L, 3 * d)[:, :, d:2 * d].unflatten(2, (nh, 128))
    v4 = qkv.view(B, L, 3 * d)[:, :, 2 * d:].unflatten(2, (nh, 128))
    ckv = rnd(B, T, 2 * d)
    lat = torch.randn(48, 25, 32, 56, device=dev)
    known, mask = torch.randn_like(lat), (torch.arange(25, device=dev) > 0).float().view(25, 1, 1).expand(25, 32, 56).contiguous()
    tok = torch.randn(2, L, 192, device=dev)
    bufs8 = H.attn_fp8_buffers(B, nh, L, dev)
    KERNELS = {
        "ln_modulate": lambda: H.ln_modulate(x, out=hb, shift=tab[:, 0], scale=tab[:, 1], row_index=rows),
        "ln_affine": lambda: H.ln_modulate(x, out=hb, ln_w=nw, ln_b=nw),
        "gemm_qkv": lambda: H.gemm(hb, wqkv, bq, out=qkv),
        "rmsnorm_rope_qk": lambda: H.rmsnorm_rope(qkv[:, 0:d], nw, qkv[:, d:2 * d], nw, rope_cos=cos, rope_sin=sin, tokens_per_batch=L, head_dim=128),
        "attn_self": lambda: H.attn_fwd(q4, k4, v4, out=ao.view(B, L, nh, 128), prescaled=True),
        "attn_self_general": lambda: (os.environ.__setitem__("FLEXAM_ATTN_FULL", "0"), H.attn_fwd(q4, k4, v4, out=ao.view(B, L, nh, 128), prescaled=True),
                                      os.environ.pop("FLEXAM_ATTN_FULL")),
        "attn_mxfp8_pack": lambda: H.attn_fp8_pack(q4, k4, v4, bufs8),
        "attn_mxfp8": lambda: H.attn_fwd_fp8(bufs8, L, out=ao.view(B, L, nh, 128)),
        "gemm_oproj_residual": lambda: H.gemm_gate_residual(ao, wo, bo, x, gate=tab[:, 2], gate_row=rows),
        "gemm_crossq": lambda: H.gemm(hb, wo, bo, out=qkv[:, 0:d]),
        "rmsnorm_q": lambda: H.rmsnorm_rope(qkv[:, 0:d], nw),
        "attn_cross": lambda: H.attn_fwd(q4, ckv[:, :, 0:d].unflatten(2, (nh, 128)), ckv[:, :, d:].unflatten(2, (nh, 128)), out=ao.view(B, L, nh, 128), prescaled=True),
        "gemm_ffn1_gelu": lambda: H.gemm(hb, w1, b1, out=ffn, epilogue=H.EPI_GELU_TANH),
        "gemm_ffn2_residual": lambda: H.gemm_gate_residual(ffn, w2, bo, x, gate=tab[:, 5], gate_row=rows),
        "gate_residual": lambda: H.gate_residual(x, hb, tab[:, 2], rows),
        "cfg_euler_blend": lambda: H.cfg_euler_blend(tok[0], tok[1], 448, 6.0, -0.02, lat, known, mask),
    }
    return KERNELS


def main():
    args = [a for a in sys.argv[1:]]
    reps = int(args.pop(0)) if args and args[0].isdigit() else 3
    want = set(args)
    KERNELS = build_kernels()
    for name, fn in KERNELS.items():
        if want and name not in want:
            continue
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
    print("ran", sorted(want) if want else "all", "x", reps)


if __name__ == "__main__":
    main()
