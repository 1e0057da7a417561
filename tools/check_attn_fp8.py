"""Dev tool: the MXFP8 self-attention (flexam_attn_fp8_pack + flexam_attn_fwd_fp8) against fp32 attention on the same bf16 inputs,
and its time next to the bf16 kernel at the production shape."""
import os, sys, math, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H
dev = torch.device("cuda:0"); BF = torch.bfloat16
LOG2E = 1.4426950408889634


def ref_attn(q, k, v):          # q prescaled (exp2 units)
    s = torch.einsum("blhd,bmhd->bhlm", q.float(), k.float()) * math.log(2.0)
    return torch.einsum("bhlm,bmhd->blhd", torch.softmax(s, dim=-1), v.float())


def case(B, Hh, L, sharp=1.0, splits=None, seed=0, outlier=False):
    g = torch.Generator(device="cpu").manual_seed(seed)
    q = torch.randn(B, L, Hh, 128, generator=g) * (128 ** -0.5 * LOG2E * sharp)
    k = torch.randn(B, L, Hh, 128, generator=g)
    v = torch.randn(B, L, Hh, 128, generator=g)
    if outlier:
        k[..., 5] *= 20; q[..., 77] *= 10; v[..., 100] *= 30
    q, k, v = (t.to(BF).to(dev) for t in (q, k, v))
    bufs = H.attn_fp8_pack(q, k, v)
    kw = {} if splits is None else dict(kv_splits=splits[0], split_from_unit=splits[1])
    o8 = H.attn_fwd_fp8(bufs, L, **kw).float()
    ob = H.attn_fwd(q, k, v, prescaled=True).float()
    want = ref_attn(q, k, v)
    rel = lambda a: float((a - want).norm() / want.norm())
    print(f"B={B} H={Hh} L={L} sharp={sharp} splits={splits} outlier={outlier}: rel-RMS fp8 {rel(o8):.4f}  bf16 {rel(ob):.4f}   max|err| fp8 {float((o8 - want).abs().max()):.4f}  finite {bool(torch.isfinite(o8).all())}", flush=True)
    return rel(o8)


if __name__ == "__main__":
    case(1, 1, 64)
    case(1, 2, 256)
    case(1, 2, 256, sharp=8.0)
    case(1, 1, 300)
    case(2, 3, 1111)
    case(1, 2, 1024, splits=(2, 0))
    case(1, 2, 1024, splits=(4, 5))
    case(1, 2, 2048, sharp=4.0, outlier=True)
    if len(sys.argv) > 1:
        B, Hh, L = 2, 24, 11648
        q = (torch.randn(B, L, Hh, 128, device=dev) * (128 ** -0.5 * LOG2E)).to(BF)
        k = torch.randn(B, L, Hh, 128, device=dev).to(BF)
        v = torch.randn(B, L, Hh, 128, device=dev).to(BF)
        bufs = H.attn_fp8_pack(q, k, v)
        out = torch.empty(B, L, Hh, 128, device=dev, dtype=BF)
        fns = {"bf16": lambda: H.attn_fwd(q, k, v, out=out, prescaled=True), "fp8": lambda: H.attn_fwd_fp8(bufs, L, out=out), "pack": lambda: H.attn_fp8_pack(q, k, v, bufs)}
        res = {n: [] for n in fns}
        for rnd in range(5):
            for n, fn in fns.items():
                fn(); fn()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(5): fn()
                e.record(); torch.cuda.synchronize()
                res[n].append(s.elapsed_time(e) / 5)
        for n in fns:
            print(f"{n}: median {statistics.median(res[n]) * 1e3:.0f} us")
        o8 = H.attn_fwd_fp8(bufs, L).float(); ob = H.attn_fwd(q, k, v, prescaled=True).float()
        print("production shape, fp8 vs bf16 kernel: rel-RMS", float((o8 - ob).norm() / ob.norm()))
