"""Dev tool: hunts wrong / non-repeatable results of the GEMM tail split-K at given shapes.
usage: splitk_diag.py [poison]   (FLEXAM_GEMM_SPLITK=0 in the environment = control run without the split)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H
dev = torch.device("cuda:0"); BF = torch.bfloat16
poison = "poison" in sys.argv
print("SPLITK env:", os.environ.get("FLEXAM_GEMM_SPLITK"), "poison:", poison)
for (m, n, k, lim) in ((23296, 3072, 14336, 1), (2912, 3072, 14336, 2), (160, 1024, 27648, 2), (640, 1024, 27648, 2), (23296, 3072, 3072, 2)):
    g = torch.Generator().manual_seed(m + k)
    a = torch.randint(-lim, lim + 1, (m, k), generator=g, dtype=torch.int8).float().to(BF).to(dev)
    w = torch.randint(-lim, lim + 1, (n, k), generator=g, dtype=torch.int8).float().to(BF).to(dev)
    b = torch.randint(-8, 9, (n,), generator=g).float().to(dev)
    ref = a.float() @ w.float().t() + b                  # fp32 on integers: exact (|sum| < 2^24)
    x0 = torch.randint(-5, 6, (m, n), generator=g).float().to(dev)
    want_x = x0 + ref.to(BF).float()
    st = torch.cuda.current_stream().cuda_stream
    for epi in ("f32", "resid"):
        bad_counts = []
        for it in range(6):
            if poison:
                ws = H._gemm_workspace(dev, st)
                ws[1024:].view(torch.float32).fill_(float("nan"))
            if epi == "f32":
                out = H.gemm(a, w, b, out_dtype=torch.float32)
                bad = out != ref
                bad |= out.isnan()
            else:
                x = x0.clone()
                H.gemm_gate_residual(a, w, b, x)
                bad = x != want_x
                bad |= x.isnan()
            nb = int(bad.sum())
            bad_counts.append(nb)
            if nb and it < 3:
                idx = bad.nonzero()
                tiles = torch.unique(torch.stack([idx[:, 0] // 256, idx[:, 1] // 256], 1), dim=0)
                got = (out if epi == "f32" else x)[bad][:6].tolist()
                exp = (ref if epi == "f32" else want_x)[bad][:6].tolist()
                print(f"   it {it}: {nb} wrong in {tiles.shape[0]} tiles, first tiles {tiles[:6].tolist()}, rows {idx[:4,0].tolist()} cols {idx[:4,1].tolist()} got {got} want {exp}")
        print(f"M={m} N={n} K={k} {epi}: wrong elements per launch {bad_counts}", flush=True)
