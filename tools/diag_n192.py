"""diagnostic: where does the 192-wide gate-residual epilogue go wrong?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H
BF = torch.bfloat16
os.environ["FLEXAM_GEMM_N192"] = "2"
g = torch.Generator().manual_seed(1)
for (m, n, k) in ((384, 384, 128), (2912, 3072, 3072)):
    a = torch.randint(-2, 3, (m, k), generator=g).float()
    w = torch.randint(-2, 3, (n, k), generator=g).float()
    b = torch.randint(-8, 9, (n,), generator=g).float()
    want = a @ w.t() + b
    x0 = torch.randint(-5, 6, (m, n), generator=g).float()
    x = x0.clone().cuda()
    H.gemm_gate_residual(a.to(BF).cuda(), w.to(BF).cuda(), b.cuda(), x)
    bad = (x.cpu() != x0 + want.to(BF).float())
    print(m, n, k, "bad fraction", float(bad.float().mean()))
    if bad.any():
        r, c = bad.nonzero(as_tuple=True)
        print(" bad rows mod 192 hist (by 16):", torch.bincount((r % 192) // 16, minlength=12).tolist())
        print(" bad rows mod 16 hist:", torch.bincount(r % 16, minlength=16).tolist())
        print(" bad cols mod 192 hist (by 4):", torch.bincount((c % 192) // 4, minlength=48).tolist())
        d = (x.cpu() - x0)[bad]
        wv = want.to(BF).float()[bad]
        print(" first bad: got delta", d[:8].tolist(), "want delta", wv[:8].tolist(), "at", list(zip(r[:8].tolist(), c[:8].tolist())))
        # is the delta some other column's / row's y?
        r0, c0 = int(r[0]), int(c[0])
        yy = want.to(BF).float()
        hits = (yy[r0 // 192 * 192:(r0 // 192 + 1) * 192, c0 // 192 * 192:(c0 // 192 + 1) * 192] == d[0]).nonzero()
        print(" positions in the tile whose y equals the wrong delta:", hits[:10].tolist(), "bad pos in tile", (r0 % 192, c0 % 192))
