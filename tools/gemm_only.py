"""Dev tool: run only the FFN1-shaped GEMM (and optionally attention) a few times, for rocprofv3 --pmc passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H
dev = torch.device("cuda:0"); BF = torch.bfloat16
what = sys.argv[1] if len(sys.argv) > 1 else "gemm"
g = torch.Generator().manual_seed(0)
M, d, f = 23296, 3072, 14336
if what == "gemm":
    a = (torch.randn(M, d, generator=g) * 0.5).to(BF).to(dev); w = (torch.randn(f, d, generator=g) * 0.5).to(BF).to(dev)
    b = torch.randn(f, device=dev); out = torch.empty(M, f, dtype=BF, device=dev)
    for _ in range(5):
        H.gemm(a, w, b, out=out, epilogue=1)
elif what == "vendor":                      # hipBLASLt on the same FFN1 shape (context for the counters)
    a = (torch.randn(M, d, generator=g) * 0.5).to(BF).to(dev); w = (torch.randn(f, d, generator=g) * 0.5).to(BF).to(dev)
    for _ in range(5):
        out = torch.nn.functional.linear(a, w)
elif what == "ffn2":
    a = (torch.randn(M, f, generator=g) * 0.5).to(BF).to(dev); w = (torch.randn(d, f, generator=g) * 0.5).to(BF).to(dev)
    b = torch.randn(d, device=dev); out = torch.empty(M, d, dtype=BF, device=dev)
    for _ in range(5):
        H.gemm(a, w, b, out=out)
else:
    L = 11648
    qkv = (torch.randn(2, L, 3 * d, generator=g) * 0.5).to(BF).to(dev)
    q, k, v = (qkv[:, :, i * d:(i + 1) * d].unflatten(2, (24, 128)) for i in range(3))
    o = torch.empty(2, L, 24, 128, dtype=BF, device=dev)
    for _ in range(3):
        H.attn_fwd(q, k, v, out=o, prescaled=True)      # the form the DiT engine runs
torch.cuda.synchronize()
