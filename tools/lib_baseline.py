"""Dev tool: what the vendor libraries reach on this GPU for the DiT's GEMM / attention shapes (context for the
roofline fractions in DESIGN.md; NOT used by the product path)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from flexam_amd import hip as H
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator().manual_seed(0)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


M = 23296
for name, N, K in (("qkv", 9216, 3072), ("ffn1", 14336, 3072), ("ffn2", 3072, 14336), ("oproj", 3072, 3072)):
    a = (torch.randn(M, K, generator=g) * 0.5).to(BF).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.5).to(BF).to(dev)
    b = torch.randn(N, device=dev)
    out = torch.empty(M, N, dtype=BF, device=dev)
    fl = 2.0 * M * N * K
    t_mine = timeit(lambda: H.gemm(a, w, b, out=out))
    t_lib = timeit(lambda: F.linear(a, w))
    print(f"{name:6s} M={M} N={N} K={K}: flexam {fl / t_mine / 1e12:7.1f} TF/s   torch F.linear (hipBLASLt/rocBLAS) {fl / t_lib / 1e12:7.1f} TF/s")

L, d = 11648, 3072
qkv = (torch.randn(2, L, 3 * d, generator=g) * 0.5).to(BF).to(dev)
q, k, v = (qkv[:, :, i * d:(i + 1) * d].unflatten(2, (24, 128)) for i in range(3))
o = torch.empty(2, L, 24, 128, dtype=BF, device=dev)
fl = 4.0 * 2 * 24 * L * L * 128
t_mine = timeit(lambda: H.attn_fwd(q, k, v, out=o), 5)
qt, kt, vt = (x.transpose(1, 2).contiguous() for x in (q, k, v))
try:
    t_lib = timeit(lambda: F.scaled_dot_product_attention(qt, kt, vt), 5)
    lib = f"{fl / t_lib / 1e12:7.1f} TF/s"
except Exception as e:  # noqa
    lib = f"failed: {e}"
print(f"attn   B=2 H=24 L={L} D=128: flexam {fl / t_mine / 1e12:7.1f} TF/s   torch SDPA {lib}")
