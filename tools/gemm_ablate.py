"""Dev tool: time the FFN1 / QKV / FFN2 GEMMs with a -DFLEXAM_GEMM_ABLATE build of the library (tools/probes/libflexam_ablate.so,
built by hand: hipcc ... -DFLEXAM_GEMM_ABLATE) under FLEXAM_GEMM_DEBUG bit masks.  TIMING ONLY: ablated runs compute garbage."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H
H.load_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "libflexam_ablate.so"))
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator().manual_seed(0)
M = 23296
shapes = (("qkv", 9216, 3072), ("ffn1", 14336, 3072), ("ffn2", 3072, 14336), ("oproj+gate", 3072, 3072), ("ffn2+gate", 3072, 14336))
xres = torch.randn(M, 3072, device=dev); gate = torch.randn(4, 3072, device=dev)
rows = torch.randint(0, 4, (M,), dtype=torch.int32, device=dev)
ten = {}
for name, N, K in shapes:
    ten[name] = ((torch.randn(M, K, generator=g) * 0.5).to(BF).to(dev), (torch.randn(N, K, generator=g) * 0.5).to(BF).to(dev),
                 torch.randn(N, device=dev), torch.empty(M, N, dtype=BF, device=dev))
for mask in [int(x) for x in sys.argv[1:]] or [0]:
    os.environ["FLEXAM_GEMM_DEBUG"] = str(mask)
    line = f"debug mask {mask:2d}:"
    for name, N, K in shapes:
        a, w, b, out = ten[name]
        fn = (lambda: H.gemm_gate_residual(a, w, b, xres, gate=gate, gate_row=rows)) if name.endswith("+gate") else (lambda: H.gemm(a, w, b, out=out))
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        line += f"  {name} {2.0 * M * N * K / dt / 1e12:7.1f} TF/s"
    print(line, flush=True)
