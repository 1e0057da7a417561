"""Dev tool: fp8 FFN2 (gate-residual) GEMM time over tile height (FLEXAM_GEMM_MT) and L2 group height (FLEXAM_GEMM_GM) at the two
token counts of the bench (23296 = 2 x 11648, 45760 = 2 x 22880)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H
dev = torch.device("cuda:0")
N, K = 3072, 14336
for M in (23296, 45760):
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    a8, sa = H.quantize_rows_fp8(a)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
    w8, sw = H.quantize_rows_fp8(w)
    x = torch.zeros(M, N, device=dev)
    gate = torch.ones(2, N, device=dev)
    bias = torch.zeros(N, device=dev)
    for mt in ("", "8", "7", "6", "5", "4"):
        for gm in ("", "2", "8"):
            os.environ.pop("FLEXAM_GEMM_MT", None); os.environ.pop("FLEXAM_GEMM_GM", None)
            if mt: os.environ["FLEXAM_GEMM_MT"] = mt
            if gm: os.environ["FLEXAM_GEMM_GM"] = gm
            f = lambda: H.gemm_fp8_gate_residual(a8, sa, w8, sw, bias, x, gate=gate, rows_per_batch=M // 2)
            f(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10): f()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 10
            print(f"M {M} MT {mt or 'auto'} GM {gm or '4'}: {dt * 1e3:.3f} ms  {2.0 * M * N * K / dt / 1e12:.0f} TF/s")
