"""Dev tool: per-call time distribution of the N = K = 3072 GEMM over fresh allocations (hunting the intermittent slow
run noted in profiles/r1e_gemm_notes.txt #10)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H
dev = torch.device("cuda:0"); BF = torch.bfloat16
M, N, K = 23296, 3072, 3072
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 12
calls = 20
keep = []
for trial in range(trials):
    if trial % 3 == 2:
        keep.append(torch.empty((trial * 7 + 1) * 1000_003, dtype=torch.uint8, device=dev))     # shift later allocations
    a = (torch.randn(M, K, device=dev) * 0.5).to(BF)
    w = (torch.randn(N, K, device=dev) * 0.5).to(BF)
    b = torch.randn(N, device=dev)
    out = torch.empty(M, N, dtype=BF, device=dev)
    for _ in range(3):
        H.gemm(a, w, b, out=out)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(calls)]
    torch.cuda.synchronize()
    for s, e in ev:
        s.record(); H.gemm(a, w, b, out=out); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in ev)
    print(f"trial {trial:2d}: a@{a.data_ptr() & 0xffffffff:#x} out@{out.data_ptr() & 0xffffffff:#x}  min {ts[0]:.3f}  med {ts[calls // 2]:.3f}  max {ts[-1]:.3f} ms  slow(>2x) {sum(t > 2 * ts[0] for t in ts)}")
    del a, w, b, out
