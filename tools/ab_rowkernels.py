"""Dev tool: row-kernel variants of tools/probes/rowkernel_probe.hip against the library's ln_modulate / rmsnorm at the DiT shape, round-robin medians."""
import ctypes, os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H
P = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "librowkernel_probe.so"))
dev = torch.device("cuda:0")
M, C = 23296, 3072
x = torch.randn(M, C, device=dev)
tab = torch.randn(4, 6, C, device=dev) * 0.1
ri = (torch.arange(M, device=dev) % 2).to(torch.int32)
out = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
q = torch.randn(M, C, device=dev).to(torch.bfloat16)
w = torch.ones(C, device=dev)
st = lambda: torch.cuda.current_stream().cuda_stream
arms = {
    "lib ln_modulate": lambda: H.ln_modulate(x, out=out, shift=tab[:, 0], scale=tab[:, 1], row_index=ri),
    "v0 (lib shape, 4 rows/block)": lambda: P.probe_ln(0, ctypes.c_void_p(x.data_ptr()), ctypes.c_int64(M), ctypes.c_void_p(tab.data_ptr()), ctypes.c_void_p(ri.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(st())),
    "v1 dense 16B loads, 8B stores": lambda: P.probe_ln(1, ctypes.c_void_p(x.data_ptr()), ctypes.c_int64(M), ctypes.c_void_p(tab.data_ptr()), ctypes.c_void_p(ri.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(st())),
    "v0, 8 rows/block": lambda: P.probe_ln(2, ctypes.c_void_p(x.data_ptr()), ctypes.c_int64(M), ctypes.c_void_p(tab.data_ptr()), ctypes.c_void_p(ri.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(st())),
    "v0, 2 rows/block": lambda: P.probe_ln(3, ctypes.c_void_p(x.data_ptr()), ctypes.c_int64(M), ctypes.c_void_p(tab.data_ptr()), ctypes.c_void_p(ri.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(st())),
    "v0, 1 row/block": lambda: P.probe_ln(5, ctypes.c_void_p(x.data_ptr()), ctypes.c_int64(M), ctypes.c_void_p(tab.data_ptr()), ctypes.c_void_p(ri.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(st())),
    "fp32->bf16 copy (same bytes)": lambda: P.probe_ln(4, ctypes.c_void_p(x.data_ptr()), ctypes.c_int64(M), ctypes.c_void_p(tab.data_ptr()), ctypes.c_void_p(ri.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(st())),
    "lib rmsnorm (no rope)": lambda: H.rmsnorm_rope(q, w),
    "w1 rmsnorm wave per row": lambda: P.probe_rms(ctypes.c_void_p(q.data_ptr()), ctypes.c_int64(M), ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(st())),
}
bytes_ = {k: (M * C * 4.0 if "rms" in k else M * C * 6.0) for k in arms}
res = {k: [] for k in arms}
names = list(arms)
for rnd in range(7):
    for k in (names if rnd % 2 == 0 else names[::-1]):
        arms[k](); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): arms[k]()
        torch.cuda.synchronize()
        res[k].append((time.perf_counter() - t0) / 20)
for k in names:
    t = statistics.median(res[k])
    print(f"{k:34s} {t * 1e6:7.1f} us  {bytes_[k] / t / 1e9:7.0f} GB/s algorithmic")
