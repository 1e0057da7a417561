"""Dev tool: anatomy of ONE GEMM launch from in-kernel stamps (diagnostic build `python tools/build_attn_variants.py --src=gemm.hip
gemmstamps=-DFLEXAM_GEMM_STAMPS`): every workgroup writes the 100 MHz counter at kernel entry, when the first K block of its first tile has
landed, at the end of its last K loop and at exit.  Prints, per block GEMM shape at M rows (default 2912 = one rank of eight): how far the
workgroups' starts are spread (dispatch ramp), the wait for the first operands, the K loops, the epilogue, the spread of the exits, and the
launch as the stamps see it against its HIP-event time.  usage: gemm_launch_anatomy.py [M ...]"""
import ctypes, os, statistics, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch
from flexam_amd import hip as H
lib = H.load_library(os.path.join(root, "tools", "probes", "libflexam_var_gemmstamps.so"))
lib.flexam_debug_gemm_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator().manual_seed(0)
d, f = 3072, 14336
NWG = H.num_cus()


def stamps():
    buf = (ctypes.c_ulonglong * (4 * NWG))()
    assert lib.flexam_debug_gemm_stamps(buf, NWG) == 0
    return [[buf[4 * i + j] * 0.01 for j in range(4)] for i in range(NWG)]          # microseconds


for M in [int(a) for a in sys.argv[1:]] or [2912]:
    r = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(BF).to(dev)
    x, hmid = r(M, d), r(M, f)
    w_o, w_qkv, w_f1, w_f2 = r(d, d), r(3 * d, d), r(f, d), r(d, f)
    out_q, out_f = torch.empty(M, 3 * d, dtype=BF, device=dev), torch.empty(M, f, dtype=BF, device=dev)
    xres = torch.zeros(M, d, device=dev)
    b3, bf_, bd = torch.zeros(3 * d, device=dev), torch.zeros(f, device=dev), torch.zeros(d, device=dev)
    gate = torch.randn(4, d, device=dev)
    rows = (torch.arange(M, device=dev) % 2).to(torch.int32)
    cases = {"qkv": lambda: H.gemm(x, w_qkv, b3, out=out_q), "cross-q": lambda: H.gemm(x, w_o, bd, out=out_q[:, :d]),
             "ffn1 + gelu": lambda: H.gemm(x, w_f1, bf_, out=out_f, epilogue=H.EPI_GELU_TANH),
             "ffn2 + residual": lambda: H.gemm_gate_residual(hmid, w_f2, bd, xres, gate=gate, gate_row=rows),
             "o-proj + residual": lambda: H.gemm_gate_residual(x, w_o, bd, xres, gate=gate, gate_row=rows)}
    print(f"== M = {M} (microseconds; p50 / max over the workgroups that ran)")
    for name, fn in cases.items():
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        torch.cuda.synchronize()
        import time as _t
        t0 = _t.perf_counter()
        for _ in range(40):
            fn()
        torch.cuda.synchronize()
        bb = (_t.perf_counter() - t0) / 40 * 1e6                      # back to back, no events in between: what a launch costs in a stream of launches
        fn(); torch.cuda.synchronize()
        st = [w for w in stamps() if w[3] > w[0] > 0]
        t00 = min(w[0] for w in st)
        start = [w[0] - t00 for w in st]; first = [w[1] - w[0] for w in st]; loop = [w[2] - w[1] for w in st]; epi = [w[3] - w[2] for w in st]
        tend = max(w[3] for w in st); tail = [tend - w[3] for w in st]
        med = statistics.median
        print(f"{name:18s} {len(st):3d} wgs | start spread {med(start):5.1f} / {max(start):5.1f} | first operands {med(first):5.1f} / {max(first):5.1f} | "
              f"K loops (+ epilogues between) {med(loop):6.1f} / {max(loop):6.1f} | last epilogue {med(epi):5.1f} / {max(epi):5.1f} | idle before the launch ends {med(tail):5.1f} / {max(tail):5.1f} | "
              f"stamps {tend - t00:6.1f}  event {s.elapsed_time(e) * 1e3:6.1f}  back-to-back {bb:6.1f}", flush=True)
