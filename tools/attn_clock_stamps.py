"""Dev tool: the in-kernel clock of the self-attention loop (MI355X_MICROARCH.md, DVFS give-back item 6) for both MFMA bodies.  Needs
the diagnostic build `python tools/build_attn_variants.py stamps=-DFLEXAM_ATTN_BODY16,-DFLEXAM_ATTN_STAMPS`: every workgroup stamps
s_memtime (shader cycles) and s_memrealtime (100 MHz) around its tile loop into a buffer nothing else reads.  Per body: 2.5 s of
back-to-back launches on random data, then the median over the workgroups of the last launch.  Also the cycles per call."""
import ctypes, os, sys, time, statistics
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch
from flexam_amd import hip as H
LIBS = sys.argv[1:] or ["stamps"]           # names of tools/probes/libflexam_var_<name>.so built with -DFLEXAM_ATTN_STAMPS
lib = None


def use(name):
    global lib
    lib = H.load_library(os.path.join(root, "tools", "probes", f"libflexam_var_{name}.so"))
    lib.flexam_debug_attn_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]


use(LIBS[0])
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator().manual_seed(0)
L, d = 11648, 3072
qkv = (torch.randn(2, L, 3 * d, generator=g) * 0.5).to(BF).to(dev)
q, k, v = (qkv[:, :, i * d:(i + 1) * d].unflatten(2, (24, 128)) for i in range(3))
o = torch.empty(2, L, 24, 128, dtype=BF, device=dev)
NWG = 2048          # the whole-unit workgroups of a call (blockIdx < 2048 run all keys: 182 tiles)
for rnd in range(2):
    for name, body in ([(LIBS[0], "32"), (LIBS[0], "16")] if len(LIBS) == 1 else [(n, "32") for n in LIBS]):
        use(name)
        if len(LIBS) > 1:
            print(f"-- {name}", flush=True)
        os.environ["FLEXAM_ATTN_BODY"] = body
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 2.5:
            for _ in range(20):
                H.attn_fwd(q, k, v, out=o, prescaled=True)
            torch.cuda.synchronize(); n += 20
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            H.attn_fwd(q, k, v, out=o, prescaled=True)
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / 20 * 1e3
        buf = (ctypes.c_ulonglong * (2 * NWG))()
        assert lib.flexam_debug_attn_stamps(buf, NWG) == 0
        clk = [buf[2 * i] / buf[2 * i + 1] * 100.0 for i in range(NWG) if buf[2 * i + 1] > 0]
        cyc = [buf[2 * i] for i in range(NWG) if buf[2 * i + 1] > 0]
        ghz = statistics.median(clk) / 1e3
        if body == "32" and hasattr(lib, "flexam_debug_attn_barrier_wait"):
            lib.flexam_debug_attn_barrier_wait.argtypes = [ctypes.c_void_p, ctypes.c_int]
            bb = (ctypes.c_ulonglong * (8 * NWG))()
            if lib.flexam_debug_attn_barrier_wait(bb, NWG) == 0:
                per_wave = [statistics.median(bb[8 * i + w] for i in range(NWG)) for w in range(8)]
                print("   cycles in the tile loop's s_barrier per wave (median over workgroups, of %.0f kcycles): " % (statistics.median(cyc) / 1e3)
                      + " ".join(f"w{w}:{v / 1e3:.1f}k" for w, v in enumerate(per_wave)), flush=True)
        print(f"body {body}x{body}: {us:8.1f} us/call   in-kernel clock {ghz:.3f} GHz (median of {len(clk)} workgroups, min {min(clk) / 1e3:.3f} max {max(clk) / 1e3:.3f})   "
              f"tile loop {statistics.median(cyc) / 1e3:8.1f} kcycles per work unit (182 tiles)   call = {us * ghz * 1e-3:.3f} Mcycles", flush=True)
