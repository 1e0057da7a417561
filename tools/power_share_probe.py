"""Dev tool: how the power budget couples concurrent work.  The pure-MFMA probe (tools/probes/mfma_toggle_probe.hip, random operands) on
256 / 128 / 64 workgroups (one per CU), alone and with a bandwidth-bound copy running on a second stream: does the matrix pipe run faster
per CU when fewer CUs draw MFMA power, and how much does a streaming kernel next to it take away?"""
import ctypes, os, sys, time, statistics
import torch
P = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "libmfma_toggle_probe.so"))
dev = torch.device("cuda:0")
src = (torch.randn(4096 * 8, device=dev) * 0.5).to(torch.bfloat16)
out = torch.zeros(512, device=dev)
a = torch.randn(256 * 1024 * 1024 // 4, device=dev)          # 256 MiB in, 256 MiB out per copy
b = torch.empty_like(a)
side = torch.cuda.Stream()
iters = 4000


def mfma(blocks, mode=5):
    P.run_probe(mode, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(out.data_ptr()), iters, blocks, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))


def timed(blocks, with_copy):
    torch.cuda.synchronize()
    n_copy = 0
    t0 = time.perf_counter()
    if with_copy:
        with torch.cuda.stream(side):
            for _ in range(12):
                b.copy_(a); n_copy += 1
    mfma(blocks)
    ev = torch.cuda.Event(); ev.record()
    ev.synchronize()
    t = time.perf_counter() - t0
    torch.cuda.synchronize()
    return t


for mode, name in ((5, "16x16x32"), (0, "32x32x16")):
    for blocks in (256, 128, 64):
        for with_copy in (False, True):
            ts = []
            for _ in range(7):
                torch.cuda.synchronize()
                if with_copy:
                    with torch.cuda.stream(side):
                        for _ in range(12):
                            b.copy_(a)
                t0 = time.perf_counter()
                P.run_probe(mode, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(out.data_ptr()), iters, blocks, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
                torch.cuda.current_stream().synchronize()
                ts.append(time.perf_counter() - t0)
                torch.cuda.synchronize()
            t = statistics.median(ts)
            per_block = (8 * iters * (64 * 2.0 * 16 * 16 * 32 if mode == 5 else 32 * 2.0 * 32 * 32 * 16))
            print(f"{name} {blocks:3d} workgroups{' + copy stream' if with_copy else '             '}: {t * 1e3:7.3f} ms  {blocks * per_block / t / 1e12:7.0f} TF/s total  "
                  f"{per_block / t / 1e12 * 256:7.0f} TF/s if all 256 CUs ran at this rate", flush=True)
