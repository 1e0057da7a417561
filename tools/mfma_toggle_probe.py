"""Dev tool: MFMA throughput under the power limit vs how often the operands change between consecutive instructions
(tools/probes/mfma_toggle_probe.hip), round-robin medians; each launch ~2 ms on all 256 CUs."""
import ctypes, os, sys, time, statistics
import torch
P = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "libmfma_toggle_probe.so"))
dev = torch.device("cuda:0")
src = (torch.randn(4096 * 8, device=dev) * 0.5).to(torch.bfloat16)
out = torch.zeros(512, device=dev)
iters, blocks = 2000, 256
flops = blocks * 8 * iters * 32 * 2.0 * 32 * 32 * 16
names = {0: "A and B change every MFMA", 1: "B shared by pairs", 2: "B constant", 3: "A and B constant", 4: "zero operands",
         5: "16x16x32, random operands", 6: "16x16x32, zero operands"}
flops_of = lambda m: flops if m < 5 else blocks * 8 * iters * 64 * 2.0 * 16 * 16 * 32
res = {m: [] for m in names}
run = lambda m: P.run_probe(m, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(out.data_ptr()), iters, blocks, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
for rnd in range(9):
    for m in (list(names) if rnd % 2 == 0 else list(names)[::-1]):
        run(m); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): run(m)
        torch.cuda.synchronize()
        res[m].append((time.perf_counter() - t0) / 5)
for m, n in names.items():
    t = statistics.median(res[m])
    print(f"mode {m} {n:28s} {t * 1e3:7.3f} ms  {flops_of(m) / t / 1e12:7.0f} TF/s ({flops_of(m) / t / 2.5e15:.1%} of 2.5 PF)")
