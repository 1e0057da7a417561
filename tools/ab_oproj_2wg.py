"""Dev tool: the two-workgroups-per-CU gate-residual instance (128 x 128 tiles, 4 waves; FLEXAM_GEMM_2WG=1) against the one-per-CU
instance at the o-projection shape: exact-integer check (ragged M, gate rows), then a same-process round-robin of the isolated
launch time over the L2 group heights."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H
dev = torch.device("cuda:0"); BF = torch.bfloat16
N = K = 3072


def exact(M):
    g = torch.Generator(device="cpu").manual_seed(M)
    a = torch.randint(-4, 5, (M, K), generator=g).to(BF).to(dev)
    w = torch.randint(-2, 3, (N, K), generator=g).to(BF).to(dev)
    b = torch.randint(-8, 9, (N,), generator=g).float().to(dev)
    x = torch.randint(-64, 65, (M, N), generator=g).float().to(dev)
    gate = torch.randint(-2, 3, (2, N), generator=g).float().to(dev)
    rows = (torch.arange(M) >= M // 2).to(torch.int32).to(dev)
    y = (a.float() @ w.float().t() + b).to(BF).float()
    want = x + y * gate[rows.long()]
    out = {}
    for two in ("0", "1"):
        os.environ["FLEXAM_GEMM_2WG"] = two
        xd = x.clone()
        H.gemm_gate_residual(a, w, b, xd, gate, rows)
        torch.cuda.synchronize()
        out[two] = xd
        print(f"M={M} 2WG={two}: max |diff| vs fp32 reference {float((xd - want).abs().max())}")
    assert torch.equal(out["0"], out["1"])


exact(23296)
exact(4096 + 77)

M = 23296
a = (torch.randn(M, K, device=dev) * 0.5).to(BF)
w = (torch.randn(N, K, device=dev) * 0.02).to(BF)
b = torch.randn(N, device=dev)
x = torch.randn(M, N, device=dev)
y = torch.empty(M, N, device=dev, dtype=BF)
gate = torch.randn(2, N, device=dev)
VARS = ("FLEXAM_GEMM_2WG", "FLEXAM_GEMM_GM", "FLEXAM_GEMM_2WG_DELAY")


def run(arms, fn, rounds=6):
    res = {arm: [] for arm in arms}
    for rnd in range(rounds):
        for arm in arms:
            for k in VARS: os.environ.pop(k, None)
            for kv in arm.split(","):
                k, _, v = kv.partition("="); os.environ["FLEXAM_GEMM_" + k] = v
            for _ in range(2): fn()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): fn()
            e.record(); torch.cuda.synchronize()
            res[arm].append(s.elapsed_time(e) / 10)
    for arm in arms:
        print(f"  {arm:28s} median {statistics.median(res[arm]) * 1e3:.1f} us  min {min(res[arm]) * 1e3:.1f}", flush=True)


print("gate-residual (fp32 X read-modify-write), in-tree library")
run(["2WG=0"] + [f"2WG=1,GM={g}" for g in (4, 8, 16)] + [f"2WG=1,2WG_DELAY={d}" for d in (2, 4, 7, 10)], lambda: H.gemm_gate_residual(a, w, b, x, gate, rows_per_batch=M // 2))
lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "libflexam_var_2wgall.so")
if os.path.exists(lib):
    H.load_library(lib)
    print("plain bf16 store (the K loops alone), diagnostic library -DFLEXAM_GEMM_2WG_ALL")
    run(["2WG=0"] + [f"2WG=1,GM={g}" for g in (4, 8, 16)], lambda: H.gemm(a, w, b, out=y))
