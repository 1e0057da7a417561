"""Dev tool: umT5-xxl-shaped text encoder (random weights) on two 512-token prompts: time + sanity."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import WanT5EncoderModel
torch.manual_seed(0)
with torch.device("cuda:0"):
    m = WanT5EncoderModel(vocab=256384, dim=4096, dim_attn=4096, dim_ffn=10240, num_heads=64, num_layers=24, num_buckets=32, shared_pos=False)
m = m.to(torch.bfloat16)
ids = torch.randint(1, 256384, (2, 512), device="cuda:0")
mask = torch.zeros(2, 512, dtype=torch.long, device="cuda:0")
mask[0, :77] = 1
mask[1, :300] = 1
for it in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = m(ids, mask)[0]
    torch.cuda.synchronize()
    print(f"umT5-xxl encode {tuple(ids.shape)} -> {tuple(out.shape)} {out.dtype}: {time.perf_counter() - t0:.3f} s, "
          f"finite={bool(torch.isfinite(out.float()).all())}, mem={torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
