"""Dev tool: the DiT's self-attention call a few times under FLEXAM_ATTN_WIDE (for rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H
dev = torch.device("cuda:0"); BF = torch.bfloat16
g = torch.Generator().manual_seed(0)
q, k, v = [(torch.randn(2, 11648, 24, 128, generator=g) * 0.5).to(BF).to(dev) for _ in range(3)]
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    H.attn_fwd(q, k, v, prescaled=True)
torch.cuda.synchronize()
