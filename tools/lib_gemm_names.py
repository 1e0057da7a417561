"""Dev tool: run the vendor GEMM on the DiT shapes (for rocprofv3 --kernel-trace, to read the library kernel's tile config)."""
import torch
import torch.nn.functional as F
dev = torch.device("cuda:0"); BF = torch.bfloat16
M = 23296
for N, K in ((9216, 3072), (14336, 3072), (3072, 14336), (3072, 3072)):
    a = torch.randn(M, K, device=dev).to(BF); w = torch.randn(N, K, device=dev).to(BF)
    for _ in range(3):
        F.linear(a, w)
torch.cuda.synchronize()
