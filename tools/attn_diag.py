"""Dev tool: localise attention layout bugs with one-hot softmax rows and structured V."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexam_amd import hip as H
dev = torch.device("cuda:0"); BF = torch.bfloat16
Lq, Lk = 32, 64
q = torch.zeros(1, Lq, 1, 128); q[..., 0] = 8.0
bad_key, bad_d = [], []
for kstar in range(Lk):
    k = torch.zeros(1, Lk, 1, 128); k[..., 0] = -8.0; k[0, kstar, 0, 0] = 8.0
    vkey = torch.arange(Lk).float().view(1, Lk, 1, 1).expand(1, Lk, 1, 128).contiguous()
    o = H.attn_fwd(q.to(BF).to(dev), k.to(BF).to(dev), vkey.to(BF).to(dev)).float().cpu()
    if not torch.all(o == kstar):
        bad_key.append((kstar, sorted(set(o.flatten().tolist()))[:8], int((o != kstar).sum())))
vd = torch.arange(128).float().view(1, 1, 1, 128).expand(1, Lk, 1, 128).contiguous()
k = torch.zeros(1, Lk, 1, 128)
o = H.attn_fwd(q.to(BF).to(dev), k.to(BF).to(dev), vd.to(BF).to(dev)).float().cpu()[0, :, 0]
print("V=d test: rows equal?", bool((o == o[0]).all()), " first row:", o[0].tolist())
print("one-hot key test: bad cases", len(bad_key))
for b in bad_key[:16]:
    print(b)
