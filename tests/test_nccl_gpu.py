"""GPU, >= 2 devices only: the multi-rank DiT paths under RCCL (`nccl` backend), one process per GPU -- the first time
`dist.all_to_all` / the async K|V all-gather run on real collectives.  Skipped on the one-GPU boxes of the build rounds."""
import socket

import pytest
import torch
import torch.multiprocessing as mp

from oracle import cases as C
from oracle import dit as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["allgather", "ulysses"])
def test_two_ranks_under_rccl_match_single_process(mode, monkeypatch):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL refuses two ranks on one device)")
    from test_sp_gpu import _worker
    monkeypatch.setenv("FLEXAM_SP_MODE", mode)
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret, False, False, "nccl")) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    out0, lat0 = ret[0]
    torch.testing.assert_close(out0, ret[1][0], rtol=0, atol=0)
    torch.testing.assert_close(lat0, ret[1][1], rtol=0, atol=0)
    from flexam_amd import Wan2_2Transformer3DModel_FlexAM
    cfg = dict(O.DIT_TINY)
    kw = dict(cfg)
    kw.pop("eps")
    m = Wan2_2Transformer3DModel_FlexAM(**kw)
    m.load_state_dict(C.dit_weights(cfg, 7), strict=True)
    m = m.to("cuda:0")
    case = C.dit_case(cfg, 41, per_token_t=True)
    d = {k: ([u.cuda() for u in v] if isinstance(v, list) else (v.cuda() if torch.is_tensor(v) else v)) for k, v in case.items()}
    single = m(**d).float().cpu()
    rel = ((out0 - single).pow(2).mean().sqrt() / single.pow(2).mean().sqrt()).item()
    print(f"RCCL 2 ranks, {mode}: rel-rms vs single process {rel:.2e}")
    assert rel < 2e-3 and bool(torch.isfinite(lat0).all())
