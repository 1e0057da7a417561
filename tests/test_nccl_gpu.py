"""GPU, >= 2 devices only: the multi-rank DiT paths under RCCL (`nccl` backend), one process per GPU -- `dist.all_to_all`, the
async K|V all-gather pieces and `device_id=` initialisation on real collectives.  Skipped on the one-GPU boxes of the build
rounds; on an N-GPU box every case with world <= N runs.  Both layouts SURVEY 8(e) asks for: 2 CFG rows x N/2 token chunks
(cfg_parallel None = the default) and pure N-way token chunks with the CFG pair batched (cfg_parallel False).

No multi-GPU box existed in any build round: these cases have NEVER executed.  Their first contact with real RCCL must not stop the
rest of a `pytest -x` run, so a failing case is reported as XFAIL with its reason (a passing one as a plain pass);
FLEXAM_TEST_NCCL_STRICT=1 turns failures into failures."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,cfg_parallel,mode", [
    (2, None, "allgather"), (2, False, "allgather"), (2, False, "ulysses"),
    (4, None, "allgather"), (4, False, "allgather"), (4, False, "allgather-p1-wait"), (4, None, "ulysses"), (4, False, "ulysses"),
    (8, None, "allgather"), (8, False, "allgather"), (8, None, "ulysses"), (4, False, "ulysses-o2"), (8, False, "ulysses-o2")])
def test_ranks_under_rccl_match_single_process(world, cfg_parallel, mode, monkeypatch):
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs >= {world} GPUs (RCCL refuses two ranks on one device)")
    from test_sp_gpu import rel_rms, ranks_agree, run_world, single_process
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sp = world // 2 if (cfg_parallel is None and world % 2 == 0 and (world == 2 or not mode.startswith("ulysses"))) else world
    wide = mode.startswith("ulysses") and sp > 2        # the all-to-all needs heads % ranks == 0: four 128-wide heads
    if wide and 4 % sp:
        pytest.skip("the four-head test model does not divide over this many sequence-parallel ranks")
    case = (cfg_parallel, mode.replace("-o2", "-ov2"))                   # -o2: attention per sample too (FLEXAM_SP_OVERLAP=2)
    # one spawned world per case here (on the one-GPU boxes test_sp_gpu.py batches the cases of a world): under real RCCL a hung case must
    # not take the others with it
    strict = os.environ.get("FLEXAM_TEST_NCCL_STRICT") == "1"
    try:
        out0, lat0 = ranks_agree(run_world(world, [case], wide, backend="nccl"), case)
        single, _ = single_process(wide)
        rel = rel_rms(out0, single)
        print(f"RCCL {world} ranks, cfg_parallel={cfg_parallel}, {mode}: rel-rms vs single process {rel:.2e}")
        ok = rel < 4e-3 and bool(torch.isfinite(lat0).all())
        why = f"rel-rms vs single process {rel:.2e} (bound 4e-3)"
    except Exception as e:                                   # noqa: BLE001  (a hung / crashed rank surfaces here through run_world's timeout)
        if strict:
            raise
        ok, why = False, f"{type(e).__name__}: {e}"
    if not ok and not strict:
        pytest.xfail(f"first execution under real RCCL failed -- {why}")
    assert ok, why
