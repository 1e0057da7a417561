"""GPU (one device is enough): RCCL's stream semantics against kernels this library launches through RAW stream handles.

The multi-rank engine enqueues its kernels with ctypes on `torch.cuda.current_stream().cuda_stream` and lets `torch.distributed`
run the K|V all-gathers asynchronously on RCCL's own stream (DiTEngine._allgather_attention).  What keeps that correct is c10d's
contract: the collective's stream waits for the work already enqueued on the current stream when it is issued, `Work.wait()`
makes the current stream wait for the collective, and nothing else orders the two.  Two ranks cannot share a device under RCCL,
so this runs the REAL backend with a ONE-rank group (`device_id=` initialisation, `all_gather_into_tensor(async_op=True)`,
`all_to_all_single`, grouped `all_to_all`: each degenerates to a device copy on RCCL's stream) in the engine's pattern on buffers large
enough for every step to take tens of microseconds:
    producer kernel (writes send) -> async gather (reads send, writes cat) | reader kernel (reads send concurrently)
    -> wait -> consumer kernel (reads cat) -> next iteration's producer overwrites send
with integer-valued fp32 data, so the accumulated result is exact and any missing ordering edge shows as a wrong sum."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(port, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        from flexam_amd import hip
        from flexam_amd.dist import SeqGather, all_to_all_blocks, all_to_all_chunks
        n = 16 << 20                                             # 64 MiB of fp32 per buffer
        g = torch.Generator().manual_seed(5)
        base = torch.randint(-8, 9, (n,), generator=g).float().to(dev)
        send, cat, side, acc, acc_side = (torch.zeros(n, device=dev) for _ in range(5))
        iters = 24
        for i in range(iters):
            hip.axpby(send, float(i + 1), base, 0.0)             # producer: send = (i + 1) * base   (raw-handle launch)
            w = dist.all_gather_into_tensor(cat, send, async_op=True)
            hip.axpby(side, 1.0, send, 0.0)                      # reads send while the collective reads it
            hip.axpby(acc_side, 1.0, side, 1.0)
            w.wait()
            hip.axpby(acc, 1.0, cat, 1.0)                        # consumer: acc += cat
        want = base * (iters * (iters + 1) // 2)
        ok_gather = bool(torch.equal(acc, want)) and bool(torch.equal(acc_side, want))
        # the engine's other collectives on the same backend: SeqGather (head-token gather), all_to_all_single / grouped all_to_all (ulysses)
        acc.zero_()
        x3 = send.view(1, 4096, -1)
        for i in range(8):
            hip.axpby(send, float(i + 1), base, 0.0)
            sg = SeqGather(x3)
            hip.axpby(side, 1.0, send, 0.0)
            full = sg.finish()
            hip.axpby(acc, 1.0, full.reshape(-1), 1.0)
        ok_seq = bool(torch.equal(acc, base * 36))
        acc.zero_()
        for i in range(8):
            hip.axpby(send, float(i + 1), base, 0.0)
            all_to_all_chunks(cat.view(1, -1), send.view(1, -1), None)
            hip.axpby(acc, 1.0, cat, 1.0)
            hip.axpby(send, float(-(i + 1)), base, 0.0)
            all_to_all_blocks([cat], [send], None)
            hip.axpby(acc, 2.0, cat, 1.0)                        # acc += 2 * (-(i + 1) * base)
        ok_a2a = bool(torch.equal(acc, base * (-36)))
        acc.zero_(); acc_side.zero_()
        for i in range(8):                                       # the pipelined all-to-all exchange: async grouped send / recv, waited for later
            hip.axpby(send, float(i + 1), base, 0.0)
            w = all_to_all_blocks([cat], [send], None, async_op=True)
            hip.axpby(side, 1.0, send, 0.0)
            hip.axpby(acc_side, 1.0, side, 1.0)
            w.wait()
            hip.axpby(acc, 1.0, cat, 1.0)
        ok_a2a = ok_a2a and bool(torch.equal(acc, base * 36)) and bool(torch.equal(acc_side, base * 36))
        ret["ok"] = (ok_gather, ok_seq, ok_a2a)
    finally:
        dist.destroy_process_group()


def test_async_collectives_are_ordered_against_raw_handle_launches():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    p = ctx.Process(target=_worker, args=(port, ret))
    p.start()
    p.join(600)
    assert p.exitcode == 0
    assert ret["ok"] == (True, True, True), ret["ok"]
