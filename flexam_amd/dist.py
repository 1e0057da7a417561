"""Sequence-parallel plumbing for the FlexAM DiT: the part of the reference that is *missing*
(`FlexAM/dist` was swallowed by the reference's .gitignore; only its call sites exist:
wan_transformer3d_FlexAM.py:22-24, 801-815, 919-920, 970-975, 1103-1104).

Design: one process per GPU, contiguous token chunks per rank (L/N tokens; attention is
full 3-D so chunks need not be frame aligned, RoPE rows are looked up by the *global* token index),
weights replicated, every op token-local except self-attention, which needs all keys/values.  Two
exchange schemes per block (DiTEngine.sp_mode):
  * "allgather" (default; the collective BASELINE.json's north_star names): all-gather of the post-norm, post-RoPE K|V
    ([L/N, 2C] bf16 per CFG row and rank) -- (N-1) x 2 C elements per token arrive.  Default form (r6): ONE gather per block, waited
    for, ONE attention call of the local queries over all keys; with VIDEOX_ATTENTION_TYPE=SAGE_ATTENTION the MXFP8 key / value records
    are what travels.  FLEXAM_SP_OVERLAP=1: the gather in head-group pieces started as soon as K|V exist, the rank attending to its
    LOCAL chunk underneath (flexam_attn_fwd_partial), then to the gathered chunks, partial softmaxes merged (flexam_attn_merge);
  * "ulysses" (FLEXAM_SP_MODE=ulysses; the heads must divide over the ranks): all-to-all of q|k|v so that every rank holds
    ALL tokens of H/N heads, attention, all-to-all of the output back to token chunks -- (N-1)/N of 4 C elements per
    token leave a rank, each peer link carries 1/N of it.  q|k|v are written in the send layout by the RMSNorm+RoPE kernel
    and the returning blocks are read in place by the output projection (per-K-block A offsets): no pack / unpack passes.
Plus one all-gather of the head output per step.

These helpers are backend-agnostic torch.distributed code (RCCL on GPUs; the CPU tests run them
under gloo with world_size 2)."""
from typing import Tuple

import torch
import torch.distributed as dist


_SUBGROUPS = {}      # (generation of the default process group, member tuple) -> process group
_GENERATION = [None, 0]   # [the default process group the cache was filled under, counter]


def _generation() -> int:
    """Bumps when the default process group is not the one the cache was filled under (destroy_process_group + a new
    init_process_group in a long-lived process): handles of a destroyed communicator must not be handed out again."""
    world = dist.group.WORLD if dist.is_initialized() else None
    if world is not _GENERATION[0]:
        _GENERATION[0] = world
        _GENERATION[1] += 1
        _SUBGROUPS.clear()
    return _GENERATION[1]


def subgroup(members) -> "dist.ProcessGroup":
    """The process group of `members` (global ranks), created on first use and cached: torch.distributed.new_group builds a new
    communicator on every call (collective over ALL ranks, never freed), so a caller that re-selects a layout -- bench.py's layout
    probe, repeated enable_multi_gpus_inference calls -- must not create them again.  Every rank must ask for the same member sets in
    the same order the first time (new_group's own rule).  The cache belongs to ONE default process group: after
    destroy_process_group() and a re-init it starts empty (clear_subgroups() does the same by hand)."""
    key = (_generation(), tuple(int(r) for r in members))
    g = _SUBGROUPS.get(key)
    if g is None:
        g = _SUBGROUPS[key] = dist.new_group(list(key[1]))
    return g


def clear_subgroups() -> None:
    """Forgets the cached sub-communicators (they are not destroyed here: destroy_process_group() does that)."""
    _SUBGROUPS.clear()


def live_subgroups() -> int:
    """Number of sub-communicators this process has created through subgroup() under the current default process group
    (the world group is not counted)."""
    _generation()
    return len(_SUBGROUPS)


class LoopbackGroup:
    """ONE process standing in for rank `rank` of a `size`-rank group (bench.py --emulate-rank, tests): every collective issued on
    it is replaced by device copies of the SAME sizes on a side stream -- the place RCCL's own stream has in a real run -- and returns
    a Work-like object, so a rank's launch sequence, buffer shapes, partial / merge calls and stream dependencies are the real ones
    while nothing leaves the GPU.  The "received" chunks are copies of the local one: timing and plumbing only, never results."""

    def __init__(self, size: int, rank: int = 0, copies: bool = True, link_gbps: float = None, latency_us: float = 10.0):
        if not 0 <= rank < size:
            raise ValueError(f"LoopbackGroup: rank {rank} outside 0..{size - 1}")
        self.size, self.rank = int(size), int(rank)
        # copies = False: a collective keeps its stream plumbing (side stream, event, Work) but moves nothing -- the rank's step with NO
        # stand-in for the exchange at all (its "received" buffers keep whatever they held: timing only)
        self.copies = bool(copies)
        # link_gbps: an ASSUMED rate of one direction of one xGMI link (GB/s).  Every collective then also holds its side stream for
        # bytes-per-link / rate + latency_us (flexam_delay_us): the mesh is point-to-point, every peer's chunk / block travels on its own
        # link, all links in parallel -- so the time of an all-gather is ONE chunk's bytes over one link, of an all-to-all ONE block's.
        self.link_gbps = None if not link_gbps else float(link_gbps)
        self.latency_us = float(latency_us)
        self._filled = set()                           # (copies = False) buffers that received their one-off fill
        self._stream = None

    _side_streams = {}                                 # one side stream per device for ALL loopback groups of the process

    def side_stream(self, device):
        """The stream the stand-ins run on -- the place RCCL's own stream has in a real run.  ONE per device and process (ROCm maps
        streams onto a few hardware queues; a fresh stream per group landed on the compute stream's queue every few groups and the
        "collective" then serialised with compute: two timing regimes 10 ms per step apart, profiles/r6u_*), high priority like RCCL's."""
        key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
        st = LoopbackGroup._side_streams.get(key)
        if st is None:
            st = LoopbackGroup._side_streams[key] = torch.cuda.Stream(device=device, priority=-1)
        self._stream = st
        return st

    def run(self, fn, ref: torch.Tensor, async_op: bool, link_bytes: int = 0):
        """`fn()` (the copies) after everything enqueued so far on the caller's stream; returns a Work (async) or None (joined).
        link_bytes: what ONE link carries for this collective (its transfer time under `link_gbps`)."""
        if ref.device.type != "cuda":                      # CPU tensors (unit tests of the plumbing): synchronous
            if self.copies:
                fn()
            return _LoopbackWork(None) if async_op else None
        cur = torch.cuda.current_stream(ref.device)
        side = self.side_stream(ref.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            # copies = False: every receive buffer is filled ONCE (the first collective that targets it), so that what the kernels read
            # is data of the workload's kind -- uninitialised memory would be timed on the attention kernel's rescale path
            key = (ref.data_ptr(), ref.numel())
            if self.copies or key not in self._filled:
                fn()
                self._filled.add(key)
            if self.link_gbps and self.size > 1:
                from . import hip
                hip.delay_us(link_bytes / (self.link_gbps * 1e3) + self.latency_us)
            done = torch.cuda.Event()
            done.record(side)
        work = _LoopbackWork(done)
        if async_op:
            return work
        work.wait()
        return None


class _LoopbackWork:
    def __init__(self, event):
        self.event = event

    def wait(self):
        if self.event is not None:
            torch.cuda.current_stream().wait_event(self.event)
        return True


def group_size(group=None) -> int:
    return group.size if isinstance(group, LoopbackGroup) else dist.get_world_size(group)


def group_rank(group=None) -> int:
    return group.rank if isinstance(group, LoopbackGroup) else dist.get_rank(group)


def group_backend(group=None) -> str:
    """"nccl" (RCCL), "gloo", ... or "loopback"."""
    return "loopback" if isinstance(group, LoopbackGroup) else dist.get_backend(group)


def all_gather_into_tensor(out: torch.Tensor, inp: torch.Tensor, group=None, async_op: bool = False):
    """dist.all_gather_into_tensor (rank-major concatenation on dim 0) on a real group; on a LoopbackGroup every rank's slot of `out`
    receives a copy of `inp` (one launch, the bytes a real gather writes)."""
    if isinstance(group, LoopbackGroup):
        src = inp.contiguous()
        return group.run(lambda: out.view(group.size, *src.shape).copy_(src.unsqueeze(0).expand(group.size, *src.shape)), out, async_op,
                         link_bytes=src.numel() * src.element_size())        # every peer's chunk on its own link, all in parallel
    return dist.all_gather_into_tensor(out, inp, group=group, async_op=async_op)


_SP_CONTEXT = [None]


class sequence_parallel_context:
    """Set by the engine around the block-module calls of a sequence-parallel forward (the reference re-binds every
    `block.self_attn.forward` to a USP forward at this point, wan_transformer3d_FlexAM.py:807-815): blocks receive THIS RANK'S token
    chunk [B, L/N, C] with the global `seq_lens` / `grid_sizes` (wan_transformer3d_FlexAM.py:970-975), and a self-attention forward --
    the native one, or a caller's re-bound one -- finds here what it needs for the exchange: group, rank, size, the global token
    offset of the chunk and the RoPE tables of the WHOLE sequence.  `current_sp_context()` is None outside such a forward."""

    def __init__(self, group, rank: int, size: int, token_offset: int, seq_len: int, rope_cos=None, rope_sin=None):
        self.info = dict(group=group, rank=rank, size=size, token_offset=token_offset, seq_len=seq_len, rope_cos=rope_cos, rope_sin=rope_sin)

    def __enter__(self):
        self._prev = _SP_CONTEXT[0]
        _SP_CONTEXT[0] = self.info
        return self.info

    def __exit__(self, *exc):
        _SP_CONTEXT[0] = self._prev
        return False


def current_sp_context():
    return _SP_CONTEXT[0]


def padded_len(seq_len: int, world: int) -> int:
    """The reference pads the token sequence to a multiple of the sequence-parallel size with zero tokens that attention masks as
    keys (wan_transformer3d_FlexAM.py:919-925: `seq_len = ceil(seq_len / sp) * sp`, k_lens = the real lengths)."""
    return -(-int(seq_len) // int(world)) * int(world)


def chunk_bounds(seq_len: int, rank: int, world: int) -> Tuple[int, int]:
    """[start, end) of this rank's token chunk of the PADDED sequence (padded_len): every rank holds the same number of rows; the
    rows at or beyond `seq_len` (only ever at the end of the last chunks) are the reference's zero pad tokens."""
    lc = padded_len(seq_len, world) // world
    return rank * lc, (rank + 1) * lc


def real_tokens(seq_len: int, rank: int, world: int) -> int:
    """How many of this rank's chunk rows are real tokens (the rest are pads)."""
    s, e = chunk_bounds(seq_len, rank, world)
    return max(0, min(e, seq_len) - s)


def all_gather_seq(local: torch.Tensor, group=None, out: torch.Tensor = None, scratch: torch.Tensor = None) -> torch.Tensor:
    """local [B, Lc, X] on every rank -> [B, N*Lc, X] with rank r's rows at [r*Lc, (r+1)*Lc)."""
    world = group_size(group)
    b, lc, x = local.shape
    if scratch is None:
        scratch = torch.empty(world * b, lc, x, device=local.device, dtype=local.dtype)
    all_gather_into_tensor(scratch.view(world * b, lc, x), local.contiguous(), group=group)   # rank-major concat on dim 0
    if out is None:
        out = torch.empty(b, world * lc, x, device=local.device, dtype=local.dtype)
    out.view(b, world, lc, x).copy_(scratch.view(world, b, lc, x).transpose(0, 1))
    return out


class SeqGather:
    """Asynchronous form of all_gather_seq: start() launches the collective (overlaps with whatever the caller
    enqueues next on its compute stream), finish() waits for it and returns [B, N*Lc, X].  With B == 1 the
    rank-major concatenation already IS the token order, so no re-layout pass is needed."""

    def __init__(self, local: torch.Tensor, group=None, out: torch.Tensor = None, scratch: torch.Tensor = None):
        self.world = group_size(group)
        self.shape = tuple(local.shape)
        b, lc, x = self.shape
        self.out = out
        if b == 1:
            self.scratch = out.view(self.world * b, lc, x) if out is not None else torch.empty(self.world, lc, x, device=local.device, dtype=local.dtype)
        else:
            self.scratch = scratch if scratch is not None else torch.empty(self.world * b, lc, x, device=local.device, dtype=local.dtype)
        self.work = all_gather_into_tensor(self.scratch.view(self.world * b, lc, x), local.contiguous(), group=group, async_op=True)

    def finish(self) -> torch.Tensor:
        self.work.wait()
        b, lc, x = self.shape
        if b == 1:
            return self.scratch.view(1, self.world * lc, x)
        out = self.out if self.out is not None else torch.empty(b, self.world * lc, x, device=self.scratch.device, dtype=self.scratch.dtype)
        out.view(b, self.world, lc, x).copy_(self.scratch.view(self.world, b, lc, x).transpose(0, 1))
        return out


def all_to_all_chunks(out: torch.Tensor, inp: torch.Tensor, group=None, async_op: bool = False):
    """inp [N, ...] (chunk j goes to rank j) -> out [N, ...] (chunk i came from rank i), both contiguous.  RCCL: one
    all_to_all_single (pairwise sends over the xGMI mesh).  Other backends (the gloo test runs) may lack all-to-all on device
    tensors: the same result is assembled from an all-gather of every rank's send buffer."""
    if isinstance(group, LoopbackGroup):
        return group.run(lambda: out.copy_(inp), out, async_op, link_bytes=inp[0].numel() * inp.element_size())
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "nccl":
        return dist.all_to_all_single(out, inp, group=group, async_op=async_op)
    rank = dist.get_rank(group)
    everything = torch.empty((world,) + tuple(inp.shape), device=inp.device, dtype=inp.dtype)
    dist.all_gather_into_tensor(everything.view(world * inp.shape[0], *inp.shape[1:]), inp.contiguous(), group=group)
    out.copy_(everything[:, rank])
    return None


def all_to_all_blocks(outs, ins, group=None, async_op: bool = False, packed=None):
    """ins[j] (contiguous) goes to rank j, outs[i] (contiguous, anywhere in memory) receives rank i's block: the list form lets
    every received block land where the consumer wants it (no unpack pass).  RCCL: one grouped send/recv (dist.all_to_all);
    async_op = True returns its Work (wait() makes the caller's stream wait for the exchange), so the next sample's exchange
    travels under this sample's attention.  Other backends (the gloo runs of the tests): the same result from an all-gather of
    every rank's stacked blocks, finished before returning (None: nothing to wait for)."""
    if isinstance(group, LoopbackGroup):                                  # block i "arrives" as a copy of the block sent to rank i
        def copies():
            if packed is not None:                                        # packed = (out, inp) with out[i] = outs[i], inp[i] = ins[i]: the same bytes, one copy
                packed[0].copy_(packed[1])
                return
            for o, t in zip(outs, ins):
                o.copy_(t)
        return group.run(copies, outs[0], async_op, link_bytes=ins[0].numel() * ins[0].element_size())    # one block per peer link (and direction)
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "nccl":
        return dist.all_to_all(list(outs), list(ins), group=group, async_op=async_op)
    rank = dist.get_rank(group)
    mine = torch.stack([t.contiguous() for t in ins])                      # [dst, ...]
    everything = torch.empty((world,) + tuple(mine.shape), device=mine.device, dtype=mine.dtype)
    dist.all_gather_into_tensor(everything.view(world * mine.shape[0], *mine.shape[1:]), mine, group=group)
    for i, o in enumerate(outs):
        o.copy_(everything[i, rank])
    return None


def shard_rows(full: torch.Tensor, batch: int, seq_len: int, rank: int, world: int, chunk: int = None) -> torch.Tensor:
    """Per-token vector [B*L] (e.g. the AdaLN row index) -> this rank's [B*Lc] slice of the padded sequence; pad tokens repeat the
    last real token's entry (the reference pads `t` with its last element, wan_transformer3d_FlexAM.py:930-934).  `chunk`: rows per
    rank when the engine pads to more than a multiple of the ranks (MXFP8 key records: whole 64-key tiles per rank)."""
    s, e = (rank * chunk, (rank + 1) * chunk) if chunk is not None else chunk_bounds(seq_len, rank, world)
    v = full.view(batch, seq_len)
    if e > seq_len:
        v = torch.cat([v, v[:, -1:].expand(batch, e - seq_len)], dim=1)
    return v[:, s:e].contiguous().view(-1)


def shard_streams(jobs, encode, group=None):
    """Independent objects shard across ranks with no data-path collective: the conditioning streams
    of one clip (PIPE.py:655-822: control, depth, cos levels, masked video, reference image) are VAE-encoded
    round-robin -- stream j on rank j % N -- and every result is then broadcast from its owner (a few MB of
    latents per stream).  `jobs`: list of (key, tensor_or_None); `encode(tensor) -> latent`.  Returns {key: latent};
    None inputs stay None.  Every rank must call this with the same job list."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    live = [(k, v) for k, v in jobs if v is not None]
    out = {k: None for k, _ in jobs}
    mine = {k: encode(v) for j, (k, v) in enumerate(live) if j % world == rank}
    if world == 1:
        out.update(mine)
        return out
    ref = next(iter(mine.values())) if mine else None
    shapes = [None] * world
    dist.all_gather_object(shapes, {k: (tuple(t.shape), str(t.dtype).replace("torch.", "")) for k, t in mine.items()}, group=group)
    device = ref.device if ref is not None else torch.device("cuda", torch.cuda.current_device())
    for j, (k, _) in enumerate(live):
        owner = j % world
        shape, dt = shapes[owner][k]
        t = mine[k].contiguous() if owner == rank else torch.empty(shape, device=device, dtype=getattr(torch, dt))
        dist.broadcast(t, src=dist.get_global_rank(group, owner) if group is not None else owner, group=group)
        out[k] = t
    return out


def get_sequence_parallel_world_size(group=None) -> int:
    """The reference's `get_sequence_parallel_world_size()` (missing FlexAM/dist; call site wan_transformer3d_FlexAM.py:802): inside a
    sequence-parallel forward the size of the group the engine runs on (a CFG half under the 2 x N/2 layout)."""
    ctx = current_sp_context()
    if ctx is not None and group is None:
        return ctx["size"]
    return group_size(group) if (isinstance(group, LoopbackGroup) or dist.is_initialized()) else 1


def get_sequence_parallel_rank(group=None) -> int:
    ctx = current_sp_context()
    if ctx is not None and group is None:
        return ctx["rank"]
    return group_rank(group) if (isinstance(group, LoopbackGroup) or dist.is_initialized()) else 0
