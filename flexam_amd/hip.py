"""ctypes binding of libflexam_hip.so (C ABI in include/flexam_hip.h) + thin tensor wrappers.

PyTorch is plumbing here: it owns device memory and the stream; every function below passes raw
device pointers, sizes and strides to the HIP library and raises RuntimeError on a non-zero
return code.  There is NO fallback: if the shared library is missing, or a tensor is not on a
GPU, the call fails loudly (a silent CPU/eager path would void every parity claim).
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libflexam_hip.so")

EPI_NONE, EPI_GELU_TANH = 0, 1

_P, _I, _L, _F = c_void_p, c_int, c_int64, c_float
_SIGNATURES = {
    "flexam_version": ([], c_int),
    "flexam_arch": ([], c_char_p),
    "flexam_last_error": ([], c_char_p),
    "flexam_device_check": ([], c_int),
    "flexam_device_cus": ([], c_int),
    "flexam_set_cu_budget": ([_I], c_int),
    "flexam_gemm_bf16": ([_P, _L, _P, _L, _P, _P, _L, _L, _L, _L, _I, _I, _P, _P, _L, _P], c_int),
    "flexam_gemm_bf16_gate_residual": ([_P, _L, _P, _L, _P, _P, _L, _P, _L, _P, _L, _L, _L, _L, _P, _P, _L, _P], c_int),
    "flexam_quantize_rows_fp8": ([_P, _L, _P, _L, _P, _L, _I, _P], c_int),
    "flexam_gemm_fp8": ([_P, _L, _P, _P, _L, _P, _P, _P, _L, _L, _L, _L, _I, _P], c_int),
    "flexam_gemm_fp8_gate_residual": ([_P, _L, _P, _P, _L, _P, _P, _P, _L, _P, _L, _P, _L, _L, _L, _L, _P], c_int),
    "flexam_gemm_fp8_gelu_q": ([_P, _L, _P, _P, _L, _P, _P, _P, _P, _L, _L, _L, _L, _P], c_int),
    "flexam_attn_fwd": ([_P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _L, _L, _I, _I, _I, _I, _I, _F, _P], c_int),
    "flexam_attn_fwd_lastkey": ([_P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _L, _L, _I, _I, _I, _I, _I, _F, _F, _P], c_int),
    "flexam_attn_fwd_splitkv": ([_P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _L, _L, _I, _I, _I, _I, _I, _F, _I, _I, _P, _P, _P], c_int),
    "flexam_attn_fwd_partial": ([_P, _L, _L, _P, _L, _L, _P, _L, _L, _I, _I, _I, _I, _I, _F, _I, _I, _P, _P, _P], c_int),
    "flexam_attn_merge": ([_P, _L, _L, _I, _I, _I, _I, _F, _I, _P, _P, _P], c_int),
    "flexam_attn_fp8_pack": ([_P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _P, _P, _I, _I, _I, _I, _P], c_int),
    "flexam_rmsnorm_rope_mx": ([_P, _L, _P, _P, _L, _P, _P, _P, _P, _L, _I, _F, _P, _P, _L, _L, _I, _I, _P], c_int),
    "flexam_attn_fwd_fp8": ([_P, _P, _P, _P, _L, _L, _I, _I, _I, _I, _I, _I, _P, _P, _P], c_int),
    "flexam_attn_fwd_fp8_chunked": ([_P, _P, _P, _P, _L, _L, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P], c_int),
    "flexam_ln_modulate": ([_P, _L, _L, _I, _F, _P, _P, _L, _P, _L, _P, _P, _P, _L, _P], c_int),
    "flexam_ln_modulate_fp8": ([_P, _L, _L, _I, _F, _P, _P, _L, _P, _L, _P, _P, _P, _L, _P, _P, _F, _F, _P], c_int),
    "flexam_gate_residual": ([_P, _L, _P, _L, _P, _L, _P, _L, _L, _I, _P], c_int),
    "flexam_rmsnorm_rope": ([_P, _L, _P, _L, _P, _P, _L, _P, _L, _P, _L, _I, _F, _P, _P, _L, _L, _I, _P], c_int),
    "flexam_rmsnorm_rope_scatter": ([_P, _L, _P, _P, _L, _P, _P, _L, _P, _P, _P, _L, _L, _I, _L, _L, _I, _F, _P, _P, _L, _L, _I, _P], c_int),
    "flexam_mod_table": ([_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P], c_int),
    "flexam_small_linear_f32": ([_P, _L, _P, _I, _L, _P, _P, _L, _I, _I, _I, _I, _P], c_int),
    "flexam_sinusoid_embed": ([_P, _P, _I, _I, _P], c_int),
    "flexam_patchify": ([_P, _I, _I, _I, _I, _I, _P, _L, _I, _L, _P], c_int),
    "flexam_unpatchify": ([_P, _L, _L, _I, _I, _I, _I, _P, _I, _P], c_int),
    "flexam_cfg_euler_blend": ([_P, _P, _L, _L, _F, _F, _P, _P, _P, _I, _I, _I, _I, _P], c_int),
    "flexam_axpby_f32": ([_P, _F, _P, _F, _L, _P], c_int),
    "flexam_checksum": ([_P, _L, _P, _P], c_int),
    "flexam_cfg_velocity": ([_P, _P, _L, _L, _F, _P, _I, _I, _I, _I, _P], c_int),
    "flexam_lincomb_f32": ([_P, _L, _I, _P, _P, _P], c_int),
    "flexam_mask_blend_f32": ([_P, _P, _P, _I, _L, _P], c_int),
    "flexam_pack_cl": ([_P, _I, _I, _I, _I, _I, _P, _I, _I, _P], c_int),
    "flexam_unpack_cl": ([_P, _I, _L, _I, _I, _I, _I, _P, _P], c_int),
    "flexam_groupnorm_silu_cl": ([_P, _L, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P, _I, _P, _I, _P], c_int),
    "flexam_vae_prep_cl": ([_P, _I, _L, _I, _I, _I, _I, _P, _I, _P, _I, _I, _I, _P], c_int),
    "flexam_upsample2x_cl": ([_P, _I, _L, _I, _I, _I, _I, _I, _P, _I, _P], c_int),
    "flexam_dupup_add_cl": ([_P, _L, _I, _I, _I, _I, _P, _L, _I, _I, _I, _P], c_int),
    "flexam_deinterleave_cl": ([_P, _I, _L, _I, _I, _I, _I, _P, _I, _P], c_int),
    "flexam_tapsum_cl": ([_P, _L, _I, _I, _I, _I, _I, _P, _P, _L, _P], c_int),
    "flexam_phase_dupup_cl": ([_P, _L, _L, _P, _L, _I, _I, _I, _I, _P, _L, _I, _I, _I, _P], c_int),
    "flexam_softmax_rows": ([_P, _L, _L, _I, _F, _P, _L, _I, _P], c_int),
    "flexam_scatter_add_cl": ([_P, _L, _P, _L, _I, _I, _I, _I, _P], c_int),
    "flexam_vae_unpatchify_clamp": ([_P, _L, _I, _I, _I, _P, _I, _I, _F, _F, _P], c_int),
    "flexam_pack_affine_cl": ([_P, _I, _I, _I, _I, _P, _P, _P, _I, _P], c_int),
    "flexam_raster_keys": ([_P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P], c_int),
    "flexam_raster_resolve": ([_P, _P, _L, _I, _I, _I, _I, _P, _P, _P], c_int),
    "flexam_t5_norm": ([_P, _L, _L, _I, _F, _P, _P, _L, _I, _P], c_int),
    "flexam_softmax_bias_rows": ([_P, _L, _L, _I, _F, _P, _L, _P, _P, _L, _I, _P], c_int),
    "flexam_mul_bf16": ([_P, _P, _P, _L, _P], c_int),
    "flexam_vae_patchify_cl": ([_P, _I, _I, _I, _I, _I, _P, _I, _I, _P], c_int),
    "flexam_space_to_depth_cl": ([_P, _I, _L, _I, _I, _I, _I, _P, _I, _I, _P], c_int),
    "flexam_avgdown_add_cl": ([_P, _L, _I, _I, _I, _I, _P, _L, _I, _I, _I, _I, _P], c_int),
    "flexam_delay_us": ([_F, _P], c_int),
    "flexam_fn_id": ([c_char_p], c_int),
    "flexam_fn_count": ([], c_int),
    "flexam_fn_name": ([_I], c_char_p),
    "flexam_replay": ([_P, _L, _P, _P], c_int),
}

_lib = None


def load_library(path: str = None) -> ctypes.CDLL:
    """Loads libflexam_hip.so and declares every entry point.  No compute call is made, so this
    also works on a host without a GPU (used by the CPU test that checks the exported symbols)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: build it with `python -m flexam_amd.build` "
                           "(flexam_amd has no CPU or eager fallback)")
    lib = ctypes.CDLL(path)
    for name, (argtypes, restype) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.argtypes = argtypes
        fn.restype = restype
    _lib = lib
    return lib


def lib():
    """The library -- or, while a command list is being recorded (`record()`), the recorder in front of it."""
    if _rec is not None:
        return _rec
    return _lib if _lib is not None else load_library()


# ----------------------------------------------------------------------------- command lists (csrc/replay.hip)
REPLAY_MAX_ARGS = 26


class _Arg(ctypes.Union):
    _fields_ = [("i", c_int64), ("f", ctypes.c_double), ("p", c_void_p)]


class _Cmd(ctypes.Structure):
    _fields_ = [("fn", ctypes.c_int32), ("nargs", ctypes.c_int32), ("a", _Arg * REPLAY_MAX_ARGS)]


# entry points that read HOST arrays during the call (flexam_hip.h): a recorded pointer to a temporary host array would dangle
_NOT_RECORDABLE = ("flexam_lincomb_f32",)
_rec = None              # the active _Recorder (one at a time, per process: recording happens on the thread that drives the engine)
_FN_IDS = {}


def _fn_id(name: str) -> int:
    if name not in _FN_IDS:
        real = _lib if _lib is not None else load_library()
        _FN_IDS[name] = int(real.flexam_fn_id(name.encode()))
    return _FN_IDS[name]


class Plan:
    """A recorded stretch of the engine's work: C segments (arrays of flexam_cmd, each re-issued by ONE flexam_replay call) interleaved
    with host operations (Python callables: collectives, waits, torch copies -- whatever the engine wrapped in `host_op`).  `keep` holds
    every tensor whose pointer sits in a command, so no recorded address can be handed to anybody else while the plan lives."""

    def __init__(self):
        self.items, self.keep, self.launches = [], [], 0

    def run(self):
        real = _lib if _lib is not None else load_library()
        st = _stream()
        failed = c_int64(-1)
        for kind, a, b in self.items:
            if kind == "c":
                rc = real.flexam_replay(a, b, ctypes.byref(failed), st)
                if rc != 0:
                    name = real.flexam_fn_name(a[failed.value].fn) if failed.value >= 0 else b"?"
                    raise RuntimeError(f"flexam_replay: command {failed.value} ({(name or b'?').decode()}) failed with code {rc}: "
                                       f"{real.flexam_last_error().decode()}")
            else:
                a()


class _Recorder:
    """Stands in for the library while a plan is recorded: every stream-ordered entry point is CALLED (the recording step is a real
    step) and appended to the open C segment as (function id, argument words); anything else (queries, flexam_last_error) passes
    through.  The stream argument is not recorded: a replay runs on the stream it is issued on."""

    def __init__(self, plan: Plan):
        self.plan, self.seg = plan, []
        self._wrapped = {}

    def __getattr__(self, name):
        real = _lib if _lib is not None else load_library()
        fn = getattr(real, name)
        fid = _fn_id(name) if name.startswith("flexam_") and name not in ("flexam_fn_id", "flexam_fn_name", "flexam_fn_count", "flexam_replay") else -1
        if fid < 0:
            return fn
        if name in _NOT_RECORDABLE:
            raise RuntimeError(f"{name} takes host arrays and cannot be part of a recorded launch plan")
        w = self._wrapped.get(name)
        if w is None:
            kinds = _SIGNATURES[name][0][:-1]

            def w(*args, _fn=fn, _fid=fid, _kinds=kinds, _name=name):
                rc = _fn(*args)
                if rc == 0:
                    if len(args) != len(_kinds) + 1:
                        raise RuntimeError(f"{_name}: {len(args)} arguments recorded, the signature has {len(_kinds) + 1}")
                    self.seg.append((_fid, _kinds, args[:-1]))
                return rc
            self._wrapped[name] = w
        return w

    def flush(self):
        if not self.seg:
            return
        arr = (_Cmd * len(self.seg))()
        for c, (fid, kinds, args) in zip(arr, self.seg):
            c.fn, c.nargs = fid, len(args)
            for j, (k, v) in enumerate(zip(kinds, args)):
                if k is _P:
                    c.a[j].p = v
                elif k is _F:
                    c.a[j].f = float(v)
                else:
                    c.a[j].i = int(v)
        self.plan.items.append(("c", arr, len(self.seg)))
        self.plan.launches += len(self.seg)
        self.seg = []


class record:
    """`with hip.record() as plan:` -- everything the block of code launches through this module is executed AND recorded into `plan`
    (Plan.run() re-issues it).  Valid for code whose launches depend only on things that do not change between runs: buffer addresses,
    shapes, scalars (the caller's business: DiTEngine keys its plans on all of them).  Host-side work in between goes through
    `host_op`."""

    def __init__(self):
        self.plan = Plan()

    def __enter__(self):
        global _rec
        if _rec is not None:
            raise RuntimeError("flexam_amd.hip.record: already recording")
        _rec = _Recorder(self.plan)
        return self.plan

    def __exit__(self, et, ev, tb):
        global _rec
        rec, _rec = _rec, None
        if et is None:
            rec.flush()
        return False


def host_op(fn):
    """Runs `fn()` now; while a plan is recorded it also closes the open C segment and becomes a host step of the plan (run again, in
    this place, by every Plan.run()).  For what is not a call into the library: collectives, Work.wait(), torch copies.  A host step is
    opaque: library calls it makes itself (an emulated collective's delay kernel on its side stream) are part of the step, not of a C
    segment -- recording is suspended while it runs."""
    global _rec
    if _rec is None:
        return fn()
    rec = _rec
    rec.flush()
    rec.plan.items.append(("py", fn, None))
    _rec = None
    try:
        return fn()
    finally:
        _rec = rec


def delay_us(us: float):
    """Emulation aid: the current stream waits `us` microseconds (flexam_delay_us)."""
    _check(lib().flexam_delay_us(float(us), _stream()), "flexam_delay_us")


def recording() -> bool:
    return _rec is not None


def _check(rc: int, what: str):
    if rc != 0:
        msg = lib().flexam_last_error().decode()
        raise RuntimeError(f"{what} failed with code {rc}: {msg}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t, dtype=None):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("flexam_amd.hip: tensor is not on a GPU (there is no CPU path)")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"flexam_amd.hip: expected {dtype}, got {t.dtype}")
    if _rec is not None:
        _rec.plan.keep.append(t)                   # a recorded address stays this tensor's for as long as the plan lives
    return t.data_ptr()


def _raw(t):
    """data_ptr() of scratch the wrappers pass without a dtype check (workspaces, byte images); kept alive by a plan being recorded."""
    if _rec is not None:
        _rec.plan.keep.append(t)
    return t.data_ptr()


def _rows(t):
    """2-D view requirements: last dim contiguous; returns (rows, cols, row_stride)."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise RuntimeError(f"flexam_amd.hip: expected a 2-D row-major view, got shape {tuple(t.shape)} stride {t.stride()}")
    return t.shape[0], t.shape[1], t.stride(0)


BF16, F32, I32, I64 = torch.bfloat16, torch.float32, torch.int32, torch.int64


def device_check():
    _check(lib().flexam_device_check(), "flexam_device_check")


def set_cu_budget(cus: int = 0):
    """Plan the library's persistent grids for `cus` CUs of the current device (0 = all); see flexam_set_cu_budget."""
    _check(lib().flexam_set_cu_budget(int(cus)), "flexam_set_cu_budget")


def num_cus() -> int:
    """Compute units the library plans its grids for on the current device (flexam_device_cus: 256 on MI355X)."""
    return int(lib().flexam_device_cus())


# ----------------------------------------------------------------------------- GEMM
_GEMM_WS = {}
GEMM_WS_BYTES = 256 * 256 * 256 * 4                 # FLEXAM_GEMM_WS_BYTES
WS_CACHE_SLOTS = 8                                  # (device, stream) pairs that keep their scratch; older ones are dropped


def _ws_slot(cache: dict, key, make):
    """Scratch per (device, stream) -- launches that can run concurrently never share it -- in a small LRU: a program that
    creates streams per call does not grow memory without bound.  Dropping an entry waits for its device first (rare: the
    9th distinct stream), so no kernel still writes the slabs when the allocator hands them out again."""
    hit = cache.pop(key, None)
    if hit is None:
        while len(cache) >= WS_CACHE_SLOTS:
            old = next(iter(cache))
            torch.cuda.synchronize(old[0])
            del cache[old]
        hit = make()
    cache[key] = hit                                # re-inserted last = most recently used
    return hit


def _gemm_workspace(device, stream: int):
    """Tail split-K scratch (64 MiB of partial-sum slabs) handed to every GEMM call, per (device, stream)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), stream)
    return _ws_slot(_GEMM_WS, key, lambda: torch.empty(GEMM_WS_BYTES, device=device, dtype=torch.uint8))


def gemm(a, w, bias=None, out=None, epilogue=EPI_NONE, out_dtype=BF16, a_koff=None, m=None, k=None):
    """out[M,N] = epi(a[M,K] @ w[N,K]^T + bias).  a, w bf16 2-D views (row stride free).
    With a_koff (int64 [K/64]) `a` is only a base view: rows are `m`, K = `k` (implicit conv)."""
    am, ak, lda = _rows(a)
    wn, wk, ldw = _rows(w)
    M = am if m is None else m
    K = wk if k is None else k
    if a_koff is None and ak != K:
        raise RuntimeError(f"gemm: K mismatch a {ak} vs w {K}")
    if out is None:
        out = torch.empty(M, wn, device=a.device, dtype=out_dtype)
    om, on, ldc = _rows(out)
    if (om, on) != (M, wn):
        raise RuntimeError(f"gemm: out shape {tuple(out.shape)} != ({M}, {wn})")
    st = _stream()
    ws = _gemm_workspace(a.device, st)
    _check(lib().flexam_gemm_bf16(_ptr(a, BF16), lda, _ptr(w, BF16), ldw, _ptr(bias, F32), _ptr(out), ldc, M, wn, K, epilogue,
                                  1 if out.dtype == F32 else 0, _ptr(a_koff, I64), _raw(ws), ws.numel(), st), "flexam_gemm_bf16")
    return out


def gemm_gate_residual(a, w, bias, x, gate=None, gate_row=None, rows_per_batch=0, a_koff=None):
    """x[M,N] (fp32, in place) += bf16(a @ w^T + bias) * gate[row].  With a_koff `a` is a base view
    (implicit conv) and M, K come from x and w."""
    am, ak, lda = _rows(a)
    N, K, ldw = _rows(w)
    M, xn, ldx = _rows(x)
    if (a_koff is None and (ak != K or am != M)) or xn != N:
        raise RuntimeError("gemm_gate_residual: shape mismatch")
    gate_ld = gate.stride(0) if gate is not None else 0
    st = _stream()
    ws = _gemm_workspace(a.device, st)
    _check(lib().flexam_gemm_bf16_gate_residual(_ptr(a, BF16), lda, _ptr(w, BF16), ldw, _ptr(bias, F32), _ptr(x, F32), ldx,
                                                _ptr(gate, F32), gate_ld, _ptr(gate_row, I32), rows_per_batch, M, N, K,
                                                _ptr(a_koff, I64), _raw(ws), ws.numel(), st), "flexam_gemm_bf16_gate_residual")
    return x


# ----------------------------------------------------------------------------- fp8 GEMM (BASELINE configs[4])
U8 = torch.uint8


def quantize_rows_fp8(x, q=None, scale=None):
    """x [M, K] bf16 rows -> (q [M, K] uint8 holding OCP e4m3 bytes, scale [M] fp32) with x ~ q * scale[:, None]."""
    M, K, ldx = _rows(x)
    if q is None:
        q = torch.empty(M, K, device=x.device, dtype=U8)
    if scale is None:
        scale = torch.empty(M, device=x.device, dtype=F32)
    _check(lib().flexam_quantize_rows_fp8(_ptr(x, BF16), ldx, _ptr(q, U8), q.stride(0), _ptr(scale, F32), M, K, _stream()),
           "flexam_quantize_rows_fp8")
    return q, scale


def gemm_fp8(a8, a_scale, w8, w_scale, bias=None, out=None, epilogue=EPI_NONE):
    """out[M,N] bf16 = epi((a8 @ w8^T) * a_scale[:, None] * w_scale[None, :] + bias); a8 [M,K], w8 [N,K] e4m3 bytes."""
    M, K, lda = _rows(a8)
    N, wk, ldw = _rows(w8)
    if wk != K:
        raise RuntimeError(f"gemm_fp8: K mismatch a {K} vs w {wk}")
    if out is None:
        out = torch.empty(M, N, device=a8.device, dtype=BF16)
    _check(lib().flexam_gemm_fp8(_ptr(a8, U8), lda, _ptr(a_scale, F32), _ptr(w8, U8), ldw, _ptr(w_scale, F32), _ptr(bias, F32),
                                 _ptr(out, BF16), out.stride(0), M, N, K, epilogue, _stream()), "flexam_gemm_fp8")
    return out


def gemm_fp8_gelu_q(a8, a_scale, w8, w_scale, bias, out_scale, q_out):
    """q_out[M,N] (e4m3 bytes) = e4m3(gelu_tanh((a8 @ w8^T) * scales + bias) / out_scale[:, None]): FFN1 writing FFN2's A operand."""
    M, K, lda = _rows(a8)
    N, wk, ldw = _rows(w8)
    qm, qn, ldq = _rows(q_out)
    if wk != K or qm != M or qn != N:
        raise RuntimeError("gemm_fp8_gelu_q: shape mismatch")
    _check(lib().flexam_gemm_fp8_gelu_q(_ptr(a8, U8), lda, _ptr(a_scale, F32), _ptr(w8, U8), ldw, _ptr(w_scale, F32), _ptr(bias, F32),
                                        _ptr(out_scale, F32), _ptr(q_out, U8), ldq, M, N, K, _stream()), "flexam_gemm_fp8_gelu_q")
    return q_out


def gemm_fp8_gate_residual(a8, a_scale, w8, w_scale, bias, x, gate=None, gate_row=None, rows_per_batch=0):
    """x[M,N] (fp32, in place) += bf16((a8 @ w8^T) * scales + bias) * gate[row]."""
    M, K, lda = _rows(a8)
    N, wk, ldw = _rows(w8)
    xm, xn, ldx = _rows(x)
    if wk != K or xm != M or xn != N:
        raise RuntimeError("gemm_fp8_gate_residual: shape mismatch")
    _check(lib().flexam_gemm_fp8_gate_residual(_ptr(a8, U8), lda, _ptr(a_scale, F32), _ptr(w8, U8), ldw, _ptr(w_scale, F32), _ptr(bias, F32),
                                               _ptr(x, F32), ldx, _ptr(gate, F32), gate.stride(0) if gate is not None else 0,
                                               _ptr(gate_row, I32), rows_per_batch, M, N, K, _stream()), "flexam_gemm_fp8_gate_residual")
    return x


# ----------------------------------------------------------------------------- attention
def attn_split_plan(batch_heads: int, lq: int, lk: int, n_cu: int = 256):
    """(kv_splits, split_from_unit) for flexam_attn_fwd_splitkv.  W = batch_heads * ceil(lq/256) work units of ceil(lk/64) key
    tiles; the launch takes ceil(W/n_cu) rounds.  Cutting only the units of the last, partial round into S key ranges turns
    that round into ceil(rem*S/n_cu)/S of a round (+ 4 % per pass for the partial outputs and the merge); S is kept only if the
    whole launch gets more than 2 % shorter and every range keeps at least 8 key tiles."""
    w = batch_heads * ((lq + 255) // 256)
    tiles = (lk + 63) // 64
    full, rem = (w // n_cu) * n_cu, w % n_cu
    if rem == 0:
        return 1, w
    best, best_cost = 1, float(w // n_cu + 1)
    for s in (2, 3, 4, 5, 6, 8):
        if tiles // s < 8:
            break
        passes = -(-rem * s // n_cu)
        cost = w // n_cu + passes * (1.0 / s + 0.04)
        if cost < best_cost * 0.98:
            best, best_cost = s, cost
    return (best, full) if best > 1 else (1, w)


def attn_kv_splits(batch_heads: int, lq: int, lk: int, n_cu: int = 256) -> int:
    return attn_split_plan(batch_heads, lq, lk, n_cu)[0]


_ATTN_WS = {}


ATTN_PRESCALED = -1.0      # flexam_hip.h FLEXAM_ATTN_PRESCALED


def attn_fwd_lastkey(q, k, v, last_key_multiplicity, out=None, softmax_scale=None, prescaled=False):
    """attn_fwd with the last key counted `last_key_multiplicity` times (identical trailing context rows folded into one, see
    flexam_hip.h); same layouts as attn_fwd, no split-KV (short contexts)."""
    B, Lq, H, D = q.shape
    Lk = k.shape[1]
    for t in (q, k, v):
        if t.stride(3) != 1 or t.stride(2) != D:
            raise RuntimeError("attn_fwd_lastkey: heads must be packed along the row (stride(2) == head_dim, stride(3) == 1)")
    if out is None:
        out = torch.empty(B, Lq, H, D, device=q.device, dtype=BF16)
    scale = ATTN_PRESCALED if prescaled else (softmax_scale if softmax_scale is not None else D ** -0.5)
    _check(lib().flexam_attn_fwd_lastkey(_ptr(q, BF16), q.stride(0), q.stride(1), _ptr(k, BF16), k.stride(0), k.stride(1),
                                         _ptr(v, BF16), v.stride(0), v.stride(1), _ptr(out, BF16), out.stride(0), out.stride(1),
                                         B, H, Lq, Lk, D, scale, float(last_key_multiplicity), _stream()), "flexam_attn_fwd_lastkey")
    return out


def attn_fwd(q, k, v, out=None, softmax_scale=None, kv_splits=None, split_from_unit=None, prescaled=False):
    """q [B,Lq,H,128], k/v [B,Lk,H,128] bf16 (arbitrary batch/row strides, head dim contiguous and
    heads packed: stride(2) == 128) -> out [B,Lq,H,128] bf16.  kv_splits None: plan from attn_split_plan (only the last,
    partial round of the CUs is split); an explicit kv_splits splits every unit unless split_from_unit is given too.
    prescaled: q already carries softmax_scale * log2(e) (folded into its producer before the rounding to bf16)."""
    B, Lq, H, D = q.shape
    Lk = k.shape[1]
    for t in (q, k, v):
        if t.stride(3) != 1 or t.stride(2) != D:
            raise RuntimeError("attn_fwd: heads must be packed along the row (stride(2) == head_dim, stride(3) == 1)")
    if out is None:
        out = torch.empty(B, Lq, H, D, device=q.device, dtype=BF16)
    scale = ATTN_PRESCALED if prescaled else (softmax_scale if softmax_scale is not None else D ** -0.5)
    units = B * H * ((Lq + 255) // 256)
    if kv_splits is None:
        S, from_unit = attn_split_plan(B * H, Lq, Lk, num_cus())
    else:
        S, from_unit = int(kv_splits), (0 if split_from_unit is None else int(split_from_unit))
    if S <= 1:
        _check(lib().flexam_attn_fwd(_ptr(q, BF16), q.stride(0), q.stride(1), _ptr(k, BF16), k.stride(0), k.stride(1),
                                     _ptr(v, BF16), v.stride(0), v.stride(1), _ptr(out, BF16), out.stride(0), out.stride(1),
                                     B, H, Lq, Lk, D, scale, _stream()), "flexam_attn_fwd")
        return out
    n = units - from_unit
    st = _stream()
    slot, key = (q.device.index if q.device.index is not None else torch.cuda.current_device(), st), (S, n)
    if _ATTN_WS.get(slot, (None,))[0] != key:   # per-shape scratch per (device, stream), reused across launches (stream-ordered)
        _ATTN_WS.pop(slot, None)
        _ws_slot(_ATTN_WS, slot, lambda: (key, torch.empty(S, n, 256, D, device=q.device, dtype=F32),
                                          torch.empty(S, n, 256, 2, device=q.device, dtype=F32)))
    _, ws_o, ws_ml = _ws_slot(_ATTN_WS, slot, None)
    _check(lib().flexam_attn_fwd_splitkv(_ptr(q, BF16), q.stride(0), q.stride(1), _ptr(k, BF16), k.stride(0), k.stride(1),
                                         _ptr(v, BF16), v.stride(0), v.stride(1), _ptr(out, BF16), out.stride(0), out.stride(1),
                                         B, H, Lq, Lk, D, scale, S, from_unit, _ptr(ws_o, F32), _ptr(ws_ml, F32), _stream()),
           "flexam_attn_fwd_splitkv")
    return out


def attn_effective_splits(lk: int, splits: int) -> int:
    """Key ranges a request for `splits` really gives: ranges hold whole 64-key tiles, empty trailing ranges are dropped."""
    tiles = (lk + 63) // 64
    splits = max(1, min(int(splits), tiles))
    per = -(-tiles // splits)
    return -(-tiles // per)


def attn_partial_workspace(B, H, Lq, n_slots, device):
    units = B * H * ((Lq + 255) // 256)
    return (torch.empty(n_slots, units, 256, 128, device=device, dtype=F32), torch.empty(n_slots, units, 256, 2, device=device, dtype=F32))


def attn_fwd_partial(q, k, v, ws, slot0, kv_splits=1, softmax_scale=None, prescaled=False):
    """Partial softmax of q against THESE keys into workspace slots slot0 .. slot0 + kv_splits - 1 (see flexam_hip.h);
    returns the number of slots written."""
    B, Lq, H, D = q.shape
    Lk = k.shape[1]
    for t in (q, k, v):
        if t.stride(3) != 1 or t.stride(2) != D:
            raise RuntimeError("attn_fwd_partial: heads must be packed along the row (stride(2) == head_dim, stride(3) == 1)")
    S = attn_effective_splits(Lk, kv_splits)
    ws_o, ws_ml = ws
    if slot0 + S > ws_o.shape[0] or ws_o.shape[1] != B * H * ((Lq + 255) // 256):
        raise RuntimeError("attn_fwd_partial: workspace too small for these slots / this query shape")
    scale = ATTN_PRESCALED if prescaled else (softmax_scale if softmax_scale is not None else D ** -0.5)
    _check(lib().flexam_attn_fwd_partial(_ptr(q, BF16), q.stride(0), q.stride(1), _ptr(k, BF16), k.stride(0), k.stride(1),
                                         _ptr(v, BF16), v.stride(0), v.stride(1), B, H, Lq, Lk, D, scale, S, slot0,
                                         _ptr(ws_o, F32), _ptr(ws_ml, F32), _stream()), "flexam_attn_fwd_partial")
    return S


def attn_merge(out, ws, n_slots, softmax_scale=None, prescaled=False):
    """out [B, Lq, H, 128] bf16 = the softmax over the union of the key sets of workspace slots 0 .. n_slots - 1."""
    B, Lq, H, D = out.shape
    if out.stride(3) != 1 or out.stride(2) != D:
        raise RuntimeError("attn_merge: heads must be packed along the row")
    scale = ATTN_PRESCALED if prescaled else (softmax_scale if softmax_scale is not None else D ** -0.5)
    _check(lib().flexam_attn_merge(_ptr(out, BF16), out.stride(0), out.stride(1), B, H, Lq, D, scale, n_slots, _ptr(ws[0], F32),
                                   _ptr(ws[1], F32), _stream()), "flexam_attn_merge")
    return out


ATTN8_REC_BYTES = 18432


def attn_fp8_buffers(B, H, L, device):
    """(q8, qs, kv8) for attn_fp8_pack / attn_fwd_fp8 at this shape (see flexam_hip.h)."""
    lp, tiles = -(-L // 256) * 256, -(-L // 64)
    return (torch.zeros(B, H, lp, 128, device=device, dtype=torch.uint8), torch.zeros(B, H, lp, device=device, dtype=torch.int32),
            torch.zeros(B, H, tiles, ATTN8_REC_BYTES, device=device, dtype=torch.uint8))      # zeros: rmsnorm_rope_mx never writes the padding rows


def _check_fp8_bufs(bufs, B, H, L, device, who):
    """The MXFP8 operand buffers are raw byte images the C side cannot size-check: refuse anything that is not what attn_fp8_buffers(B,
    H, L) makes (shape, dtype, device, contiguity) -- a buffer set of another (B, L) would be out-of-bounds device traffic, not an
    error.  The kernels also rely on the padding rows past L staying zero, which attn_fp8_buffers' torch.zeros guarantees and nothing
    here ever writes."""
    if not isinstance(bufs, (tuple, list)) or len(bufs) != 3:
        raise RuntimeError(f"{who}: bufs must be the (q8, qs, kv8) triple of attn_fp8_buffers")
    q8, qs, kv8 = bufs
    lp, tiles = -(-L // 256) * 256, -(-L // 64)
    want = ((q8, (B, H, lp, 128), torch.uint8), (qs, (B, H, lp), torch.int32), (kv8, (B, H, tiles, ATTN8_REC_BYTES), torch.uint8))
    for t, shape, dt in want:
        if tuple(t.shape) != shape or t.dtype != dt or not t.is_contiguous() or t.device != torch.device(device):
            raise RuntimeError(f"{who}: operand buffers are not attn_fp8_buffers(B={B}, H={H}, L={L}) on {device}: got {tuple(t.shape)} {t.dtype} "
                               f"on {t.device}, want {shape} {dt}")


def attn_fp8_pack(q, k, v, bufs=None):
    """q, k, v [B, L, H, 128] bf16 (q prescaled by softmax_scale * log2 e) -> the MXFP8 operand buffers of attn_fwd_fp8."""
    B, L, H, D = v.shape
    for t in (q, k, v):
        if t is not None and (t.stride(3) != 1 or t.stride(2) != D or t.shape != v.shape):
            raise RuntimeError("attn_fp8_pack: q, k, v must be [B, L, H, 128] with packed heads")
    if (q is None) != (k is None):
        raise RuntimeError("attn_fp8_pack: q and k go together (both None: V only, after rmsnorm_rope_mx)")
    if bufs is None:
        bufs = attn_fp8_buffers(B, H, L, v.device)
    _check_fp8_bufs(bufs, B, H, L, v.device, "attn_fp8_pack")
    q8, qs, kv8 = bufs
    qk = (lambda t: (t.stride(0), t.stride(1))) if q is not None else (lambda t: (0, 0))
    _check(lib().flexam_attn_fp8_pack(_ptr(q, BF16), *qk(q), _ptr(k, BF16), *qk(k), _ptr(v, BF16),
                                      v.stride(0), v.stride(1), _raw(q8), _raw(qs), _raw(kv8), B, H, L, D, _stream()),
           "flexam_attn_fp8_pack")
    return bufs


def rmsnorm_rope_mx(q, wq, k, wk, bufs, rope_cos, rope_sin, tokens_per_batch, token_offset=0, eps=1e-6, heads=24):
    """RMSNorm + RoPE of q and k ([M, 3072] bf16 views) written as the MXFP8 operands of attn_fwd_fp8 (Q rows, K image and scales);
    the V half of `bufs` comes from attn_fp8_pack(None, None, v, bufs)."""
    M, C, ldq = _rows(q)
    if tokens_per_batch <= 0 or M % tokens_per_batch or C != heads * 128:
        raise RuntimeError(f"rmsnorm_rope_mx: {M} rows of {C} columns do not tile batches of {tokens_per_batch} tokens x {heads} heads x 128")
    _check_fp8_bufs(bufs, M // tokens_per_batch, heads, tokens_per_batch, q.device, "rmsnorm_rope_mx")
    q8, qs, kv8 = bufs
    _check(lib().flexam_rmsnorm_rope_mx(_ptr(q, BF16), ldq, _ptr(wq, F32), _ptr(k, BF16), k.stride(0), _ptr(wk, F32), _raw(q8),
                                        _raw(qs), _raw(kv8), M, C, eps, _ptr(rope_cos, F32), _ptr(rope_sin, F32),
                                        tokens_per_batch, token_offset, heads, 128, _stream()), "flexam_rmsnorm_rope_mx")
    return bufs


def attn_fwd_fp8(bufs, L, out=None, kv_splits=None, split_from_unit=None):
    """Self-attention from packed MXFP8 operands (attn_fp8_pack) -> out [B, L, H, 128] bf16."""
    q8, qs, kv8 = bufs
    B, H, D = q8.shape[0], q8.shape[1], 128
    _check_fp8_bufs(bufs, B, H, L, q8.device, "attn_fwd_fp8")
    if out is None:
        out = torch.empty(B, L, H, D, device=q8.device, dtype=BF16)
    if out.stride(3) != 1 or out.stride(2) != D or tuple(out.shape) != (B, L, H, D):
        raise RuntimeError(f"attn_fwd_fp8: out must be [B={B}, L={L}, H={H}, 128] with heads packed along the row, got {tuple(out.shape)}")
    units = B * H * ((L + 255) // 256)
    if kv_splits is None:
        S, from_unit = attn_split_plan(B * H, L, L, num_cus())
    else:
        S, from_unit = int(kv_splits), (0 if split_from_unit is None else int(split_from_unit))
    ws_o = ws_ml = None
    if S > 1:
        n = units - from_unit
        st = _stream()
        slot, key = (q8.device.index if q8.device.index is not None else torch.cuda.current_device(), st), (S, n)
        if _ATTN_WS.get(slot, (None,))[0] != key:
            _ATTN_WS.pop(slot, None)
            _ws_slot(_ATTN_WS, slot, lambda: (key, torch.empty(S, n, 256, D, device=q8.device, dtype=F32),
                                              torch.empty(S, n, 256, 2, device=q8.device, dtype=F32)))
        _, ws_o, ws_ml = _ws_slot(_ATTN_WS, slot, None)
    _check(lib().flexam_attn_fwd_fp8(_raw(q8), _raw(qs), _raw(kv8), _ptr(out, BF16), out.stride(0), out.stride(1), B, H, L, D,
                                     max(S, 1), from_unit if S > 1 else 0, _ptr(ws_o, F32), _ptr(ws_ml, F32), _stream()), "flexam_attn_fwd_fp8")
    return out


def attn_fwd_fp8_chunked(q8, qs, kv8_chunks, lq, lk, out=None):
    """Attention of `lq` local queries (q8 / qs of attn_fp8_buffers(B, H, lq)) over `lk` keys whose MXFP8 records come in CHUNKS:
    kv8_chunks uint8 [n_chunks, B, H, chunk_tiles, ATTN8_REC_BYTES], chunk c holding the keys [c, c + 1) * chunk_tiles * 64 -- the
    rank-major result of all-gathering every sequence-parallel rank's own records.  -> out [B, lq, H, 128] bf16."""
    B, H, D = q8.shape[0], q8.shape[1], 128
    lp = -(-lq // 256) * 256
    if tuple(q8.shape) != (B, H, lp, D) or tuple(qs.shape) != (B, H, lp) or q8.dtype != U8 or qs.dtype != torch.int32 or not (q8.is_contiguous() and qs.is_contiguous()):
        raise RuntimeError(f"attn_fwd_fp8_chunked: q8 / qs are not the query buffers of attn_fp8_buffers(B={B}, H={H}, L={lq})")
    if (kv8_chunks.dim() != 5 or kv8_chunks.dtype != U8 or not kv8_chunks.is_contiguous() or tuple(kv8_chunks.shape[1:3]) != (B, H)
            or kv8_chunks.shape[4] != ATTN8_REC_BYTES or kv8_chunks.device != q8.device):
        raise RuntimeError(f"attn_fwd_fp8_chunked: records must be contiguous uint8 [chunks, B={B}, H={H}, chunk_tiles, {ATTN8_REC_BYTES}], got {tuple(kv8_chunks.shape)}")
    n_chunks, chunk_tiles = kv8_chunks.shape[0], kv8_chunks.shape[3]
    if not 0 < lk <= n_chunks * chunk_tiles * 64:
        raise RuntimeError(f"attn_fwd_fp8_chunked: {lk} keys do not fit {n_chunks} chunks of {chunk_tiles} tiles")
    if out is None:
        out = torch.empty(B, lq, H, D, device=q8.device, dtype=BF16)
    if out.stride(3) != 1 or out.stride(2) != D or tuple(out.shape) != (B, lq, H, D):
        raise RuntimeError(f"attn_fwd_fp8_chunked: out must be [B={B}, L={lq}, H={H}, 128] with heads packed along the row, got {tuple(out.shape)}")
    units = B * H * ((lq + 255) // 256)
    S, from_unit = attn_split_plan(B * H, lq, lk, num_cus())
    ws_o = ws_ml = None
    if S > 1:
        n = units - from_unit
        st = _stream()
        slot, key = (q8.device.index if q8.device.index is not None else torch.cuda.current_device(), st), (S, n)
        if _ATTN_WS.get(slot, (None,))[0] != key:
            _ATTN_WS.pop(slot, None)
            _ws_slot(_ATTN_WS, slot, lambda: (key, torch.empty(S, n, 256, D, device=q8.device, dtype=F32),
                                              torch.empty(S, n, 256, 2, device=q8.device, dtype=F32)))
        _, ws_o, ws_ml = _ws_slot(_ATTN_WS, slot, None)
    _check(lib().flexam_attn_fwd_fp8_chunked(_raw(q8), _raw(qs), _raw(kv8_chunks), _ptr(out, BF16), out.stride(0), out.stride(1), B, H, lq, lk,
                                             chunk_tiles, D, max(S, 1), from_unit if S > 1 else 0, _ptr(ws_o, F32), _ptr(ws_ml, F32), _stream()),
           "flexam_attn_fwd_fp8_chunked")
    return out


# ----------------------------------------------------------------------------- DiT row kernels
def ln_modulate(x, out=None, eps=1e-6, shift=None, scale=None, row_index=None, rows_per_batch=0, ln_w=None, ln_b=None):
    M, C, ldx = _rows(x)
    if out is None:
        out = torch.empty(M, C, device=x.device, dtype=BF16)
    tab_ld = shift.stride(0) if shift is not None else 0
    if shift is not None and (shift.stride(-1) != 1 or scale.stride(-1) != 1 or scale.stride(0) != tab_ld):
        raise RuntimeError("ln_modulate: shift/scale must be row views of one table")
    _check(lib().flexam_ln_modulate(_ptr(x, F32), ldx, M, C, eps, _ptr(shift, F32), _ptr(scale, F32), tab_ld,
                                    _ptr(row_index, I32), rows_per_batch, _ptr(ln_w, F32), _ptr(ln_b, F32), _ptr(out, BF16),
                                    out.stride(0), _stream()), "flexam_ln_modulate")
    return out


def ln_modulate_fp8(x, q_out, row_scale, eps=1e-6, shift=None, scale=None, row_index=None, rows_per_batch=0, ln_w=None, ln_b=None,
                    next_scale=None, next_wnorm=0.0, next_bias=0.0):
    """ln_modulate with the output written as e4m3 bytes q_out [M, C] (uint8, row stride free) + row_scale [M] fp32.  next_scale [M]
    (optional): the output scale of the GEMM + GELU this row feeds (gemm_fp8_gelu_q), from the row's L2 norm and the bounds
    next_wnorm >= max |w_j|_2, next_bias >= max |b_j| of that GEMM (see flexam_hip.h)."""
    M, C, ldx = _rows(x)
    tab_ld = shift.stride(0) if shift is not None else 0
    _check(lib().flexam_ln_modulate_fp8(_ptr(x, F32), ldx, M, C, eps, _ptr(shift, F32), _ptr(scale, F32), tab_ld, _ptr(row_index, I32),
                                        rows_per_batch, _ptr(ln_w, F32), _ptr(ln_b, F32), _ptr(q_out, torch.uint8), q_out.stride(0),
                                        _ptr(row_scale, F32), _ptr(next_scale, F32), float(next_wnorm), float(next_bias), _stream()),
           "flexam_ln_modulate_fp8")
    return q_out, row_scale


def gate_residual(x, y, gate=None, row_index=None, rows_per_batch=0):
    M, C, ldx = _rows(x)
    gate_ld = gate.stride(0) if gate is not None else 0
    _check(lib().flexam_gate_residual(_ptr(x, F32), ldx, _ptr(y, BF16), y.stride(0), _ptr(gate, F32), gate_ld,
                                      _ptr(row_index, I32), rows_per_batch, M, C, _stream()), "flexam_gate_residual")
    return x


def rmsnorm_rope(q, wq, k=None, wk=None, eps=1e-6, rope_cos=None, rope_sin=None, tokens_per_batch=0, token_offset=0,
                 head_dim=128, q_out=None, k_out=None):
    """In place by default.  q/k: 2-D bf16 views [M, C]."""
    M, C, ldq = _rows(q)
    q_out = q if q_out is None else q_out
    k_out = k if k_out is None else k_out
    _check(lib().flexam_rmsnorm_rope(_ptr(q, BF16), ldq, _ptr(q_out, BF16), q_out.stride(0), _ptr(wq, F32),
                                     _ptr(k, BF16), k.stride(0) if k is not None else 0,
                                     _ptr(k_out, BF16), k_out.stride(0) if k_out is not None else 0, _ptr(wk, F32),
                                     M, C, eps, _ptr(rope_cos, F32), _ptr(rope_sin, F32), tokens_per_batch, token_offset,
                                     head_dim, _stream()), "flexam_rmsnorm_rope")
    return q_out, k_out


def rmsnorm_rope_scatter(q, wq, k, wk, v, q_out, k_out, v_out, ld_out, out_bs, col_block, block_stride, eps=1e-6, rope_cos=None,
                         rope_sin=None, tokens_per_batch=0, token_offset=0, head_dim=128):
    """q/k/v: 2-D bf16 views [M, C] (q, v optional); *_out: bf16 tensors whose data pointers are the bases of the scattered
    layout described in flexam_hip.h (element (m, col) -> (m // tpb) * out_bs + (m % tpb) * ld_out + (col // col_block) *
    block_stride + col % col_block)."""
    M, C, ldk = _rows(k)
    _check(lib().flexam_rmsnorm_rope_scatter(_ptr(q, BF16), q.stride(0) if q is not None else 0, _ptr(wq, F32), _ptr(k, BF16), ldk,
                                             _ptr(wk, F32), _ptr(v, BF16), v.stride(0) if v is not None else 0, _ptr(q_out, BF16),
                                             _ptr(k_out, BF16), _ptr(v_out, BF16), ld_out, out_bs, col_block, block_stride, M, C, eps,
                                             _ptr(rope_cos, F32), _ptr(rope_sin, F32), tokens_per_batch, token_offset, head_dim,
                                             _stream()), "flexam_rmsnorm_rope_scatter")


def mod_table(mod, e, out, rows_per_batch, scale_mask, mdens=None, dens=None, dens_slots=-1):
    """mod [nblk,nj,C], e [R,nj,C], mdens [nblk,nslot,C], dens [B,nslot,C] -> out [nblk,R,nj,C] fp32."""
    nblk, nj, C = mod.shape
    R = e.shape[0]
    nslot = mdens.shape[1] if mdens is not None else 0
    for t in (mod, e, out, mdens, dens):
        if t is not None and not t.is_contiguous():
            raise RuntimeError("mod_table: tensors must be contiguous")
    _check(lib().flexam_mod_table(_ptr(mod, F32), _ptr(e, F32), _ptr(mdens, F32), _ptr(dens, F32), _ptr(out, F32), nblk, R, nj,
                                  nslot, C, rows_per_batch, scale_mask, dens_slots, _stream()), "flexam_mod_table")
    return out


def small_linear(x, w, b=None, silu_in=False, out=None):
    """fp32 y[M,N] = silu?(x[M,K]) @ w[N,K]^T + b, M <= 32; w bf16 or fp32 (each output sums in the same order whatever M is)."""
    M, K, ldx = _rows(x)
    N, wk, ldw = _rows(w)
    if wk != K:
        raise RuntimeError("small_linear: K mismatch")
    if out is None:
        out = torch.empty(M, N, device=x.device, dtype=F32)
    if w.dtype not in (BF16, F32):
        raise RuntimeError("small_linear: weight must be bf16 or fp32")
    _check(lib().flexam_small_linear_f32(_ptr(x, F32), ldx, _ptr(w), 1 if w.dtype == BF16 else 0, ldw, _ptr(b, F32),
                                         _ptr(out, F32), out.stride(0), M, N, K, 1 if silu_in else 0, _stream()),
           "flexam_small_linear_f32")
    return out


def sinusoid_embed(t, dim, out=None):
    R = t.numel()
    if out is None:
        out = torch.empty(R, dim, device=t.device, dtype=F32)
    _check(lib().flexam_sinusoid_embed(_ptr(t.contiguous(), F32), _ptr(out, F32), R, dim, _stream()), "flexam_sinusoid_embed")
    return out


def patchify(src, dst, col0=0, row0=0):
    """src [C,F,H,W] (fp32/bf16, contiguous) -> dst[row0 + token, col0 + c*4+ph*2+pw] (bf16 2-D)."""
    C, F, H, W = src.shape
    if not src.is_contiguous():
        raise RuntimeError("patchify: src must be contiguous")
    _check(lib().flexam_patchify(_ptr(src), 1 if src.dtype == BF16 else 0, C, F, H, W, _ptr(dst, BF16), dst.stride(0), col0, row0,
                                 _stream()), "flexam_patchify")
    return dst


def unpatchify(tok, tok0, C, F, H, W, out=None, dtype=F32):
    if out is None:
        out = torch.empty(C, F, H, W, device=tok.device, dtype=dtype)
    _check(lib().flexam_unpatchify(_ptr(tok, F32), tok.stride(0), tok0, C, F, H, W, _ptr(out), 1 if out.dtype == BF16 else 0,
                                   _stream()), "flexam_unpatchify")
    return out


def cfg_euler_blend(tok_uncond, tok_cond, tok0, guidance, dt, latents, known=None, mask=None):
    C, F, H, W = latents.shape
    _check(lib().flexam_cfg_euler_blend(_ptr(tok_uncond, F32), _ptr(tok_cond, F32), tok_uncond.stride(0), tok0, guidance, dt,
                                        _ptr(latents, F32), _ptr(known, F32), _ptr(mask, F32), C, F, H, W, _stream()),
           "flexam_cfg_euler_blend")
    return latents


def axpby(y, a, x, b):
    """y = a*x + b*y (fp32, same shape, contiguous)."""
    if y.shape != x.shape or not y.is_contiguous() or not x.is_contiguous():
        raise RuntimeError("axpby: contiguous tensors of equal shape required")
    _check(lib().flexam_axpby_f32(_ptr(y, F32), a, _ptr(x, F32), b, y.numel(), _stream()), "flexam_axpby_f32")
    return y


def checksum(t: torch.Tensor):
    """Content fingerprint (two 64-bit sums of per-(index, word) hashes) of a device tensor; synchronises (host logic only)."""
    return checksums([t])[0]


def checksums(tensors):
    """Fingerprints of several device tensors with ONE readback: every launch adds into its own slot of one buffer."""
    ts = [t.contiguous() for t in tensors]
    if not ts:
        return []
    out = torch.zeros(len(ts), 2, device=ts[0].device, dtype=I64)
    for i, t in enumerate(ts):
        _check(lib().flexam_checksum(_ptr(t), t.numel() * t.element_size(), out[i].data_ptr(), _stream()), "flexam_checksum")
    return [tuple(r) for r in out.tolist()]


def cfg_velocity(tok_uncond, tok_cond, tok0, guidance, out):
    """out [C,F,H,W] fp32 = unpatchify(u + g (c - u)); tok_cond None: out = unpatchify(u)."""
    C, F, H, W = out.shape
    _check(lib().flexam_cfg_velocity(_ptr(tok_uncond, F32), _ptr(tok_cond, F32), tok_uncond.stride(0), tok0, guidance, _ptr(out, F32),
                                     C, F, H, W, _stream()), "flexam_cfg_velocity")
    return out


def lincomb(out, terms):
    """out = sum(c * t for c, t in terms): fp32 contiguous tensors of out's shape; out may be one of them."""
    n = len(terms)
    for _, t in terms:
        if t.shape != out.shape or t.dtype != F32 or not t.is_contiguous() or t.device != out.device:
            raise RuntimeError("lincomb: fp32 contiguous tensors of equal shape on one device required")
    if not out.is_contiguous() or out.dtype != F32:
        raise RuntimeError("lincomb: out must be fp32 contiguous")
    ptrs = (c_void_p * n)(*[t.data_ptr() for _, t in terms])
    coefs = (c_float * n)(*[float(c) for c, _ in terms])
    _check(lib().flexam_lincomb_f32(_ptr(out, F32), out.numel(), n, ctypes.cast(ptrs, c_void_p), ctypes.cast(coefs, c_void_p),
                                    _stream()), "flexam_lincomb_f32")
    return out


def mask_blend(x, known, mask):
    """x [C, ...] = (1 - mask) * known + mask * x with mask [...] broadcast over the leading channel dim."""
    C = x.shape[0]
    fhw = x.numel() // C
    if mask.numel() != fhw or known.shape != x.shape:
        raise RuntimeError("mask_blend: shape mismatch")
    _check(lib().flexam_mask_blend_f32(_ptr(x, F32), _ptr(known, F32), _ptr(mask, F32), C, fhw, _stream()), "flexam_mask_blend_f32")
    return x


# ----------------------------------------------------------------------------- channels-last conv helpers
def pack_cl(src, dst, c0=0):
    """src [C,F,H,W] -> interior of dst [F,H+2,W+2,Cp] (bf16) at channel offset c0."""
    C, F, H, W = src.shape
    if not src.is_contiguous() or dst.shape[:3] != (F, H + 2, W + 2) or not dst.is_contiguous():
        raise RuntimeError("pack_cl: bad layout")
    _check(lib().flexam_pack_cl(_ptr(src), 1 if src.dtype == BF16 else 0, C, F, H, W, _ptr(dst, BF16), dst.shape[3], c0, _stream()),
           "flexam_pack_cl")
    return dst


def unpack_cl(src, C, F, H, W, out=None):
    """src [F*(H+2)*(W+2), ld] (fp32/bf16 rows) -> [C,F,H,W] fp32."""
    if out is None:
        out = torch.empty(C, F, H, W, device=src.device, dtype=F32)
    _check(lib().flexam_unpack_cl(_ptr(src), 1 if src.dtype == BF16 else 0, src.stride(0), C, F, H, W, _ptr(out, F32), _stream()),
           "flexam_unpack_cl")
    return out


def groupnorm_silu_cl(x, C, F, H, W, groups, gamma, beta, dst, residual=None, eps=1e-5, stats=None):
    """x [F*(H+2)*(W+2), ld] fp32 -> dst [F,H+2,W+2,Cp] bf16 interior; residual: bf16 padded image."""
    if stats is None:
        stats = torch.empty(2 * groups, device=x.device, dtype=F32)
    _check(lib().flexam_groupnorm_silu_cl(_ptr(x, F32), x.stride(0), C, F, H, W, groups, eps, _ptr(gamma, F32), _ptr(beta, F32),
                                          _ptr(stats, F32), _ptr(residual, BF16), residual.shape[-1] if residual is not None else 0,
                                          _ptr(dst, BF16), dst.shape[-1], _stream()), "flexam_groupnorm_silu_cl")
    return dst


# ----------------------------------------------------------------------------- VAE decoder helpers
def vae_prep_cl(src, C, T, H, W, dst, mode=0, gamma=None, t0=0, compact=False):
    """src rows [T*(H+2)*(W+2), ld] (fp32/bf16) -> dst image [*, H+2, W+2, Cp] (frame offset t0) or compact [T*H*W, Cp]."""
    _check(lib().flexam_vae_prep_cl(_ptr(src), 1 if src.dtype == BF16 else 0, src.stride(0), C, T, H, W, _ptr(gamma, F32), mode,
                                    _ptr(dst, BF16), dst.shape[-1], t0, 1 if compact else 0, _stream()), "flexam_vae_prep_cl")
    return dst


def upsample2x_cl(src, C, T, H, W, dst, interleave=False):
    _check(lib().flexam_upsample2x_cl(_ptr(src), 1 if src.dtype == BF16 else 0, src.stride(0), C, T, H, W, 1 if interleave else 0,
                                      _ptr(dst, BF16), dst.shape[-1], _stream()), "flexam_upsample2x_cl")
    return dst


def dupup_add_cl(x_main, Co, To, Ho, Wo, x_in, Ci, ft, drop):
    _check(lib().flexam_dupup_add_cl(_ptr(x_main, F32), x_main.stride(0), Co, To, Ho, Wo, _ptr(x_in, F32), x_in.stride(0), Ci, ft, drop,
                                     _stream()), "flexam_dupup_add_cl")
    return x_main


def deinterleave_cl(src, C, T, H, W, dst):
    """src rows [T*(H+2)*(W+2), 2C] -> dst padded image [2T, H+2, W+2, Cp]: frame 2t + s = channels [sC, (s+1)C) of frame t."""
    _check(lib().flexam_deinterleave_cl(_ptr(src), 1 if src.dtype == BF16 else 0, src.stride(0), C, T, H, W, _ptr(dst, BF16), dst.shape[-1],
                                        _stream()), "flexam_deinterleave_cl")
    return dst


def phase_dupup_cl(phases, x_main, Co, To, Ho, Wo, x_in, Ci, ft, drop):
    """phases [4, rows of the padded (Ho/2, Wo/2) image, Co] fp32 -> x_main rows of the padded (Ho, Wo) image, + the DupUp3D shortcut of x_in."""
    if phases.dim() != 3 or phases.shape[0] != 4 or phases.stride(2) != 1 or phases.shape[2] != Co:
        raise RuntimeError("phase_dupup_cl: phases must be [4, rows, Co] fp32")
    _check(lib().flexam_phase_dupup_cl(_ptr(phases, F32), phases.stride(1), phases.stride(0), _ptr(x_main, F32), x_main.stride(0), Co, To, Ho, Wo,
                                       _ptr(x_in, F32), x_in.stride(0), Ci, ft, drop, _stream()), "flexam_phase_dupup_cl")
    return x_main


def tapsum_cl(y, T, H, W, kt, Co, bias, out):
    """y [(kt - 1 + T) * (H+2) * (W+2), >= kt*9*Co] fp32 per-tap products -> out rows [T * (H+2) * (W+2), Co] fp32 (interior positions)."""
    _check(lib().flexam_tapsum_cl(_ptr(y, F32), y.stride(0), T, H, W, kt, Co, _ptr(bias, F32), _ptr(out, F32), out.stride(0), _stream()),
           "flexam_tapsum_cl")
    return out


def softmax_rows(s, scale, out, n_valid):
    M = s.shape[0]
    _check(lib().flexam_softmax_rows(_ptr(s, F32), s.stride(0), M, n_valid, scale, _ptr(out, BF16), out.stride(0), out.shape[1], _stream()),
           "flexam_softmax_rows")
    return out


def scatter_add_cl(x, y, C, T, H, W):
    _check(lib().flexam_scatter_add_cl(_ptr(x, F32), x.stride(0), _ptr(y, BF16), y.stride(0), C, T, H, W, _stream()), "flexam_scatter_add_cl")
    return x


def vae_unpatchify_clamp(src, T, H, W, video, f0, lo=-1.0, hi=1.0):
    _check(lib().flexam_vae_unpatchify_clamp(_ptr(src, F32), src.stride(0), T, H, W, _ptr(video, F32), video.shape[1], f0, lo, hi, _stream()),
           "flexam_vae_unpatchify_clamp")
    return video


def pack_affine_cl(src, mul, add, dst):
    C, T, H, W = src.shape
    _check(lib().flexam_pack_affine_cl(_ptr(src.contiguous(), F32), C, T, H, W, _ptr(mul, F32), _ptr(add, F32), _ptr(dst, BF16),
                                       dst.shape[-1], _stream()), "flexam_pack_affine_cl")
    return dst


# ----------------------------------------------------------------------------- VAE encoder helpers
def vae_patchify_cl(video, f0, T, dst, t0=0):
    """video [3, Ftot, 2H, 2W] fp32 frames f0..f0+T -> dst image [*, H+2, W+2, Cp] frames t0.., 12 channels (c r q)."""
    _, ftot, h2, w2 = video.shape
    _check(lib().flexam_vae_patchify_cl(_ptr(video, F32), ftot, f0, T, h2 // 2, w2 // 2, _ptr(dst, BF16), dst.shape[-1], t0, _stream()),
           "flexam_vae_patchify_cl")
    return dst


def space_to_depth_cl(src, C, T, H, W, dst, Cs, t0=0):
    """src rows [T*(H+2)*(W+2), ld] -> dst image [*, H/2+2, W/2+2, 4*Cs]."""
    _check(lib().flexam_space_to_depth_cl(_ptr(src), 1 if src.dtype == BF16 else 0, src.stride(0), C, T, H, W, _ptr(dst, BF16), Cs, t0,
                                          _stream()), "flexam_space_to_depth_cl")
    return dst


def avgdown_add_cl(x_main, Co, To, Ho, Wo, x_in, Ci, Ti, ft, fs):
    _check(lib().flexam_avgdown_add_cl(_ptr(x_main, F32), x_main.stride(0), Co, To, Ho, Wo, _ptr(x_in, F32), x_in.stride(0), Ci, Ti, ft, fs,
                                       _stream()), "flexam_avgdown_add_cl")
    return x_main


# ----------------------------------------------------------------------------- umT5 text encoder helpers
def t5_norm(x, w, out, eps=1e-6):
    """x [M, C] fp32 rows -> out [M, C] (bf16 or fp32) = w * x * rsqrt(mean(x^2) + eps)."""
    M, C, ldx = _rows(x)
    _check(lib().flexam_t5_norm(_ptr(x, F32), ldx, M, C, eps, _ptr(w, F32), _ptr(out), out.stride(0), 1 if out.dtype == F32 else 0,
                                _stream()), "flexam_t5_norm")
    return out


def softmax_bias_rows(s, out, n_valid, scale=1.0, bias=None, key_mask=None):
    """out [M, Npad] bf16 = softmax(scale * s[:, :n_valid] + bias) over keys with key_mask != 0."""
    M = s.shape[0]
    _check(lib().flexam_softmax_bias_rows(_ptr(s, F32), s.stride(0), M, n_valid, scale, _ptr(bias, F32), bias.stride(0) if bias is not None else 0,
                                          _ptr(key_mask, F32), _ptr(out, BF16), out.stride(0), out.shape[1], _stream()),
           "flexam_softmax_bias_rows")
    return out


def mul_bf16(a, b, out=None):
    if out is None:
        out = torch.empty_like(a)
    if a.shape != b.shape or not (a.is_contiguous() and b.is_contiguous() and out.is_contiguous()):
        raise RuntimeError("mul_bf16: contiguous bf16 tensors of equal shape required")
    _check(lib().flexam_mul_bf16(_ptr(a, BF16), _ptr(b, BF16), _ptr(out, BF16), a.numel(), _stream()), "flexam_mul_bf16")
    return out


# ----------------------------------------------------------------------------- conditioning rasteriser (csrc/raster.hip)
def raster_keys(points, visible, height, width, half, y_min=0, mask=None, keys=None):
    """points [T, N, 3] fp32 (u, v, depth), visible [T, N] uint8 / bool or None, mask [T, H, W] fp32 or None -> keys [T, H, W] int64
    (the uint64 key image of flexam_raster_keys: per pixel the nearest drawn point, all ones = none)."""
    if points.dim() != 3 or points.shape[2] != 3 or not points.is_contiguous():
        raise RuntimeError(f"raster_keys: contiguous points [T, N, 3] required, got {tuple(points.shape)}")
    T, N, _ = points.shape
    if visible is not None:
        if visible.dtype == torch.bool:
            visible = visible.view(U8)
        if visible.shape != (T, N) or not visible.is_contiguous():
            raise RuntimeError(f"raster_keys: visible must be contiguous [T, N] = {(T, N)}, got {tuple(visible.shape)}")
    if mask is not None and (mask.shape != (T, height, width) or not mask.is_contiguous()):
        raise RuntimeError(f"raster_keys: mask must be contiguous [T, H, W] = {(T, height, width)}, got {tuple(mask.shape)}")
    if keys is None:
        keys = torch.empty(T, height, width, device=points.device, dtype=torch.int64)
    elif keys.shape != (T, height, width) or keys.dtype != torch.int64 or not keys.is_contiguous():
        raise RuntimeError("raster_keys: keys must be a contiguous int64 [T, H, W] buffer")
    _check(lib().flexam_raster_keys(_ptr(points, F32), _ptr(visible, U8), T, N, height, width, half, y_min, _ptr(mask, F32), _ptr(keys), _stream()),
           "flexam_raster_keys")
    keys.raster_points = N                         # the point count these keys index: raster_resolve holds its colour table to it
    return keys


def raster_resolve(keys, colors, out_u8=None, out_f32=None, want_u8=False, want_f32=True, n_points=None):
    """keys [T, H, W] (raster_keys), colors [N, 3] or [T, N, 3] uint8 -> (bytes [T, H, W, 3] or None, planes [3, T, H, W] fp32 = byte / 255 or None).
    N must be the point count the keys were made with (raster_keys attaches it to its result; `n_points` for keys from elsewhere): a
    shorter table would be read past its end, a longer per-frame one at another frame's rows."""
    T, H, W = keys.shape
    n_keys = n_points if n_points is not None else getattr(keys, "raster_points", None)
    if colors.dtype != U8 or colors.shape[-1] != 3 or not colors.is_contiguous() or colors.dim() not in (2, 3):
        raise RuntimeError(f"raster_resolve: contiguous uint8 colours [N, 3] or [T, N, 3] required, got {tuple(colors.shape)} {colors.dtype}")
    if colors.dim() == 3 and colors.shape[0] != T:
        raise RuntimeError("raster_resolve: per-frame colours need one table per frame")
    stride = colors.shape[1] * 3 if colors.dim() == 3 else 0
    N = colors.shape[-2]
    if n_keys is not None and N != n_keys:
        raise RuntimeError(f"raster_resolve: the keys index {n_keys} points, the colour table has {N} rows")
    if out_u8 is None and want_u8:
        out_u8 = torch.empty(T, H, W, 3, device=keys.device, dtype=U8)
    if out_f32 is None and want_f32:
        out_f32 = torch.empty(3, T, H, W, device=keys.device, dtype=F32)
    for o, shp in ((out_u8, (T, H, W, 3)), (out_f32, (3, T, H, W))):
        if o is not None and (tuple(o.shape) != shp or not o.is_contiguous()):
            raise RuntimeError(f"raster_resolve: output must be contiguous {shp}, got {tuple(o.shape)}")
    _check(lib().flexam_raster_resolve(_ptr(keys, torch.int64), _ptr(colors, U8), stride, N, T, H, W, _ptr(out_u8, U8), _ptr(out_f32, F32), _stream()),
           "flexam_raster_resolve")
    return out_u8, out_f32
