"""CPU: the C-ABI library builds for gfx950, loads without a GPU, and exports exactly the entry
points include/flexam_hip.h declares (no compute calls here)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "flexam_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(flexam_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def libpath():
    from flexam_amd import build
    if not os.path.exists("/opt/rocm/bin/hipcc") and os.path.exists(build.LIB):
        return build.LIB
    return build.build(verbose=False)


def test_header_declares_the_hot_path_entry_points():
    syms = declared_symbols()
    for must in ("flexam_gemm_bf16", "flexam_attn_fwd", "flexam_rmsnorm_rope", "flexam_ln_modulate", "flexam_gate_residual",
                 "flexam_patchify", "flexam_unpatchify", "flexam_cfg_euler_blend", "flexam_version", "flexam_arch",
                 "flexam_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol(libpath):
    out = subprocess.check_output(["nm", "-D", "--defined-only", libpath], text=True)
    exported = set(re.findall(r" T (flexam_[a-z0-9_]+)", out))
    missing = [s for s in declared_symbols() if s not in exported]
    assert not missing, f"declared in include/flexam_hip.h but not exported: {missing}"
    undeclared = sorted(exported - set(declared_symbols()))
    assert not undeclared, f"exported but not declared in the header: {undeclared}"


def test_ctypes_binding_covers_header_and_loads(libpath):
    from flexam_amd import hip
    assert sorted(hip._SIGNATURES) == declared_symbols()
    lib = hip.load_library(libpath)
    assert lib.flexam_version() >= 1
    assert lib.flexam_arch() == b"gfx950"


def test_code_object_targets_gfx950(libpath):
    data = open(libpath, "rb").read()
    assert b"gfx950" in data and b"gfx942" not in data and b"sm_" not in data


def test_no_cpu_fallback_without_gpu():
    """Product ops must fail loudly on CPU tensors rather than route through torch or the oracle."""
    import torch
    from flexam_amd import hip
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        hip.gemm(torch.zeros(4, 64, dtype=torch.bfloat16), torch.zeros(4, 64, dtype=torch.bfloat16))


def test_replay_table_follows_the_header_and_the_binding(libpath):
    """Command lists (csrc/replay.hip): the committed dispatch table is what flexam_amd/gen_replay.py makes of the header; every
    stream-ordered entry point has an id, the ids are the header order, and the argument kinds the generator derives from the C
    prototypes are the ones the ctypes binding declares (pointer / integer / float per position) -- a recorded call is replayed with
    exactly the words it was made with."""
    import ctypes
    from flexam_amd import gen_replay, hip
    header = open(os.path.join(ROOT, "include", "flexam_hip.h")).read()
    assert open(gen_replay.OUT).read() == gen_replay.generate(header), "csrc/replay_table.inc is stale: python -m flexam_amd.gen_replay"
    lib = hip.load_library(libpath)
    kinds = gen_replay.arg_kinds(header)
    assert lib.flexam_fn_count() == len(kinds) >= 50
    of = {hip._P: "p", hip._I: "i", hip._L: "i", hip._F: "f"}
    for i, name in enumerate(kinds):
        assert lib.flexam_fn_id(name.encode()) == i and lib.flexam_fn_name(i) == name.encode()
        argtypes = hip._SIGNATURES[name][0]
        assert argtypes[-1] is hip._P and "".join(of[t] for t in argtypes[:-1]) == kinds[name], name
        assert len(argtypes) - 1 <= hip.REPLAY_MAX_ARGS
    for name, (argtypes, _) in hip._SIGNATURES.items():             # and nothing that takes a stream is missing from the table
        if argtypes and argtypes[-1] is hip._P and name not in kinds:
            assert name in ("flexam_replay",), name
    assert lib.flexam_fn_id(b"flexam_last_error") == -1 and lib.flexam_fn_id(b"flexam_replay") == -1
    assert ctypes.sizeof(hip._Cmd) == 8 + 8 * hip.REPLAY_MAX_ARGS
    # argument checks of the list itself run without a GPU: an empty list is fine, a wrong id / word count is refused before any launch
    failed = ctypes.c_int64(7)
    assert lib.flexam_replay(None, 0, ctypes.byref(failed), None) == 0 and failed.value == -1
    cmds = (hip._Cmd * 2)()
    cmds[0].fn, cmds[0].nargs = lib.flexam_fn_count(), 0
    assert lib.flexam_replay(cmds, 1, ctypes.byref(failed), None) == -1 and failed.value == 0 and b"unknown" not in lib.flexam_last_error()
    cmds[0].fn, cmds[0].nargs = lib.flexam_fn_id(b"flexam_axpby_f32"), 2                  # takes 5 words + the stream
    assert lib.flexam_replay(cmds, 1, ctypes.byref(failed), None) == -1 and b"flexam_axpby_f32" in lib.flexam_last_error()
    # a well-formed command reaches the entry point's own argument check (null pointers: refused there, index reported)
    cmds[1].fn, cmds[1].nargs = lib.flexam_fn_id(b"flexam_gemm_bf16"), 15
    cmds[0].fn, cmds[0].nargs = cmds[1].fn, 15
    rc = lib.flexam_replay(cmds, 2, ctypes.byref(failed), None)
    assert rc == -1 and failed.value == 0 and b"gemm: null pointer" in lib.flexam_last_error()


def test_recorder_builds_the_words_the_call_was_made_with(libpath, monkeypatch):
    """hip.record() without a GPU: a stand-in library function is 'called' through the recorder and the plan holds one C segment whose
    words equal the arguments (pointer, integers, a float), minus the stream; host_op splits segments; failures are not recorded."""
    import ctypes
    from flexam_amd import hip
    lib = hip.load_library(libpath)
    calls = []

    class Fake:
        def __getattr__(self, name):
            real = getattr(lib, name)
            if name != "flexam_axpby_f32":
                return real
            def f(*a):
                calls.append(a)
                return 0 if a[0] else -1
            return f
    monkeypatch.setattr(hip, "_lib", Fake())
    marks = []
    with hip.record() as plan:
        assert hip.recording()
        hip.lib().flexam_axpby_f32(4096, 1.5, 8192, -2.0, 77, 1234)
        hip.host_op(lambda: marks.append("host"))
        hip.lib().flexam_axpby_f32(0, 1.0, 0, 1.0, 1, 1234)                      # fails (rc -1): executed, not recorded
        hip.lib().flexam_axpby_f32(16, 0.25, 32, 0.0, 5, 1234)
        assert hip.lib().flexam_last_error is lib.flexam_last_error              # queries pass through
    assert not hip.recording() and len(calls) == 3 and marks == ["host"]
    assert [k for k, _, _ in plan.items] == ["c", "py", "c"] and plan.launches == 2
    seg = plan.items[0][1]
    assert seg[0].fn == lib.flexam_fn_id(b"flexam_axpby_f32") and seg[0].nargs == 5
    assert (seg[0].a[0].p, seg[0].a[1].f, seg[0].a[2].p, seg[0].a[3].f, seg[0].a[4].i) == (4096, 1.5, 8192, -2.0, 77)
    seg2 = plan.items[2][1]
    assert (seg2[0].a[0].p, seg2[0].a[1].f, seg2[0].a[4].i) == (16, 0.25, 5)
