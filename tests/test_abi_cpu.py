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
