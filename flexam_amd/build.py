"""Builds libflexam_hip.so (gfx950 only) in-tree with hipcc.  `python -m flexam_amd.build`.

No JIT, no torch cpp_extension: the library is a plain C-ABI shared object (include/flexam_hip.h)
that travels to the GPU box with the source tree."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libflexam_hip.so")
SOURCES = ["api.hip", "gemm.hip", "gemm_fp8.hip", "attn.hip", "dit_elementwise.hip", "conv_cl.hip", "vae.hip", "text_encoder.hip", "raster.hip", "replay.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=fast", "-Wno-unused-result",
         "-I" + CSRC, "-I" + os.path.join(ROOT, "include")]


def _stale(obj, src):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    deps = [src, os.path.join(CSRC, "common.h"), os.path.join(ROOT, "include", "flexam_hip.h")]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".inc")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    from . import gen_replay
    if gen_replay.write() and verbose:                 # csrc/replay_table.inc follows include/flexam_hip.h (committed; rewritten only when the header moved)
        print("regenerated", gen_replay.OUT, flush=True)
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs, procs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, src):
            cmd = [hipcc, *FLAGS, "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    if procs or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
