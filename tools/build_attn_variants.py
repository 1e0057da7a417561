"""Dev tool: builds variants of the library that differ only in csrc/attn.hip compile-time switches, for tools/ab_attn_variants.py.
usage: build_attn_variants.py NAME=-DFOO=1,-DBAR=2 NAME2= ...   -> tools/probes/libflexam_var_NAME.so (the other objects come from
flexam_amd/build, so run `python -m flexam_amd.build` first)."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, "flexam_amd", "csrc"); objdir = os.path.join(root, "flexam_amd", "build"); out = os.path.join(root, "tools", "probes")
flags = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=fast", "-Wno-unused-result", "-I" + csrc, "-I" + os.path.join(root, "include"), "-I" + out]      # tools/probes holds attn_body16.inc (the diagnostic 16x16x32 body)
others = [os.path.join(objdir, f) for f in sorted(os.listdir(objdir)) if f.endswith(".o") and f != "attn.o"]
procs = []
for spec in sys.argv[1:]:
    name, _, defs = spec.partition("=")
    obj = os.path.join("/tmp", f"attn_var_{name}.o")
    # every variant is a diagnostic build: the ablation / stamp switches of attn.hip refuse to compile without this macro
    cmd = ["/opt/rocm/bin/hipcc", *flags, "-DFLEXAM_DIAGNOSTIC_BUILD", *[d for d in defs.split(",") if d], "-c", os.path.join(csrc, "attn.hip"), "-o", obj]
    procs.append((name, obj, subprocess.Popen(cmd)))
for name, obj, p in procs:
    if p.wait() != 0:
        raise SystemExit(f"variant {name} failed to compile")
    lib = os.path.join(out, f"libflexam_var_{name}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj, *others])
    print(lib)
