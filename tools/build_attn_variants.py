"""Dev tool: builds variants of the library that differ only in the compile-time switches of ONE source (csrc/attn.hip, or --src=gemm.hip
...), for tools/ab_attn_variants.py / ab_step.py.  usage: build_attn_variants.py [--src=gemm.hip] NAME=-DFOO=1,-DBAR=2 NAME2= ...
-> tools/probes/libflexam_var_NAME.so (the other objects come from flexam_amd/build, so run `python -m flexam_amd.build` first)."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, "flexam_amd", "csrc"); objdir = os.path.join(root, "flexam_amd", "build"); out = os.path.join(root, "tools", "probes")
flags = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=fast", "-Wno-unused-result", "-I" + csrc, "-I" + os.path.join(root, "include"), "-I" + out]      # tools/probes holds attn_body16.inc (the diagnostic 16x16x32 body)
src = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--src=")), "attn.hip")
others = [os.path.join(objdir, f) for f in sorted(os.listdir(objdir)) if f.endswith(".o") and f != src.replace(".hip", ".o")]
procs = []
for spec in [a for a in sys.argv[1:] if not a.startswith("--src=")]:
    name, _, defs = spec.partition("=")
    obj = os.path.join("/tmp", f"attn_var_{name}.o")
    # every variant is a diagnostic build: the ablation / stamp switches of attn.hip refuse to compile without this macro
    cmd = ["/opt/rocm/bin/hipcc", *flags, "-DFLEXAM_DIAGNOSTIC_BUILD", *[d for d in defs.split(",") if d], "-c", os.path.join(csrc, src), "-o", obj]
    procs.append((name, obj, subprocess.Popen(cmd)))
for name, obj, p in procs:
    if p.wait() != 0:
        raise SystemExit(f"variant {name} failed to compile")
    lib = os.path.join(out, f"libflexam_var_{name}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj, *others])
    print(lib)
