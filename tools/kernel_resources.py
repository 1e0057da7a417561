"""Register / scratch footprint of every kernel in libflexam_hip.so, read from the code objects the library carries (the AMDGPU
metadata notes hipcc writes: .vgpr_count, .agpr_count, .vgpr_spill_count, .private_segment_fixed_size, LDS).  Library for
tests/test_isa_resources_cpu.py (no shipped hot-loop instance may spill) and a CLI:  python tools/kernel_resources.py [substring]"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "flexam_amd", "libflexam_hip.so")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
FIELDS = ("agpr_count", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
          "group_segment_fixed_size")


def demangle(names):
    filt = os.path.join(LLVM, "llvm-cxxfilt")
    if not os.path.exists(filt):
        return list(names)
    out = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return out[:len(names)]


def kernel_resources(lib=LIB):
    """{demangled kernel name: {field: int}} over every gfx950 code object bundled in `lib`."""
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        data = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(MAGIC, data)] + [len(data)]
        for i in range(len(starts) - 1):
            bundle, co = os.path.join(tmp, f"b{i}.bin"), os.path.join(tmp, f"co{i}.o")
            open(bundle, "wb").write(data[starts[i]:starts[i + 1]])
            subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input", bundle,
                                   "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output", co], stderr=subprocess.DEVNULL)
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
            for entry in re.split(r"\n\s*- \.agpr_count:", "\n" + notes)[1:]:
                entry = ".agpr_count:" + entry
                name = re.search(r"\.name:\s+(\S+)", entry)
                if not name:
                    continue
                vals = {}
                for f in FIELDS:
                    m = re.search(r"\." + f + r":\s+(\d+)", entry)
                    vals[f] = int(m.group(1)) if m else 0
                res[name.group(1)] = vals
    names = list(res)
    return {d: res[n] for n, d in zip(names, demangle(names))}


if __name__ == "__main__":
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    for name, v in sorted(kernel_resources().items()):
        if pat in name:
            short = name.replace("(anonymous namespace)::", "")
            short = re.sub(r"\((anonymous namespace::)?\w*Params.*$", "", short)
            print(f"{short[:110]:110s} vgpr {v['vgpr_count']:3d} agpr {v['agpr_count']:3d} spill {v['vgpr_spill_count']:3d} scratch {v['private_segment_fixed_size']:4d} B  lds {v['group_segment_fixed_size']}")
