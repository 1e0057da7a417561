"""Dev tool: rocprofv3 PMC passes (tools/pmc_pass.sh) -> one table per kernel: mean counters per dispatch, average duration from
the kernel trace, and the derived figures the roofline lines quote:
  hbm_bytes   = 2 x FETCH_SIZE + WRITE_SIZE  (KiB -> bytes; gfx950 reports half of a wide streaming read: MI355X_MICROARCH.md, HBM)
  GB/s        = hbm_bytes / average duration
  mfma_busy   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)
  clock_ghz   = GRBM_GUI_ACTIVE / 8 / duration
usage: pmc_table.py <dir> [--json out.json] [--attn-traffic out.json [record-name]]
  --attn-traffic: the self-attention call's HBM bytes per launch (attn_fwd_kernel + attn_merge_kernel) with the sha of the attention
  sources they were measured on -> profiles/head_attn_traffic.json, the only file bench.py's roofline.traffic reads"""
import collections, csv, glob, json, subprocess, sys
root = sys.argv[1]


def demangle(name: str) -> str:
    """rocprofv3 leaves some template instances mangled (_ZN12_GLOBAL__N_1...); c++filt where the box has it, else the two
    GEMM kernels by pattern."""
    import re
    if not name.startswith("_Z"):
        return name
    try:
        out = subprocess.run(["c++filt", name], capture_output=True, text=True, timeout=5).stdout.strip()
        if out and out != name:
            return out
    except Exception:
        pass
    m = re.match(r"_ZN12_GLOBAL__N_1\d+(gemm_bf16_kernel|gemm_splitk_finish_kernel)ILi(\d)E(DF16b|f)Li(\d)E(?:Lb(\d)E)?", name)
    if m:
        epi = {"0": "EPI_NONE", "1": "EPI_GELU", "2": "EPI_GATE_RESIDUAL"}[m.group(2)]
        return (f"{m.group(1)}<{epi}, {'bf16' if m.group(3) == 'DF16b' else 'float'}, MT={m.group(4)}"
                + (f", TAIL={m.group(5)}" if m.group(5) is not None else "") + ">")
    return name

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[demangle(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = {}
for f in glob.glob(root + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[demangle(r["Name"])] = (float(r["AverageNs"]), int(r["Calls"]))
table = {}
for k, cs in sorted(agg.items()):
    if not any(t in k for t in ("anonymous namespace", "flexam", "_GLOBAL__N_", "gemm_", "attn_")) or "at::native" in k:
        continue
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    row = dict(counters={c: round(v, 1) for c, v in sorted(m.items())})
    if k in dur:
        ns, calls = dur[k]
        row["avg_us"], row["calls"] = round(ns / 1e3, 2), calls
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            b = (2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0
            row["hbm_bytes"], row["hbm_gbs"] = round(b), round(b / ns, 1)
        if "GRBM_GUI_ACTIVE" in m:
            row["clock_ghz"] = round(m["GRBM_GUI_ACTIVE"] / 8 / ns, 3)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
                row["mfma_busy"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * m["GRBM_GUI_ACTIVE"] / 8), 4)
    table[k] = row
for k, row in table.items():
    print(k[:110])
    print("   ", {a: b for a, b in row.items() if a != "counters"})
    print("   ", row["counters"])
if "--json" in sys.argv:
    json.dump(table, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)

if "--attn-traffic" in sys.argv:
    import os
    i = sys.argv.index("--attn-traffic")
    dst = sys.argv[i + 1]
    record = sys.argv[i + 2] if len(sys.argv) > i + 2 and not sys.argv[i + 2].startswith("--") else os.path.basename(root.rstrip("/"))
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, repo)
    from benchlib.kernels import attn_source_sha
    main = [k for k in table if "attn_fwd_kernel<0,true,true" in k.replace("(anonymous namespace)::", "").replace(" ", "")]   # <KIND = self, PRE, FULL, ...>
    merge = [k for k in table if "attn_merge_kernel" in k]
    if not main or "hbm_bytes" not in table[main[0]]:
        sys.exit("pmc_table --attn-traffic: no attn_fwd_kernel<0, true, true> row with FETCH_SIZE / WRITE_SIZE in " + root)
    b = table[main[0]]["hbm_bytes"] + (table[merge[0]].get("hbm_bytes", 0) if merge else 0)
    json.dump({"kernel": "one self-attention call = ONE attn_fwd_kernel<0, true, true> launch + attn_merge_kernel", "hbm_bytes_per_launch": b,
               "parts": {main[0]: table[main[0]]["hbm_bytes"], **({merge[0]: table[merge[0]].get("hbm_bytes", 0)} if merge else {})},
               "gfx950_fetch_correction": 2.0, "algorithmic_bytes_per_launch": 2 * 11648 * 24 * 128 * 2 * 4, "shape": [2, 11648, 24, 128],
               "source_sha16": attn_source_sha(repo), "record": record,
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes (tools/pmc_pass.sh over tools/kernel_driver.py), mean per dispatch"},
              open(dst, "w"), indent=1)
