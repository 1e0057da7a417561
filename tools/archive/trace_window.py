"""Dev tool: a window of a rocprofv3 --kernel-trace CSV in start order: queue, kernel, start offset, duration, grid.  usage: trace_window.py <kernel_trace.csv> [fraction]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = ("gemm_bf16", "attn_fwd", "ln_modulate", "rmsnorm")
rows = [r for r in rows if any(k in r["Kernel_Name"] for k in keep)]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.8
w = rows[int(len(rows) * frac):int(len(rows) * frac) + 44]
t0 = int(w[0]["Start_Timestamp"])
for r in w:
    nm = r["Kernel_Name"]
    short = "attn" if "attn_fwd" in nm else "ln" if "ln_mod" in nm else "rmsnorm" if "rmsnorm" in nm else "gemm " + (nm.split("gemm_bf16_kernelILi")[1][:14] if "kernelILi" in nm else nm[-40:])
    q = r.get("Queue_Id", "?")
    s = (int(r["Start_Timestamp"]) - t0) / 1e3
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"q{q:>3s} {short:24s} start {s:9.1f} us  end {s + d:9.1f}  dur {d:8.1f} us  grid {r.get('Grid_Size', '?')}")
