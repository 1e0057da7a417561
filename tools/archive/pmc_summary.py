"""Dev tool: summarise rocprofv3 --pmc CSVs (counter_collection) per kernel: mean per dispatch."""
import csv, glob, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"][:60]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    if not any(x in k for x in ("gemm", "attn", "Cijk")):
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} mean {sum(v)/len(v):.4g}  (n={len(v)})")
for f in glob.glob(root + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(x in r["Name"] for x in ("gemm", "attn", "Cijk")):
            print("trace:", r["Name"][:60], "calls", r["Calls"], "avg_us", float(r["AverageNs"]) / 1e3)
