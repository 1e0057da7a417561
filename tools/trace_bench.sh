#!/bin/bash
# usage (GPU box, repo root): bash tools/trace_bench.sh <outdir> [bench.py args...]   (environment is passed through)
# rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-vae --no-clip --no-kernel-timing --emulate-rank 0`
# (the emulated-rank legs of the default line launch the same kernels at other shapes: they would pollute the per-kernel averages);
# copies the kernel stats CSV to <outdir>/kernel_stats.csv and prints the top rows
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run through gpurun, from the repo root}"
out=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p "$R/$out"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$out/trace" -- python3 "$R/bench.py" --steps 4 --warmup 1 --no-cpu-baseline --no-vae --no-clip --no-kernel-timing --emulate-rank 0 "$@" > "$R/$out/bench.json" 2> "$R/$out/trace.log" || { tail -20 "$R/$out/trace.log"; exit 1; }
f=$(find "$R/$out/trace" -name "*kernel_stats.csv" | head -1)
cp "$f" "$R/$out/kernel_stats.csv"
python3 - "$R/$out/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:9.2f} ms  {r['Percentage']}%")
PY
tail -1 "$R/$out/bench.json" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step (profiled)', d['ms_per_step'])"
