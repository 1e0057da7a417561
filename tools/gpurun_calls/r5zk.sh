set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5zk
mkdir -p $O
cd $R
for v in "default:" "sage:--sage" "fp8:--fp8" "fp8_sage:--fp8 --sage" "fp8_sage_oproj:--fp8 --sage --fp8-oproj" "blob:--mask blob" "704x1280:--height 704 --width 1280"; do
  name=${v%%:*}; flags=${v#*:}
  python bench.py $flags --no-cpu-baseline --no-vae --no-clip --emulate-rank 0 > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r5zk/bench_$name.json").read().strip().split("\n")[-1])
print("$name", round(d["value"],3), "steps/s", round(d["ms_per_step"],2), "ms", d["dtype"], "finite", d["finite"], "roofline", round(d["roofline"]["frac"],3))
PY
done
