set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5n
mkdir -p $O
cd $R
python bench.py --height 704 --width 1280 --no-cpu-baseline > $O/bench_704x1280_full.json 2> $O/bench_704.err; tail -3 $O/bench_704.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5n/bench_704x1280_full.json").read().strip().split("\n")[-1])
for k in ("value","ms_per_step","vae_decode_sec","vae_encode_sec_per_stream","conditioning_encode_sec_8_streams","sec_per_clip","finite"):
    print(k, d[k])
print(d["roofline"]["frac"], d["clip_end_to_end"])
PY
python bench.py --mask blob --no-cpu-baseline --no-clip > $O/bench_blob.json 2> $O/bench_blob.err
python bench.py --fp8 --no-cpu-baseline --no-clip > $O/bench_fp8.json 2> $O/bench_fp8.err
python - <<'PY'
import json
for n in ("bench_blob","bench_fp8"):
    d=json.loads(open(f"gpurun_out/r5n/{n}.json").read().strip().split("\n")[-1])
    print(n, round(d["value"],3), round(d["ms_per_step"],2), d["dtype"], d["config"]["workload"][-90:])
PY
