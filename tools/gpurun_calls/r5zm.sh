set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5zm
mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -2 $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5zm/bench_default.json").read().strip().split("\n")[-1])
for k in ("value","ms_per_step","vae_decode_sec","conditioning_encode_sec_8_streams","sec_per_clip","sec_per_clip_from_tracks","dit_block_executed_mfma_frac"):
    print(k, d[k])
print(d["roofline"]["frac"], d["clip_end_to_end"]["sec"], d["conditioning_raster"]["sec"], d["cpu_baseline"]["value"])
PY
