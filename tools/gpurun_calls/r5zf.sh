set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5zf
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q --durations=8 > $O/suite.txt 2>&1
tail -14 $O/suite.txt | cut -c1-200
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5zf/bench_default.json").read().strip().split("\n")[-1])
for k in ("value","ms_per_step","vae_decode_sec","vae_encode_sec_per_stream","conditioning_encode_sec_8_streams","sec_per_clip","dit_block_executed_mfma_frac"):
    print(k, d[k])
print(d["roofline"]["frac"], d["clip_end_to_end"]["sec"], d["emulated_ranks"]["predicted_scaling_no_comm"])
PY
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5zf/bench_default.json").read().strip().split("\n")[-1])
print(json.dumps(d["conditioning_raster"])[:700])
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
