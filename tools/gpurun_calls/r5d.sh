set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5d
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q --durations=25 > $O/suite.txt 2>&1
tail -32 $O/suite.txt | cut -c1-200
python -m pytest tests/test_full_depth_gpu.py -m gpu -q -s > $O/full_depth.txt 2>&1
grep -E "psnr|rel-rms|passed|failed" $O/full_depth.txt | cut -c1-300
for n in 8 4 2; do for f in 1 0; do FLEXAM_SP_FUSED_QKV=$f python tools/emulate_rank.py $n 0 4 1 2>&1 | tail -1; done; done > $O/emulate_fusedqkv.txt 2>&1
cat $O/emulate_fusedqkv.txt
python tools/oracle_threads_probe.py 2>&1 | grep cpu_baseline > $O/cpu_baseline_threads.txt; cat $O/cpu_baseline_threads.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5d/bench_default.json").read().strip().split("\n")[-1])
for k in ("value","ms_per_step","vae_decode_sec","vae_encode_sec_per_stream","conditioning_encode_sec_8_streams","sec_per_clip","dit_block_executed_mfma_frac"):
    print(k, d[k])
print(d["roofline"]["frac"], d["clip_end_to_end"])
PY
