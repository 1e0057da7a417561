set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5g
mkdir -p $O
cd $R
python -m pytest tests/test_dit_gpu.py tests/test_bench_launch.py tests/test_block_seam_gpu.py tests/test_sampler_gpu.py -m gpu -q -s > $O/tests.txt 2>&1
grep -E "odd latent|padded self|passed|failed|Error" $O/tests.txt | cut -c1-250 | tail -12
python bench.py --fp8 --sage --no-vae --no-clip --no-cpu-baseline > $O/bench_fp8_sage.json 2> $O/bench_fp8_sage.err
python bench.py --height 704 --width 1280 --no-vae --no-clip --no-cpu-baseline > $O/bench_704x1280.json 2> $O/bench_704.err
python - <<'PY'
import json
for n in ("bench_fp8_sage","bench_704x1280"):
    d=json.loads(open(f"gpurun_out/r5g/{n}.json").read().strip().split("\n")[-1])
    print(n, round(d["value"],3), round(d["ms_per_step"],2), d["dtype"], round(d["roofline"]["frac"],3), "emulated" in str(d.keys()))
PY
