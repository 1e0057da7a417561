set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5v
mkdir -p $O
cd $R
python -m pytest tests/test_hip_kernels.py tests/test_dit_gpu.py tests/test_full_width_gpu.py -m gpu -q -x > $O/tests.txt 2>&1; tail -3 $O/tests.txt
FLEXAM_AB_A=$R/tools/probes/libflexam_var_base.so python tools/ab_gemm.py 7 > $O/ab_gemm.txt 2>&1; cat $O/ab_gemm.txt
python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-vae --no-clip --emulate-rank 0 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5v/bench.json").read().strip().split("\n")[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d.get("dit_block_executed_mfma_frac"))
print({k:v for k,v in d.get("kernels",{}).items()} if isinstance(d.get("kernels"),dict) else "")
PY
