set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5zj
mkdir -p $O
cd $R
python -m pytest tests/test_hip_kernels.py tests/test_raster_gpu.py tests/test_vae_gpu.py -m gpu -q > $O/tests.txt 2>&1; tail -2 $O/tests.txt
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-vae --no-clip --emulate-rank 0 > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.loads(open('gpurun_out/r5zj/bench.json').read().strip().split('\n')[-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'])"
