set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5f
mkdir -p $O
cd $R
python -m pytest tests/test_vae_gpu.py tests/test_sp_gpu.py tests/test_pipeline_pixels_gpu.py -m gpu -q > $O/tests.txt 2>&1; tail -3 $O/tests.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json; echo
python tools/parity_30_layers_L2912.py > $O/parity_30_layers_L2912.txt 2>&1; cat $O/parity_30_layers_L2912.txt | cut -c1-400
