set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5q
mkdir -p $O
cd $R
python -m pytest tests/test_bench_launch.py tests/test_conv_helpers_gpu.py -m gpu -q > $O/tests.txt 2>&1; tail -4 $O/tests.txt
