set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5ze
mkdir -p $O
cd $R
python -m pytest tests/test_raster_gpu.py tests/test_abi_errors_gpu.py -m gpu -q -x > $O/tests.txt 2>&1; tail -15 $O/tests.txt
python tools/raster_bench.py > $O/raster_bench.txt 2>&1; cat $O/raster_bench.txt
