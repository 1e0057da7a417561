set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5zh
mkdir -p $O
cd $R
python -m pytest tests/test_raster_gpu.py -m gpu -q > $O/tests.txt 2>&1; tail -2 $O/tests.txt
python tools/raster_bench.py 2 > $O/raster_bench_step2.txt 2>&1; cat $O/raster_bench_step2.txt
python tools/raster_bench.py 1 > $O/raster_bench_step1.txt 2>&1; cat $O/raster_bench_step1.txt
