set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5zg
mkdir -p $O
cd $R
python -m pytest tests/test_raster_gpu.py tests/test_pipeline_pixels_gpu.py -m gpu -q -s > $O/tests.txt 2>&1; grep -E "passed|failed|tracks ->|Error" $O/tests.txt | tail -8
