set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5zn
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/raster_bench.py > $O/raster_bench.txt 2>&1
f=$(find $O/trace -name '*kernel_stats.csv' | head -1); cp $f $O/raster_kernel_stats.csv; rm -rf $O/trace
head -8 $O/raster_kernel_stats.csv | cut -c1-200
