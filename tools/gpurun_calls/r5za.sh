set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5z
mkdir -p $O
cd $R
bash tools/trace_bench.sh gpurun_out/r5z/trace2 > $O/trace_bench2.txt 2>&1; tail -16 $O/trace_bench2.txt
rm -rf $O/trace2/trace
