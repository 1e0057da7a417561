set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5z
mkdir -p $O
cd $R
bash tools/trace_bench.sh gpurun_out/r5z/trace > $O/trace_bench.txt 2>&1; tail -16 $O/trace_bench.txt
rm -rf $O/trace/trace
bash tools/pmc_pass.sh gpurun_out/r5z/pmc tools/kernel_driver.py 3 > $O/pmc_pass.log 2>&1; tail -3 $O/pmc_pass.log
python tools/pmc_table.py $O/pmc --attn-traffic $O/head_attn_traffic.json r5z_block_kernels_pmc.txt > $O/block_kernels_pmc.txt 2>&1
cat $O/head_attn_traffic.json | head -12
find $O/pmc -name "*.csv" -size +200k -delete; du -sh $O/pmc
