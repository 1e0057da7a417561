# r6m (as r6i, after the chunked MXFP8 records): records of HEAD in one call -- GPU suite, the driver's bench command, kernel trace of the step, counter passes + attention traffic record, variant lines
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6m
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -q -m gpu > $O/gpu_suite.txt 2>&1; echo "gpu_suite rc $?" >> $O/rc.txt
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?" >> $O/rc.txt
bash tools/trace_bench.sh gpurun_out/r6m/trace > $O/trace_bench.txt 2>&1; rm -rf $O/trace/trace
bash tools/pmc_pass.sh gpurun_out/r6m/pmc tools/kernel_driver.py 3 > $O/pmc_pass.log 2>&1; tail -3 $O/pmc_pass.log
python tools/pmc_table.py $O/pmc --attn-traffic $O/head_attn_traffic.json r6m_block_kernels_pmc.txt > $O/block_kernels_pmc.txt 2>&1
find $O/pmc -name "*.csv" -size +200k -delete
for v in "--mask blob" "--sage" "--fp8" "--fp8 --sage" "--fp8 --sage --fp8-oproj" "--height 704 --width 1280" "--fp8 --height 704 --width 1280 --emulate-rank 8" "--fp8 --sage --emulate-rank 8" "--logit-scale 6"; do
  echo "== bench.py $v" >> $O/bench_lines.txt
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-vae --no-clip $v 2>> $O/bench_lines.err | tail -1 >> $O/bench_lines.txt
done
python __graft_entry__.py --smoke > $O/smoke.txt 2>&1; echo "smoke rc $?" >> $O/rc.txt
cat $O/rc.txt; tail -n 6 $O/gpu_suite.txt; tail -n 18 $O/trace_bench.txt; cat $O/head_attn_traffic.json | head -8
