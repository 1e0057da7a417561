# r6zs: final validation of HEAD -- GPU suite + the driver's bench command (5 emulated layouts, link model)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6zs
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -q -m gpu > $O/gpu_suite.txt 2>&1; echo "gpu_suite rc $?" >> $O/rc.txt
( time python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt; echo "bench rc $?" >> $O/rc.txt
python __graft_entry__.py --smoke > $O/smoke.txt 2>&1; echo "smoke rc $?" >> $O/rc.txt
cat $O/rc.txt; tail -n 3 $O/gpu_suite.txt; cat $O/bench_time.txt; tail -1 $O/smoke.txt
