# r6b: full GPU suite with launch plans (replay) on by default + the driver's bench command
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6b
mkdir -p $O
cd $R
python -m pytest tests/test_replay_gpu.py -x -q -m gpu -s > $O/tests_replay.txt 2>&1; echo "tests_replay rc $?" >> $O/rc.txt
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_replay_gpu.py > $O/gpu_suite.txt 2>&1; echo "gpu_suite rc $?" >> $O/rc.txt
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?" >> $O/rc.txt
cat $O/rc.txt; tail -n 5 $O/tests_replay.txt; tail -n 8 $O/gpu_suite.txt
