# r6q: the link model of the emulated ranks (flexam_delay_us, LoopbackGroup link_gbps): tests + the default bench line with predicted_scaling_at_link_GBps
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6q
mkdir -p $O
cd $R
python -m pytest tests/test_replay_gpu.py tests/test_bench_launch.py -q -m gpu -k "delay or emulated_rank_of_four or replay" > $O/tests.txt 2>&1; echo "tests rc $?" >> $O/rc.txt
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?" >> $O/rc.txt
cat $O/rc.txt; tail -4 $O/tests.txt
