set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5m
mkdir -p $O
cd $R
python -m pytest tests/test_bench_launch.py -m gpu -q > $O/tests.txt 2>&1; tail -3 $O/tests.txt
for i in 0 1 2; do python tools/emulate_rank.py 8 $i 6 2 2>&1 | tail -1; done > $O/emulate8.txt; cat $O/emulate8.txt
for i in 0 1 2; do python tools/emulate_rank.py 4 $i 6 2 2>&1 | tail -1; done > $O/emulate4.txt; cat $O/emulate4.txt
