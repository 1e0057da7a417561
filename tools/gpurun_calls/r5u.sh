set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5u
mkdir -p $O
cd $R
FAST="--no-cpu-baseline --no-vae --no-clip --emulate-rank 0"
python -m pytest tests/test_dit_gpu.py -m gpu -q -x > $O/tests_fwd.txt 2>&1; tail -2 $O/tests_fwd.txt
FLEXAM_EW_REVERSE=1 python -m pytest tests/test_dit_gpu.py -m gpu -q -x > $O/tests_rev.txt 2>&1; tail -2 $O/tests_rev.txt
for rep in 1 2; do
for arm in 0 1; do
  FLEXAM_EW_REVERSE=$arm python bench.py --steps 12 --warmup 3 $FAST --no-kernel-timing > $O/bench_rev${arm}_$rep.json 2> $O/bench_rev${arm}_$rep.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r5u/bench_rev${arm}_$rep.json").read().strip().split("\n")[-1])
print("reverse=${arm} rep $rep", d["ms_per_step"], d["value"])
PY
done
done
cd /tmp; export TMPDIR=/tmp
for arm in 0 1; do
  export FLEXAM_EW_REVERSE=$arm
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$arm -o p -- python3 $R/bench.py --steps 4 --warmup 1 $FAST --no-kernel-timing > $O/prof$arm.log 2>&1
  f=$(find $O/prof$arm -name '*kernel_stats.csv' | head -1)
  head -16 $f | cut -c1-160 > $O/kernel_stats_rev$arm.txt
  cp $f $O/kernel_stats_rev$arm.csv
  rm -rf $O/prof$arm
done
unset FLEXAM_EW_REVERSE
grep -E "ln_modulate|rmsnorm_rope|Li0EDF16bLi8ELb0ELi2ELi4E|Li0EDF16bLi7" $O/kernel_stats_rev0.txt $O/kernel_stats_rev1.txt | cut -c1-220
