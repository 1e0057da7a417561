# r6e: kernel trace of one emulated rank of 8 (default layout: cfg2 x sp4, one K|V gather waited for; and cfg2 x sp4 all-to-all)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6e
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for l in 0 2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace$l -- python3 $R/tools/emulate_rank.py 8 $l 5 1 > $O/emulate_rank_$l.txt 2>&1
  f=$(find $O/trace$l -name '*kernel_stats.csv' | head -1); cp $f $O/emulated_rank_of_8_layout${l}_kernel_stats.csv; rm -rf $O/trace$l
  tail -1 $O/emulate_rank_$l.txt
done
head -22 $O/emulated_rank_of_8_layout0_kernel_stats.csv | cut -c1-160
