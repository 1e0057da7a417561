set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5a
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q --durations=60 > $O/suite.txt 2>&1
tail -5 $O/suite.txt
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/vae_dec -- python3 $R/tools/vae_bench.py 25 decode > $O/vae_dec.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/vae_enc -- python3 $R/tools/vae_bench.py 25 encode > $O/vae_enc.log 2>&1
for d in vae_dec vae_enc; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); cp $f $O/${d}_kernel_stats.csv; f=$(find $O/$d -name "*kernel_trace.csv" | head -1); gzip -c $f > $O/${d}_kernel_trace.csv.gz; rm -rf $O/$d; done
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 3000 $O/bench_default.json
