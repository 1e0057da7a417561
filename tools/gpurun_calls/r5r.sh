set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5r
mkdir -p $O
cd $R
python tools/vae_bench.py 25 all 2>&1 | grep -E "^decode|^encode|^band|TFLOP" > $O/vae_all.txt; cat $O/vae_all.txt
