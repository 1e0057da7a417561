set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5i
mkdir -p $O
cd $R
python tools/ab_prep_pix.py > $O/prep_pix.txt 2>&1; cat $O/prep_pix.txt
python tools/vae_bench.py 25 decode 2>&1 | grep -E "^decode|TFLOP" > $O/vae.txt; python tools/vae_bench.py 25 encode 2>&1 | grep -E "^encode" >> $O/vae.txt; cat $O/vae.txt
