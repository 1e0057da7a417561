set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5w
mkdir -p $O
cd $R
python tools/ab_gemm_stagger.py 0,20,40,60,80,120,160,240 5 > $O/stagger.txt 2>&1; cat $O/stagger.txt
