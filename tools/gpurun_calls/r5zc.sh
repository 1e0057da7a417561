set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5zc
mkdir -p $O
cd $R
FLEXAM_AB_ROUNDS=9 python tools/ab_env.py FLEXAM_GEMM_MT 0 8 7 6 5 > $O/mt.txt 2>&1; cat $O/mt.txt
