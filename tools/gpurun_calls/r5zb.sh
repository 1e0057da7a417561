set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5zb
mkdir -p $O
cd $R
FLEXAM_AB_ROUNDS=15 python tools/ab_env.py FLEXAM_GEMM_GM 4 6 8 12 16 > $O/gm2.txt 2>&1; cat $O/gm2.txt
