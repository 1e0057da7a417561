# r6x: EXPERIMENT -- de-phased epilogues: workgroup groups of the persistent GEMM start g x D microseconds apart (FLEXAM_GEMM_STAGGER=groups:us)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6x
mkdir -p $O
cd $R
FLEXAM_AB_ROUNDS=9 timeout 600 python tools/ab_env.py FLEXAM_GEMM_STAGGER 0 2:5 2:10 2:20 4:5 4:10 3:8 > $O/stagger_M23296.txt 2>&1
cat $O/stagger_M23296.txt
