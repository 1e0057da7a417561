set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5zi
mkdir -p $O
cd $R
FLEXAM_GEMM_MT=7 FLEXAM_AB_A=$R/flexam_amd/libflexam_hip.so FLEXAM_AB_B=$R/tools/probes/libflexam_var_ntmajor.so python tools/ab_gemm.py 9 > $O/ntmajor_mt7.txt 2>&1; cat $O/ntmajor_mt7.txt
