# r6zf: the same ceilings, arms of (nearly) equal power only: product vs half the attention fragment reads vs half the GEMM fragment reads
# (compile-time ablations, WRONG results by design), groups of 20 steps; + the GEMM shapes alone
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6zf
mkdir -p $O
cd $R
timeout 1500 python tools/ab_step.py lib=tree lib=halffrag lib=gemmhalf --steps=20 --rounds=3 2>&1 | grep -v amdgpu.ids > $O/ab_step_ceilings.txt; echo "ab_step rc $?" >> $O/rc.txt
FLEXAM_AB_A=$R/flexam_amd/libflexam_hip.so FLEXAM_AB_B=$R/tools/probes/libflexam_var_gemmhalf.so timeout 600 python tools/ab_gemm.py 7 2>&1 | grep -v amdgpu.ids > $O/ab_gemm_half_frag.txt; echo "ab_gemm rc $?" >> $O/rc.txt
cat $O/rc.txt $O/ab_step_ceilings.txt $O/ab_gemm_half_frag.txt
