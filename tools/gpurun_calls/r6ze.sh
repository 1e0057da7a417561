# r6ze: ceilings of the two register-tile restructurings the r5 verdict names (item 3), as timing ablations at UNCHANGED occupancy:
# attention with half the K / V fragment reads per MFMA (what 64 q rows per wave would read), GEMM with half its fragment reads (more than
# 128 x 128 wave tiles would save); in the step, groups of 20 steps per arm, socket power and shader clock beside each; WRONG results by design
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6ze
mkdir -p $O
cd $R
timeout 1500 python tools/ab_step.py lib=tree lib=halffrag lib=gemmabl,FLEXAM_GEMM_DEBUG=0 lib=gemmabl,FLEXAM_GEMM_DEBUG=8 --steps=20 --rounds=3 2>&1 | grep -v amdgpu.ids > $O/ab_step_ceilings.txt; echo "ab_step rc $?" >> $O/rc.txt
timeout 300 python tools/power_attn.py lib=tree lib=halffrag 2>&1 | grep -v amdgpu.ids > $O/power_attn_isolated.txt; echo "power_attn rc $?" >> $O/rc.txt
cat $O/rc.txt $O/ab_step_ceilings.txt $O/power_attn_isolated.txt
