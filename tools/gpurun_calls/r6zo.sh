# r6zo: X (fp32 residual) with / without non-temporal hints in the gate-residual GEMM epilogue: one emulated rank of eight and of four (X = 36 / 72 MB:
# cacheable), and the single-GPU step (286 MB: the r4 finding)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6zo
mkdir -p $O
cd $R
for rep in 1 2; do
  for nt in 1 0; do
    for w in 8 4; do
      echo -n "world $w FLEXAM_GEMM_X_NT=$nt: " >> $O/rank.txt
      FLEXAM_GEMM_X_NT=$nt python tools/emulate_rank.py $w 0 20 5 2>/dev/null | tail -1 >> $O/rank.txt
    done
  done
done
timeout 600 python tools/ab_step.py FLEXAM_GEMM_X_NT=1 FLEXAM_GEMM_X_NT=0 --steps=10 --rounds=3 2>&1 | grep -v amdgpu.ids > $O/ab_step_single_gpu.txt
cat $O/rank.txt $O/ab_step_single_gpu.txt
