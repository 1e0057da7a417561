# r6zp: the GELU output (FFN1 -> FFN2) with / without non-temporal stores at a rank's sizes (83 / 167 MB) and on one GPU (668 MB); exact tests first
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6zp
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_hip_kernels.py tests/test_full_width_gpu.py -q -m gpu -k "gemm or exact or gelu" > $O/tests.txt 2>&1; echo "tests rc $?" >> $O/rank.txt
for rep in 1 2; do
  for nt in 1 0; do
    for w in 8 4; do
      echo -n "world $w FLEXAM_GEMM_C_NT=$nt: " >> $O/rank.txt
      FLEXAM_GEMM_C_NT=$nt python tools/emulate_rank.py $w 0 20 5 2>/dev/null | tail -1 | sed 's/, host enqueue.*//' >> $O/rank.txt
    done
  done
done
timeout 600 python tools/ab_step.py FLEXAM_GEMM_C_NT=1 FLEXAM_GEMM_C_NT=0 --steps=10 --rounds=3 2>&1 | grep -v amdgpu.ids > $O/ab_step_single_gpu.txt
tail -2 $O/tests.txt; cat $O/rank.txt $O/ab_step_single_gpu.txt
