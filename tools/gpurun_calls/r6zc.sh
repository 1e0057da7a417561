# r6zc: 192-wide gate-residual epilogue with all three X batches of its tile in flight: exact tests, then A/B against HEAD's gemm.hip at M = 2912 / 5824
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6zc
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_full_width_gpu.py -q -m gpu -k "gemm or exact or gate" > $O/tests.txt 2>&1; echo "tests rc $?" >> $O/rc.txt
for m in 2912 5824; do
  echo "== M=$m" >> $O/ab.txt
  FLEXAM_AB_M=$m FLEXAM_AB_A=$R/tools/probes/libflexam_var_gemmbase.so timeout 600 python tools/ab_gemm.py 9 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
done
cat $O/rc.txt; tail -2 $O/tests.txt; cat $O/ab.txt
