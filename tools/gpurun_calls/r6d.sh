# r6d: the 192 x 192 tile after the shadow-lane fix: exact tests, kernel-level A/B at M = 2912 / 5824, emulated rank of 8 with / without it
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6d
mkdir -p $O
cd $R
python -m pytest tests/test_hip_kernels.py -q -m gpu -k "gemm" > $O/tests_gemm.txt 2>&1; echo "tests_gemm rc $?" >> $O/rc.txt
python -m pytest tests/test_full_width_gpu.py -q -m gpu -s -k "per_rank_row_counts or ffn" > $O/tests_exact.txt 2>&1; echo "tests_exact rc $?" >> $O/rc.txt
for m in 2912 5824 11648; do echo "== M=$m" >> $O/gemm_n192_ab.txt; FLEXAM_AB_M=$m python tools/ab_env.py FLEXAM_GEMM_N192 0 1 2 >> $O/gemm_n192_ab.txt 2>&1; done
for v in 0 1; do for l in 0 1 2; do FLEXAM_GEMM_N192=$v python tools/emulate_rank.py 8 $l 6 2 2>&1 | tail -1 | sed "s/^/N192=$v: /" >> $O/emulated_rank_n192_ab.txt; done; done
cat $O/rc.txt; tail -n 3 $O/tests_gemm.txt $O/tests_exact.txt; cat $O/gemm_n192_ab.txt $O/emulated_rank_n192_ab.txt
