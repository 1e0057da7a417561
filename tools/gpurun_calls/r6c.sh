# r6c: the 192 x 192 and 384 x 160 tile shapes (exact tests, encode A/B, emulated rank of 8 with / without the 192-wide tiles), packed all-to-all host path
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c
mkdir -p $O
cd $R
python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "gemm" > $O/tests_gemm.txt 2>&1; echo "tests_gemm rc $?" >> $O/rc.txt
python -m pytest tests/test_full_width_gpu.py -x -q -m gpu -s -k "per_rank_row_counts or ffn" > $O/tests_exact.txt 2>&1; echo "tests_exact rc $?" >> $O/rc.txt
python -m pytest tests/test_vae_gpu.py tests/test_conv_helpers_gpu.py tests/test_raster_gpu.py -x -q -m gpu > $O/tests_vae.txt 2>&1; echo "tests_vae rc $?" >> $O/rc.txt
python -m pytest tests/test_sp_gpu.py -q -m gpu -k "ulysses" > $O/tests_sp_ulysses.txt 2>&1; echo "tests_sp_ulysses rc $?" >> $O/rc.txt
python -m pytest tests/test_bench_launch.py -q -m gpu -k "emulated_rank_of_four" > $O/tests_bench_emul.txt 2>&1; echo "tests_bench_emul rc $?" >> $O/rc.txt
python tools/vae_encode_ab.py FLEXAM_GEMM_N160_TALL 0 1 0 1 > $O/vae_encode_tall_ab.txt 2>&1
for v in 0 1; do for l in 0 1 2; do FLEXAM_GEMM_N192=$v python tools/emulate_rank.py 8 $l 6 2 2>&1 | tail -1 | sed "s/^/N192=$v: /" >> $O/emulated_rank_n192_ab.txt; done; done
cat $O/rc.txt; tail -n 3 $O/tests_gemm.txt $O/tests_exact.txt $O/tests_vae.txt $O/tests_sp_ulysses.txt $O/tests_bench_emul.txt; cat $O/vae_encode_tall_ab.txt $O/emulated_rank_n192_ab.txt
