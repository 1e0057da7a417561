set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5h
mkdir -p $O
cd $R
python -m pytest tests/test_dit_gpu.py tests/test_bench_launch.py::test_one_gpu_line_carries_an_emulated_rank_of_four tests/test_abi_errors_gpu.py tests/test_hip_kernels.py -m gpu -q -s > $O/tests.txt 2>&1
grep -E "odd latent|padded self|passed|failed|Error" $O/tests.txt | cut -c1-250 | tail -12
python tools/ab_prep_pix.py > $O/prep_pix.txt 2>&1; cat $O/prep_pix.txt
