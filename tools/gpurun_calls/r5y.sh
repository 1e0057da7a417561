set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5y
mkdir -p $O
cd $R
python -m pytest tests/test_hip_kernels.py tests/test_vae_gpu.py tests/test_conv_helpers_gpu.py -m gpu -q -x > $O/tests.txt 2>&1; tail -3 $O/tests.txt
python tools/ab_vae_variants.py base > $O/ab_vae.txt 2>&1; tail -4 $O/ab_vae.txt
