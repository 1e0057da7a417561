set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5t
mkdir -p $O
cd $R
python -m pytest tests/test_vae_gpu.py tests/test_pipeline_pixels_gpu.py tests/test_conv_helpers_gpu.py tests/test_full_width_gpu.py::test_vae_decode_chunk_true_widths_sixteenth_area -m gpu -q > $O/tests.txt 2>&1; tail -3 $O/tests.txt
python tools/ab_prep_pix.py > $O/prep_span.txt 2>&1; cat $O/prep_span.txt
python tools/vae_bench.py 25 decode 2>&1 | grep -E "^decode|TFLOP" > $O/vae.txt; python tools/vae_bench.py 25 encode 2>&1 | grep -E "^encode" >> $O/vae.txt; cat $O/vae.txt
rocprofv3 --kernel-trace --stats -d $O/prof_enc -o enc -- python tools/vae_bench.py 25 encode > $O/prof_enc.log 2>&1
head -45 $(find $O/prof_enc -name '*kernel_stats.csv' | head -1) > $O/enc_kernels.txt
find $O/prof_enc -name '*kernel_stats.csv' | head -1) > $O/enc_kernels.txt
find $O/prof_enc -name '*.csv' ! -name '*kernel_stats.csv' -delete; find $O/prof_enc -name '*.db' -delete
head -30 $O/enc_kernels.txt
