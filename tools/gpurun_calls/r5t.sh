set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5t
mkdir -p $O
cd $R
python -m pytest tests/test_vae_gpu.py tests/test_pipeline_pixels_gpu.py tests/test_conv_helpers_gpu.py tests/test_full_width_gpu.py::test_vae_decode_chunk_true_widths_sixteenth_area -m gpu -q > $O/tests.txt 2>&1; tail -3 $O/tests.txt
python tools/ab_prep_pix.py > $O/prep_span.txt 2>&1; cat $O/prep_span.txt
python tools/vae_bench.py 25 decode 2>&1 | grep -E "^decode|TFLOP" > $O/vae.txt; python tools/vae_bench.py 25 encode 2>&1 | grep -E "^encode" >> $O/vae.txt; cat $O/vae.txt
