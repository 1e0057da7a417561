set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5k
mkdir -p $O
cd $R
python -m pytest tests/test_vae_gpu.py tests/test_pipeline_pixels_gpu.py tests/test_conv_helpers_gpu.py tests/test_full_width_gpu.py::test_vae_decode_chunk_true_widths_sixteenth_area -m gpu -q > $O/tests.txt 2>&1; tail -3 $O/tests.txt
python tools/vae_bench.py 25 decode 2>&1 | grep -E "^decode|TFLOP" > $O/vae.txt; python tools/vae_bench.py 25 encode 2>&1 | grep -E "^encode" >> $O/vae.txt; cat $O/vae.txt
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/vae_dec -- python3 $R/tools/vae_bench.py 25 decode > $O/vae_dec.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/vae_enc -- python3 $R/tools/vae_bench.py 25 encode > $O/vae_enc.log 2>&1
for d in vae_dec vae_enc; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); cp $f $O/${d}_kernel_stats.csv; rm -rf $O/$d; done
head -24 $O/vae_dec_kernel_stats.csv | cut -c1-200
