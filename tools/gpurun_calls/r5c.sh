set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5c
mkdir -p $O
cd $R
python -m pytest tests/test_vae_gpu.py tests/test_pipeline_pixels_gpu.py tests/test_full_width_gpu.py::test_vae_decode_chunk_true_widths_sixteenth_area tests/test_conv_helpers_gpu.py -m gpu -q -s --durations=10 > $O/tests.txt 2>&1
grep -E "psnr|rel-rms|passed|failed|Error|error" $O/tests.txt | cut -c1-250 | tail -40
for v in "FLEXAM_VAE_UPCONV=image FLEXAM_VAE_HEADCONV=implicit" "FLEXAM_VAE_UPCONV=phase FLEXAM_VAE_HEADCONV=implicit" "FLEXAM_VAE_UPCONV=phase FLEXAM_VAE_HEADCONV=fold"; do echo "$v"; env $v python tools/vae_bench.py 25 decode 2>&1 | grep -E "^decode|TFLOP"; done > $O/vae_variants.txt 2>&1
for c in 24 48; do echo "ENC_CHUNK=$c"; FLEXAM_VAE_ENC_CHUNK=$c python tools/vae_bench.py 25 encode 2>&1 | grep -E "^encode"; done >> $O/vae_variants.txt 2>&1
cat $O/vae_variants.txt
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/vae_dec -- python3 $R/tools/vae_bench.py 25 decode > $O/vae_dec.log 2>&1
FLEXAM_VAE_ENC_CHUNK=48 rocprofv3 --kernel-trace --stats --output-format csv -d $O/vae_enc -- python3 $R/tools/vae_bench.py 25 encode > $O/vae_enc.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/emu8_ag -- python3 $R/tools/emulate_rank.py 8 0 > $O/emu8_ag.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/emu8_a2a -- python3 $R/tools/emulate_rank.py 8 1 > $O/emu8_a2a.log 2>&1
for d in vae_dec vae_enc emu8_ag emu8_a2a; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); cp $f $O/${d}_kernel_stats.csv; f=$(find $O/$d -name "*kernel_trace.csv" | head -1); gzip -c $f > $O/${d}_kernel_trace.csv.gz; rm -rf $O/$d; done
tail -2 $O/emu8_ag.log $O/emu8_a2a.log
cd $R
python tools/power_model.py 2.5 247.7 > $O/power_model.txt 2>&1; tail -22 $O/power_model.txt | cut -c1-260
