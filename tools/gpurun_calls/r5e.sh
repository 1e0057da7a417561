set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5e
mkdir -p $O
cd $R
python -m pytest tests/test_vae_gpu.py tests/test_pipeline_pixels_gpu.py tests/test_full_width_gpu.py::test_vae_decode_chunk_true_widths_sixteenth_area -m gpu -q -s > $O/tests.txt 2>&1
grep -E "slices|passed|failed|Error" $O/tests.txt | cut -c1-250 | tail -12
for v in "FLEXAM_VAE_SLICE_MB=0 FLEXAM_VAE_ENC_CHUNK=48" "FLEXAM_VAE_SLICE_MB=320 FLEXAM_VAE_ENC_CHUNK=48" "FLEXAM_VAE_SLICE_MB=320 FLEXAM_VAE_ENC_CHUNK=96" "FLEXAM_VAE_SLICE_MB=160 FLEXAM_VAE_ENC_CHUNK=96" "FLEXAM_VAE_SLICE_MB=640 FLEXAM_VAE_ENC_CHUNK=96"; do echo "$v"; env $v python tools/vae_bench.py 25 encode 2>&1 | grep -E "^encode \("; done > $O/vae_slices.txt 2>&1
for v in "FLEXAM_VAE_SLICE_MB=0 FLEXAM_VAE_DEC_CHUNK=2" "FLEXAM_VAE_SLICE_MB=320 FLEXAM_VAE_DEC_CHUNK=2" "FLEXAM_VAE_SLICE_MB=320 FLEXAM_VAE_DEC_CHUNK=4" "FLEXAM_VAE_SLICE_MB=320 FLEXAM_VAE_DEC_CHUNK=8" "FLEXAM_VAE_SLICE_MB=500 FLEXAM_VAE_DEC_CHUNK=8" "FLEXAM_VAE_SLICE_MB=160 FLEXAM_VAE_DEC_CHUNK=8"; do echo "$v"; env $v python tools/vae_bench.py 25 decode 2>&1 | grep -E "^decode"; done >> $O/vae_slices.txt 2>&1
cat $O/vae_slices.txt
bash tools/pmc_pass.sh gpurun_out/r5e/pmc tools/kernel_driver.py 3 > $O/pmc_pass.log 2>&1
python tools/pmc_table.py $O/pmc --attn-traffic $O/head_attn_traffic.json r5e_block_kernels_pmc.txt > $O/block_kernels_pmc.txt 2>&1
cat $O/head_attn_traffic.json | head -20
find $O/pmc -name "*.csv" -size +200k -delete; du -sh $O/pmc
bash tools/trace_bench.sh gpurun_out/r5e/trace > $O/trace_bench.txt 2>&1; tail -16 $O/trace_bench.txt
rm -rf $O/trace/trace
