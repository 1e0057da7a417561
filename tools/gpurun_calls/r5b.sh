set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5b
mkdir -p $O
cd $R
python tools/oracle_threads_probe.py > $O/oracle_threads.txt 2>&1
tail -14 $O/oracle_threads.txt
python -m pytest tests/test_vae_gpu.py tests/test_pipeline_pixels_gpu.py tests/test_full_depth_gpu.py tests/test_sp_gpu.py tests/test_full_width_gpu.py tests/test_fp8_gpu.py tests/test_attn_fp8_gpu.py tests/test_bench_launch.py -m gpu -q -s --durations=40 > $O/tests.txt 2>&1
grep -E "psnr|rel-rms|passed|failed|rows|Error|error" $O/tests.txt | cut -c1-300 | tail -90
for c in 4 8 24 48 96; do echo "ENC_CHUNK=$c"; FLEXAM_VAE_ENC_CHUNK=$c python tools/vae_bench.py 25 encode 2>&1 | grep -E "^encode"; done > $O/vae_chunks.txt 2>&1
for c in 1 2 3 4 6 8; do echo "DEC_CHUNK=$c"; FLEXAM_VAE_DEC_CHUNK=$c python tools/vae_bench.py 25 decode 2>&1 | grep -E "^decode|TFLOP"; done >> $O/vae_chunks.txt 2>&1
cat $O/vae_chunks.txt
python bench.py --no-vae --no-clip --no-cpu-baseline --emulate-rank 8 > $O/bench_emulate8.json 2> $O/bench_emulate8.err; tail -c 2500 $O/bench_emulate8.json
python bench.py --no-vae --no-clip --no-cpu-baseline --no-kernel-timing --emulate-rank 4 > $O/bench_emulate4.json 2> $O/bench_emulate4.err
python bench.py --no-vae --no-clip --no-cpu-baseline --no-kernel-timing --emulate-rank 2 > $O/bench_emulate2.json 2> $O/bench_emulate2.err
python bench.py --no-vae --no-clip --no-cpu-baseline --logit-scale 6 > $O/bench_logit6.json 2> $O/bench_logit6.err
python tools/power_model.py 2.5 > $O/power_model.txt 2>&1; tail -20 $O/power_model.txt
